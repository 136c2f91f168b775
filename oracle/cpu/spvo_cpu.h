/*
 * spvo_cpu.h -- C ABI of the CPU restatement of the hot path (TEST INFRASTRUCTURE: the compiled second oracle and the
 * CPU timing baseline of bench.py; the product path -- libspvo.so -- never loads this library).
 *
 * Plain C++17 + OpenMP, no third-party code.  Every function restates a stage of the reference
 * (src/odml_visual_odometry/src/feature_detection_neural_network.cpp = "nn.cpp",
 *  src/odml_visual_odometry/src/feature_detection_base.cpp = "base.cpp",
 *  include/odml_visual_odometry/ceres_cost_function.hpp = "cost.hpp") with the same decisions the Python files of oracle/ pin;
 * tests/test_cpu_backend.py checks the two restatements against each other (integer outputs bit for bit).
 */
#ifndef SPVO_CPU_H
#define SPVO_CPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct spvo_cpu spvo_cpu;

typedef struct {
  int net_height, net_width;     /* multiples of 8 (hpp:296)                      */
  float conf_thresh;             /* nn.cpp:203, strict >                          */
  int dist_thresh, border_remove, max_keypoints;   /* nn.cpp:239-257, hpp:368     */
  int bug_compat_p;              /* base.cpp:95,111 (see include/spvo.h)          */
  int num_threads;               /* OpenMP threads, 0 = all                       */
} spvo_cpu_config;

void spvo_cpu_default_config(spvo_cpu_config *cfg);
int spvo_cpu_create(const spvo_cpu_config *cfg, spvo_cpu **out);
void spvo_cpu_destroy(spvo_cpu *c);
const char *spvo_cpu_last_error(void);
int spvo_cpu_threads(const spvo_cpu *c);

int spvo_cpu_load_weights(spvo_cpu *c, const char *path);   /* the same .spvw file spvo_load_weights reads (FP32 engines) */

/* base.cpp:68-121 + nn.cpp:139-161 */
int spvo_cpu_preprocess(spvo_cpu *c, const uint8_t *img, int rows, int cols, size_t stride, double P[12], uint8_t *resized);
/* nn.cpp:163-176: input [batch,1,H,W] -> det [batch,65,H/8,W/8], desc [batch,256,H/8,W/8] (NCHW, L2-normalised) */
int spvo_cpu_forward(spvo_cpu *c, const float *input, int batch, float *det, float *desc);
int spvo_cpu_heatmap(spvo_cpu *c, const float *det, float *heat);                       /* nn.cpp:266-326 */
int spvo_cpu_nms(spvo_cpu *c, const float *heat, int32_t *xy, int *n);                  /* nn.cpp:188-262 */
int spvo_cpu_sample_descriptors(spvo_cpu *c, const float *desc_nchw, const int32_t *xy, int n, float *out);   /* nn.cpp:366-431 */
/* addStereoImagePair (nn.cpp:449-498) for one image: all of the above; xy [cap][2] floats, desc [cap][256] */
int spvo_cpu_detect(spvo_cpu *c, const uint8_t *img, int rows, int cols, size_t stride, double P[12], float *xy, float *desc, int *n);

/* base.cpp:434-491 (cv::BFMatcher NORM_L2): selector 0 = NN, 1 = KNN; train_idx -1 = no match */
int spvo_cpu_match(spvo_cpu *c, const float *a, int na, const float *b, int nb, int selector, int cross_check, float ratio,
                   int32_t *train_idx, float *distance);

int spvo_cpu_triangulate(const double P_l[12], const double P_r[12], const float *xy_l, const float *xy_r, int n, float *xyz);   /* base.cpp:211-223 */
int spvo_cpu_pnp_ransac(const double K[9], const float *xyz, const float *xy, int n, int iterations, double reproj_error, uint32_t seed,
                        double rvec[3], double tvec[3], int32_t *inliers, int *n_inliers, int *ok);                               /* base.cpp:227-239 */
typedef struct { float X[3]; float uv[2]; int32_t cam; int32_t inverse; } spvo_cpu_obs;   /* = spvo_obs */
typedef struct { int iterations, converged, usable; double initial_cost, final_cost; } spvo_cpu_refine_summary;
int spvo_cpu_pnp_refine(const double P_l[12], const double P_r[12], const spvo_cpu_obs *obs, int n_obs, int max_iterations, double huber_delta,
                        double q[4], double t[3], spvo_cpu_refine_summary *summary);                                              /* base.cpp:282-375 */

/* ---- the whole stereoCallback (visual_odometry_node.cpp:150-262) on the CPU: state machine of hpp:96-178 */
typedef struct {
  float t_detect_ms, t_match_ms, t_solve_ms, t_total_ms;   /* the reference's latency CSV columns (node.cpp:246-258) */
  int n_kp_l, n_kp_r, n_stereo, n_temporal, n_joined, n_inliers, pnp_ok, accepted, refined, lm_iterations;
  double q[4], t[3];                                       /* cam0_curr_T_cam0_prev; identity on the first frame */
} spvo_cpu_step_result;
int spvo_cpu_frontend_reset(spvo_cpu *c, int selector, int cross_check, float stereo_threshold, float min_disparity, int refinement_degree);
int spvo_cpu_frontend_step(spvo_cpu *c, const uint8_t *img_l, const uint8_t *img_r, int rows, int cols, size_t stride,
                           const double P_l[12], const double P_r[12], spvo_cpu_step_result *res);
/* The classic front end of BASELINE config 1 on the CPU (ClassicFeatureFrontEnd with ORB / ORB / BF, classic.cpp:7-120; parameters of
 * launch/visual_odometry_classic.launch): the following spvo_cpu_frontend_step calls detect ORB keypoints at the native resolution, match
 * by Hamming distance and solve as above.  orb_cpu.inc says what is restated and the one thing that cannot be (OpenCV's learned test pairs). */
int spvo_cpu_frontend_reset_classic(spvo_cpu *c, int selector, int cross_check, float stereo_threshold, int refinement_degree);
/* ORB alone: xy [cap][2] (level-0 coordinates), angle_response_octave [cap][3], desc [cap][32]; *n = keypoints found */
int spvo_cpu_orb_tables(float *pattern, float *taps);
int spvo_cpu_orb(spvo_cpu *c, const uint8_t *img, int rows, int cols, size_t stride, float *xy, float *angle_response_octave, uint8_t *desc, int cap, int *n);
/* introspection for the parity tests: maps_of_indices[match_type] of the last step */
int spvo_cpu_frontend_map(spvo_cpu *c, int match_type, int32_t *out, int cap);
/* ... and the keypoints of deque position -4..-1 (prevL, prevR, currL, currR: hpp:66-72); returns their number */
int spvo_cpu_frontend_keypoints(spvo_cpu *c, int position, float *xy, int cap);

#ifdef __cplusplus
}
#endif
#endif
