/*
 * spvo.h -- C ABI of the MI355X-native SuperPoint stereo-VO front end.
 *
 * This is the drop-in boundary beneath the reference's C++ `FeatureFrontEnd`
 * (reference: src/odml_visual_odometry/include/odml_visual_odometry/
 * feature_detection.hpp:96-178, "hpp" below).  The reference calls TensorRT,
 * OpenCV and Ceres directly from that class; here the class in
 * superpoint-stereo-visual-odometry_amd/host/ keeps the same public API and
 * forwards every heavy call to the entry points below, which run hand-written
 * gfx950 HIP kernels.  Plain pointers and sizes only; no C++, OpenCV, ROS or
 * torch types.  All functions return 0 on success or a negative spvo_status;
 * nothing throws across this boundary (the reference logs ROS_ERROR and
 * returns: neural_network.cpp:53-55,96-100).  A context is NOT thread-safe and
 * is bound to one HIP device (the reference is single-threaded: node.cpp:445-449).
 *
 * "nn.cpp"   = src/odml_visual_odometry/src/feature_detection_neural_network.cpp
 * "base.cpp" = src/odml_visual_odometry/src/feature_detection_base.cpp
 * "cost.hpp" = src/odml_visual_odometry/include/odml_visual_odometry/ceres_cost_function.hpp
 */
#ifndef SPVO_H
#define SPVO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spvo_ctx spvo_ctx;

typedef enum {
  SPVO_OK = 0,
  SPVO_ERR_INVALID = -1,   /* bad argument / shape */
  SPVO_ERR_DEVICE = -2,    /* HIP runtime error, no gfx950 device, ... */
  SPVO_ERR_IO = -3,        /* weight file missing / malformed (nn.cpp:53-55) */
  SPVO_ERR_STATE = -4,     /* call order (no weights loaded, slot empty, ...) */
  SPVO_ERR_CAPACITY = -5   /* caller buffer too small */
} spvo_status;

/* Constructor arguments of SuperPointFeatureFrontEnd that matter below the
 * class (hpp:272-295); defaults are the reference's laptop launch file
 * (launch/visual_odometry_superpoint.launch:3-26). */
typedef struct {
  int device;          /* HIP device ordinal                                  */
  int net_height;      /* input_height, multiple of 8 (hpp:296)     [360]     */
  int net_width;       /* input_width,  multiple of 8               [1176]    */
  int max_batch;       /* images per network call: 1 or 2 (hpp:342-344) [2]   */
  float conf_thresh;   /* heat > conf_thresh, strict (nn.cpp:203)   [0.015]   */
  int dist_thresh;     /* NMS Chebyshev radius (nn.cpp:246-254)     [4]       */
  int border_remove;   /* nn.cpp:239-244                            [4]       */
  int max_keypoints;   /* hpp:368 (static constexpr 1000)           [1000]    */
  int bug_compat_p;    /* 1: keep the reference's no-op principal-point shift
                          (base.cpp:95,111 writes at<float> into a CV_64F P);
                          0: apply the intended cx/cy -= crop offset.   [1]   */
} spvo_config;

void spvo_default_config(spvo_config *cfg);

int spvo_create(const spvo_config *cfg, spvo_ctx **out);
void spvo_destroy(spvo_ctx *ctx);
const char *spvo_last_error(const spvo_ctx *ctx);   /* ctx may be NULL */

/* Replaces loadTrtEngine (nn.cpp:43-137): load a .spvw plan + fp32 weights
 * (written by spvo/weights.py) and repack them on the device. */
int spvo_load_weights(spvo_ctx *ctx, const char *path);

/* ------------------------------------------------------------------ stages
 * One entry point per reference stage so that each can be parity-checked
 * alone.  Unless a name ends in _dev every pointer is HOST memory and the call
 * is synchronous (copies in, runs on the context's stream, copies out).      */

/* Precision of the loaded engine: 0 = FP32, 1 = FP16 (what engine_generation.py:13-56 selects with trtexec --fp16
 * and the engine file name carries, nn.cpp:44-49), 2 = INT8 (an extension: calibrated activation scales travel in
 * the engine file); SPVO_ERR_STATE (negative) before spvo_load_weights.  FP16 and INT8 engines keep fp32 bindings
 * (nn.cpp:117): every entry point takes and returns the same types. */
int spvo_engine_precision(const spvo_ctx *ctx);

/* Opt-in evaluation mode for FP32 engines loaded AFTER the call (environment default: SPVO_FP32_SPLIT=1): every fp32
 * operand of the convolutions is carried as three bf16 pieces (exact: 3 x 8 = 24 significand bits) and every product is
 * the sum of its six leading partial products on the bf16 matrix pipe with fp32 accumulation
 * (csrc/conv_bf16x3.hip.h).  Results agree with the native fp32 engine to fp32 rounding level; the engine file, the
 * bindings and spvo_engine_precision (0) do not change.  Covers convolution + L2-norm graphs (the VGG SuperPoint of
 * nn.cpp:43-137); other graphs fail at spvo_load_weights. */
int spvo_set_fp32_split(spvo_ctx *ctx, int enable);

/* preprocessImageImpl (base.cpp:68-121) + preprocessImage (nn.cpp:139-161):
 * centre-crop to the network aspect ratio, cv::resize(INTER_LINEAR) 8-bit
 * fixed-point semantics, scale P rows 0-1.  `P` (3x4 row-major, f64) is
 * updated in place.  `resized_u8` (net_height*net_width) may be NULL. */
int spvo_preprocess(spvo_ctx *ctx, const uint8_t *img, int rows, int cols, size_t stride,
                    double P[12], uint8_t *resized_u8);

/* runNeuralNetwork (nn.cpp:163-176): input [batch,1,H,W] f32 in [0,1];
 * det [batch,65,H/8,W/8] NCHW; desc_nhwc [batch,H/8,W/8,256] (unit L2 norm
 * over the last axis; the reference's NCHW `output_desc` transposed, which is
 * what nn.cpp:339-342 computes on the CPU before sampling). Either output may
 * be NULL. */
int spvo_forward(spvo_ctx *ctx, const float *input, int batch, float *det, float *desc_nhwc);

/* Fetch an intermediate activation of the last spvo_forward as dense NCHW
 * [batch,C,h,w]; tensor ids are the plan's (spvo/weights.py). Test hook. */
int spvo_debug_tensor(spvo_ctx *ctx, int tensor_id, int batch, float *out, size_t out_floats);

/* postprocessDetectionAndDescription, detector half (nn.cpp:266-326):
 * det [65,H/8,W/8] -> heat [H,W]. */
int spvo_heatmap(spvo_ctx *ctx, const float *det, float *heat);

/* processOneHeatmap (nn.cpp:188-262): threshold, rank (confidence desc, then
 * column-major pixel index asc), exact greedy NMS, border filter, cap.
 * xy: [max_keypoints][2] int32 (x, y) in rank order; *n: count. */
int spvo_nms(spvo_ctx *ctx, const float *heat, int32_t *xy, int *n);

/* bilinearInterpolationDesc (nn.cpp:366-431) for n keypoints on one image:
 * desc_nhwc [H/8,W/8,256] -> out [n,256] (re-normalised, nn.cpp:428). */
int spvo_sample_descriptors(spvo_ctx *ctx, const float *desc_nhwc, const int32_t *xy, int n,
                            float *out);

typedef struct {
  int n;            /* keypoints found (<= max_keypoints)                      */
  float *xy;        /* caller buffer [max_keypoints][2]: x, y (integers stored
                       as float, like cv::KeyPoint::pt; nn.cpp:243)            */
  float *desc;      /* caller buffer [max_keypoints][256]                      */
} spvo_features;

/* addStereoImagePair (nn.cpp:449-498), everything but the deque bookkeeping:
 * preprocess both images, run the network, post-process.  P_l/P_r are updated
 * in place (nn.cpp:465-466 clones then mutates).  The device copies of the
 * keypoints/descriptors are kept in feature slots `slot_l`/`slot_r` (0..9: the
 * caller's ring of prevL, prevR, currL, currR of hpp:66-72, plus the slots of up to
 * three pairs submitted ahead) for spvo_match_slots.
 * `resized_l`/`resized_r` (net_height*net_width u8, what nn.cpp:154 pushes to
 * images_dq) may be NULL. */
int spvo_detect(spvo_ctx *ctx, const uint8_t *img_l, const uint8_t *img_r, int rows, int cols,
                size_t stride, double P_l[12], double P_r[12], int slot_l, int slot_r,
                spvo_features *out_l, spvo_features *out_r, uint8_t *resized_l,
                uint8_t *resized_r);

/* Same, with both images already resident in device memory (u8, `stride` bytes
 * per row) and no host copies of descriptors (out_*->desc may be NULL).  The context
 * works on its own NON-BLOCKING streams: the images must be complete in device memory
 * when the call is made (it does not order itself behind work the caller queued on the
 * NULL stream or any other stream), and must stay untouched until the call -- for the
 * asynchronous form, the matching spvo_detect_wait -- has returned. */
int spvo_detect_dev(spvo_ctx *ctx, const void *d_img_l, const void *d_img_r, int rows, int cols,
                    size_t stride, double P_l[12], double P_r[12], int slot_l, int slot_r,
                    spvo_features *out_l, spvo_features *out_r);

/* Asynchronous form of spvo_detect_dev: _submit returns as soon as the whole detector chain (and,
 * with spvo_set_prematch, the two standard matches) is enqueued; _wait blocks until the OLDEST
 * submission has finished and hands out what spvo_detect_dev would have.  At most six submissions
 * may be in flight: the post-processing of one then overlaps with the network of the next.
 * Meanwhile the caller may run spvo_match_slots on precomputed matches and
 * spvo_solve_stereo_odometry for pairs already waited for: the ROS node receives the next image
 * pairs while it is still solving the current one.  The slots named here are rewritten while the
 * submission is in flight and must differ from those of other submissions in flight; the temporal
 * partner of a submission is the left slot of the submission before it.  All other entry points
 * that touch the detector's buffers (spvo_detect, spvo_forward, spvo_nms, ...) return
 * SPVO_ERR_STATE while a submission is in flight. */
int spvo_detect_dev_submit(spvo_ctx *ctx, const void *d_img_l, const void *d_img_r, int rows, int cols,
                           size_t stride, int slot_l, int slot_r);
int spvo_detect_wait(spvo_ctx *ctx, double P_l[12], double P_r[12], spvo_features *out_l,
                     spvo_features *out_r);

/* Trunk pairing (extension; off by default): with `on`, a submission whose network would only queue behind an earlier one is HELD
 * until the next submission arrives, and the network then runs for both stereo pairs in one set of launches (four images per layer:
 * every layer's launch, first loads and last stores are paid once per two pairs; 678 instead of 738 us per pair for the VGG fp32
 * forward pass at 360x1176).  Waiting for a held pair launches it alone, so nothing ever blocks; results do not depend on the
 * grouping.  It pays when the caller hands pairs over at least four ahead (each launch then finds its predecessor still running);
 * with fewer it costs throughput, which is why the caller decides.  At most six submissions may be in flight. */
int spvo_set_trunk_pairing(spvo_ctx *ctx, int on);

/* The asynchronous form for images in HOST memory -- what a ROS node holds (cv_bridge::toCvCopy, node.cpp:163-168).
 * _submit copies the two images into pinned staging buffers of the submission (the caller's buffers are free when it
 * returns), queues the host-to-device copies and the whole detector chain behind them and returns; up to six
 * submissions may be in flight, exactly as with spvo_detect_dev_submit (same slot rules).  `extras`: bit 0 = the resized
 * u8 images (nn.cpp:154, images_dq), bit 1 = the descriptors (descriptors_dq) also travel back, into pinned mirrors of
 * the submission.  _collect completes the OLDEST submission (of either kind) like spvo_detect_wait and hands out what was
 * requested: results are bit-identical to spvo_detect on the same images. */
int spvo_detect_submit(spvo_ctx *ctx, const uint8_t *img_l, const uint8_t *img_r, int rows, int cols, size_t stride,
                       int slot_l, int slot_r, int extras);
int spvo_detect_collect(spvo_ctx *ctx, double P_l[12], double P_r[12], spvo_features *out_l, spvo_features *out_r,
                        uint8_t *resized_l, uint8_t *resized_r);

/* spvo_detect_collect without its host copies: completes the OLDEST submission and hands out POINTERS into the submission's
 * pinned host mirrors -- n[i] keypoints, xy[i] (n[i] x 2 floats), desc[i] (n[i] x 256 floats; NULL unless `extras` bit 1 was
 * set at spvo_detect_submit), resized[i] (net_height x net_width u8; NULL unless bit 0 was set), i = 0 left, 1 right.  The
 * kernels that produce these results write them there; a caller that owns the final containers (descriptors_dq / images_dq,
 * nn.cpp:154, 494-498) copies each of them ONCE, when it suits it.  The pointers stay valid until seven more submissions
 * have been made on this context (each submission owns one of eight sets of mirrors) or the context is destroyed. */
typedef struct {
  int n[2];
  const float *xy[2];
  const float *desc[2];
  const uint8_t *resized[2];
  int token;   /* the submission's set of mirrors (spvo_detect_mirrors_wait) */
} spvo_detect_mirrors;
int spvo_detect_collect_mirrors(spvo_ctx *ctx, double P_l[12], double P_r[12], spvo_detect_mirrors *out);
/* n, xy and resized are complete when spvo_detect_collect_mirrors returns.  The descriptors travel to their mirror BESIDE the
 * submission's matches (a copy kernel on a stream of its own: the matcher reads the device copy): desc[] may be read once this
 * call has returned.  A front end that fills descriptors_dq while the solver runs never waits here. */
int spvo_detect_mirrors_wait(spvo_ctx *ctx, const spvo_detect_mirrors *m);

typedef enum { SPVO_SELECT_NN = 0, SPVO_SELECT_KNN = 1 } spvo_selector;

/* matchDescriptors (base.cpp:434-491) = cv::BFMatcher(NORM_L2) match / knnMatch
 * k=2 + ratio test.  For every query row i: train_idx[i] = matched train row or
 * -1 (this is maps_of_indices, base.cpp:483-491) and distance[i] = L2 distance
 * of the matched pair (valid where train_idx[i] >= 0).  NN + cross_check
 * (base.cpp:27-28) is cv::batchDistance's crosscheck as BFMatcher runs it:
 * every train row votes for its nearest query row and a query row keeps the
 * nearest of its voters -- every mutual nearest-neighbour pair plus the pairs
 * that procedure adds; unmatched rows get -1.  KNN keeps i iff
 * d0 < ratio*d1 (base.cpp:469); with nb < 2 nothing is kept. */
int spvo_match(spvo_ctx *ctx, const float *desc_a, int na, const float *desc_b, int nb,
               int selector, int cross_check, float ratio, int32_t *train_idx, float *distance);

/* ------------------------------------------------------- classic front end: ORB
 * detectKeypoints + describeKeypoints of ClassicFeatureFrontEnd for DetectorType::ORB / DescriptorType::ORB
 * (feature_detection_classic.cpp:12-25, 66-68: cv::ORB::create(2000, 1.2f, 8, 31, 0, 2, FAST_SCORE, 31, 20)) on one 8-bit
 * image in host memory: 8-level pyramid, FAST-9 corners with non-maximum suppression, the best `nfeatures` split over the levels,
 * intensity-centroid direction, 256-bit steered BRIEF on the smoothed level.  OpenCV is not available to this build: the algorithm
 * is the published one as restated by oracle/cpu/orb_cpu.inc (its header lists the open choices and the one deviation, the
 * test-pair table) and the kernels reproduce that restatement bit for bit.  Keypoints come level by level, best response first,
 * in level-0 pixel coordinates; `n` receives their number (<= nfeatures), of which min(n, cap) are written. */
typedef struct {
  float x, y;        /* level-0 coordinates                                  */
  float angle;       /* radians, atan2 of the patch's intensity centroid      */
  float response;    /* FAST score                                            */
  int32_t octave;    /* pyramid level                                         */
} spvo_orb_keypoint;
int spvo_orb_detect(spvo_ctx *ctx, const uint8_t *img, int rows, int cols, size_t stride, int nfeatures,
                    spvo_orb_keypoint *keypoints, uint8_t *descriptors /* [cap][32] */, int cap, int *n);
/* the tables the descriptor uses: 256 x (x1, y1, x2, y2) test pairs and the 7 smoothing taps (either may be NULL) */
int spvo_orb_tables(float *pattern /* [1024] */, float *taps /* [7] */);

/* The same for BINARY descriptors: cv::BFMatcher(NORM_HAMMING), what initMatcher (base.cpp:17-21) builds for the ORB / BRISK /
 * AKAZE descriptors of ClassicFeatureFrontEnd (classic.cpp:66-79) and matchDescriptors (base.cpp:434-500) runs on them.
 * Rows of `desc_bytes` bytes (ORB 32, BRISK 64, AKAZE 61; at most 64), distance = number of differing bits (exact), reported
 * as a float like cv::DMatch::distance; selector / cross_check / ratio and the meaning of train_idx / distance as in
 * spvo_match. */
int spvo_match_hamming(spvo_ctx *ctx, const uint8_t *desc_a, int na, const uint8_t *desc_b, int nb, int desc_bytes,
                       int selector, int cross_check, float ratio, int32_t *train_idx, float *distance);

/* Same on the device-resident descriptors of two feature slots. */
int spvo_match_slots(spvo_ctx *ctx, int slot_a, int slot_b, int selector, int cross_check,
                     float ratio, int32_t *train_idx, float *distance);

/* Optional latency hiding for the reference's fixed call order (node.cpp:175-198: detect, then
 * match CURR_LEFT->CURR_RIGHT, then CURR_LEFT->PREV_LEFT): when enabled, spvo_detect* enqueues
 * those two matches (slot_l -> slot_r, slot_l -> the previous call's slot_l) with these selector
 * parameters in the same GPU submission, and spvo_match_slots returns the stored result when it
 * is asked for exactly that match (same slots, same slot contents, same parameters).  Results
 * are identical with it on or off. */
int spvo_set_prematch(spvo_ctx *ctx, int enable, int selector, int cross_check, float ratio);

/* Extension (BASELINE config 5): build the matcher's candidate shortlist with an fp8 (e4m3) distance GEMM instead of
 * the fp32 one.  The GEMM only prunes; the result is EXACT: a first pass re-scores a statistical window canonically, a second
 * pass every column whose rigorous lower bound (from the per-row norms of the fp8 rounding residuals) does not exceed the
 * second canonical distance found -- indices and distances are the brute-force ones on every row (csrc/match.hip.h). */
int spvo_set_match_fp8(spvo_ctx *ctx, int enable);
/* 1 / 0: the fp8 shortlist is on / off in this context (what the host class's setMatchFp8 asked for); negative: error. */
int spvo_get_match_fp8(const spvo_ctx *ctx);

/* cv::triangulatePoints + convertPointsFromHomogeneous (base.cpp:211-223):
 * DLT null vector of the 4x4 system in f64, stored f32, then x/w.
 * xy_l, xy_r: [n][2] f32; xyz: [n][3] f32. */
int spvo_triangulate(spvo_ctx *ctx, const double P_l[12], const double P_r[12], const float *xy_l,
                     const float *xy_r, int n, float *xyz);

/* Deterministic P3P-RANSAC standing in for cv::solvePnPRansac(..., true, 500,
 * 2.0, 0.999, inliers, USAC_ACCURATE) (base.cpp:237-239).  K = P_l[:, :3].
 * rvec/tvec: in = motion prior, out = model.  inliers: caller buffer [n].
 * Returns 0 with *ok = 0 when no model with >= 4 inliers was found. */
typedef struct {
  int iterations;          /* [500]   */
  double reproj_error;     /* [2.0]   */
  double confidence;       /* [0.999] (kept for signature parity; the loop is not
                              terminated early so results do not depend on it) */
  uint32_t seed;           /* sample stream seed [0] */
} spvo_ransac_opts;

int spvo_pnp_ransac(spvo_ctx *ctx, const double K[9], const float *xyz, const float *xy, int n,
                    const spvo_ransac_opts *opts, double rvec[3], double tvec[3],
                    int32_t *inliers, int *n_inliers, int *ok);

/* One residual block of the refinement problem (CostFunctor32, cost.hpp:8-58). */
typedef struct {
  float X[3];          /* 3-D point (cv::Vec3f)                 */
  float uv[2];         /* observation (cv::Point2f)             */
  int32_t cam;         /* 0: P_l, 1: P_r                        */
  int32_t inverse;     /* inverse_transformation_ (cost.hpp:35) */
} spvo_obs;

typedef struct {
  int max_iterations;      /* [40] base.cpp:362 */
  double huber_delta;      /* [1.0] base.cpp:286 */
} spvo_refine_opts;

typedef struct {
  int iterations;
  int converged;           /* termination_type == CONVERGENCE (base.cpp:366-367) */
  int usable;
  double initial_cost, final_cost;
} spvo_refine_summary;

/* ceres::Solve on the CostFunctor32 blocks (base.cpp:282-375): Levenberg-
 * Marquardt with Huber loss, quaternion local parameterisation, f64.
 * q: Eigen coefficient order x,y,z,w (base.cpp:302); in = start, out = result. */
int spvo_pnp_refine(spvo_ctx *ctx, const double P_l[12], const double P_r[12],
                    const spvo_obs *obs, int n_obs, const spvo_refine_opts *opts, double q[4],
                    double t[3], spvo_refine_summary *summary);

/* The numeric body of solveStereoOdometry after the correspondence join (base.cpp:209-375) in ONE
 * submission: triangulate -> RANSAC -> gating (base.cpp:241-272; decided on the host when the results are collected) -> residual blocks in the order
 * of base.cpp:291-356 -> LM refinement -> "not converged => keep the RANSAC pose" (base.cpp:366-374).
 * Identical to calling spvo_triangulate, spvo_pnp_ransac and spvo_pnp_refine in sequence with the
 * host-side glue in between, minus the round trips. */
typedef struct {
  int n;                        /* joined correspondences (base.cpp:156-207)            */
  const float *xy_cl, *xy_cr;   /* [n][2] current left / right                          */
  const float *xy_pl, *xy_pr;   /* [n][2] previous left / right                         */
  const float *prev_xyz;        /* [n][3] previous-frame 3-D point of each correspondence,
                                   or NULL before the first solved frame (base.cpp:323)  */
  const int32_t *prev_valid;    /* [n] 1 where prev_xyz[i] exists (base.cpp:326-332)     */
  double P_l[12], P_r[12];
  double rvec_pred[3], tvec_pred[3];   /* motion prior (hpp:156-157)                     */
  int frame_count;              /* hpp:158                                               */
  int refinement_degree;        /* hpp:143                                               */
  spvo_ransac_opts ransac;
  spvo_refine_opts refine;
  /* Round 6 (both optional, zero = as before): */
  const int32_t *prev_index;    /* [n] instead of prev_xyz / prev_valid: index of each correspondence's previous-frame point among the
                                   points the PREVIOUS spvo_solve_submit of this context triangulated (its xyz output, still on the
                                   device), -1 where there is none (base.cpp:323-332).  The host need not have collected them.   */
  int late_prior;               /* 1: rvec_pred / tvec_pred / frame_count are not known yet (the previous frame's solve is still in
                                   flight) -- they are ignored here and handed to spvo_solve_wait_prior instead.
                                   2: the same, and the chain's LAST kernel is held back: it goes out in one launch with the next
                                   submission's hypotheses (beside which it runs), or alone when this solve is waited for first --
                                   for callers that wait for frame k only after they have submitted frame k + 2 (see below)       */
} spvo_solve_input;

typedef struct {
  double q[4], t[3];            /* cam0_prev_T_cam0_curr to invert (base.cpp:377-385); q = x,y,z,w */
  double rvec[3], tvec[3];      /* pose after gating = the new motion prior when `accepted`       */
  int pnp_ok;                   /* solvePnPRansac's return value                                  */
  int accepted;                 /* do_optmz (base.cpp:243-272)                                    */
  int refined;                  /* refinement ran, converged and was kept                         */
  int n_inliers;
  spvo_refine_summary summary;
} spvo_solve_output;

int spvo_solve_stereo_odometry(spvo_ctx *ctx, const spvo_solve_input *in, spvo_solve_output *out,
                               float *xyz /* [n][3] triangulated points */,
                               int32_t *inliers /* [n] RANSAC inliers, ascending */);

/* The same call in two halves, for a caller with something else to do in between (collecting the next pair's detector
 * output, publishing, bookkeeping).  spvo_solve_submit stages the inputs in pinned memory -- the caller's arrays are free
 * again when it returns -- and enqueues the whole chain on the solver's stream; spvo_solve_wait blocks until the OLDEST
 * pending solve is done and hands out what spvo_solve_stereo_odometry would have.  While a solve is pending the stand-alone
 * solver entry points (spvo_triangulate, spvo_pnp_ransac, spvo_pnp_refine) answer SPVO_ERR_STATE.
 *
 * Up to THREE solves may be pending (round 6).  Nothing the device computes for frame k needs frame k - 1's POSE: the RANSAC's
 * minimal solver is prior-free (as cv::solvePnPRansac's P3P is, base.cpp:237-239), the refinement starts from the RANSAC
 * pose, and the one step that does need the motion prior -- the gate, base.cpp:241-272: three subtractions and a compare --
 * is evaluated by the wait on the host (a submission that carries its prior, late_prior = 0, has it evaluated on the device instead, so
 * that a rejected frame skips its refinement).  What frame k needs of frame k - 1 are its 3-D points (base.cpp:323-332), and
 * `prev_index` refers to them where they lie.  So a caller may submit frame k (late_prior = 1, prev_index) BEFORE it waits
 * for frame k - 1, and hands the prior -- known once k - 1 has been collected -- to spvo_solve_wait_prior.  Results are
 * those of the one-piece call, bit for bit (tests/test_gpu_odometry.py, tests/test_gpu_host.py).  The last kernel of a
 * late_prior = 2 submission's chain (selection, residual blocks, refinement: one workgroup, ~90 us) is held back and goes out in
 * ONE launch with the hypotheses of the next submission, beside which it runs -- or alone, when the solve is waited for
 * first: a caller that waits for frame k only after it has submitted frame k + 2 never waits for the solver's stream, whose
 * work per frame is then max(hypotheses, tail) instead of their sum.  The reference has no counterpart: solveStereoOdometry
 * (base.cpp:125-399) is one blocking call. */
int spvo_solve_submit(spvo_ctx *ctx, const spvo_solve_input *in);
int spvo_solve_wait(spvo_ctx *ctx, spvo_solve_output *out, float *xyz, int32_t *inliers);
int spvo_solve_wait_prior(spvo_ctx *ctx, const double rvec_pred[3], const double tvec_pred[3], int frame_count,
                          spvo_solve_output *out, float *xyz, int32_t *inliers);
int spvo_solve_pending(spvo_ctx *ctx);   /* solves submitted and not waited for yet (0 .. 3) */

/* ------------------------------------------------------- multi-GPU: pose gather
 * The path shards by stereo stream: one process per GPU, each with its own FeatureFrontEnd
 * state; nothing but the resulting relative poses -- q (x, y, z, w) + t of cam0_curr_T_cam0_prev, base.cpp:377-385,
 * 7 doubles -- is ever exchanged.  The communicator is RCCL over xGMI (ncclAllGather on a stream of its own); librccl
 * is opened when the first communicator is created, not when this library is loaded.  Errors: spvo_last_error(NULL). */
typedef struct spvo_comm spvo_comm;
#define SPVO_COMM_ID_BYTES 128

/* Rank 0 creates the id (ncclGetUniqueId) and hands it to the other ranks out of band (ROS parameter server, a file,
 * MPI, torchrun's store); every rank then calls spvo_comm_create -- collectively, like ncclCommInitRank. */
int spvo_comm_unique_id(unsigned char id[SPVO_COMM_ID_BYTES]);
/* SPVO_OK when librccl can be opened and has the entry points this library uses (dlopen + dlsym: no bootstrap state is created,
 * unlike spvo_comm_unique_id): what every rank checks BEFORE the ranks enter spvo_comm_create together.  A failure INSIDE
 * ncclCommInitRank (a peer that died after this check) is not recoverable collectively: the launcher's job (bench.py: spawn_ranks /
 * torch.distributed.run stop the job when a rank dies). */
int spvo_comm_available(void);
int spvo_comm_create(int device, int rank, int world, const unsigned char id[SPVO_COMM_ID_BYTES], spvo_comm **out);
/* TEST transport, no GPU: ranks exchange through files in `dir` (world-size-2 CPU tests of the N > 1 code path). */
int spvo_comm_create_host(const char *dir, int rank, int world, spvo_comm **out);
int spvo_comm_rank(const spvo_comm *comm);
int spvo_comm_world(const spvo_comm *comm);
void spvo_comm_destroy(spvo_comm *comm);

/* all[r][7] = rank r's pose.  Collective and synchronous (host memory in, host memory out). */
int spvo_pose_allgather(spvo_comm *comm, const double pose[7], double *all /* [world][7] */);
/* Batched form: n poses per rank (the same n on every rank) in ONE collective; all[r][i][7].  56 bytes per frame are
 * pure latency, so a throughput-oriented caller gathers once per batch of frames. */
int spvo_pose_allgather_n(spvo_comm *comm, const double *poses /* [n][7] */, int n, double *all /* [world][n][7] */);

/* ---------------------------------------------------------------- plumbing */
void *spvo_stream(spvo_ctx *ctx);                 /* hipStream_t of the context */
int spvo_synchronize(spvo_ctx *ctx);

/* Per-stage HIP-event timing on the context's stream (bench.py's roofline leg).
 * Stage names: "conv:<op index>", "net", "post", "match", ...; see DESIGN.md. */
int spvo_profile_enable(spvo_ctx *ctx, int on);
int spvo_profile_reset(spvo_ctx *ctx);
/* Restrict the timing to ONE stage (e.g. "conv:1"); NULL or "" = every stage again.  Two event records per step
 * instead of two per kernel: the way to time a kernel inside a throughput measurement without slowing it down. */
int spvo_profile_only(spvo_ctx *ctx, const char *stage);
int spvo_profile_count(spvo_ctx *ctx);
int spvo_profile_get(spvo_ctx *ctx, int i, char *name, size_t name_cap, double *total_ms,
                     long long *calls, double *flops_per_call, double *bytes_per_call);
/* Which kernel family the loaded engine runs a "conv:<op index>" stage on ("conv_wino4_kernel", "conv_wino2_kernel",
 * "conv_mfma_kernel", "conv_f16_kernel", ...), and how many multiply-adds the matrix pipe executes per multiply-add of the
 * direct convolution (Winograd F(4x4,3x3): 0.25, F(2x2,3x3): 4/9, split bf16x3 mode: 6, otherwise 1). */
int spvo_profile_stage_kernel(spvo_ctx *ctx, const char *stage, char *name, size_t name_cap, double *executed_per_algorithmic);

/* ------------------------------------------------------- diagnostic switches (A/B measurements, parity debugging)
 * Which kernels / streams an engine uses is decided by the library from the plan and the sizes.  The decisions can be overridden
 * for measurements and tests -- through THIS call only: the library reads no environment variable for them, so the environment
 * of the process that hosts it (a ROS node) cannot change kernels by accident.  Process-wide; a value takes effect for contexts
 * created / engines loaded afterwards.  `name` is one of the names INTEGRATION.md lists ("winograd", "wino4", "wino_narrow",
 * "wino_dynamic", "winograd_min_tiles", "wino4_min_tiles", "merge_siblings", "heads_fused", "heads_on_net", "heads_split",
 * "match_fused", "fp32_split", "prematch", "spin_wait", "trunk_timing", "solve_timing", "nms_first", "upload_side"); SPVO_ERR_INVALID for any other.
 * spvo_get_tuning returns the value set, or `dflt`; spvo_clear_tuning forgets every value. */
int spvo_set_tuning(const char *name, int value);
int spvo_get_tuning(const char *name, int dflt);
void spvo_clear_tuning(void);

#ifdef __cplusplus
}
#endif
#endif /* SPVO_H */
