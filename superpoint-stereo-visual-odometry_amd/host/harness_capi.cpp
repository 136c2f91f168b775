// harness_capi.cpp -- a flat C wrapper around the C++ SuperPointFeatureFrontEnd so that the
// Python tests and bench.py can drive the host class exactly as visual_odometry_node.cpp does
// (node.cpp:150-262).  Test/bench scaffolding only: a ROS build links the class directly.
#include <cstring>

#include "feature_detection.hpp"
#include "vo_io.hpp"

extern "C" {

void *spvo_host_create(const char *models_dir, const char *prefix, const char *machine, int selector_knn, int cross_check, int batch,
                       int height, int width, float conf_thresh, int dist_thresh, int border_remove, float stereo_threshold,
                       float min_disparity, int refinement_degree, int verbose, int precision) {
  SuperPointFeatureFrontEnd::setModelsDir(models_dir ? models_dir : "");
  auto *fe = new SuperPointFeatureFrontEnd(MatcherType::BF, selector_knn ? SelectorType::KNN : SelectorType::NN, cross_check != 0, prefix, batch,
                                           machine, precision == 2 ? TRT_INT8 : precision == 1 ? TRT_FP16 : TRT_FP32, height, width, conf_thresh, dist_thresh, /*num_threads=*/6, border_remove,
                                           stereo_threshold, min_disparity, refinement_degree, verbose != 0);
  return fe;
}

// the class-level options of the host mirror (feature_detection.hpp: setDevice, setMaxKeypoints, setMatchFp8): for the front ends created
// afterwards; a negative value hands the decision back to the environment variable
void spvo_host_set_options(int device, int max_keypoints, int match_fp8) {
  FeatureFrontEnd::setDevice(device);
  SuperPointFeatureFrontEnd::setMaxKeypoints(max_keypoints);
  SuperPointFeatureFrontEnd::setMatchFp8(match_fp8);
}
int spvo_host_max_keypoints(void *h) { return static_cast<SuperPointFeatureFrontEnd *>(h)->maxKeypoints(); }

void spvo_host_destroy(void *h) { delete static_cast<SuperPointFeatureFrontEnd *>(h); }

int spvo_host_engine_loaded(void *h) { return static_cast<SuperPointFeatureFrontEnd *>(h)->engineLoaded() ? 1 : 0; }

const char *spvo_host_last_error(void *h) { return static_cast<SuperPointFeatureFrontEnd *>(h)->lastError().c_str(); }

void spvo_host_set_seed(void *h, unsigned seed) { static_cast<SuperPointFeatureFrontEnd *>(h)->ransac_seed = seed; }

// node.cpp:163-175
void spvo_host_add_stereo_pair(void *h, const uint8_t *img_l, const uint8_t *img_r, int rows, int cols, const double *P_l, const double *P_r) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  cv::Mat l(rows, cols, CV_8UC1), r(rows, cols, CV_8UC1), pl(3, 4, CV_64FC1), pr(3, 4, CV_64FC1);
  std::memcpy(l.data, img_l, (size_t)rows * cols);
  std::memcpy(r.data, img_r, (size_t)rows * cols);
  std::memcpy(pl.data, P_l, 12 * sizeof(double));
  std::memcpy(pr.data, P_r, 12 * sizeof(double));
  fe->addStereoImagePair(l, r, pl, pr);
}

// Host images kept as cv::Mat objects, like the messages cv_bridge hands to the node: created once, then passed by (shallow) header
// copy -- addStereoImagePair replaces the caller's header with the resized image, the stored one keeps the original.
void *spvo_host_make_image(const uint8_t *img, int rows, int cols) {
  auto *m = new cv::Mat(rows, cols, CV_8UC1);
  for (int r = 0; r < rows; ++r) std::memcpy(m->ptr<uint8_t>(r), img + (size_t)r * cols, (size_t)cols);
  return m;
}
void spvo_host_free_image(void *m) { delete static_cast<cv::Mat *>(m); }

void spvo_host_add_stereo_pair_mat(void *h, void *img_l, void *img_r, const double *P_l, const double *P_r) {
  cv::Mat l = *static_cast<cv::Mat *>(img_l), r = *static_cast<cv::Mat *>(img_r), pl(3, 4, CV_64FC1), pr(3, 4, CV_64FC1);
  std::memcpy(pl.data, P_l, 12 * sizeof(double));
  std::memcpy(pr.data, P_r, 12 * sizeof(double));
  static_cast<SuperPointFeatureFrontEnd *>(h)->addStereoImagePair(l, r, pl, pr);
}

void spvo_host_prefetch_mat(void *h, void *img_l, void *img_r) {
  static_cast<SuperPointFeatureFrontEnd *>(h)->prefetchStereoImagePair(*static_cast<cv::Mat *>(img_l), *static_cast<cv::Mat *>(img_r));
}

void spvo_host_add_stereo_pair_dev(void *h, const void *d_l, const void *d_r, int rows, int cols, size_t stride, const double *P_l, const double *P_r,
                                   int host_descriptors) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  cv::Mat pl(3, 4, CV_64FC1), pr(3, 4, CV_64FC1);
  std::memcpy(pl.data, P_l, 12 * sizeof(double));
  std::memcpy(pr.data, P_r, 12 * sizeof(double));
  fe->addStereoImagePairDevice(d_l, d_r, rows, cols, stride, pl, pr, host_descriptors != 0);
}

void spvo_host_prefetch_dev(void *h, const void *d_l, const void *d_r, int rows, int cols, size_t stride) {
  static_cast<SuperPointFeatureFrontEnd *>(h)->prefetchStereoImagePairDevice(d_l, d_r, rows, cols, stride);
}

void *spvo_host_ctx(void *h) { return static_cast<SuperPointFeatureFrontEnd *>(h)->context(); }

void spvo_host_match(void *h, int match_type) { static_cast<SuperPointFeatureFrontEnd *>(h)->matchDescriptors((MatchType)match_type); }

// out: q (x, y, z, w), t of cam0_curr_T_cam0_prev
void spvo_host_solve(void *h, double *q, double *t) {
  tf2::Transform T;
  static_cast<SuperPointFeatureFrontEnd *>(h)->solveStereoOdometry(T);
  q[0] = T.getRotation().x(); q[1] = T.getRotation().y(); q[2] = T.getRotation().z(); q[3] = T.getRotation().w();
  t[0] = T.getOrigin().x(); t[1] = T.getOrigin().y(); t[2] = T.getOrigin().z();
}

// the two halves of spvo_host_solve (solveStereoOdometrySubmit / Collect); both return 1 on success
int spvo_host_solve_submit(void *h) { return static_cast<SuperPointFeatureFrontEnd *>(h)->solveStereoOdometrySubmit() ? 1 : 0; }
int spvo_host_solve_collect(void *h, double *q, double *t) {
  tf2::Transform T;
  if (!static_cast<SuperPointFeatureFrontEnd *>(h)->solveStereoOdometryCollect(T)) return 0;
  q[0] = T.getRotation().x(); q[1] = T.getRotation().y(); q[2] = T.getRotation().z(); q[3] = T.getRotation().w();
  t[0] = T.getOrigin().x(); t[1] = T.getOrigin().y(); t[2] = T.getOrigin().z();
  return 1;
}
int spvo_host_solve_pending(void *h) { return static_cast<SuperPointFeatureFrontEnd *>(h)->solvesPending(); }

void spvo_host_clear(void *h) { static_cast<SuperPointFeatureFrontEnd *>(h)->clearLagecyData(); }

// setDeferredHostCopies (feature_detection.hpp): 0 = images_dq / descriptors_dq are filled inside addStereoImagePair
void spvo_host_set_deferred_copies(void *h, int on) { static_cast<SuperPointFeatureFrontEnd *>(h)->setDeferredHostCopies(on != 0); }

int spvo_host_dq_size(void *h) { return (int)static_cast<SuperPointFeatureFrontEnd *>(h)->keypoints_dq.size(); }

// position: -4..-1 (ImagePosition)
int spvo_host_keypoints(void *h, int position, float *xy, int cap) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  if ((int)fe->keypoints_dq.size() + position < 0) return -1;
  const auto &k = fe->keypoints_dq.end()[position];
  for (int i = 0; i < (int)k.size() && i < cap; ++i) { xy[2 * i] = k[i].pt.x; xy[2 * i + 1] = k[i].pt.y; }
  return (int)k.size();
}

int spvo_host_descriptors(void *h, int position, float *desc, int cap) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  fe->completeHostCopies();   // the harness reads the deque itself (feature_detection.hpp: setDeferredHostCopies)
  if ((int)fe->descriptors_dq.size() + position < 0) return -1;
  const auto &d = fe->descriptors_dq.end()[position];
  const int n = d.rows < cap ? d.rows : cap;
  if (n) std::memcpy(desc, d.data, (size_t)n * 256 * sizeof(float));
  return d.rows;
}

int spvo_host_image(void *h, int position, uint8_t *out, int cap) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  fe->completeHostCopies();   // the harness reads the deque itself (feature_detection.hpp: setDeferredHostCopies)
  if ((int)fe->images_dq.size() + position < 0) return -1;
  const auto &m = fe->images_dq.end()[position];
  if (m.rows * m.cols <= cap) std::memcpy(out, m.data, (size_t)m.rows * m.cols);
  return m.rows * m.cols;
}

// The deques exactly as a reader of the PUBLIC members finds them right now: no completeHostCopies() first (the test of the default
// setDeferredHostCopies(false): images_dq.back() / descriptors_dq.back() are filled when addStereoImagePair returns, nn.cpp:154, 494-498)
int spvo_host_descriptors_raw(void *h, int position, float *desc, int cap) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  if ((int)fe->descriptors_dq.size() + position < 0) return -1;
  const auto &d = fe->descriptors_dq.end()[position];
  const int n = d.rows < cap ? d.rows : cap;
  if (n && d.cols == 256) std::memcpy(desc, d.data, (size_t)n * 256 * sizeof(float));
  return d.rows;
}

int spvo_host_image_raw(void *h, int position, uint8_t *out, int cap) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  if ((int)fe->images_dq.size() + position < 0) return -1;
  const auto &m = fe->images_dq.end()[position];
  if (m.rows * m.cols <= cap && m.data) std::memcpy(out, m.data, (size_t)m.rows * m.cols);
  return m.rows * m.cols;
}

int spvo_host_matches(void *h, int match_type, int *query, int *train, float *dist, int cap) {
  const auto &m = static_cast<SuperPointFeatureFrontEnd *>(h)->cv_DMatches_list[match_type];
  for (int i = 0; i < (int)m.size() && i < cap; ++i) { query[i] = m[i].queryIdx; train[i] = m[i].trainIdx; dist[i] = m[i].distance; }
  return (int)m.size();
}

int spvo_host_map(void *h, int match_type, int *out, int cap) {
  const auto &m = static_cast<SuperPointFeatureFrontEnd *>(h)->mapsOfIndices()[match_type];
  for (int i = 0; i < (int)m.size() && i < cap; ++i) out[i] = m[i];
  return (int)m.size();
}

int spvo_host_inliers(void *h, int which, int *out, int cap) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  const auto &v = which == 0 ? fe->inliersPnp() : fe->inliersPostmatching();
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
}

// ---- result files (vo_io.hpp): pure host code, no GPU
static tf2::Transform make_tf(const double *q, const double *t) {
  tf2::Transform T;
  T.setRotation(tf2::Quaternion(q[0], q[1], q[2], q[3]));
  T.setOrigin(tf2::Vector3(t[0], t[1], t[2]));
  return T;
}

// integrates n front-end outputs (q xyzw, t) and writes the KITTI pose file; returns lines written
int spvo_host_write_kitti(const char *dir, int kitti_eval_id, int seq_start, const double *base_q, const double *base_t, const double *q, const double *t,
                          int n, double *final_pose /* q(4) t(3) of world_T_base */) {
  const tf2::Transform base_T_cam0 = make_tf(base_q, base_t);
  PoseIntegrator integ(base_T_cam0);
  KittiPoseWriter wr(base_T_cam0, seq_start);
  if (!wr.open(dir, kitti_eval_id)) return -1;
  for (int i = 0; i < n; ++i) wr.write(integ.integrate(make_tf(q + 4 * i, t + 3 * i)));
  wr.close();
  const tf2::Transform &P = integ.pose();
  final_pose[0] = P.getRotation().x(); final_pose[1] = P.getRotation().y(); final_pose[2] = P.getRotation().z(); final_pose[3] = P.getRotation().w();
  final_pose[4] = P.getOrigin().x(); final_pose[5] = P.getOrigin().y(); final_pose[6] = P.getOrigin().z();
  return n - (seq_start < n ? seq_start : n);
}

int spvo_host_write_latency(const char *dir, const char *prefix, int batch, int height, int width, const char *precision, int kitti_eval_id,
                            const float *rows, int n, char *name_out, int name_cap) {
  const std::string name = LatencyCsv::fileName(prefix, batch, height, width, precision, kitti_eval_id);
  LatencyCsv csv;
  if (!csv.open(std::string(dir) + "/" + name)) return -1;
  for (int i = 0; i < n; ++i) csv.row(rows[4 * i], rows[4 * i + 1], rows[4 * i + 2], rows[4 * i + 3]);
  csv.close();
  std::strncpy(name_out, name.c_str(), name_cap - 1);
  name_out[name_cap - 1] = 0;
  return n;
}

// constructs the classic front end exactly as visual_odometry_node.cpp:353-360 does and offers it one stereo pair; returns the
// number of deque entries it produced (0 in a build without OpenCV) and the error it logged
int spvo_host_classic_probe(char *err, int cap) {
  ClassicFeatureFrontEnd fe(detector_name_to_type.at("ORB"), descriptor_name_to_type.at("ORB"), matcher_name_to_type.at("BF"),
                            selector_name_to_type.at("KNN"), true, 2.0f, 2.0f, 4, false, 0, 0);
  cv::Mat l(16, 16, CV_8UC1), r(16, 16, CV_8UC1), pl(3, 4, CV_64FC1), pr(3, 4, CV_64FC1);
  fe.addStereoImagePair(l, r, pl, pr);
  std::strncpy(err, fe.lastError().c_str(), cap - 1);
  err[cap - 1] = 0;
  return ClassicFeatureFrontEnd::available() ? (int)fe.keypoints_dq.size() : -(int)fe.keypoints_dq.size();
}

// The classic front end's matching route without its detectors (which need OpenCV): a ClassicFeatureFrontEnd(ORB, ORB, BF,
// selector, cross_check) gets the four feature sets of two stereo frames pushed into its public deques -- binary descriptors,
// `nbytes` per row -- and runs matchDescriptors(match_type) (base.cpp:434-500 with NORM_HAMMING, base.cpp:17-21).
// sets: prevL, prevR, currL, currR; n[4] rows each, desc[i] = n[i] x nbytes.  Writes maps_of_indices[match_type]; returns its size.
int spvo_host_classic_match(int knn, int cross_check, int match_type, const uint8_t *const desc[4], const int n[4], int nbytes, int *map_out, int cap) {
  ClassicFeatureFrontEnd fe(detector_name_to_type.at("ORB"), descriptor_name_to_type.at("ORB"), matcher_name_to_type.at("BF"),
                            selector_name_to_type.at(knn ? "KNN" : "NN"), cross_check != 0, 2.0f, 2.0f, 4, false, 0, 0);
  for (int i = 0; i < 4; ++i) {
    cv::Mat d(n[i], nbytes, CV_8UC1);
    if (n[i]) std::memcpy(d.data, desc[i], (size_t)n[i] * nbytes);
    std::vector<cv::KeyPoint> kp((size_t)n[i]);
    fe.keypoints_dq.push_back(kp);
    fe.descriptors_dq.push_back(d);
    fe.images_dq.push_back(cv::Mat());
  }
  fe.matchDescriptors((MatchType)match_type);
  const std::vector<int> &m = fe.mapsOfIndices().at((MatchType)match_type);
  for (int i = 0; i < (int)m.size() && i < cap; ++i) map_out[i] = m[i];
  return (int)m.size();
}

// stereoCallback (node.cpp:150-262) replayed on a ClassicFeatureFrontEnd(ORB, ORB, BF, selector, cross_check, stereo_threshold, ..)
// exactly as node.cpp:353-360 constructs it, over n stereo pairs in host memory (native resolution, launch/visual_odometry_classic.launch).
// poses: n x 7 (q xyzw, t of cam0_curr_T_cam0_prev; identity for frame 0); stats: n x 4 (keypoints left, right, stereo matches,
// PnP inliers); seconds: wall time of frames warm .. n-1.  Returns the number of frames processed, negative on failure.
int spvo_host_classic_sequence(int n, const uint8_t *const *imgs_l, const uint8_t *const *imgs_r, int rows, int cols, const double *P_l, const double *P_r, int knn,
                               int cross_check, float stereo_threshold, int refinement_degree, int warm, double *poses, int *stats, double *seconds) {
  ClassicFeatureFrontEnd fe(detector_name_to_type.at("ORB"), descriptor_name_to_type.at("ORB"), matcher_name_to_type.at("BF"),
                            selector_name_to_type.at(knn ? "KNN" : "NN"), cross_check != 0, stereo_threshold, stereo_threshold, refinement_degree, false, 0, 0);
  timespec t0{}, t1{};
  for (int k = 0; k < n; ++k) {
    if (k == warm) clock_gettime(CLOCK_MONOTONIC, &t0);
    cv::Mat l(rows, cols, CV_8UC1), r(rows, cols, CV_8UC1), pl(3, 4, CV_64FC1), pr(3, 4, CV_64FC1);
    std::memcpy(l.data, imgs_l[k], (size_t)rows * cols);
    std::memcpy(r.data, imgs_r[k], (size_t)rows * cols);
    std::memcpy(pl.data, P_l, 12 * sizeof(double));
    std::memcpy(pr.data, P_r, 12 * sizeof(double));
    fe.addStereoImagePair(l, r, pl, pr);
    if (fe.keypoints_dq.size() < 2) return -(k + 1);
    fe.matchDescriptors(CURR_LEFT_CURR_RIGHT);
    double *p = poses + 7 * k;
    p[0] = p[1] = p[2] = 0; p[3] = 1; p[4] = p[5] = p[6] = 0;
    int inl = 0;
    if (fe.keypoints_dq.size() >= 4) {
      fe.matchDescriptors(CURR_LEFT_PREV_LEFT);
      tf2::Transform T;
      T.setIdentity();
      fe.solveStereoOdometry(T);
      p[0] = T.getRotation().x(); p[1] = T.getRotation().y(); p[2] = T.getRotation().z(); p[3] = T.getRotation().w();
      p[4] = T.getOrigin().x(); p[5] = T.getOrigin().y(); p[6] = T.getOrigin().z();
      inl = (int)fe.inliersPnp().size();
    }
    if (stats) {
      stats[4 * k] = (int)fe.keypoints_dq.end()[-2].size(); stats[4 * k + 1] = (int)fe.keypoints_dq.end()[-1].size();
      stats[4 * k + 2] = (int)fe.cv_DMatches_list[CURR_LEFT_CURR_RIGHT].size(); stats[4 * k + 3] = inl;
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = n > warm ? (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) : 0.0;
  return n;
}

// bit 0 pnp ok, bit 1 accepted by the gate, bit 2 refinement kept; LM iterations in bits 8..
int spvo_host_last_solve(void *h) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  return (fe->lastPnpOk() ? 1 : 0) | (fe->lastAccepted() ? 2 : 0) | (fe->lastRefined() ? 4 : 0) | (fe->lastLmIterations() << 8);
}

int spvo_host_frame_count(void *h) { return static_cast<SuperPointFeatureFrontEnd *>(h)->frameCount(); }

// ---- a block of stereoCallbacks in one call (bench.py's timed region: no interpreter between two frames, as in the reference's C++ node)
// What one frame of a block leaves behind.  latency_ms: from the FIRST call that handed the pair over (its announcement through
// prefetchStereoImagePairDevice, or addStereoImagePairDevice itself when nothing was announced) to the moment its pose was in
// the caller's hands -- the reference's t_total (visual_odometry_node.cpp:246-258) when pairs are handed over one at a time, and what
// look-ahead and a deferred solve cost on top of it when they are not.
struct SpvoFrameRecord {
  double q[4], t[3];        // cam0_curr_T_cam0_prev (identity for a frame without a pose: the first of a sequence)
  double latency_ms;
  int has_pose, pnp_ok, accepted, refined, lm_iterations, pnp_inliers, stereo_matches, keypoints_left;
};

static double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// n stereoCallbacks (node.cpp:150-262) on device-resident pairs: frame k of the block is pair (first + k) % cycle of d_l / d_r.  depth > 0:
// the next `depth` pairs are announced ahead (prefetchStereoImagePairDevice), before the current one is collected; deferred != 0: a frame's
// solve is handed over (solveStereoOdometrySubmit) and its pose collected while the next frame is processed -- the block's last one before
// the call returns, so that every frame of the block has its record.  Returns the number of frames processed.
int spvo_host_run_device_block(void *h, const void *const *d_l, const void *const *d_r, int cycle, int rows, int cols, size_t stride, const double *P_l,
                               const double *P_r, long first, int n, int depth, int deferred, SpvoFrameRecord *rec) {
  auto *fe = static_cast<SuperPointFeatureFrontEnd *>(h);
  if (!fe || !d_l || !d_r || cycle <= 0 || n < 0 || !rec || depth < 0 || depth > 4) return -1;
  cv::Mat pl(3, 4, CV_64FC1), pr(3, 4, CV_64FC1);
  std::memcpy(pl.data, P_l, 12 * sizeof(double));
  std::memcpy(pr.data, P_r, 12 * sizeof(double));
  constexpr int RINGN = 16;
  // when frame g was first handed over (ring by g % 16: at most five pairs are ever announced and uncollected), and up to which frame
  // announcements have been made: kept across calls -- the last `depth` pairs a block announces are collected by the next block
  static struct { void *h; double t_first[RINGN]; long up_to; } st = {nullptr, {}, -1};
  if (st.h != h || st.up_to < first - 1 || st.up_to > first + depth) { st.h = h; st.up_to = first - 1; }
  double *t_first = st.t_first;
  long &announced_up_to = st.up_to;
  auto record_pose = [&](int k, const tf2::Transform &T, bool ok) {
    SpvoFrameRecord &r = rec[k];
    r.has_pose = ok ? 1 : 0;
    r.q[0] = T.getRotation().x(); r.q[1] = T.getRotation().y(); r.q[2] = T.getRotation().z(); r.q[3] = T.getRotation().w();
    r.t[0] = T.getOrigin().x(); r.t[1] = T.getOrigin().y(); r.t[2] = T.getOrigin().z();
    r.latency_ms = now_ms() - t_first[(first + k) % RINGN];
    r.pnp_ok = fe->lastPnpOk(); r.accepted = fe->lastAccepted(); r.refined = fe->lastRefined(); r.lm_iterations = fe->lastLmIterations();
    r.pnp_inliers = (int)fe->inliersPnp().size();
  };
  const bool collect_first = spvo_get_tuning("solve_collect_first", 0) != 0;
  const int keep = spvo_get_tuning("solve_keep", 1) == 2 ? 2 : 1;
  std::deque<int> pend_q;   // frames of the block whose deferred solves are in flight, oldest first
  for (int k = 0; k < n; ++k) {
    const long g = first + k;
    SpvoFrameRecord &r = rec[k];
    std::memset(&r, 0, sizeof r);
    r.q[3] = 1;
    if (depth > 0)
      for (long a = g; a <= g + depth; ++a) {
        if (a > announced_up_to) { t_first[a % RINGN] = now_ms(); announced_up_to = a; }
        fe->prefetchStereoImagePairDevice(d_l[a % cycle], d_r[a % cycle], rows, cols, stride);   // (no-op if already announced)
      }
    else
      t_first[g % RINGN] = now_ms();
    fe->addStereoImagePairDevice(d_l[g % cycle], d_r[g % cycle], rows, cols, stride, pl, pr, false);
    fe->matchDescriptors(CURR_LEFT_CURR_RIGHT);
    r.keypoints_left = fe->keypoints_dq.size() >= 2 ? (int)fe->keypoints_dq.end()[CURR_LEFT].size() : 0;
    r.stereo_matches = (int)fe->cv_DMatches_list[CURR_LEFT_CURR_RIGHT].size();
    if (fe->keypoints_dq.size() < 4) {   // node.cpp:188-193: the first pair of a sequence has no pose
      r.latency_ms = now_ms() - t_first[g % RINGN];
      continue;
    }
    fe->matchDescriptors(CURR_LEFT_PREV_LEFT);
    tf2::Transform T;
    T.setIdentity();
    if (!deferred) {
      fe->solveStereoOdometry(T);
      record_pose(k, T, true);
      continue;
    }
    // this frame's chain goes out FIRST (it needs nothing of the previous frame's result: feature_detection.hpp), then the previous
    // frame's pose is collected -- the solver's stream always has the next chain queued behind the running one
    auto collect_one = [&]() {
      const bool ok = fe->solveStereoOdometryCollect(T);
      if (!pend_q.empty()) { record_pose(pend_q.front(), T, ok); pend_q.pop_front(); }
    };
    if (collect_first && fe->solvePending()) collect_one();   // (diagnostic, tuning "solve_collect_first": rounds 1-5's order, for A/B runs)
    const bool submitted = fe->solveStereoOdometrySubmit();
    if (submitted) pend_q.push_back(k);
    // `keep` solves stay pending behind a submit (tuning "solve_keep", default 1; 2: the pose of frame k is collected behind the submit of
    // frame k + 2 -- frame k's tail kernel then went out in ONE launch with frame k + 1's hypotheses and is long done: the small engines' setting)
    while (fe->solvesPending() > (submitted ? keep : 0)) collect_one();
  }
  while (fe->solvePending()) {   // every frame of the block has its record when the call returns
    tf2::Transform T;
    T.setIdentity();
    const bool ok = fe->solveStereoOdometryCollect(T);
    if (!pend_q.empty()) { record_pose(pend_q.front(), T, ok); pend_q.pop_front(); }
  }
  return n;
}

}  // extern "C"
