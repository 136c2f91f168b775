// feature_detection.cpp -- host side of the front end: the bookkeeping the reference does in
//   src/odml_visual_odometry/src/feature_detection_base.cpp            ("base.cpp")
//   src/odml_visual_odometry/src/feature_detection_neural_network.cpp  ("nn.cpp")
// with every numeric step forwarded to the C ABI (include/spvo.h).  Error convention as in
// the reference: log and return, never throw (nn.cpp:53-55, 96-100, 490-491).
#include "feature_detection.hpp"
#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>

// ------------------------------------------------------------------------- tf2lite
#ifndef SPVO_USE_OPENCV
namespace tf2lite {
double Vector3::length() const { return std::sqrt(x() * x() + y() * y() + z() * z()); }

static void rotate(const Quaternion &q, const double v[3], double out[3]) {
  // unit quaternion rotation, same polynomial as Eigen's toRotationMatrix
  const double x = q.x(), y = q.y(), z = q.z(), w = q.w();
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  out[0] = (1 - (tyy + tzz)) * v[0] + (txy - twz) * v[1] + (txz + twy) * v[2];
  out[1] = (txy + twz) * v[0] + (1 - (txx + tzz)) * v[1] + (tyz - twx) * v[2];
  out[2] = (txz - twy) * v[0] + (tyz + twx) * v[1] + (1 - (txx + tyy)) * v[2];
}

Transform Transform::inverse() const {
  Transform r;
  const double n = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z() + q.w() * q.w());
  r.q = Quaternion(-q.x() / n, -q.y() / n, -q.z() / n, q.w() / n);
  const double v[3] = {t.x(), t.y(), t.z()};
  double o[3];
  rotate(r.q, v, o);
  r.t = Vector3(-o[0], -o[1], -o[2]);
  return r;
}

Transform Transform::operator*(const Transform &o) const {
  Transform r;
  r.q = Quaternion(q.w() * o.q.x() + q.x() * o.q.w() + q.y() * o.q.z() - q.z() * o.q.y(),
                   q.w() * o.q.y() - q.x() * o.q.z() + q.y() * o.q.w() + q.z() * o.q.x(),
                   q.w() * o.q.z() + q.x() * o.q.y() - q.y() * o.q.x() + q.z() * o.q.w(),
                   q.w() * o.q.w() - q.x() * o.q.x() - q.y() * o.q.y() - q.z() * o.q.z());
  const double v[3] = {o.t.x(), o.t.y(), o.t.z()};
  double rv[3];
  rotate(q, v, rv);
  r.t = Vector3(rv[0] + t.x(), rv[1] + t.y(), rv[2] + t.z());
  return r;
}
}  // namespace tf2lite
#endif

// ------------------------------------------------------------------------- logging
void FeatureFrontEnd::logError(const std::string &msg) {
  last_error_ = msg;
  static const bool quiet = std::getenv("SPVO_QUIET") != nullptr && std::strcmp(std::getenv("SPVO_QUIET"), "0") != 0;   // benchmarks with untrained weights trip the gating message every frame
  if (!quiet) std::fprintf(stderr, "[ERROR] %s\n", msg.c_str());  // ROS_ERROR stand-in
}
void FeatureFrontEnd::logInfo(const std::string &msg) const {
  if (verbose_) std::fprintf(stderr, "[ INFO] %s\n", msg.c_str());
}

// ------------------------------------------------------------------------- base.cpp:10-33
void FeatureFrontEnd::initMatcher() {
  if (matcher_type_ == MatcherType::BF) {
    // cv::BFMatcher::create(norm_type, cross_check_ & (selector_type_ != KNN))   base.cpp:27-28
    matcher_cross_check_ = cross_check_ && (selector_type_ != SelectorType::KNN);
    // NORM_HAMMING for the binary descriptors of the classic front end (ORB / BRISK / AKAZE, base.cpp:17-21): spvo_match_hamming
    // on the host matrices of descriptors_dq; NORM_L2 (SuperPoint / SIFT): spvo_match_slots on the device-resident feature slots
    matcher_hamming_ = descriptor_type_ != DescriptorType::SIFT && descriptor_type_ != DescriptorType::SuperPoint;
    matcher_ready_ = true;
  } else {
    logError("[initMatcher] FLANN matcher is not implemented on the GPU path (base.cpp:29-32); use BF");
  }
}

static int g_device = -1;   // FeatureFrontEnd::setDevice
void FeatureFrontEnd::setDevice(int device) { g_device = device; }
static int configured_device() {
  if (g_device >= 0) return g_device;
  if (const char *dev = std::getenv("SPVO_DEVICE")) return std::atoi(dev);
  return 0;
}

bool FeatureFrontEnd::ensureContext() {
  if (ctx_) return true;
  spvo_config cfg;
  spvo_default_config(&cfg);
  cfg.device = configured_device();
  solve_timing_ = spvo_get_tuning("solve_timing", 0) != 0;
  solve_hold_tail_ = spvo_get_tuning("solve_keep", 1) == 2;
  if (input_height_ > 0 && input_width_ > 0) {
    cfg.net_height = (input_height_ + 7) / 8 * 8;   // only the pre-processing geometry matters to a context without an engine
    cfg.net_width = (input_width_ + 7) / 8 * 8;
  }
  if (const char *bc = std::getenv("SPVO_BUG_COMPAT_P")) cfg.bug_compat_p = std::atoi(bc);
  if (spvo_create(&cfg, &ctx_) != SPVO_OK) {
    logError(std::string("spvo_create: ") + spvo_last_error(nullptr));
    ctx_ = nullptr;
    return false;
  }
  return true;
}

// ------------------------------------------------------------------------- base.cpp:68-121
void FeatureFrontEnd::preprocessImageImpl(cv::Mat &img, cv::Mat &projection_matrix) {
  if (img.type() != CV_8UC1 || projection_matrix.type() != CV_64FC1 || projection_matrix.rows != 3 || projection_matrix.cols != 4) {
    logError("preprocessImageImpl: expected a CV_8UC1 image and a 3x4 CV_64F projection matrix (node.cpp:91,163-168)");
    return;
  }
  if (input_height_ <= 0 || input_width_ <= 0 || input_height_ % 8 || input_width_ % 8) {
    logError("preprocessImageImpl: the target size must be a positive multiple of 8 (hpp:296)");
    return;
  }
  if (!ensureContext()) return;
  cv::Mat out(input_height_, input_width_, CV_8UC1);
  if (spvo_preprocess(ctx_, img.data, img.rows, img.cols, (size_t)img.step, projection_matrix.ptr<double>(0), out.data) != SPVO_OK) {
    logError(std::string("spvo_preprocess: ") + spvo_last_error(ctx_));
    return;
  }
  img = out;   // the reference crops and resizes the caller's image in place (base.cpp:89,105,115)
}

// ------------------------------------------------------------------------- base.cpp:35-66
void FeatureFrontEnd::clearLagecyData() {
  while (!solve_q_.empty() && ctx_) {   // solves handed over and never collected: complete them, their results are dropped with the rest
    spvo_solve_output so;
    std::vector<float> xyz((size_t)std::max(solve_q_.front().n, 1) * 3);
    std::vector<int32_t> inl(std::max(solve_q_.front().n, 1));
    const double zero[3] = {0, 0, 0};
    (void)spvo_solve_wait_prior(ctx_, zero, zero, 0, &so, xyz.data(), inl.data());
    solve_q_.pop_front();
  }
  solve_q_.clear();
  prev_points_on_device_ = false;
  completeHostCopies();
  images_dq.clear();
  keypoints_dq.clear();
  descriptors_dq.clear();
  slots_dq_.clear();
  for (auto &m : cv_DMatches_list) m.clear();
  projection_matrix_l_.release();
  projection_matrix_r_.release();
  for (int k = 0; k < 3; ++k) r_vec_pred[k] = t_vec_pred[k] = 0;
  frame_count = 0;
  for (auto &m : maps_of_indices) m.clear();
  prev_left_points_3d_inited = false;
  inliers_postmatching.clear();
  inliers_pnp.clear();
  prev_left_points_3d.clear();
  map_from_prev_left_matched_to_prev_valid_index.clear();
  map_from_curr_valid_to_prev_left_matched_index.clear();
  map_from_curr_left_matched_to_curr_valid_index.clear();
}

// ------------------------------------------------------------------------- base.cpp:434-500
void FeatureFrontEnd::matchDescriptors(const MatchType match_type) {
  const int p0 = match_type_to_positions[match_type].first, p1 = match_type_to_positions[match_type].second;
  const int n_dq = (int)keypoints_dq.size();
  if (n_dq + p0 < 0 || n_dq + p1 < 0) {
    logError("matchDescriptors: not enough frames for " + MatchType_str[match_type]);
    return;
  }
  const std::vector<cv::KeyPoint> &keypoints0 = keypoints_dq.end()[p0];
  if (matcher_hamming_ && !ensureContext()) return;
  if (!ctx_ || !matcher_ready_ || (!matcher_hamming_ && slots_dq_.size() != keypoints_dq.size())) {
    logError("matchDescriptors: front end not initialised");
    return;
  }
  if (descriptors_dq.end()[p0].rows < 10) std::fprintf(stderr, "[ WARN] descriptors0.rows == %d < 10\n", descriptors_dq.end()[p0].rows);
  if (descriptors_dq.end()[p1].rows < 10) std::fprintf(stderr, "[ WARN] descriptors1.rows == %d < 10\n", descriptors_dq.end()[p1].rows);

  std::vector<cv::DMatch> &cv_Dmatches = cv_DMatches_list[match_type];
  cv_Dmatches.clear();
  const int n0 = (int)keypoints0.size();
  std::vector<int32_t> train(std::max(n0, 1), -1);
  std::vector<float> dist(std::max(n0, 1), 0.f);
  const int sel = selector_type_ == SelectorType::KNN ? SPVO_SELECT_KNN : SPVO_SELECT_NN;
  int rc;
  if (matcher_hamming_) {   // binary descriptors: host matrices (CV_8U, one row per keypoint), packed row by row if they are views
    const cv::Mat &d0 = descriptors_dq.end()[p0], &d1 = descriptors_dq.end()[p1];
    const int nbytes = d0.rows ? d0.cols : d1.cols;
    auto rows_of = [nbytes](const cv::Mat &m, std::vector<uint8_t> &buf) -> const uint8_t * {
      if (m.rows == 0) return nullptr;
      if ((size_t)m.step == (size_t)nbytes) return m.ptr<uint8_t>(0);
      buf.resize((size_t)m.rows * nbytes);
      for (int r = 0; r < m.rows; ++r) std::memcpy(buf.data() + (size_t)r * nbytes, m.ptr<uint8_t>(r), nbytes);
      return buf.data();
    };
    std::vector<uint8_t> b0, b1;
    if ((d0.rows && d0.depth() != CV_8U) || (d1.rows && d1.depth() != CV_8U) || (d0.rows && d1.rows && d0.cols != d1.cols) || d0.rows != n0) {
      logError("matchDescriptors: binary descriptors expected (CV_8U, one row per keypoint, equal widths)");
      return;
    }
    rc = spvo_match_hamming(ctx_, rows_of(d0, b0), d0.rows, rows_of(d1, b1), d1.rows, nbytes, sel, matcher_cross_check_ ? 1 : 0, knn_threshold_, train.data(), dist.data());
  } else {
    completeImageCopies();   // the GPU is still matching (spvo_match_slots waits for it): images_dq's share of the deferred copies fits here
    rc = spvo_match_slots(ctx_, slots_dq_.end()[p0], slots_dq_.end()[p1], sel, matcher_cross_check_ ? 1 : 0, knn_threshold_, train.data(), dist.data());
  }
  if (rc != SPVO_OK) {
    logError(std::string("matchDescriptors: ") + spvo_last_error(ctx_));
    return;
  }
  for (int i = 0; i < n0; ++i)
    if (train[i] >= 0) {
      cv::DMatch m;
      m.queryIdx = i; m.trainIdx = train[i]; m.imgIdx = 0; m.distance = dist[i];
      cv_Dmatches.push_back(m);
    }

  if (match_type == MatchType::CURR_LEFT_CURR_RIGHT)  // base.cpp:475-481
    maps_of_indices[MatchType::PREV_LEFT_PREV_RIGHT] = maps_of_indices[MatchType::CURR_LEFT_CURR_RIGHT];

  std::vector<int> &map = maps_of_indices.at(match_type);
  map.clear();
  map.resize(keypoints0.size(), -1);
  for (const cv::DMatch &m : cv_Dmatches) map.at(m.queryIdx) = m.trainIdx;

  if (verbose_) logInfo(std::to_string(cv_Dmatches.size()) + " matches for " + MatchType_str[match_type]);
  if (cv_Dmatches.size() < 10) std::fprintf(stderr, "[ WARN] %zu matches < 10 for %s\n", cv_Dmatches.size(), MatchType_str[match_type].c_str());
}

// ------------------------------------------------------------------------- base.cpp:125-399
static double host_now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }

void FeatureFrontEnd::solveStereoOdometry(tf2::Transform &cam0_curr_T_cam0_prev) {
  if (solveStereoOdometrySubmit()) solveStereoOdometryCollect(cam0_curr_T_cam0_prev);   // (Collect fills the deques while the kernels run)
  else completeHostCopies();
}

// First half (extension): the correspondence join (base.cpp:127-207) and the hand-over of the numeric part to spvo_solve_submit.
// Since round 6 this half needs NOTHING of the previous frame's result: its 3-D points are referred to by index where they lie on the
// device (spvo_solve_input::prev_index), the motion prior and the frame count are only needed by the gate, which the second half
// evaluates (spvo_solve_wait_prior).  So the next frames may be submitted BEFORE this one has been collected: up to three solves in flight,
// collected oldest first (with two kept pending the library runs a frame's hypotheses and the previous frame's tail kernel as ONE launch).  The join's own maps (base.cpp:388-392) roll here; pose, prior, frame count and the points' host copy roll in
// solveStereoOdometryCollect.
bool FeatureFrontEnd::solveStereoOdometrySubmit() {
  const bool timing = solve_timing_;   // diagnostic (spvo_set_tuning "solve_timing", read when this front end created its context)
  double *acc = solve_timing_acc_;
  long &calls = solve_timing_calls_;
  const double th0 = timing ? host_now_us() : 0;
  if (solve_q_.size() >= 3) {
    logError("solveStereoOdometrySubmit: three solves are in flight, collect the oldest one first");
    return false;
  }
  if (keypoints_dq.size() < 4 || !ensureContext()) {
    logError("solveStereoOdometry needs two stereo frames");
    return false;
  }
  const std::vector<cv::DMatch> &stereo = cv_DMatches_list[CURR_LEFT_CURR_RIGHT];
  const size_t num_curr_stereo = stereo.size();
  std::vector<float> kp_cl, kp_cr, kp_pl, kp_pr;  // n x 2 each
  kp_cl.reserve(2 * num_curr_stereo); kp_cr.reserve(2 * num_curr_stereo);
  kp_pl.reserve(2 * num_curr_stereo); kp_pr.reserve(2 * num_curr_stereo);
  inliers_postmatching.clear();
  inliers_postmatching.reserve(num_curr_stereo);

  if (refinement_degree_ >= 3) {
    map_from_curr_left_matched_to_curr_valid_index.clear();
    map_from_curr_left_matched_to_curr_valid_index.resize(keypoints_dq.end()[CURR_LEFT].size(), -1);
    map_from_curr_valid_to_prev_left_matched_index.clear();
    map_from_curr_valid_to_prev_left_matched_index.reserve(num_curr_stereo);
  }
  const auto &map_temporal = maps_of_indices[CURR_LEFT_PREV_LEFT];
  const auto &map_prev = maps_of_indices[PREV_LEFT_PREV_RIGHT];
  // the reference indexes with .at(): out-of-range would throw there; here it is reported
  for (const auto &m : stereo) {
    const int i_cl = m.queryIdx;
    if (i_cl >= (int)map_temporal.size()) { logError("maps_of_indices out of range"); return false; }
    if (map_temporal[i_cl] == -1) continue;
    const cv::Point2f &a = keypoints_dq.end()[CURR_LEFT][i_cl].pt;
    const cv::Point2f &b = keypoints_dq.end()[CURR_RIGHT][m.trainIdx].pt;
    if (std::abs(a.y - b.y) > stereo_threshold_ || std::abs(a.x - b.x) < min_disparity_) continue;  // base.cpp:169-172
    const int i_pl = map_temporal[i_cl];
    if (i_pl >= (int)map_prev.size()) { logError("maps_of_indices out of range"); return false; }
    if (map_prev[i_pl] == -1) continue;
    const cv::Point2f &c = keypoints_dq.end()[PREV_LEFT][i_pl].pt;
    const cv::Point2f &d = keypoints_dq.end()[PREV_RIGHT][map_prev[i_pl]].pt;
    kp_cl.push_back(a.x); kp_cl.push_back(a.y);
    kp_cr.push_back(b.x); kp_cr.push_back(b.y);
    kp_pl.push_back(c.x); kp_pl.push_back(c.y);
    kp_pr.push_back(d.x); kp_pr.push_back(d.y);
    inliers_postmatching.push_back(i_cl);
    if (refinement_degree_ >= 3) {
      map_from_curr_left_matched_to_curr_valid_index[i_cl] = (int)(kp_cl.size() / 2) - 1;
      map_from_curr_valid_to_prev_left_matched_index.push_back(i_pl);
    }
  }
  const int n = (int)(kp_cl.size() / 2);

  // everything numeric from here to the refined pose is ONE call into the C ABI:
  // triangulation (base.cpp:211-223), PnP-RANSAC (227-239), gating (241-272), residual blocks
  // (291-356), refinement and the "not converged => keep RANSAC" rule (358-375)
  const double *Pl = projection_matrix_l_.ptr<double>(0), *Pr = projection_matrix_r_.ptr<double>(0);
  std::vector<int32_t> prev_index;
  if (refinement_degree_ >= 3 && prev_points_on_device_) {  // base.cpp:323-332: the previous frame's 3-D point of each correspondence, by index
    prev_index.assign(std::max(n, 1), -1);
    for (int vi = 0; vi < n; ++vi) {
      const int matched_prev = map_from_curr_valid_to_prev_left_matched_index[vi];
      if (matched_prev < 0 || matched_prev >= (int)map_from_prev_left_matched_to_prev_valid_index.size()) continue;
      prev_index[vi] = map_from_prev_left_matched_to_prev_valid_index[matched_prev];   // (-1: that keypoint had no valid correspondence then)
    }
  }
  spvo_solve_input si;
  std::memset(&si, 0, sizeof si);
  si.n = n;
  si.xy_cl = kp_cl.data(); si.xy_cr = kp_cr.data(); si.xy_pl = kp_pl.data(); si.xy_pr = kp_pr.data();
  si.prev_index = prev_index.empty() ? nullptr : prev_index.data();
  // r_vec_pred / t_vec_pred / frame_count: with an earlier solve still in flight they are not final yet and are handed over by
  // solveStereoOdometryCollect (late prior: the gate is evaluated there); with none in flight -- the synchronous call sequence -- they go along
  // now and the device gates (a frame the gate rejects then skips its refinement).  Same decision, same results either way.
  si.late_prior = solve_q_.empty() ? 0 : (solve_hold_tail_ ? 2 : 1);   // (2: this frame's tail kernel goes out with the next frame's hypotheses, include/spvo.h)
  for (int k = 0; k < 3; ++k) { si.rvec_pred[k] = r_vec_pred[k]; si.tvec_pred[k] = t_vec_pred[k]; }
  si.frame_count = frame_count;
  for (int k = 0; k < 12; ++k) { si.P_l[k] = Pl[k]; si.P_r[k] = Pr[k]; }
  si.refinement_degree = refinement_degree_;
  si.ransac = spvo_ransac_opts{500, 2.0, 0.999, ransac_seed};   // base.cpp:239
  si.refine = spvo_refine_opts{40, 1.0};                        // base.cpp:286, 362
  const double th1 = timing ? host_now_us() : 0;
  const int solve_rc = spvo_solve_submit(ctx_, &si);   // the inputs are staged: the vectors above may go
  if (solve_rc != SPVO_OK) {
    logError(std::string("spvo_solve_submit: ") + spvo_last_error(ctx_));
    return false;
  }
  solve_q_.emplace_back();
  solve_q_.back().n = n;
  solve_q_.back().inliers_postmatching = inliers_postmatching;
  if (refinement_degree_ >= 3) {  // base.cpp:388-392 (the maps of the join; the points themselves stay on the device for the next frame's prev_index)
    map_from_prev_left_matched_to_prev_valid_index = map_from_curr_left_matched_to_curr_valid_index;
    prev_points_on_device_ = true;
  }
  if (timing) {
    const double th2 = host_now_us();
    acc[0] += th1 - th0; acc[1] += th2 - th1;
    if (++calls % 200 == 0) {
      std::fprintf(stderr, "[host solve timing] join %.1f us, submit %.1f us\n", acc[0] / 200, acc[1] / 200);
      acc[0] = acc[1] = 0;
    }
  }
  return true;
}

// Second half (extension): waits for the solve, then base.cpp:241-272 (prior update), 377-396 (output, state roll)
bool FeatureFrontEnd::solveStereoOdometryCollect(tf2::Transform &cam0_curr_T_cam0_prev) {
  if (solve_q_.empty()) {
    logError("solveStereoOdometryCollect: no solve pending");
    return false;
  }
  const int n = solve_q_.front().n;
  inliers_postmatching.swap(solve_q_.front().inliers_postmatching);   // (introspection: the frame whose pose this call returns)
  solve_q_.pop_front();
  std::vector<float> &pts3d = solve_pts3d_;
  pts3d.assign((size_t)std::max(n, 1) * 3, 0.f);
  inliers_pnp.assign(std::max(n, 1), 0);
  spvo_solve_output so;
  completeHostCopies();   // the solver's kernels are running: the bulk copies into images_dq / descriptors_dq cost nothing here
  // the gate (base.cpp:241-272) is evaluated in here, against the prior and the frame count as they stand NOW (every earlier frame collected)
  const int solve_rc = spvo_solve_wait_prior(ctx_, r_vec_pred, t_vec_pred, frame_count, &so, pts3d.data(), inliers_pnp.data());
  if (solve_rc != SPVO_OK) {
    logError(std::string("spvo_solve_wait_prior: ") + spvo_last_error(ctx_));
    return false;
  }
  inliers_pnp.resize(so.n_inliers);
  last_pnp_ok_ = so.pnp_ok != 0; last_accepted_ = so.accepted != 0; last_refined_ = so.refined != 0; last_lm_iterations_ = so.summary.iterations;
  if (!so.pnp_ok) logError("solvePnPRansac failed! Identity transformation will be applied.");
  else if (!so.accepted) logError("solvePnPRansac succeeded but acceleration is abnormally large!");
  else for (int k = 0; k < 3; ++k) { r_vec_pred[k] = so.rvec[k]; t_vec_pred[k] = so.tvec[k]; }   // base.cpp:269-270
  if (so.accepted && refinement_degree_ > 0 && !so.refined) logError("summary.IsSolutionUsable() == false or NOT CONVERGENT");
  const double q_opt[4] = {so.q[0], so.q[1], so.q[2], so.q[3]};
  const double t_opt[3] = {so.t[0], so.t[1], so.t[2]};

  tf2::Transform cam0_prev_T_cam0_curr;
  cam0_prev_T_cam0_curr.setRotation(tf2::Quaternion(q_opt[0], q_opt[1], q_opt[2], q_opt[3]));
  cam0_prev_T_cam0_curr.setOrigin(tf2::Vector3(t_opt[0], t_opt[1], t_opt[2]));
  cam0_curr_T_cam0_prev = cam0_prev_T_cam0_curr.inverse();

  if (refinement_degree_ >= 3) {  // base.cpp:393-394 (host copy of the points; the index maps rolled at the submit)
    prev_left_points_3d.assign(pts3d.begin(), pts3d.begin() + (size_t)n * 3);
    prev_left_points_3d_inited = true;
  }
  ++frame_count;
  return true;
}

cv::Mat FeatureFrontEnd::visualizeMatches(const MatchType match_type) {
  completeHostCopies();
  if (images_dq.size() < 4) return cv::Mat();
  return images_dq.end()[match_type_to_positions[match_type].second].clone();
}

cv::Mat FeatureFrontEnd::visualizeInliers(const ImagePosition image_position) {
  if (image_position != CURR_LEFT) logError("inlier visualization for " + ImagePosition_str.at(image_position) + " is not implemented yet");
  completeHostCopies();
  if (images_dq.empty()) return cv::Mat();
  return images_dq.end()[image_position].clone();
}

// ------------------------------------------------------------------------- classic front end (classic.cpp)
// The reference's constructor hands `stereo_threshold` to the base class a second time in the place of `min_disparity`
// (hpp:203-206), so the classic launch file's min_disparity is never used; kept, because the stereo gate of
// solveStereoOdometry (base.cpp:169-172) then behaves as the reference's does.
ClassicFeatureFrontEnd::ClassicFeatureFrontEnd()
    : ClassicFeatureFrontEnd(DetectorType::ShiTomasi, DescriptorType::ORB, MatcherType::BF, SelectorType::NN, true, 2.0f, 1.0f, 4, true, 120, 392) {}

ClassicFeatureFrontEnd::ClassicFeatureFrontEnd(const DetectorType detector_type, const DescriptorType descriptor_type, const MatcherType matcher_type,
                                               const SelectorType selector_type, const bool cross_check, const float stereo_threshold,
                                               const float /*min_disparity*/, const int refinement_degree, const bool verbose, const int input_height,
                                               const int input_width)
    : FeatureFrontEnd(detector_type, descriptor_type, matcher_type, selector_type, cross_check, stereo_threshold, stereo_threshold, refinement_degree,
                      verbose, input_height, input_width) {
  initDetector();
  initDescriptor();
  initMatcher();
}

ClassicFeatureFrontEnd::~ClassicFeatureFrontEnd() {
  if (ctx_) spvo_destroy(ctx_);
  ctx_ = nullptr;
}

#ifdef SPVO_USE_OPENCV
bool ClassicFeatureFrontEnd::available() { return true; }

void ClassicFeatureFrontEnd::initDetector() {   // parameters: classic.cpp:7-56
  if (detector_type_ == DetectorType::ORB) detector_ = cv::ORB::create(2000, 1.2f, 8, 31, 0, 2, cv::ORB::FAST_SCORE, 31, 20);
  else if (detector_type_ == DetectorType::BRISK) detector_ = cv::BRISK::create();
  else if (detector_type_ == DetectorType::AKAZE) detector_ = cv::AKAZE::create();
  else if (detector_type_ == DetectorType::SIFT) detector_ = cv::SIFT::create();
  else if (detector_type_ == DetectorType::FAST) detector_ = cv::FastFeatureDetector::create(10, true);
  else if (detector_type_ == DetectorType::ShiTomasi) detector_ = cv::GFTTDetector::create(1000, 0.03, 7.5, 5, false, 0.04);
  else logError("[initDetector] Detector is not implemented");
}

void ClassicFeatureFrontEnd::initDescriptor() {   // classic.cpp:58-79
  if (descriptor_type_ == DescriptorType::ORB) extractor_ = cv::ORB::create();
  else if (descriptor_type_ == DescriptorType::BRISK) extractor_ = cv::BRISK::create(30, 3, 1.0f);
  else if (descriptor_type_ == DescriptorType::AKAZE) extractor_ = cv::AKAZE::create();
  else if (descriptor_type_ == DescriptorType::SIFT) extractor_ = cv::SIFT::create();
  else logError("[initDescriptor] Decscriptor is not implemented");
}

std::vector<cv::KeyPoint> ClassicFeatureFrontEnd::detectKeypoints(const cv::Mat &img) {
  std::vector<cv::KeyPoint> keypoints;
  if (detector_) detector_->detect(img, keypoints);
  return keypoints;
}

cv::Mat ClassicFeatureFrontEnd::describeKeypoints(std::vector<cv::KeyPoint> &keypoints, const cv::Mat &img) {
  cv::Mat descriptors;
  if (extractor_) extractor_->compute(img, keypoints, descriptors);
  return descriptors;
}

void ClassicFeatureFrontEnd::addStereoImagePair(cv::Mat &img_l, cv::Mat &img_r, const cv::Mat &projection_matrix_l, const cv::Mat &projection_matrix_r) {
  if (img_l.rows != img_r.rows || img_l.cols != img_r.cols) {
    logError("input images shape doesn't match!");
    return;
  }
  projection_matrix_l_ = projection_matrix_l.clone();
  projection_matrix_r_ = projection_matrix_r.clone();
  if (input_height_ > 0 && input_width_ > 0) {   // 0 = native resolution (launch/visual_odometry_classic.launch)
    preprocessImageImpl(img_l, projection_matrix_l_);
    preprocessImageImpl(img_r, projection_matrix_r_);
  }
  cv::Mat *imgs[2] = {&img_l, &img_r};
  for (cv::Mat *im : imgs) {
    images_dq.push_back(*im);
    keypoints_dq.push_back(detectKeypoints(*im));
    descriptors_dq.push_back(describeKeypoints(keypoints_dq.back(), *im));
  }
  if (verbose_) logInfo(std::to_string(keypoints_dq.end()[-2].size()) + ", " + std::to_string(keypoints_dq.end()[-1].size()) + " keypoints for img_l and img_r");
  while (images_dq.size() > NUM_IMAGE_POSITIONS) {
    images_dq.pop_front();
    keypoints_dq.pop_front();
    descriptors_dq.pop_front();
  }
}
#else
// Without OpenCV: ORB + ORB (the reference's baseline configuration, launch/visual_odometry_classic.launch) runs on the GPU
// through spvo_orb_detect -- detection and description are one pass there, so detectKeypoints keeps the descriptors for the
// describeKeypoints call that follows on the same image; the other detector / descriptor types are OpenCV features2d calls and
// stay unavailable.
bool ClassicFeatureFrontEnd::available() { return true; }
void ClassicFeatureFrontEnd::initDetector() {
  if (detector_type_ != DetectorType::ORB) logError("[initDetector] only ORB runs without OpenCV (build with SPVO_USE_OPENCV for the other detectors of classic.cpp:7-56)");
}
void ClassicFeatureFrontEnd::initDescriptor() {
  if (descriptor_type_ != DescriptorType::ORB) logError("[initDescriptor] only ORB runs without OpenCV (build with SPVO_USE_OPENCV for the other descriptors of classic.cpp:58-79)");
}

std::vector<cv::KeyPoint> ClassicFeatureFrontEnd::detectKeypoints(const cv::Mat &img) {
  std::vector<cv::KeyPoint> keypoints;
  orb_desc_ = cv::Mat();
  if (detector_type_ != DetectorType::ORB || descriptor_type_ != DescriptorType::ORB) {
    logError("ClassicFeatureFrontEnd: this detector / descriptor pair is an OpenCV features2d call (classic.cpp:7-79) -- built without SPVO_USE_OPENCV, only ORB + ORB runs");
    return keypoints;
  }
  if (!ensureContext()) return keypoints;
  if (img.depth() != CV_8U || img.rows <= 0) {
    logError("detectKeypoints: 8-bit single-channel image expected");
    return keypoints;
  }
  constexpr int NFEATURES = 2000;   // classic.cpp:13
  std::vector<spvo_orb_keypoint> kp(NFEATURES);
  cv::Mat desc(NFEATURES, 32, CV_8UC1);
  int n = 0;
  if (spvo_orb_detect(ctx_, img.ptr<uint8_t>(0), img.rows, img.cols, (size_t)img.step, NFEATURES, kp.data(), desc.ptr<uint8_t>(0), NFEATURES, &n) != SPVO_OK) {
    logError(std::string("spvo_orb_detect: ") + spvo_last_error(ctx_));
    return keypoints;
  }
  keypoints.reserve(n);
  float level_scale[8];
  level_scale[0] = 1.f;
  for (int l = 1; l < 8; ++l) level_scale[l] = level_scale[l - 1] * 1.2f;
  for (int i = 0; i < n; ++i) {
    cv::KeyPoint k(cv::Point2f(kp[i].x, kp[i].y), 31.f * level_scale[kp[i].octave & 7]);
    k.angle = kp[i].angle * 57.29577951308232f;   // cv::KeyPoint::angle is in degrees
    if (k.angle < 0) k.angle += 360.f;
    k.response = kp[i].response;
    k.octave = kp[i].octave;
    keypoints.push_back(k);
  }
  orb_desc_ = cv::Mat(n, 32, CV_8UC1);
  if (n) std::memcpy(orb_desc_.ptr<uint8_t>(0), desc.ptr<uint8_t>(0), (size_t)n * 32);
  return keypoints;
}

cv::Mat ClassicFeatureFrontEnd::describeKeypoints(std::vector<cv::KeyPoint> &keypoints, const cv::Mat &) {
  if (orb_desc_.rows != (int)keypoints.size()) {
    logError("describeKeypoints: call detectKeypoints on the same image first (ORB detects and describes in one pass here)");
    return cv::Mat();
  }
  return orb_desc_;
}

void ClassicFeatureFrontEnd::addStereoImagePair(cv::Mat &img_l, cv::Mat &img_r, const cv::Mat &projection_matrix_l, const cv::Mat &projection_matrix_r) {
  if (img_l.rows != img_r.rows || img_l.cols != img_r.cols) {
    logError("input images shape doesn't match!");
    return;
  }
  if (detector_type_ != DetectorType::ORB || descriptor_type_ != DescriptorType::ORB) {
    logError("ClassicFeatureFrontEnd: this detector / descriptor pair is an OpenCV features2d call (classic.cpp:7-79) -- built without SPVO_USE_OPENCV, only ORB + ORB runs");
    return;
  }
  if (!ensureContext()) return;   // no device: logged, nothing pushed (nn.cpp:53-55 convention)
  projection_matrix_l_ = projection_matrix_l.clone();
  projection_matrix_r_ = projection_matrix_r.clone();
  if (input_height_ > 0 && input_width_ > 0) {   // 0 = native resolution (launch/visual_odometry_classic.launch)
    preprocessImageImpl(img_l, projection_matrix_l_);
    preprocessImageImpl(img_r, projection_matrix_r_);
  }
  cv::Mat *imgs[2] = {&img_l, &img_r};
  for (cv::Mat *im : imgs) {
    images_dq.push_back(*im);
    keypoints_dq.push_back(detectKeypoints(*im));
    descriptors_dq.push_back(describeKeypoints(keypoints_dq.back(), *im));
  }
  if (verbose_) logInfo(std::to_string(keypoints_dq.end()[-2].size()) + ", " + std::to_string(keypoints_dq.end()[-1].size()) + " keypoints for img_l and img_r");
  while (images_dq.size() > NUM_IMAGE_POSITIONS) {
    images_dq.pop_front();
    keypoints_dq.pop_front();
    descriptors_dq.pop_front();
  }
}
#endif

// ------------------------------------------------------------------------- SuperPoint front end
static std::string g_models_dir;
static int g_max_keypoints = -1, g_match_fp8 = -1;

void SuperPointFeatureFrontEnd::setModelsDir(const std::string &dir) { g_models_dir = dir; }
void SuperPointFeatureFrontEnd::setMaxKeypoints(int cap) { g_max_keypoints = cap; }
void SuperPointFeatureFrontEnd::setMatchFp8(int on) { g_match_fp8 = on; }

SuperPointFeatureFrontEnd::SuperPointFeatureFrontEnd()
    : SuperPointFeatureFrontEnd(MatcherType::BF, SelectorType::NN, true, "superpoint_pretrained", 2, "laptop", TRT_FP32, 120, 392,
                                0.015f, 4, 12, 4, 2.0f, 1.0f, 4, true) {}

SuperPointFeatureFrontEnd::SuperPointFeatureFrontEnd(const MatcherType matcher_type, const SelectorType selector_type, const bool cross_check,
                                                     const std::string model_name_prefix, const int model_batch_size, const std::string machine_name,
                                                     const TensorRtPrecision trt_precision, const int input_height, const int input_width,
                                                     const float conf_thresh, const int dist_thresh, const int num_threads, const int border_remove,
                                                     const float stereo_threshold, const float min_disparity, const int refinement_degree,
                                                     const bool verbose)
    : FeatureFrontEnd(DetectorType::SuperPoint, DescriptorType::SuperPoint, matcher_type, selector_type, cross_check, stereo_threshold, min_disparity,
                      refinement_degree, verbose, input_height, input_width),
      model_name_prefix_(model_name_prefix), model_batch_size_(model_batch_size), machine_name_(machine_name), trt_precision_(trt_precision),
      conf_thresh_(conf_thresh), dist_thresh_(dist_thresh), border_remove_(border_remove), num_threads_(num_threads) {
  initMatcher();
  loadEngine();
}

SuperPointFeatureFrontEnd::~SuperPointFeatureFrontEnd() {
  completeHostCopies();   // (their sources are mirrors the context owns)
  drainPrefetch();
  if (ctx_) spvo_destroy(ctx_);
  ctx_ = nullptr;
}

void SuperPointFeatureFrontEnd::loadEngine() {
  spvo_config cfg;
  spvo_default_config(&cfg);
  cfg.device = configured_device();
  solve_timing_ = spvo_get_tuning("solve_timing", 0) != 0;
  solve_hold_tail_ = spvo_get_tuning("solve_keep", 1) == 2;
  cfg.net_height = input_height_;
  cfg.net_width = input_width_;
  // model_batch_size_ only changes how the reference batches its TensorRT calls (hpp:342-344);
  // both images always go through the network together here.
  cfg.max_batch = 2;
  cfg.conf_thresh = conf_thresh_;
  cfg.dist_thresh = dist_thresh_;
  cfg.border_remove = border_remove_;
  if (g_max_keypoints > 0) max_keypoints_ = g_max_keypoints;
  else if (const char *mk = std::getenv("SPVO_MAX_KEYPOINTS")) max_keypoints_ = std::max(1, std::atoi(mk));
  cfg.max_keypoints = max_keypoints_;
  if (const char *bc = std::getenv("SPVO_BUG_COMPAT_P")) cfg.bug_compat_p = std::atoi(bc);
  if (model_batch_size_ != 1 && model_batch_size_ != 2) {
    logError("Wrong batch size (" + std::to_string(model_batch_size_) + ")");  // nn.cpp:490
    return;
  }
  if (spvo_create(&cfg, &ctx_) != SPVO_OK) {
    logError(std::string("spvo_create: ") + spvo_last_error(nullptr));
    ctx_ = nullptr;
    return;
  }
  std::string dir = g_models_dir;
  if (dir.empty())
    if (const char *e = std::getenv("SPVO_MODELS_DIR")) dir = e;
  const std::string model_name_full = dir + "/" + machine_name_ + "/" + model_name_prefix_ + "_" + std::to_string(model_batch_size_) + "_" +
                                      std::to_string(input_height_) + "_" + std::to_string(input_width_) + "_" +
                                      trt_precision_enum2string.at(trt_precision_) + ".spvw";
  if (spvo_load_weights(ctx_, model_name_full.c_str()) != SPVO_OK) {
    logError(spvo_last_error(ctx_));  // "no such engine file: ..." (nn.cpp:53-55): object stays half-initialised
    return;
  }
  if (spvo_engine_precision(ctx_) != (int)trt_precision_) {   // the name promises what trtexec was told (--fp16 or not)
    logError("engine file `" + model_name_full + "` was not built for " + trt_precision_enum2string.at(trt_precision_));
    return;
  }
  logInfo("engine file `" + model_name_full + "` loaded");
  // stereoCallback always asks for CURR_LEFT->CURR_RIGHT and CURR_LEFT->PREV_LEFT right after the
  // detector (node.cpp:196-198): have them enqueued in the detector's own submission
  if (g_match_fp8 >= 0 ? g_match_fp8 != 0 : std::getenv("SPVO_MATCH_FP8") != nullptr) spvo_set_match_fp8(ctx_, 1);   // fp8 shortlist GEMM (config 5; the GEMM only prunes, two-pass exact re-rank)
  if (matcher_ready_ && spvo_get_tuning("prematch", 1))
    spvo_set_prematch(ctx_, 1, selector_type_ == SelectorType::KNN ? SPVO_SELECT_KNN : SPVO_SELECT_NN, matcher_cross_check_ ? 1 : 0, knn_threshold_);
  for (int i = 0; i < 2; ++i) {
    xy_buf_[i].assign((size_t)max_keypoints_ * 2, 0.f);
    desc_buf_[i].assign((size_t)max_keypoints_ * output_desc_channel_, 0.f);
  }
  engine_loaded_ = true;
}

void SuperPointFeatureFrontEnd::addStereoImagePair(cv::Mat &img_l, cv::Mat &img_r, const cv::Mat &projection_matrix_l,
                                                   const cv::Mat &projection_matrix_r) {
  completeHostCopies();   // a pair whose bulk copies are still owed (no solve followed it): before its mirrors can be reused
  if (!engine_loaded_) {
    logError("addStereoImagePair: no engine loaded");
    return;
  }
  if (img_l.type() != CV_8UC1 || img_r.type() != CV_8UC1 || img_l.rows != img_r.rows || img_l.cols != img_r.cols ||
      (size_t)img_l.step != (size_t)img_r.step || projection_matrix_l.type() != CV_64FC1 || projection_matrix_r.type() != CV_64FC1 ||
      projection_matrix_l.rows != 3 || projection_matrix_l.cols != 4) {
    logError("addStereoImagePair: expected two equal-sized CV_8UC1 images and 3x4 CV_64F projection matrices (node.cpp:91,163-168)");
    return;
  }
  projection_matrix_l_ = projection_matrix_l.clone();  // nn.cpp:465-466
  projection_matrix_r_ = projection_matrix_r.clone();

  // One code path for the plain call and for a pair announced with prefetchStereoImagePair: the images go through
  // spvo_detect_submit (pinned staging, asynchronous copies on the network stream; resized images and descriptors come
  // back in the submission's pinned mirrors) and spvo_detect_collect.  Submissions of OTHER pairs still in flight are
  // drained first -- the reference processes pairs strictly in call order.
  const bool hit = !prefetch_q_.empty() && prefetch_q_.front().host && prefetch_q_.front().l == img_l.data && prefetch_q_.front().r == img_r.data &&
                   prefetch_q_.front().rows == img_l.rows && prefetch_q_.front().cols == img_l.cols;
  int slot_l, slot_r;
  if (hit) {
    slot_l = prefetch_q_.front().slot_l;
    slot_r = prefetch_q_.front().slot_r;
    prefetch_q_.pop_front();
  } else {
    if (!prefetch_q_.empty()) {
      drainPrefetch();
      logError("addStereoImagePair: the prefetched pair was not the one passed in; prefetch discarded");
    }
    pickSlots(&slot_l, &slot_r);
    if (spvo_detect_submit(ctx_, img_l.data, img_r.data, img_l.rows, img_l.cols, (size_t)img_l.step, slot_l, slot_r, 3) != SPVO_OK) {
      logError(std::string("spvo_detect_submit: ") + spvo_last_error(ctx_));
      return;
    }
  }
  // The results sit in the submission's pinned mirrors (the kernels wrote them there): keypoints and the image handed back to the
  // caller are copied now; the matrices of images_dq / descriptors_dq get their final size now and their contents once --
  // at once, or (setDeferredHostCopies, the default) while the solver's kernels run.
  spvo_detect_mirrors mr;
  const int rc = spvo_detect_collect_mirrors(ctx_, projection_matrix_l_.ptr<double>(0), projection_matrix_r_.ptr<double>(0), &mr);
  if (rc != SPVO_OK) {
    logError(std::string("spvo_detect_collect_mirrors: ") + spvo_last_error(ctx_));
    return;
  }
  pending_mirrors_ = mr;
  const size_t img_bytes = (size_t)input_height_ * input_width_;
  cv::Mat res[2] = {cv::Mat(input_height_, input_width_, CV_8UC1), cv::Mat(input_height_, input_width_, CV_8UC1)};
  for (int i = 0; i < 2; ++i) std::memcpy(res[i].data, mr.resized[i], img_bytes);
  // the reference mutates the caller's images in place (crop + resize + convertTo float,
  // base.cpp:89,105,115; nn.cpp:159); hand back the resized image, keep u8
  img_l = res[0];
  img_r = res[1];
  const int slots[2] = {slot_l, slot_r};
  for (int i = 0; i < 2; ++i) {
    cv::Mat im(input_height_, input_width_, CV_8UC1);                       // nn.cpp:154: images_dq holds its own copy
    cv::Mat d(mr.n[i], output_desc_channel_, CV_32FC1);
    pending_copies_.push_back(PendingCopy{mr.resized[i], im, img_bytes, false});
    pending_copies_.push_back(PendingCopy{mr.desc[i], d, (size_t)mr.n[i] * output_desc_channel_ * sizeof(float), true});
    images_dq.push_back(im);
    std::vector<cv::KeyPoint> kps;
    kps.reserve(mr.n[i]);
    for (int k = 0; k < mr.n[i]; ++k) kps.emplace_back(cv::Point2f(mr.xy[i][2 * k], mr.xy[i][2 * k + 1]), 1.f);  // nn.cpp:243
    keypoints_dq.push_back(std::move(kps));
    descriptors_dq.push_back(d);
    slots_dq_.push_back(slots[i]);
  }
  if (!defer_host_copies_) {
    completeHostCopies();
  }
  if (verbose_) logInfo(std::to_string(keypoints_dq.end()[-2].size()) + ", " + std::to_string(keypoints_dq.end()[-1].size()) + " keypoints for img_l and img_r");
  while (images_dq.size() > 4) {  // nn.cpp:494-498
    images_dq.pop_front();
    keypoints_dq.pop_front();
    descriptors_dq.pop_front();
    slots_dq_.pop_front();
  }
}

void SuperPointFeatureFrontEnd::completeImageCopies() {
  for (PendingCopy &pc : pending_copies_)
    if (!pc.descriptors && pc.bytes && pc.src && pc.dst.data) { std::memcpy(pc.dst.data, pc.src, pc.bytes); pc.bytes = 0; }
}

void SuperPointFeatureFrontEnd::completeHostCopies() {
  if (pending_copies_.empty()) return;
  completeImageCopies();
  if (ctx_ && spvo_detect_mirrors_wait(ctx_, &pending_mirrors_) != SPVO_OK)   // the descriptors travel behind the matches: long there by now
    logError(std::string("spvo_detect_mirrors_wait: ") + spvo_last_error(ctx_));
  for (PendingCopy &pc : pending_copies_)
    if (pc.bytes && pc.src && pc.dst.data) std::memcpy(pc.dst.data, pc.src, pc.bytes);
  pending_copies_.clear();
}

void SuperPointFeatureFrontEnd::prefetchStereoImagePair(const cv::Mat &img_l, const cv::Mat &img_r) {
  completeHostCopies();   // a pair whose bulk copies are still owed (no solve followed it): before its mirrors can be reused
  if (!engine_loaded_ || prefetch_q_.size() >= 5) return;
  if (img_l.type() != CV_8UC1 || img_r.type() != CV_8UC1 || img_l.rows != img_r.rows || img_l.cols != img_r.cols || (size_t)img_l.step != (size_t)img_r.step) return;
  for (const auto &q : prefetch_q_)   // already announced
    if (q.host && q.l == img_l.data && q.r == img_r.data && q.rows == img_l.rows && q.cols == img_l.cols) return;
  Prefetch pf;
  pickSlots(&pf.slot_l, &pf.slot_r);
  if (spvo_detect_submit(ctx_, img_l.data, img_r.data, img_l.rows, img_l.cols, (size_t)img_l.step, pf.slot_l, pf.slot_r, 3) != SPVO_OK) {
    logError(std::string("spvo_detect_submit: ") + spvo_last_error(ctx_));
    return;
  }
  pf.l = img_l.data; pf.r = img_r.data;
  pf.rows = img_l.rows; pf.cols = img_l.cols; pf.stride = (size_t)img_l.step; pf.host = true;
  prefetch_q_.push_back(pf);
  notePrefetchDepth();
}

void SuperPointFeatureFrontEnd::addStereoImagePairDevice(const void *d_img_l, const void *d_img_r, int rows, int cols, size_t stride,
                                                         const cv::Mat &projection_matrix_l, const cv::Mat &projection_matrix_r,
                                                         bool host_descriptors) {
  completeHostCopies();   // a pair whose bulk copies are still owed (no solve followed it): before its mirrors can be reused
  if (!engine_loaded_) {
    logError("addStereoImagePairDevice: no engine loaded");
    return;
  }
  if (projection_matrix_l.type() != CV_64FC1 || projection_matrix_r.type() != CV_64FC1 || projection_matrix_l.rows != 3 ||
      projection_matrix_l.cols != 4) {
    logError("addStereoImagePairDevice: expected 3x4 CV_64F projection matrices");
    return;
  }
  projection_matrix_l_ = projection_matrix_l.clone();
  projection_matrix_r_ = projection_matrix_r.clone();
  int slot_l, slot_r;
  spvo_features fl{0, xy_buf_[0].data(), host_descriptors ? desc_buf_[0].data() : nullptr};
  spvo_features fr{0, xy_buf_[1].data(), host_descriptors ? desc_buf_[1].data() : nullptr};
  int rc;
  const bool hit = !prefetch_q_.empty() && !prefetch_q_.front().host && prefetch_q_.front().l == d_img_l && prefetch_q_.front().r == d_img_r &&
                   prefetch_q_.front().rows == rows && prefetch_q_.front().cols == cols && prefetch_q_.front().stride == stride;
  if (!prefetch_q_.empty() && !hit) {  // a different pair was announced: drain and drop what is in flight
    drainPrefetch();
    logError("addStereoImagePairDevice: the prefetched pair was not the one passed in; prefetch discarded");
  }
  if (hit) {
    slot_l = prefetch_q_.front().slot_l;
    slot_r = prefetch_q_.front().slot_r;
    prefetch_q_.pop_front();
    rc = spvo_detect_wait(ctx_, projection_matrix_l_.ptr<double>(0), projection_matrix_r_.ptr<double>(0), &fl, &fr);
  } else {
    pickSlots(&slot_l, &slot_r);
    rc = spvo_detect_dev(ctx_, d_img_l, d_img_r, rows, cols, stride, projection_matrix_l_.ptr<double>(0),
                         projection_matrix_r_.ptr<double>(0), slot_l, slot_r, &fl, &fr);
  }
  if (rc != SPVO_OK) {
    logError(std::string("spvo_detect_dev: ") + spvo_last_error(ctx_));
    return;
  }
  const spvo_features *f[2] = {&fl, &fr};
  const cv::Mat empty;
  const cv::Mat *res[2] = {&empty, &empty};
  const int slots[2] = {slot_l, slot_r};
  pushFeatures(f, res, slots, host_descriptors);
}

void SuperPointFeatureFrontEnd::prefetchStereoImagePairDevice(const void *d_img_l, const void *d_img_r, int rows, int cols, size_t stride) {
  completeHostCopies();   // a pair whose bulk copies are still owed (no solve followed it): before its mirrors can be reused
  if (!engine_loaded_ || prefetch_q_.size() >= 5) return;
  for (const auto &q : prefetch_q_)   // already announced
    if (!q.host && q.l == d_img_l && q.r == d_img_r && q.rows == rows && q.cols == cols && q.stride == stride) return;
  Prefetch pf;
  pickSlots(&pf.slot_l, &pf.slot_r);
  if (spvo_detect_dev_submit(ctx_, d_img_l, d_img_r, rows, cols, stride, pf.slot_l, pf.slot_r) != SPVO_OK) {
    logError(std::string("spvo_detect_dev_submit: ") + spvo_last_error(ctx_));
    return;
  }
  pf.l = d_img_l; pf.r = d_img_r;
  pf.rows = rows; pf.cols = cols; pf.stride = stride;
  prefetch_q_.push_back(pf);
  notePrefetchDepth();
}

// A caller that keeps FIVE pairs announced -- the one it is about to collect and four ahead of it, each announced when it arrives --
// gets trunk pairing (spvo_set_trunk_pairing: two pairs per set of network launches).  From then on the library holds a pair whose
// network would only queue, until its successor arrives; with fewer pairs ahead each held pair would leave the network stream idle,
// so shallower look-ahead never switches it on.
void SuperPointFeatureFrontEnd::notePrefetchDepth() {
  if (!trunk_pairing_ && prefetch_q_.size() >= 5 && ctx_ && spvo_set_trunk_pairing(ctx_, 1) == SPVO_OK) trunk_pairing_ = true;
}

void SuperPointFeatureFrontEnd::drainPrefetch() {
  while (ctx_ && !prefetch_q_.empty()) {
    spvo_features dl{0, nullptr, nullptr}, dr{0, nullptr, nullptr};
    spvo_detect_wait(ctx_, nullptr, nullptr, &dl, &dr);
    prefetch_q_.pop_front();
  }
}

void SuperPointFeatureFrontEnd::pickSlots(int *slot_l, int *slot_r) {
  // device slots: a ring of 8 pairs -- the previous and the current pair plus up to four pairs announced ahead (and margin)
  *slot_l = 2 * next_pair_;
  *slot_r = *slot_l + 1;
  next_pair_ = (next_pair_ + 1) % 8;
}

void SuperPointFeatureFrontEnd::pushFeatures(const spvo_features *f[2], const cv::Mat *images[2], const int slots[2], bool host_descriptors) {
  for (int i = 0; i < 2; ++i) {
    images_dq.push_back(images[i]->empty() ? cv::Mat() : images[i]->clone());  // nn.cpp:154
    std::vector<cv::KeyPoint> kps;
    kps.reserve(f[i]->n);
    for (int k = 0; k < f[i]->n; ++k) kps.emplace_back(cv::Point2f(f[i]->xy[2 * k], f[i]->xy[2 * k + 1]), 1.f);  // nn.cpp:243
    keypoints_dq.push_back(std::move(kps));
    cv::Mat d;
    if (host_descriptors) {
      d.create(f[i]->n, output_desc_channel_, CV_32FC1);
      if (f[i]->n) std::memcpy(d.data, f[i]->desc, (size_t)f[i]->n * output_desc_channel_ * sizeof(float));
    } else {
      d.create(f[i]->n, 0, CV_32FC1);   // n x 0 header: the descriptors live in the device slot
    }
    descriptors_dq.push_back(d);
    slots_dq_.push_back(slots[i]);
  }
  if (verbose_) logInfo(std::to_string(keypoints_dq.end()[-2].size()) + ", " + std::to_string(keypoints_dq.end()[-1].size()) + " keypoints for img_l and img_r");
  while (images_dq.size() > 4) {  // nn.cpp:494-498
    images_dq.pop_front();
    keypoints_dq.pop_front();
    descriptors_dq.pop_front();
    slots_dq_.pop_front();
  }
}
