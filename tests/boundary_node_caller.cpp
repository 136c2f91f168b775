// A caller that goes through the front-end interface the way visual_odometry_node.cpp does: construction from launch
// parameters (node.cpp:330-403), one stereoCallback (node.cpp:150-262), a sequence restart (node.cpp:316).  ROS plumbing
// (topics, cv_bridge, publishers) is left out; every use of the interface -- class names, constructor argument lists,
// methods, public members, the cv:: and tf2:: types that cross it -- is kept.  Compiled twice by
// tests/test_boundary_cpu.py: against the stand-in types and, with -DSPVO_USE_OPENCV, against OpenCV / tf2 shaped headers.
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "feature_detection.hpp"

static std::unique_ptr<FeatureFrontEnd> feature_front_end_ptr;
static int seq = 0;

static std::unique_ptr<FeatureFrontEnd> construct(bool is_classic, const std::string &detector_type, const std::string &descriptor_type,
                                                  const std::string &matcher_type, const std::string &selector_type, double stereo_threshold,
                                                  double min_disparity, int refinement_degree, bool verbose, int image_height, int image_width,
                                                  const std::string &model_name_prefix, int model_batch_size, const std::string &machine_name,
                                                  const std::string &trt_precision, double conf_thresh, int dist_thresh, int num_threads,
                                                  int border_remove) {
  if (is_classic)
    return std::make_unique<ClassicFeatureFrontEnd>(detector_name_to_type.at(detector_type), descriptor_name_to_type.at(descriptor_type),
                                                    matcher_name_to_type.at(matcher_type), selector_name_to_type.at(selector_type), true, stereo_threshold,
                                                    min_disparity, refinement_degree, verbose, image_height, image_width);
  return std::make_unique<SuperPointFeatureFrontEnd>(matcher_name_to_type.at(matcher_type), selector_name_to_type.at(selector_type), true, model_name_prefix,
                                                     model_batch_size, machine_name, trt_precision_string2enum.at(trt_precision), image_height, image_width,
                                                     conf_thresh, dist_thresh, num_threads, border_remove, stereo_threshold, min_disparity,
                                                     refinement_degree, verbose);
}

// cameraInfoToPMatrix (node.cpp:84-98): a 3x4 CV_64F matrix
static cv::Mat p_matrix(const double P[12]) {
  cv::Mat m(3, 4, CV_64FC1);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) m.at<double>(r, c) = P[4 * r + c];
  return m;
}

// stereoCallback (node.cpp:150-262); returns false on the sequence's first frame (no pose yet), else the frame's relative pose
static bool stereo_callback(cv::Mat &cv_img_l, cv::Mat &cv_img_r, const double P_l[12], const double P_r[12], tf2::Transform &cam0_curr_T_cam0_prev, double *check) {
  const cv::Mat projection_matrix_l = p_matrix(P_l), projection_matrix_r = p_matrix(P_r);
  feature_front_end_ptr->addStereoImagePair(cv_img_l, cv_img_r, projection_matrix_l, projection_matrix_r);
  const bool verbose = feature_front_end_ptr->verbose_;
  if (seq++ == 0) {
    feature_front_end_ptr->matchDescriptors(MatchType::CURR_LEFT_CURR_RIGHT);
    *check += verbose ? 0.0 : 1.0;
    return false;
  }
  for (int i = 0; i < 2; ++i) {
    feature_front_end_ptr->matchDescriptors(static_cast<MatchType>(i));
    const cv::Mat match_image = feature_front_end_ptr->visualizeMatches(static_cast<MatchType>(i));
    (void)match_image.rows;
  }
  feature_front_end_ptr->solveStereoOdometry(cam0_curr_T_cam0_prev);
  const cv::Mat inliers_image = feature_front_end_ptr->visualizeInliers(ImagePosition::CURR_LEFT);
  (void)inliers_image.cols;
  // publishOdometry (node.cpp:100-148): step length gate, inverse, composition, quaternion components
  if (cam0_curr_T_cam0_prev.getOrigin().length() > 10) { *check -= 1; return true; }
  tf2::Transform base_T_cam0;
  base_T_cam0.setIdentity();
  const tf2::Transform step = base_T_cam0 * cam0_curr_T_cam0_prev.inverse() * base_T_cam0.inverse();
  *check += step.getRotation().w() + step.getRotation().x() + step.getOrigin().z() +
            (double)feature_front_end_ptr->keypoints_dq.back().size() + (double)feature_front_end_ptr->descriptors_dq.back().rows +
            (double)feature_front_end_ptr->images_dq.back().cols + (double)feature_front_end_ptr->cv_DMatches_list[0].size();
  return true;
}

// The node's life cycle over `n_frames` stereo pairs (row-major 8-bit images, back to back): construction from launch parameters,
// one stereoCallback per pair, the goal callback's reset.  poses[k] = (qx, qy, qz, qw, tx, ty, tz) of frame k's
// cam0_curr_T_cam0_prev as solveStereoOdometry(tf2::Transform&) returned it, NaN for the first frame.  Returns the number of poses.
extern "C" int boundary_node_run(int is_classic, int n_frames, const unsigned char *l, const unsigned char *r, int rows, int cols, const double P_l[12],
                                 const double P_r[12], double min_disparity, int image_height, int image_width, double *poses, double *check_out) {
  feature_front_end_ptr = construct(is_classic != 0, "ORB", "ORB", "BF", "KNN", 2.0, min_disparity, 4, false, image_height, image_width, "superpoint_pretrained", 2,
                                    "laptop", "FP32", 0.015, 4, 6, 4);
  seq = 0;
  double acc = 0;
  int n_poses = 0;
  for (int k = 0; k < n_frames; ++k) {
    cv::Mat cv_img_l(rows, cols, CV_8UC1), cv_img_r(rows, cols, CV_8UC1);
    const unsigned char *pl = l + (size_t)k * rows * cols, *pr = r + (size_t)k * rows * cols;
    for (int y = 0; y < rows; ++y)
      for (int x = 0; x < cols; ++x) {
        cv_img_l.at<unsigned char>(y, x) = pl[(size_t)y * cols + x];
        cv_img_r.ptr<unsigned char>(y)[x] = pr[(size_t)y * cols + x];
      }
    tf2::Transform T;
    T.setIdentity();
    const bool have = stereo_callback(cv_img_l, cv_img_r, P_l, P_r, T, &acc);
    if (poses) {
      double *o = poses + 7 * k;
      if (have) {
        o[0] = T.getRotation().x(); o[1] = T.getRotation().y(); o[2] = T.getRotation().z(); o[3] = T.getRotation().w();
        o[4] = T.getOrigin().x(); o[5] = T.getOrigin().y(); o[6] = T.getOrigin().z();
      } else {
        for (int q = 0; q < 7; ++q) o[q] = __builtin_nan("");
      }
    }
    n_poses += have ? 1 : 0;
  }
  feature_front_end_ptr->clearLagecyData();   // dataLodaerGoalCallback (node.cpp:316)
  cv::Mat img(rows, cols, CV_8UC1), P = p_matrix(P_l);
  feature_front_end_ptr->preprocessImageImpl(img, P);   // public in the reference (hpp:113)
  feature_front_end_ptr.reset();
  if (check_out) *check_out = acc;
  return n_poses;
}

int boundary_node_caller(bool is_classic, unsigned char *l, unsigned char *r, int rows, int cols, const double P_l[12], const double P_r[12]) {
  double check = 0;
  std::vector<unsigned char> l2((size_t)2 * rows * cols), r2((size_t)2 * rows * cols);
  for (int k = 0; k < 2; ++k) {
    std::copy(l, l + (size_t)rows * cols, l2.begin() + (size_t)k * rows * cols);
    std::copy(r, r + (size_t)rows * cols, r2.begin() + (size_t)k * rows * cols);
  }
  boundary_node_run(is_classic, 2, l2.data(), r2.data(), rows, cols, P_l, P_r, 2.0, 360, 1176, nullptr, &check);
  return check > 0;
}
