// feature_detection.hpp -- host-side mirror of the reference's front-end interface
// (reference: src/odml_visual_odometry/include/odml_visual_odometry/feature_detection.hpp).
//
// Same enums, class names, constructor argument order, public methods and public
// data members as the reference, so visual_odometry_node.cpp's call sequence
// (node.cpp:175-244, 316, 396-403) reads unchanged.  What differs:
//   * NvInfer.h / cuda_runtime_api.h / Eigen thread pool members are gone; every heavy call
//     goes to the C ABI in include/spvo.h (hand-written gfx950 kernels);
//   * where OpenCV / tf2 / ROS headers are absent (this build image), the few types
//     the interface mentions are the stand-ins below (`cvlite`, `tf2lite`), API-compatible with
//     the subset of cv:: / tf2:: that the front end and the node use; a ROS build defines
//     SPVO_USE_OPENCV and gets the real types.
#pragma once

#include <array>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/spvo.h"

// --------------------------------------------------------------------------- OpenCV / tf2 types
// With SPVO_USE_OPENCV the interface is compiled against the real headers (a ROS build: INTEGRATION.md); without it
// against the stand-ins below.  The stand-ins expose the SAME API subset the front end and the node use -- rows / cols /
// data / step, depth(), type(), at<T>(), ptr<T>(), clone(), create(), the CV_8U / CV_32F / CV_64F codes; getRotation().x(),
// getOrigin().length() ... -- so host/*.cpp is one source for both builds (tests/test_boundary_cpu.py compiles it both ways).
#ifdef SPVO_USE_OPENCV
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <tf2/LinearMath/Transform.h>
#else
#ifndef CV_8U
#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 CV_8U
#define CV_32FC1 CV_32F
#define CV_64FC1 CV_64F
#endif
namespace cvlite {
// dense single-channel 2-D matrix with the part of cv::Mat's API this interface touches
class Mat {
public:
  struct MatStep {   // cv::Mat::step converts to size_t
    size_t p = 0;
    operator size_t() const { return p; }
  };
  int rows = 0, cols = 0;
  uint8_t *data = nullptr;
  MatStep step;

  Mat() = default;
  Mat(int r, int c, int type) { create(r, c, type); }
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type; step.p = (size_t)c * elemSize();
    const size_t bytes = (size_t)r * step.p;
    // cv::Mat::create leaves the buffer uninitialised; this stand-in zero-fills SMALL matrices (pose / projection matrices a test may
    // build element by element) and leaves image- and descriptor-sized ones alone (1 MB of memset is 30 us of a 1.1 ms frame)
    buf_ = bytes > 65536 ? std::shared_ptr<uint8_t>(new uint8_t[bytes], std::default_delete<uint8_t[]>())
                         : std::shared_ptr<uint8_t>(new uint8_t[bytes ? bytes : 1](), std::default_delete<uint8_t[]>());
    data = bytes ? buf_.get() : nullptr;
  }
  int type() const { return type_; }
  int depth() const { return type_; }   // one channel: type == depth
  int channels() const { return 1; }
  size_t elemSize() const { return type_ == CV_8U ? 1 : type_ == CV_32F ? 4 : 8; }
  bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int r = 0; r < rows && m.step.p; ++r) std::copy(data + r * step.p, data + r * step.p + m.step.p, m.data + r * m.step.p);
    return m;
  }
  void release() { *this = Mat(); }
  template <typename T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + r * step.p + c * sizeof(T)); }
  template <typename T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(data + r * step.p + c * sizeof(T)); }
  template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + r * step.p); }
  template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + r * step.p); }

private:
  int type_ = CV_8U;
  std::shared_ptr<uint8_t> buf_;
};

struct Point2f {
  float x = 0, y = 0;
  Point2f() = default;
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct KeyPoint {
  Point2f pt;
  float size = 0;
  float angle = -1;      // degrees, as in OpenCV; -1 = not applicable
  float response = 0;
  int octave = 0;
  KeyPoint() = default;
  KeyPoint(Point2f p, float s) : pt(p), size(s) {}
};
struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0;
};
}  // namespace cvlite

namespace tf2lite {
// rigid transform with the accessors of tf2::Transform / tf2::Quaternion / tf2::Vector3 the node uses
class Quaternion {
public:
  Quaternion() = default;
  Quaternion(double x, double y, double z, double w) : v_{x, y, z, w} {}
  double x() const { return v_[0]; }
  double y() const { return v_[1]; }
  double z() const { return v_[2]; }
  double w() const { return v_[3]; }

private:
  double v_[4] = {0, 0, 0, 1};
};
class Vector3 {
public:
  Vector3() = default;
  Vector3(double x, double y, double z) : v_{x, y, z} {}
  double x() const { return v_[0]; }
  double y() const { return v_[1]; }
  double z() const { return v_[2]; }
  double length() const;

private:
  double v_[3] = {0, 0, 0};
};
class Transform {
public:
  void setRotation(const Quaternion &r) { q = r; }
  void setOrigin(const Vector3 &o) { t = o; }
  const Quaternion &getRotation() const { return q; }
  const Vector3 &getOrigin() const { return t; }
  Transform inverse() const;
  Transform operator*(const Transform &o) const;
  void setIdentity() { q = Quaternion(); t = Vector3(); }

private:
  Quaternion q;
  Vector3 t;
};
}  // namespace tf2lite
namespace cv = cvlite;
namespace tf2 = tf2lite;
#endif  // SPVO_USE_OPENCV

///////////////////////////////////////////////////////////////////////////////////////
/////////////////////////// Type and macro definitions (hpp:24-90) ////////////////////
///////////////////////////////////////////////////////////////////////////////////////

enum class DetectorType { ShiTomasi, BRISK, FAST, ORB, AKAZE, SIFT, SuperPoint };
const std::unordered_map<std::string, DetectorType> detector_name_to_type = {
    {"ShiTomasi", DetectorType::ShiTomasi}, {"BRISK", DetectorType::BRISK}, {"FAST", DetectorType::FAST},
    {"ORB", DetectorType::ORB}, {"AKAZE", DetectorType::AKAZE}, {"SIFT", DetectorType::SIFT},
    {"SuperPoint", DetectorType::SuperPoint}};
enum class DescriptorType { BRISK, ORB, BRIEF, AKAZE, FREAK, SIFT, SuperPoint };
const std::unordered_map<std::string, DescriptorType> descriptor_name_to_type = {
    {"BRISK", DescriptorType::BRISK}, {"ORB", DescriptorType::ORB}, {"BRIEF", DescriptorType::BRIEF},
    {"AKAZE", DescriptorType::AKAZE}, {"FREAK", DescriptorType::FREAK}, {"SIFT", DescriptorType::SIFT},
    {"SuperPoint", DescriptorType::SuperPoint}};
enum class MatcherType { BF, FLANN };
const std::unordered_map<std::string, MatcherType> matcher_name_to_type = {{"BF", MatcherType::BF}, {"FLANN", MatcherType::FLANN}};
enum class SelectorType { NN, KNN };
const std::unordered_map<std::string, SelectorType> selector_name_to_type = {{"NN", SelectorType::NN}, {"KNN", SelectorType::KNN}};

enum ImagePosition { PREV_LEFT = -4, PREV_RIGHT = -3, CURR_LEFT = -2, CURR_RIGHT = -1, NUM_IMAGE_POSITIONS = 4 };
const std::map<int, std::string> ImagePosition_str = {
    {PREV_LEFT, "PREV_LEFT"}, {PREV_RIGHT, "PREV_RIGHT"}, {CURR_LEFT, "CURR_LEFT"}, {CURR_RIGHT, "CURR_RIGHT"}};

enum MatchType { CURR_LEFT_CURR_RIGHT = 0, CURR_LEFT_PREV_LEFT = 1, PREV_LEFT_PREV_RIGHT = 2, MATCH_TYPE_NUM = 3 };
const std::string MatchType_str[] = {"CURR_LEFT_CURR_RIGHT", "CURR_LEFT_PREV_LEFT", "PREV_LEFT_PREV_RIGHT"};
const std::array<std::pair<int, int>, MATCH_TYPE_NUM> match_type_to_positions = {
    std::pair<int, int>(CURR_LEFT, CURR_RIGHT), std::pair<int, int>(CURR_LEFT, PREV_LEFT),
    std::pair<int, int>(PREV_LEFT, PREV_RIGHT)};

// FP32 and FP16 as in the reference (hpp:124-126); INT8 is an extension (BASELINE config 5)
enum TensorRtPrecision { TRT_FP32 = 0, TRT_FP16 = 1, TRT_INT8 = 2, NUM_TRT_PRECISION_CHOICES = 3 };
const std::unordered_map<std::string, TensorRtPrecision> trt_precision_string2enum = {{"FP32", TRT_FP32}, {"FP16", TRT_FP16}, {"INT8", TRT_INT8}};
const std::array<std::string, NUM_TRT_PRECISION_CHOICES> trt_precision_enum2string = {"FP32", "FP16", "INT8"};

///////////////////////////////////////////////////////////////////////////////////////
//////////////////////////////// Abstract class (hpp:96-178) //////////////////////////
///////////////////////////////////////////////////////////////////////////////////////

class FeatureFrontEnd {
public:
  FeatureFrontEnd(const DetectorType detector_type, const DescriptorType descriptor_type,
                  const MatcherType matcher_type, const SelectorType selector_type, const bool cross_check,
                  const float stereo_threshold, const float min_disparity, const int refinement_degree,
                  const bool verbose, const int input_height, const int input_width)
      : verbose_(verbose), detector_type_(detector_type), descriptor_type_(descriptor_type),
        matcher_type_(matcher_type), selector_type_(selector_type), cross_check_(cross_check),
        stereo_threshold_(stereo_threshold), min_disparity_(min_disparity),
        refinement_degree_(refinement_degree), input_height_(input_height), input_width_(input_width) {}
  virtual ~FeatureFrontEnd() {}
  void initMatcher();
  void clearLagecyData();
  // base.cpp:68-121: centre-crop to the network aspect ratio, cv::resize(INTER_LINEAR), scale rows 0-1 of P -- on the
  // GPU (spvo_preprocess).  `img` is replaced by the resized CV_8UC1 image, as the reference does in place.
  void preprocessImageImpl(cv::Mat &img, cv::Mat &projection_matrix);
  virtual void addStereoImagePair(cv::Mat &img_l, cv::Mat &img_r, const cv::Mat &projection_matrix_l,
                                  const cv::Mat &projection_matrix_r) = 0;
  void matchDescriptors(const MatchType match_type);
  void solveStereoOdometry(tf2::Transform &cam0_curr_T_cam0_prev);
  // Extension: the same call in two halves (spvo_solve_submit / spvo_solve_wait_prior).  A caller that collects the pose of frame k
  // after it has handed frame k+1's images over keeps the solver off its critical path; results are those of the one-piece
  // call, bit for bit.  Since round 6 frame k+1 may even be SUBMITTED before frame k is collected (two solves in flight, collected
  // oldest first): no step of a frame's device chain needs the previous frame's pose, and the gate (base.cpp:241-272) is evaluated
  // by Collect against the prior as it stands then.
  bool solveStereoOdometrySubmit();
  bool solveStereoOdometryCollect(tf2::Transform &cam0_curr_T_cam0_prev);
  bool solvePending() const { return !solve_q_.empty(); }
  int solvesPending() const { return (int)solve_q_.size(); }
  // Extension: the HIP device the front ends constructed AFTERWARDS create their context on (one process per GPU sets its local
  // rank here).  < 0 (default): environment variable SPVO_DEVICE, else device 0.
  static void setDevice(int device);
  // Extension: finish whatever a front end still has to copy into images_dq / descriptors_dq (SuperPointFeatureFrontEnd defers the
  // bulk copies of a host-image pair until the solver's kernels are running: see there).  Every entry point of this class calls it
  // where it matters; code that reads the deques DIRECTLY between addStereoImagePair and solveStereoOdometry calls it first.
  virtual void completeHostCopies() {}
  virtual void completeImageCopies() {}   // the part of it that does not have to wait for the descriptors' mirror (matchDescriptors runs it while the GPU matches)
  // Drawing only (base.cpp:401-432, 502-553): out of the hot-path scope; they return the
  // stored image untouched so that the node's publish calls keep working.
  cv::Mat visualizeMatches(const MatchType match_type);
  cv::Mat visualizeInliers(const ImagePosition image_position);

  std::deque<cv::Mat> images_dq;
  std::deque<std::vector<cv::KeyPoint>> keypoints_dq;
  std::deque<cv::Mat> descriptors_dq;
  std::array<std::vector<cv::DMatch>, MATCH_TYPE_NUM> cv_DMatches_list;

  const bool verbose_;

  // introspection for tests / the latency CSV (not in the reference)
  const std::array<std::vector<int>, MATCH_TYPE_NUM> &mapsOfIndices() const { return maps_of_indices; }
  const std::vector<int> &inliersPnp() const { return inliers_pnp; }
  const std::vector<int> &inliersPostmatching() const { return inliers_postmatching; }
  const std::string &lastError() const { return last_error_; }
  int frameCount() const { return frame_count; }
  // outcome of the last solveStereoOdometry: solvePnPRansac's return value, the gate (base.cpp:243-272), refinement kept (base.cpp:366-374)
  bool lastPnpOk() const { return last_pnp_ok_; }
  bool lastAccepted() const { return last_accepted_; }
  bool lastRefined() const { return last_refined_; }
  int lastLmIterations() const { return last_lm_iterations_; }
  uint32_t ransac_seed = 0;

protected:
  const DetectorType detector_type_;
  const DescriptorType descriptor_type_;
  const MatcherType matcher_type_;
  const float knn_threshold_ = 0.8;
  const SelectorType selector_type_;
  const bool cross_check_;
  const float stereo_threshold_;
  const float min_disparity_;
  const int refinement_degree_;
  constexpr static double TIME_INTERVAL = 0.1;
  constexpr static double MAX_ACCELERATION = 8.0;
  constexpr static int IGNORE_FRAME_COUNT = 10;
  const int input_height_;
  const int input_width_;

  // replaces cv::Ptr<cv::DescriptorMatcher> matcher_: for NORM_L2 descriptors the matcher lives behind the C ABI
  bool matcher_ready_ = false;
  bool matcher_cross_check_ = false;
  struct PendingSolve { int n = 0; std::vector<int> inliers_postmatching; };
  std::deque<PendingSolve> solve_q_;       // solveStereoOdometrySubmit .. Collect, oldest first (at most three)
  bool prev_points_on_device_ = false;     // a submit has left its triangulated points in the context: the next one refers to them by index
  bool solve_timing_ = false;      // tuning "solve_timing", read when the context is created
  bool solve_hold_tail_ = false;   // tuning "solve_keep" = 2, read when the context is created: the caller keeps TWO solves pending behind every submit, so a
                                   // frame's last solver kernel is held back for the next frame's launch (spvo_solve_input::late_prior = 2)
  double solve_timing_acc_[3] = {0, 0, 0};
  long solve_timing_calls_ = 0;
  std::vector<float> solve_pts3d_;
  bool matcher_hamming_ = false;   // NORM_HAMMING descriptors of the classic front end (base.cpp:13-28): spvo_match_hamming
  // creates ctx_ without an engine: preprocessImageImpl and solveStereoOdometry of a front end that has no network
  bool ensureContext();

  cv::Mat projection_matrix_l_;
  cv::Mat projection_matrix_r_;

  double r_vec_pred[3] = {0, 0, 0};
  double t_vec_pred[3] = {0, 0, 0};
  int frame_count = 0;

  std::array<std::vector<int>, MATCH_TYPE_NUM> maps_of_indices;

  bool prev_left_points_3d_inited = false;
  std::vector<float> prev_left_points_3d;  // n x 3
  std::vector<int> map_from_prev_left_matched_to_prev_valid_index;
  std::vector<int> map_from_curr_valid_to_prev_left_matched_index;
  std::vector<int> map_from_curr_left_matched_to_curr_valid_index;

  std::vector<int> inliers_postmatching;
  std::vector<int> inliers_pnp;

  // C-ABI context and the device feature slot each deque entry lives in
  spvo_ctx *ctx_ = nullptr;
  std::deque<int> slots_dq_;
  std::string last_error_;
  bool last_pnp_ok_ = false, last_accepted_ = false, last_refined_ = false;
  int last_lm_iterations_ = 0;
  void logError(const std::string &msg);
  void logInfo(const std::string &msg) const;
};

///////////////////////////////////////////////////////////////////////////////////////
//////////////////// Classic front end (hpp:184-235, classic.cpp) /////////////////////
///////////////////////////////////////////////////////////////////////////////////////

// The CPU baseline of BASELINE config 1 (ORB / BRISK / AKAZE / SIFT / FAST / GFTT through cv::Feature2D).  Detection,
// description and Hamming / L2 matching are OpenCV calls on the host, so the class does its work only in a build with
// SPVO_USE_OPENCV; solveStereoOdometry still runs through the C ABI.  Without OpenCV (this image) the class keeps its
// place in the interface -- visual_odometry_node.cpp:353-360 constructs it when `is_classic` is set -- and
// addStereoImagePair logs an error and returns, the reference's convention for a front end that cannot run
// (nn.cpp:53-55).
class ClassicFeatureFrontEnd : public FeatureFrontEnd {
public:
  ClassicFeatureFrontEnd();
  ClassicFeatureFrontEnd(const DetectorType detector_type, const DescriptorType descriptor_type,
                         const MatcherType matcher_type, const SelectorType selector_type, const bool cross_check,
                         const float stereo_threshold, const float min_disparity, const int refinement_degree,
                         const bool verbose, const int input_height, const int input_width);
  ~ClassicFeatureFrontEnd();

  void initDetector();
  void initDescriptor();
  void addStereoImagePair(cv::Mat &img_l, cv::Mat &img_r, const cv::Mat &projection_matrix_l,
                          const cv::Mat &projection_matrix_r) override;
  std::vector<cv::KeyPoint> detectKeypoints(const cv::Mat &img);
  cv::Mat describeKeypoints(std::vector<cv::KeyPoint> &keypoints, const cv::Mat &img);
  static bool available();   // false in a build without OpenCV

private:
#ifdef SPVO_USE_OPENCV
  cv::Ptr<cv::FeatureDetector> detector_;
  cv::Ptr<cv::DescriptorExtractor> extractor_;
#else
  cv::Mat orb_desc_;   // descriptors of the image detectKeypoints saw last (spvo_orb_detect: one pass for both)
#endif
};

///////////////////////////////////////////////////////////////////////////////////////
////////////////////////// SuperPoint front end (hpp:253-391) /////////////////////////
///////////////////////////////////////////////////////////////////////////////////////

class SuperPointFeatureFrontEnd : public FeatureFrontEnd {
public:
  SuperPointFeatureFrontEnd();
  SuperPointFeatureFrontEnd(const MatcherType matcher_type, const SelectorType selector_type,
                            const bool cross_check, const std::string model_name_prefix,
                            const int model_batch_size, const std::string machine_name,
                            const TensorRtPrecision trt_precision, const int input_height,
                            const int input_width, const float conf_thresh, const int dist_thresh,
                            const int num_threads, const int border_remove, const float stereo_threshold,
                            const float min_disparity, const int refinement_degree, const bool verbose);
  ~SuperPointFeatureFrontEnd();

  // loadTrtEngine's successor: <models dir>/<machine>/<prefix>_<B>_<H>_<W>_<FP32|FP16>.spvw
  // (same naming rule as nn.cpp:44-49).  The models dir replaces ros::package::getPath:
  // environment variable SPVO_MODELS_DIR, or setModelsDir() before construction.
  void loadEngine();
  static void setModelsDir(const std::string &dir);
  // Build-side options that change RESULTS, set like the models dir: before construction, for the front ends constructed afterwards
  // (the 17-argument constructor is the reference's and has no room for them).  A value < 0 (the default) leaves the decision to the
  // environment variable named beside it, which is kept as a fallback for unchanged launch files; 0 / a positive value overrides it.
  static void setMaxKeypoints(int cap);   // keypoint cap per image (hpp:368: 1000; BASELINE config 5: 2048); env SPVO_MAX_KEYPOINTS
  static void setMatchFp8(int on);        // fp8 shortlist GEMM in the matcher (exact re-rank behind it, config 5); env SPVO_MATCH_FP8

  void addStereoImagePair(cv::Mat &img_l, cv::Mat &img_r, const cv::Mat &projection_matrix_l,
                          const cv::Mat &projection_matrix_r) override;

  // Extension (not in the reference): the stereo pair is already resident in device memory
  // (u8, `stride` bytes per row), e.g. written by a GPU camera/rectification pipeline.  Same
  // bookkeeping as addStereoImagePair; images_dq receives empty placeholders and, unless
  // `host_descriptors` is set, descriptors_dq holds n x 0 headers (the descriptors stay on the
  // device, where matchDescriptors reads them).
  void addStereoImagePairDevice(const void *d_img_l, const void *d_img_r, int rows, int cols, size_t stride,
                                const cv::Mat &projection_matrix_l, const cv::Mat &projection_matrix_r,
                                bool host_descriptors = false);
  // Extension: hand over the NEXT stereo pair early.  The detector chain for it is enqueued at once
  // and runs on the GPU while the caller is still matching / solving the current pair; the following
  // addStereoImagePairDevice with the same pointers only collects the result.  Purely a latency-hiding
  // hint: results are identical with or without it.
  void prefetchStereoImagePairDevice(const void *d_img_l, const void *d_img_r, int rows, int cols, size_t stride);
  // Extension: the same hint for HOST images (the node's message queue holds the next pairs before their turn:
  // ApproximateTime queue of 20, node.cpp:307-313).  The images are copied to pinned staging at once, so the caller may
  // reuse them; the following addStereoImagePair with the same images (same data pointers and size) only collects.
  void prefetchStereoImagePair(const cv::Mat &img_l, const cv::Mat &img_r);
  spvo_ctx *context() const { return ctx_; }
  // The resized images and the descriptors of a HOST-image pair are written by the GPU into pinned mirrors of the submission
  // (spvo_detect_collect_mirrors) and copied ONCE into the matrices of images_dq / descriptors_dq -- by default INSIDE
  // addStereoImagePair, as the reference fills both deques there (nn.cpp:154, 494-498): whatever reads the public deques after the
  // call returns (a node's drawing code, a subclass) sees what the reference would show.
  // Opt-in, setDeferredHostCopies(true): the two bulk copies (2 x 0.42 MB + 2 x 1 MB: ~0.1 ms of host time) are put off --
  // addStereoImagePair pushes matrices of the final size and returns; they are filled while the solver's kernels run
  // (solveStereoOdometry, between its submit and its wait: 864 -> 929 frames/s on the synchronous call sequence), or at the next call
  // that takes a pair, draws, clears or destroys -- always before the mirrors are reused and before anything this class hands
  // out.  A caller that opts in and reads images_dq / descriptors_dq directly between addStereoImagePair and solveStereoOdometry
  // calls completeHostCopies() first.  Keypoints, the image handed back to the caller and every index map are never deferred.
  void setDeferredHostCopies(bool on) { if (!on) completeHostCopies(); defer_host_copies_ = on; }
  void completeHostCopies() override;
  void completeImageCopies() override;

  inline int getInputHeight() const { return input_height_; }
  inline int getInputWidth() const { return input_width_; }
  bool engineLoaded() const { return engine_loaded_; }
  int maxKeypoints() const { return max_keypoints_; }

private:
  const std::string model_name_prefix_;
  const int model_batch_size_;
  const std::string machine_name_;
  const TensorRtPrecision trt_precision_;
  static constexpr int output_det_channel_ = 65;
  static constexpr int output_desc_channel_ = 256;
  const float conf_thresh_;
  const int dist_thresh_;
  const int border_remove_;
  int max_keypoints_ = 1000;   // hpp:368 (static constexpr there); setMaxKeypoints (or env SPVO_MAX_KEYPOINTS) raises it (BASELINE config 5: 2048)
  const int num_threads_;  // kept for signature parity; the work runs on the GPU
  bool engine_loaded_ = false;
  std::vector<float> xy_buf_[2], desc_buf_[2];
  struct Prefetch {
    const void *l = nullptr, *r = nullptr;   // device pointers, or the data pointers of host images
    int rows = 0, cols = 0, slot_l = 0, slot_r = 0;
    size_t stride = 0;
    bool host = false;
  };
  std::deque<Prefetch> prefetch_q_;   // pairs announced and not collected yet (at most 5: the one about to be collected + four ahead), oldest first
  int next_pair_ = 0;                 // ring of 8 slot pairs: previous, current, up to four announced ahead
  bool trunk_pairing_ = false;        // five announced pairs (the current one + four ahead) have been seen: spvo_set_trunk_pairing is on
  void notePrefetchDepth();
  void drainPrefetch();
  void pickSlots(int *slot_l, int *slot_r);
  void pushFeatures(const spvo_features *f[2], const cv::Mat *images[2], const int slots[2], bool host_descriptors);
  struct PendingCopy { const void *src = nullptr; cv::Mat dst; size_t bytes = 0; bool descriptors = false; };
  std::vector<PendingCopy> pending_copies_;   // mirror -> matrix copies not made yet (at most one pair's)
  spvo_detect_mirrors pending_mirrors_;       // ... and where they come from (spvo_detect_mirrors_wait before the descriptors are read)
  bool defer_host_copies_ = false;   // reference-observable deques by default (nn.cpp:154, 494-498); setDeferredHostCopies(true) opts in
};
