// vo_io.hpp -- the two result files of the reference pipeline, produced by the caller side of
// the front end ("next" row 2):
//   * pose integration of publishOdometry          visual_odometry_node.cpp:100-148
//   * KITTI pose file of data_processing_node      data_processing_node.cpp:36-57, 96-118, 144-188
//   * per-frame latency CSV                        visual_odometry_node.cpp:246-258, 274-303
// No ROS types: the node-side glue (topics, tf lookup of base_link -> camera_gray_left) passes
// plain transforms in.
#pragma once
#include <fstream>
#include <string>

#include "feature_detection.hpp"

// world_T_base accumulation exactly as publishOdometry does it
class PoseIntegrator {
public:
  explicit PoseIntegrator(const tf2::Transform &base_T_cam0) : base_T_cam0_(base_T_cam0) { reset(); }
  void reset() {
    world_T_base_curr_.setIdentity();
    last_valid_.setIdentity();
  }
  // returns world_T_base_curr after integrating one front-end output
  const tf2::Transform &integrate(tf2::Transform cam0_curr_T_cam0_prev);
  const tf2::Transform &pose() const { return world_T_base_curr_; }

private:
  tf2::Transform base_T_cam0_, world_T_base_curr_, last_valid_;
};

// <dir>/<id, two digits>_pred.txt, 12 numbers per line (3x4 row-major), each followed by a blank,
// default ostream precision (6 significant digits) -- what the KITTI odometry devkit reads
class KittiPoseWriter {
public:
  KittiPoseWriter(const tf2::Transform &base_T_cam0, int seq_start = 0) : base_T_cam0_(base_T_cam0), seq_start_(seq_start) {}
  static std::string fileName(int kitti_eval_id);
  bool open(const std::string &dir, int kitti_eval_id);
  void close() { file_.close(); }
  // visualOdomCallback: one world_T_base pose per frame; frames before seq_start are skipped
  void write(const tf2::Transform &world_T_base_curr);

private:
  tf2::Transform base_T_cam0_, world_T_base_start_;
  bool start_inited_ = false;
  int seq_start_ = 0, seq_count_ = 0;
  std::ofstream file_;
};

class LatencyCsv {
public:
  // <prefix>_<B>_<H>_<W>_<precision>_seq_<id>.csv   (node.cpp:285-296)
  static std::string fileName(const std::string &model_name_prefix, int batch, int height, int width,
                              const std::string &precision, int kitti_eval_id);
  bool open(const std::string &path) {
    file_.open(path);
    return file_.is_open();
  }
  void close() { file_.close(); }
  // t_detect, t_match, t_solve, t_total in ms (node.cpp:246-258)
  void row(float t_detect, float t_match, float t_solve, float t_total) {
    file_ << t_detect << "," << t_match << "," << t_solve << "," << t_total << "\n";
  }

private:
  std::ofstream file_;
};
