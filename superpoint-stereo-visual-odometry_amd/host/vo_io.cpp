#include "vo_io.hpp"

#include <cmath>

const tf2::Transform &PoseIntegrator::integrate(tf2::Transform cam0_curr_T_cam0_prev) {
  // screen out abnormal results: a step of more than 10 m is replaced by the last valid one
  // (visual_odometry_node.cpp:116-123)
  if (cam0_curr_T_cam0_prev.getOrigin().length() > 10) cam0_curr_T_cam0_prev = last_valid_;
  else last_valid_ = cam0_curr_T_cam0_prev;
  // base_prev_T_base_curr = base_T_cam0 * cam0_curr_T_cam0_prev^-1 * base_T_cam0^-1   (node.cpp:125-127)
  const tf2::Transform step = base_T_cam0_ * cam0_curr_T_cam0_prev.inverse() * base_T_cam0_.inverse();
  world_T_base_curr_ = world_T_base_curr_ * step;
  return world_T_base_curr_;
}

std::string KittiPoseWriter::fileName(int kitti_eval_id) {
  std::string name = std::to_string(kitti_eval_id) + "_pred.txt";   // data_processing_node.cpp:102-106
  if (name.size() == 10) name = "0" + name;
  return name;
}

bool KittiPoseWriter::open(const std::string &dir, int kitti_eval_id) {
  start_inited_ = false;
  seq_count_ = 0;
  file_.open(dir + "/" + fileName(kitti_eval_id));
  return file_.is_open();
}

static void rotation_matrix(const tf2::Quaternion &q, double R[9]) {
  const double n = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z() + q.w() * q.w());
  const double x = q.x() / n, y = q.y() / n, z = q.z() / n, w = q.w() / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

void KittiPoseWriter::write(const tf2::Transform &world_T_base_curr) {
  if (seq_count_ < seq_start_) {   // data_processing_node.cpp:145-148
    ++seq_count_;
    return;
  }
  if (!start_inited_) {
    start_inited_ = true;
    world_T_base_start_ = world_T_base_curr;
  }
  const tf2::Transform base_start_T_base_curr = world_T_base_start_.inverse() * world_T_base_curr;
  const tf2::Transform cam0_start_T_cam0_curr = base_T_cam0_.inverse() * base_start_T_base_curr * base_T_cam0_;
  double R[9];
  rotation_matrix(cam0_start_T_cam0_curr.getRotation(), R);
  const double t[3] = {cam0_start_T_cam0_curr.getOrigin().x(), cam0_start_T_cam0_curr.getOrigin().y(), cam0_start_T_cam0_curr.getOrigin().z()};
  for (int r = 0; r < 3; ++r) {   // data_processing_node.cpp:181-187
    for (int c = 0; c < 3; ++c) file_ << R[3 * r + c] << " ";
    file_ << t[r] << " ";
  }
  file_ << "\n";
}

std::string LatencyCsv::fileName(const std::string &model_name_prefix, int batch, int height, int width, const std::string &precision,
                                 int kitti_eval_id) {
  return model_name_prefix + "_" + std::to_string(batch) + "_" + std::to_string(height) + "_" + std::to_string(width) + "_" + precision + "_seq_" +
         std::to_string(kitti_eval_id) + ".csv";
}
