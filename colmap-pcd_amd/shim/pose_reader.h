// pose_reader.h -- image pose priors from a "pose.ply" (SURVEY.md 8f row N4), host code.
//
// Mirrors IncrementalMapperController::LoadPose (controllers/incremental_mapper.cc:920-996): every line after
// "end_header" is one image (image ids count from 1 in file order, also across skipped lines); a line that
// contains the token "nan" is skipped; the six numbers x y z roll pitch yaw are given in the LiDAR frame
// (x front, y left, z up) and converted to COLMAP's world-to-camera pose:
//   t_wc = (-y, -z, x);  R_wc = Ry(-yaw) * Rx(-pitch) * Rz(roll);  R_cw = R_wc^T;  t_cw = -R_cw t_wc;
//   stored as {t_cw.x, t_cw.y, t_cw.z, q_cw.w, q_cw.x, q_cw.y, q_cw.z}  (this order, .cc:977-978).
// The quaternion is extracted like Eigen::Quaterniond(Matrix3d) does (trace branch, else the largest diagonal).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace colmap_hip {

inline void RotationToQuaternion(const double m[3][3], double q[4] /*w x y z*/) {
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (m[2][1] - m[1][2]) * t;
    q[2] = (m[0][2] - m[2][0]) * t;
    q[3] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[1 + i] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[k][j] - m[j][k]) * t;
    q[1 + j] = (m[j][i] + m[i][j]) * t;
    q[1 + k] = (m[k][i] + m[i][k]) * t;
  }
}

// x y z roll pitch yaw (LiDAR frame) -> {t_cw, q_cw}
inline std::array<double, 7> LidarPoseToColmap(const double pose[6]) {
  const double t_wc[3] = {-pose[1], -pose[2], pose[0]};
  const double roll = pose[3], pitch = -pose[4], yaw = -pose[5];
  const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
  const double cy = std::cos(yaw), sy = std::sin(yaw);
  const double Rz[3][3] = {{cr, -sr, 0}, {sr, cr, 0}, {0, 0, 1}};
  const double Rx[3][3] = {{1, 0, 0}, {0, cp, -sp}, {0, sp, cp}};
  const double Ry[3][3] = {{cy, 0, sy}, {0, 1, 0}, {-sy, 0, cy}};
  double A[3][3], R[3][3];   // A = Ry * Rx, R = A * Rz = R_wc
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[i][j] = Ry[i][0] * Rx[0][j] + Ry[i][1] * Rx[1][j] + Ry[i][2] * Rx[2][j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = A[i][0] * Rz[0][j] + A[i][1] * Rz[1][j] + A[i][2] * Rz[2][j];
  double Rcw[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rcw[i][j] = R[j][i];
  std::array<double, 7> out{};
  for (int i = 0; i < 3; ++i) out[i] = -(Rcw[i][0] * t_wc[0] + Rcw[i][1] * t_wc[1] + Rcw[i][2] * t_wc[2]);
  RotationToQuaternion(Rcw, &out[3]);
  return out;
}

// Returns false when the file cannot be opened (the reference prints and returns false).
inline bool LoadPosePly(const std::string& path, std::map<uint32_t, std::array<double, 7>>* image_poses) {
  std::ifstream in(path);
  if (!in.is_open()) return false;
  std::string line;
  bool header_done = false;
  uint32_t image_id = 0;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (!header_done) {
      if (line == "end_header") header_done = true;
      continue;
    }
    ++image_id;
    bool has_nan = false;
    {
      std::stringstream tok(line);
      std::string s;
      while (tok >> s)
        if (s == "nan") { has_nan = true; break; }
    }
    if (has_nan) continue;
    std::stringstream ss(line);
    std::vector<double> v;
    double d;
    while (ss >> d) v.push_back(d);
    if (v.size() < 6) continue;   // the reference would index out of bounds here; a short line is skipped
    image_poses->emplace(image_id, LidarPoseToColmap(v.data()));
  }
  return true;
}

}  // namespace colmap_hip
