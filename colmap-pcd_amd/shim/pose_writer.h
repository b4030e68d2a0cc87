// pose_writer.h -- registered image poses to a "pose.ply" (SURVEY.md 8f row N4), host code.
//
// Mirrors MainWindow::SaveImagePoses (ui/main_window.cc:1078-1182), the counterpart of LoadPose (pose_reader.h):
// one line per image id 1 .. image_num; an image without a pose in the reconstruction gives the line
// "nan nan nan nan nan nan"; otherwise the world-to-camera pose {t_cw, q_cw} is turned into the camera's pose in
// the world and written in the LiDAR frame (x front, y left, z up):
//   R_wc = R(q_cw)^T, t_wc = -R_wc t_cw;   (y, x, z) Euler angles of R_wc as Eigen's eulerAngles(1, 0, 2) returns them;
//   roll = e[2], pitch = -e[1], yaw = -e[0];  a pitch outside [-pi/2, pi/2] is folded (roll + pi, pi - pitch, yaw + pi);
//   every angle is wrapped once into [-pi, pi];  x = t_wc.z, y = -t_wc.x, z = -t_wc.y;
// and all six numbers are streamed as float with the stream's default format (6 significant digits).
// Eigen is not part of this build: QuaternionToRotation restates Eigen::Quaternion::toRotationMatrix (no
// normalisation, as `quaternion.matrix()` in the reference) and EulerYXZ restates MatrixBase::eulerAngles for the
// axes (1, 0, 2) (Eigen 3.3 / 3.4: first angle in [0, pi], the other two in [-pi, pi]).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <map>
#include <string>

namespace colmap_hip {

inline void QuaternionToRotation(const double q[4] /*w x y z*/, double m[3][3]) {
  const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
  const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
  const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  m[0][0] = 1 - (tyy + tzz); m[0][1] = txy - twz;       m[0][2] = txz + twy;
  m[1][0] = txy + twz;       m[1][1] = 1 - (txx + tzz); m[1][2] = tyz - twx;
  m[2][0] = txz - twy;       m[2][1] = tyz + twx;       m[2][2] = 1 - (txx + tyy);
}

// R = Ry(e[0]) Rx(e[1]) Rz(e[2]) with e[0] in [0, pi]: eulerAngles(1, 0, 2), i.e. i = 1, j = 0, k = 2, odd permutation
inline void EulerYXZ(const double m[3][3], double e[3]) {
  const double kPi = 3.141592653589793238462643383279502884;
  e[0] = std::atan2(m[0][2], m[2][2]);
  const double c2 = std::sqrt(m[1][1] * m[1][1] + m[1][0] * m[1][0]);
  if (e[0] < 0) {
    e[0] += kPi;
    e[1] = std::atan2(-m[1][2], -c2);
  } else {
    e[1] = std::atan2(-m[1][2], c2);
  }
  const double s1 = std::sin(e[0]), c1 = std::cos(e[0]);
  e[2] = std::atan2(s1 * m[2][1] - c1 * m[0][1], c1 * m[0][0] - s1 * m[2][0]);
}

// {t_cw, q_cw (w x y z)} -> x y z roll pitch yaw in the LiDAR frame (doubles; the file holds them as float)
inline std::array<double, 6> ColmapPoseToLidar(const std::array<double, 7>& pose) {
  double Rcw[3][3], Rwc[3][3];
  QuaternionToRotation(&pose[3], Rcw);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rwc[i][j] = Rcw[j][i];
  double t_wc[3];
  for (int i = 0; i < 3; ++i) t_wc[i] = -(Rwc[i][0] * pose[0] + Rwc[i][1] * pose[1] + Rwc[i][2] * pose[2]);
  double e[3];
  EulerYXZ(Rwc, e);
  double roll = e[2], pitch = -e[1], yaw = -e[0];
  if (pitch < -M_PI / 2 || pitch > M_PI / 2) {
    roll += M_PI;
    pitch = M_PI - pitch;
    yaw += M_PI;
  }
  if (roll < -M_PI) roll += 2 * M_PI;
  else if (roll > M_PI) roll -= 2 * M_PI;
  if (pitch < -M_PI) pitch += 2 * M_PI;
  else if (pitch > M_PI) pitch -= 2 * M_PI;
  if (yaw < -M_PI) yaw += 2 * M_PI;
  else if (yaw > M_PI) yaw -= 2 * M_PI;
  return {t_wc[2], -t_wc[0], -t_wc[1], roll, pitch, yaw};
}

// image_num = OriginImagesNum() of the reference: lines for the ids 1 .. image_num.  poses: image id -> {t_cw, q_cw},
// the layout pose_reader.h produces.  Returns false when the file cannot be opened (the reference prints and returns).
inline bool SaveImagePosesPly(const std::string& path, int image_num,
                              const std::map<uint32_t, std::array<double, 7>>& poses) {
  std::ofstream out(path, std::ios::out);
  if (!out) return false;
  out << "ply" << std::endl
      << "format ascii 1.0" << std::endl
      << "element vertex " << image_num << std::endl
      << "property float x" << std::endl
      << "property float y" << std::endl
      << "property float z" << std::endl
      << "property float roll" << std::endl
      << "property float pitch" << std::endl
      << "property float yaw" << std::endl
      << "end_header" << std::endl;
  for (int i = 1; i <= image_num; ++i) {
    const auto it = poses.find((uint32_t)i);
    if (it == poses.end()) {
      out << "nan nan nan nan nan nan" << std::endl;
      continue;
    }
    const std::array<double, 6> p = ColmapPoseToLidar(it->second);
    out << static_cast<float>(p[0]) << " " << static_cast<float>(p[1]) << " " << static_cast<float>(p[2]) << " "
        << static_cast<float>(p[3]) << " " << static_cast<float>(p[4]) << " " << static_cast<float>(p[5]) << std::endl;
  }
  return static_cast<bool>(out);
}

}  // namespace colmap_hip
