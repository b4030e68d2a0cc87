// lidar_hip.h -- C++ adapters with the reference's own call-site signatures on top of the C ABI.
//
// Mirrors (names, argument meaning, error behaviour) of:
//   colmap::lidar::PointCloudProcess   src/lidar/ply.h:13-40   (Initialize, SearchNearestNeiborByKdtree, pcd_proj_)
//   colmap::LidarPoint                 src/lidar/lidar_point.h:10-50
//   BundleAdjustmentConfig::MatchClosestLidarPoint          src/optim/bundle_adjustment.cc:358-410
//   the association loops of IncrementalMapper::AdjustGlobalBundleByLidar (sfm/incremental_mapper.cc:1413-1469)
//   and BundleAdjustmentController::Run (controllers/bundle_adjustment.cc:130-185)
// Vector types are template parameters: anything indexable with operator()(int) or operator[] works, so
// Eigen::Vector3d / Eigen::Matrix<double,6,1> drop in unchanged when the reference is built against this.
// Header-only; link with -lpcdhip.  No CPU fallback: every call fails (returns false) without a gfx950 device.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pcdhip.h"
#include "pcd_proj_hip.h"
#include "ply_reader.h"

namespace colmap_hip {

enum class LidarPointType { Proj, Icp, IcpGround };  // lidar/lidar_point.h:9

// value type recorded per associated 3D point (lidar/lidar_point.h:10-50)
struct LidarPoint {
  LidarPointType type = LidarPointType::Icp;
  std::array<double, 3> xyz{};
  std::array<double, 4> abcd{};     // after Normalize(): unit normal + d
  std::array<uint8_t, 3> color{};   // GUI colour chosen at the call site
  double dist = 0.0;                // SetDist (point-to-point distance on the KD-tree paths)
  double angle = 0.0;               // SetAngle
  LidarPointType Type() const { return type; }
  const std::array<double, 3>& LidarXYZ() const { return xyz; }
  const std::array<double, 4>& LidarABCD() const { return abcd; }
  double Dist() const { return dist; }
  double Angle() const { return angle; }
};

namespace lidar {

template <typename V>
inline double vget(const V& v, int i) { return v(i); }
template <typename T, size_t N>
inline double vget(const std::array<T, N>& v, int i) { return v[i]; }
template <typename V>
inline void vset(V& v, int i, double x) { v(i) = x; }
template <typename T, size_t N>
inline void vset(std::array<T, N>& v, int i, double x) { v[i] = x; }

// Replaces lidar::PointCloudProcess: the KD-tree side (cloud_) and the depth-projection side (pcd_proj_).
class PointCloudProcess {
 public:
  explicit PointCloudProcess(const std::string& path = "", int device = 0) : path_(path), device_(device) {}
  ~PointCloudProcess() {
    pcd_proj_.reset();   // the projector refers to the cloud
    pcd_cloud_destroy(cloud_);
  }
  PointCloudProcess(const PointCloudProcess&) = delete;
  PointCloudProcess& operator=(const PointCloudProcess&) = delete;

  // lidar/ply.cc:9-31: load the PLY at path_, transform, index.  Returns false when the file cannot be
  // read (the reference prints and returns false; callers only print, sfm/incremental_mapper.cc:201-205).
  bool Initialize() {
    std::vector<float> xyz, nrm;
    if (!ReadPlyXYZNormal(path_, &xyz, &nrm)) return false;
    return InitializeFromRawCloud(xyz.data(), nrm.data(), xyz.size() / 3);
  }
  // lidar/ply.cc:9-31 with its own signature: also creates pcd_proj_ and builds its submaps (ply.cc:10, 27)
  bool Initialize(const PcdProjectionOptions& pp_options) {
    if (!Initialize()) return false;
    return BuildProjector(pp_options);
  }
  bool BuildProjector(const PcdProjectionOptions& pp_options) {
    pcd_proj_ = std::make_shared<PcdProj>(pp_options);
    return cloud_ && pcd_proj_->BuildSubMap(cloud_);
  }
  std::shared_ptr<PcdProj> pcd_proj_;   // public member as in lidar/ply.h:33; declared before cloud_ is destroyed

  // lidar/ply.cc:9-31 after pcl::io::loadPLYFile: rows as stored in the PLY (LiDAR frame, lidarpt::Point
  // AoS 32 B or two arrays).  Applies the axis swap + NaN filter of ply.cc:33-57 and builds the index.
  // Returns false on failure like the reference (the caller only prints).
  bool InitializeFromRawCloud(const float* xyz, const float* nrm, uint64_t n, bool aos32 = false) {
    pcd_cloud_options o;
    pcd_cloud_options_default(&o);
    o.device = device_;
    o.layout = aos32 ? PCD_LAYOUT_AOS32 : PCD_LAYOUT_XYZ_NRM;
    o.raw_lidar_frame = 1;
    pcd_proj_.reset();
    pcd_cloud_destroy(cloud_);
    cloud_ = nullptr;
    host_nrm_.clear();   // cached normals belong to the old cloud
    return pcd_cloud_create(xyz, nrm, n, &o, &cloud_) == PCD_OK;
  }

  // lidar/ply.h:31, ply.cc:90-107 -- unchanged signature; one query per call (correct, launch-bound)
  template <typename Vec3, typename Vec6>
  bool SearchNearestNeiborByKdtree(const Vec3& point_3d, Vec6& l_pt) {
    const double q[3] = {vget(point_3d, 0), vget(point_3d, 1), vget(point_3d, 2)};
    double l6[6];
    uint8_t ok = 0;
    if (!SearchNearestNeiborBatch(q, 1, l6, &ok) || !ok) return false;
    for (int k = 0; k < 6; ++k) vset(l_pt, k, l6[k]);   // out-param written only on success
    return true;
  }

  // batched form used by the patched loops: l6[i] = (xyz, normal) doubles, ok[i] as the bool above
  bool SearchNearestNeiborBatch(const double* q_xyz, uint64_t n, double* l6, uint8_t* ok) {
    if (!cloud_) return false;
    std::vector<double> xyz(3 * n), abcd(4 * n), dist(n), angle(n);
    std::vector<uint8_t> type(n);
    std::vector<uint32_t> idx(n);
    pcd_assoc_out out{xyz.data(), abcd.data(), type.data(), dist.data(), angle.data(), nullptr, idx.data(), nullptr};
    const double huge = 1e300;  // gate disabled: ok == SearchNearestNeiborByKdtree's return value
    if (pcd_associate(cloud_, q_xyz, n, &huge, 1, PCD_GATE_MAPPER_LOCAL, &out) != PCD_OK) return false;
    if (host_nrm_.empty() && !FetchNormals()) return false;
    for (uint64_t i = 0; i < n; ++i) {
      ok[i] = type[i] != PCD_LIDAR_NONE;
      for (int k = 0; k < 3; ++k) {
        l6[6 * i + k] = xyz[3 * i + k];
        l6[6 * i + 3 + k] = ok[i] ? (double)host_nrm_[3 * (size_t)idx[i] + k] : 0.0;
      }
    }
    return true;
  }

  pcd_cloud* handle() const { return cloud_; }
  uint64_t size() const { return pcd_cloud_size(cloud_); }

 private:
  bool FetchNormals() {
    const uint64_t n = pcd_cloud_size(cloud_);
    host_nrm_.resize(3 * n);
    std::vector<float> xyz(3 * n);
    return pcd_cloud_download(cloud_, xyz.data(), host_nrm_.data()) == PCD_OK;
  }
  std::string path_;
  int device_;
  pcd_cloud* cloud_ = nullptr;
  std::vector<float> host_nrm_;
};

}  // namespace lidar

// The three association loops, batched.  gate_mode selects the call site:
//   PCD_GATE_MAPPER_LOCAL   BundleAdjustmentConfig::MatchClosestLidarPoint (colour green 0,255,0; dist/angle stored)
//   PCD_GATE_MAPPER_GLOBAL  IncrementalMapper::AdjustGlobalBundleByLidar   (colour blue 0,0,255)
//   PCD_GATE_CONTROLLER     BundleAdjustmentController::Run                (colour blue 0,0,255)
// ground points are yellow (255,255,0) at all three.  Returns point3D_id -> LidarPoint for the points the
// reference would have passed to AddLidarPoint (rejected / not found points are absent).
// Flat form for the patched loops: the caller gathers Point3D::XYZ() (and the range schedule) straight into the
// handle's pinned staging buffers through `fill(i, xyz3, &range)`, one H2D / search / epilogue / device-side
// compaction / D2H round trip follows, and the accepted associations come back as a contiguous array of 80-byte
// records (ascending i) in pinned memory valid until the next call on this cloud.  The unordered_map inserts of
// AddLidarPoint stay at the call site:
//   for (k < n_hits) { const pcd_assoc_hit& h = hits[k]; config.AddLidarPoint(ids[h.query], ToLidarPoint(h, gate)); }
template <typename Fill>
inline bool MatchClosestLidarPointsFlat(lidar::PointCloudProcess& pcp, uint64_t n, bool per_point_range, int gate_mode,
                                        Fill fill, const pcd_assoc_hit** hits, uint64_t* n_hits) {
  double *q = nullptr, *mr = nullptr;
  if (!pcp.handle() || pcd_assoc_staging(pcp.handle(), n, &q, &mr) != PCD_OK) return false;
  double r0 = 0.0;
  for (uint64_t i = 0; i < n; ++i) fill(i, q + 3 * i, per_point_range ? mr + i : &r0);
  if (!per_point_range) mr[0] = r0;
  return pcd_associate_staged(pcp.handle(), n, per_point_range ? n : 1, gate_mode, hits, n_hits) == PCD_OK;
}

// LidarPoint of one record, with the colour the call site of `gate_mode` paints it
inline LidarPoint ToLidarPoint(const pcd_assoc_hit& h, int gate_mode) {
  LidarPoint lp;
  lp.type = h.type == PCD_LIDAR_ICP_GROUND ? LidarPointType::IcpGround : LidarPointType::Icp;
  for (int k = 0; k < 3; ++k) lp.xyz[k] = h.lidar_xyz[k];
  for (int k = 0; k < 4; ++k) lp.abcd[k] = h.abcd[k];
  if (lp.type == LidarPointType::IcpGround) lp.color = {255, 255, 0};
  else if (gate_mode == PCD_GATE_MAPPER_LOCAL) lp.color = {0, 255, 0};
  else lp.color = {0, 0, 255};
  lp.dist = h.dist;
  lp.angle = h.angle;
  return lp;
}

inline bool MatchClosestLidarPoints(lidar::PointCloudProcess& pcp, const std::vector<uint64_t>& point3D_ids,
                                    const std::vector<double>& xyz /*3 per point*/,
                                    const std::vector<double>& max_search_range /*1 or n entries*/, int gate_mode,
                                    std::unordered_map<uint64_t, LidarPoint>* lidar_maps) {
  const uint64_t n = point3D_ids.size();
  if (xyz.size() != 3 * n || !pcp.handle()) return false;
  if (gate_mode != PCD_GATE_CONTROLLER && max_search_range.size() != 1 && max_search_range.size() != n) return false;
  const bool per_point = max_search_range.size() == n && n > 1;
  const pcd_assoc_hit* hits = nullptr;
  uint64_t n_hits = 0;
  auto fill = [&](uint64_t i, double* q3, double* r) {
    q3[0] = xyz[3 * i]; q3[1] = xyz[3 * i + 1]; q3[2] = xyz[3 * i + 2];
    if (!max_search_range.empty()) *r = max_search_range[per_point ? i : 0];
  };
  if (!MatchClosestLidarPointsFlat(pcp, n, per_point, gate_mode, fill, &hits, &n_hits)) return false;
  lidar_maps->reserve(lidar_maps->size() + n_hits);
  for (uint64_t k = 0; k < n_hits; ++k) (*lidar_maps)[point3D_ids[hits[k].query]] = ToLidarPoint(hits[k], gate_mode);
  return true;
}

// BundleAdjustmentConfig::MatchVariablePoint2LidarPoint (optim/bundle_adjustment.cc:288-350): among the images
// of the point's track that were projected (PcdProj::SetNewImage(s) -> searched[image_id]), take the candidate
// whose |cos| between (X - lidar point) and the lidar normal is smallest (strict <, track order), and record it
// as a LidarPointType::Proj association: plane normalised (lidar_point.cc:39-50), dist = point-to-PLANE distance
// (lidar_point.cc:21-25; the KD-tree paths store point-to-point instead), angle (lidar_point.cc:32-37), red.
// Host code on a handful of candidates per point, like the reference.  Returns false when no image has one.
template <typename SearchedMap, typename TrackImageIds>
inline bool MatchVariablePoint2LidarPoint(const SearchedMap& searched, uint64_t point3D_id, const double* X,
                                          const TrackImageIds& track_image_ids, LidarPoint* out) {
  double angle = 360;
  std::array<double, 6> best{};
  for (const auto image_id : track_image_ids) {
    const auto it = searched.find(image_id);
    if (it == searched.end()) continue;
    const auto lp = it->second.find(point3D_id);
    if (lp == it->second.end()) continue;
    const std::array<double, 6>& c = lp->second;
    const double v[3] = {X[0] - c[0], X[1] - c[1], X[2] - c[2]};
    const double dot = v[0] * c[3] + v[1] * c[4] + v[2] * c[5];
    const double nn = std::sqrt(c[3] * c[3] + c[4] * c[4] + c[5] * c[5]);
    const double vn = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double a = std::fabs(dot / (nn * vn));
    if (a < angle) {   // NaN (zero normal / X on the lidar point) never wins, as in the reference
      angle = a;
      best = c;
    }
  }
  if (angle == 360) return false;
  LidarPoint p;
  p.type = LidarPointType::Proj;
  for (int k = 0; k < 3; ++k) p.xyz[k] = best[k];
  const double norm = std::sqrt(std::pow(best[3], 2) + std::pow(best[4], 2) + std::pow(best[5], 2));
  const double a = best[3] / norm, b = best[4] / norm, c = best[5] / norm;
  const double d = 0 - a * p.xyz[0] - b * p.xyz[1] - c * p.xyz[2];
  p.abcd = {a, b, c, d};
  p.dist = std::fabs((X[0] * a + X[1] * b + X[2] * c) + d);
  const double w[3] = {X[0] - p.xyz[0], X[1] - p.xyz[1], X[2] - p.xyz[2]};
  p.angle = std::fabs((a * w[0] + b * w[1] + c * w[2]) / std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
  p.color = {255, 0, 0};
  *out = p;
  return true;
}

}  // namespace colmap_hip
