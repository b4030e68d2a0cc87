// ba_problem.h -- host-side mirror of the reference's problem assembly, producing the flat pcd_ba_desc.
//
// Restates which residual blocks exist and which parameter blocks are constant, following
//   src/optim/bundle_adjustment.cc:601-682   SetUp / SetUpLocalByLidar / SetUpGlobalByLidar / SetUpAdjustWholeMapByLidar
//   :694-806   AddImageInSphereToProblem     :814-919  AddImageToProblem
//   :927-985   AddPointToProblem             :993-1040 AddLidarToProblem
//   :1047-1100 ParameterizeCameras           :1107-1131 ParameterizePoints
//   src/optim/bundle_adjustment.h:52-116     BundleAdjustmentOptions defaults of THIS fork
//   (refine_focal_length / principal_point / extra_params = false, lidar weights 1 / 100 / 1000)
// The scene containers (Camera, Image, Point3D, Track) are minimal stand-ins for src/base/* with the
// same accessors the assembly code uses; ids are arbitrary 32/64-bit integers as in the reference.
// After BundleAdjusterHip::SetUp*(), Evaluate() runs the HIP kernels; Ceres (or any solver) consumes the
// blocks.  Pure host code up to Evaluate(): the structure logic is unit-tested without a GPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/pcdhip.h"
#include "lidar_hip.h"

namespace colmap_hip {

typedef uint32_t camera_t;
typedef uint32_t image_t;
typedef uint64_t point3D_t;
typedef uint32_t point2D_t;
const point3D_t kInvalidPoint3DId = ~0ull;

struct Camera {
  int model_id = 0;
  std::vector<double> params;
  size_t width = 0, height = 0;
};
struct Point2D {
  double xy[2] = {0, 0};
  point3D_t point3D_id = kInvalidPoint3DId;
  bool HasPoint3D() const { return point3D_id != kInvalidPoint3DId; }
};
struct Image {
  camera_t camera_id = 0;
  double qvec[4] = {1, 0, 0, 0};
  double tvec[3] = {0, 0, 0};
  std::vector<Point2D> points2D;
  void NormalizeQvec() {  // base/image.cc: qvec /= norm (identity if zero)
    const double n = std::sqrt(qvec[0] * qvec[0] + qvec[1] * qvec[1] + qvec[2] * qvec[2] + qvec[3] * qvec[3]);
    if (n == 0) { qvec[0] = 1; qvec[1] = qvec[2] = qvec[3] = 0; }
    else for (double& v : qvec) v /= n;
  }
};
struct TrackElement { image_t image_id; point2D_t point2D_idx; };
struct Point3D {
  double xyz[3] = {0, 0, 0};
  std::vector<TrackElement> track;
  int global_opt_num = 0;      // base/point3d.h:82-86
  bool if_in_sphere = false;   // base/point3d.h:156-168
};
struct Reconstruction {
  std::unordered_map<camera_t, Camera> cameras;
  std::unordered_map<image_t, Image> images;
  std::unordered_map<point3D_t, Point3D> points3D;
};

struct BundleAdjustmentOptions {  // optim/bundle_adjustment.h:52-91 (this fork's defaults)
  bool if_add_lidar_constraint = true;
  double proj_lidar_constraint_weight = 1.0;
  double icp_lidar_constraint_weight = 100.0;
  double icp_ground_lidar_constraint_weight = 1000.0;
  int loss_function_type = PCD_LOSS_TRIVIAL;
  double loss_function_scale = 1.0;
  bool refine_focal_length = false, refine_principal_point = false, refine_extra_params = false;
  bool refine_extrinsics = true;
};

class BundleAdjustmentConfig {  // optim/bundle_adjustment.h:118-200, .cc:76-233
 public:
  void AddImage(image_t id) { image_ids_.insert(id); }
  bool HasImage(image_t id) const { return image_ids_.count(id) > 0; }
  void SetConstantCamera(camera_t id) { constant_camera_ids_.insert(id); }
  bool IsConstantCamera(camera_t id) const { return constant_camera_ids_.count(id) > 0; }
  void SetConstantPose(image_t id) { constant_poses_.insert(id); }
  bool HasConstantPose(image_t id) const { return constant_poses_.count(id) > 0; }
  void SetConstantTvec(image_t id, const std::vector<int>& idxs) { constant_tvecs_[id] = idxs; }
  bool HasConstantTvec(image_t id) const { return constant_tvecs_.count(id) > 0; }
  const std::vector<int>& ConstantTvec(image_t id) const { return constant_tvecs_.at(id); }
  void AddVariablePoint(point3D_t id) { variable_point3D_ids_.insert(id); }
  void AddConstantPoint(point3D_t id) { constant_point3D_ids_.insert(id); }
  bool HasVariablePoint(point3D_t id) const { return variable_point3D_ids_.count(id) > 0; }
  bool HasConstantPoint(point3D_t id) const { return constant_point3D_ids_.count(id) > 0; }
  void AddLidarPoint(point3D_t id, const LidarPoint& lp) { lidar_maps_[id] = lp; }
  size_t NumImages() const { return image_ids_.size(); }
  const std::unordered_set<image_t>& Images() const { return image_ids_; }
  const std::unordered_set<point3D_t>& VariablePoints() const { return variable_point3D_ids_; }
  const std::unordered_set<point3D_t>& ConstantPoints() const { return constant_point3D_ids_; }

  // optim/bundle_adjustment.cc:100-147 NumResiduals
  size_t NumResiduals(const Reconstruction& rec) const {
    size_t num_observations = 0;
    for (image_t id : image_ids_)
      for (const Point2D& p : rec.images.at(id).points2D) num_observations += p.HasPoint3D();
    auto extra = [&](point3D_t pid) {
      size_t n = 0;
      for (const TrackElement& te : rec.points3D.at(pid).track) n += !HasImage(te.image_id);
      return n;
    };
    for (point3D_t pid : variable_point3D_ids_) num_observations += extra(pid);
    for (point3D_t pid : constant_point3D_ids_) num_observations += extra(pid);
    return 2 * num_observations;
  }

  std::unordered_map<point3D_t, LidarPoint> lidar_maps_;

 private:
  std::unordered_set<camera_t> constant_camera_ids_;
  std::unordered_set<image_t> image_ids_, constant_poses_;
  std::unordered_set<point3D_t> variable_point3D_ids_, constant_point3D_ids_;
  std::unordered_map<image_t, std::vector<int>> constant_tvecs_;
};

class BundleAdjusterHip {
 public:
  enum class OptimazePhrase { Local, Global, WholeMap, NoLidar };  // (sic) optim/bundle_adjustment.h:205

  BundleAdjusterHip(const BundleAdjustmentOptions& options, const BundleAdjustmentConfig& config)
      : options_(options), config_(config) {}
  ~BundleAdjusterHip() { pcd_ba_destroy(ba_); }

  // ---- assembly (host) -------------------------------------------------------------------------
  void SetUp(Reconstruction* rec, OptimazePhrase phrase) {
    Clear();
    const bool lidar = options_.if_add_lidar_constraint && phrase != OptimazePhrase::NoLidar;
    for (image_t id : Sorted(config_.Images())) {
      if (lidar && phrase == OptimazePhrase::Global) AddImageToProblem(id, rec, /*in_sphere_only=*/true);
      else AddImageToProblem(id, rec, false);
    }
    for (point3D_t pid : Sorted(config_.VariablePoints())) AddPointToProblem(pid, rec);
    if (lidar)
      for (const auto& kv : SortedMap(config_.lidar_maps_)) AddLidarToProblem(kv.first, kv.second, rec);
    if (!(lidar && phrase == OptimazePhrase::WholeMap))   // SetUpAdjustWholeMapByLidar has no such loop (:664-682)
      for (point3D_t pid : Sorted(config_.ConstantPoints())) AddPointToProblem(pid, rec);
    ParameterizeCameras();
    ParameterizePoints(rec);
  }

  size_t NumResiduals() const { return 2 * obs_image_.size() + lidar_point_.size(); }
  // ceres::Solver::Summary::num_residuals_reduced: residual blocks whose parameter blocks are all constant drop out
  size_t NumResidualsReduced() const {
    size_t n = 0;
    for (size_t o = 0; o < obs_image_.size(); ++o) {
      const int im = obs_image_[o];
      if (!image_const_pose_[im] || !point_const_[obs_point_[o]] || CameraVariable(image_cam_[im])) n += 2;
    }
    for (int p : lidar_point_) n += point_const_[p] ? 0 : 1;
    return n;
  }
  // ceres::Solver::Summary::num_effective_parameters_reduced: tangent sizes of the variable parameter blocks --
  // 3 (quaternion) + 3 - #constant tvec entries per variable pose, 3 per variable point, the optimised subset of
  // each variable camera (ParameterizeCameras)
  size_t NumEffectiveParameters() const {
    size_t n = 0;
    for (size_t i = 0; i < poses_.size() / 7; ++i)
      if (image_used_[i] && !image_const_pose_[i]) n += 3 + 3 - __builtin_popcount(image_const_tvec_[i]);
    for (size_t p = 0; p < points_.size() / 3; ++p) n += point_const_[p] ? 0 : 3;
    for (uint8_t v : cam_refine_) n += v;
    return n;
  }
  bool CameraVariable(int cam_idx) const {
    const int k = pcd_camera_num_params(cam_model_[cam_idx]);
    for (int j = 0; j < k; ++j)
      if (cam_refine_[cam_off_[cam_idx] + j]) return true;
    return false;
  }
  size_t NumConstantPoints() const { size_t n = 0; for (uint8_t c : point_const_) n += c; return n; }

  // ---- evaluation (HIP) ------------------------------------------------------------------------
  bool Create(int device = 0) {
    if (NumResiduals() == 0) return false;                // Solve returns false (:489-491)
    pcd_ba_destroy(ba_);
    ba_ = nullptr;
    pcd_ba_desc d{};
    d.device = device;
    d.num_cameras = (int32_t)cam_model_.size(); d.cam_model = cam_model_.data(); d.cam_param_offset = cam_off_.data();
    d.cam_params = cam_params_.data(); d.cam_params_len = cam_params_.size();
    d.num_images = (int32_t)(poses_.size() / 7); d.poses = poses_.data(); d.image_camera = image_cam_.data();
    d.image_const_pose = image_const_pose_.data(); d.image_const_tvec = image_const_tvec_.data();
    d.num_points = (int32_t)(points_.size() / 3); d.points = points_.data(); d.point_const = point_const_.data();
    d.num_obs = obs_image_.size(); d.obs_image = obs_image_.data(); d.obs_point = obs_point_.data(); d.obs_xy = obs_xy_.data();
    d.num_lidar = lidar_point_.size(); d.lidar_point = lidar_point_.data(); d.lidar_abcd = lidar_abcd_.data();
    d.lidar_weight = lidar_w_.data();
    d.loss_type = options_.loss_function_type; d.loss_scale = options_.loss_function_scale;
    bool any_refined = false;
    for (uint8_t v : cam_refine_) any_refined |= v != 0;
    d.camera_refine = any_refined ? cam_refine_.data() : nullptr;   // NULL: intrinsics constant (the fork's default)
    return pcd_ba_create(&d, &ba_) == PCD_OK;
  }
  pcd_ba* handle() const { return ba_; }

  // flat arrays (also what a solver scatters back into the Reconstruction)
  std::vector<int32_t> cam_model_, cam_off_, image_cam_, obs_image_, obs_point_, lidar_point_;
  std::vector<double> cam_params_, poses_, points_, obs_xy_, lidar_abcd_, lidar_w_;
  std::vector<uint8_t> image_const_pose_, image_const_tvec_, point_const_, image_used_;
  std::vector<uint8_t> cam_refine_;      // per entry of cam_params_: 1 = optimised (pcd_ba_desc.camera_refine)
  std::vector<camera_t> camera_ids_;     // flat index -> id
  std::vector<image_t> image_ids_;       // flat index -> id
  std::vector<point3D_t> point_ids_;

 private:
  template <typename S>
  static std::vector<typename S::value_type> Sorted(const S& s) {   // deterministic order (hash order is not)
    std::vector<typename S::value_type> v(s.begin(), s.end());
    std::sort(v.begin(), v.end());
    return v;
  }
  template <typename M>
  static std::map<typename M::key_type, typename M::mapped_type> SortedMap(const M& m) {
    return std::map<typename M::key_type, typename M::mapped_type>(m.begin(), m.end());
  }
  void Clear() {
    cam_model_.clear(); cam_off_.clear(); image_cam_.clear(); obs_image_.clear(); obs_point_.clear();
    lidar_point_.clear(); cam_params_.clear(); poses_.clear(); points_.clear(); obs_xy_.clear();
    lidar_abcd_.clear(); lidar_w_.clear(); image_const_pose_.clear(); image_const_tvec_.clear();
    point_const_.clear(); image_used_.clear(); image_ids_.clear(); point_ids_.clear();
    cam_refine_.clear(); camera_ids_.clear();
    cam_index_.clear(); image_index_.clear(); point_index_.clear(); point3D_num_observations_.clear();
  }
  int CameraIndex(camera_t id, const Reconstruction* rec) {
    auto it = cam_index_.find(id);
    if (it != cam_index_.end()) return it->second;
    const Camera& c = rec->cameras.at(id);
    const int idx = (int)cam_model_.size();
    cam_index_[id] = idx;
    camera_ids_.push_back(id);
    cam_model_.push_back(c.model_id);
    cam_off_.push_back((int32_t)cam_params_.size());
    cam_params_.insert(cam_params_.end(), c.params.begin(), c.params.end());
    return idx;
  }
  // const_pose_block: the residual blocks of this image use the constant-pose functor
  int ImageIndex(image_t id, Reconstruction* rec, bool const_pose_block) {
    auto it = image_index_.find(id);
    if (it != image_index_.end()) return it->second;
    const Image& im = rec->images.at(id);
    const int idx = (int)image_cam_.size();
    image_index_[id] = idx;
    image_ids_.push_back(id);
    image_cam_.push_back(CameraIndex(im.camera_id, rec));
    poses_.insert(poses_.end(), im.qvec, im.qvec + 4);
    poses_.insert(poses_.end(), im.tvec, im.tvec + 3);
    image_const_pose_.push_back(const_pose_block ? 1 : 0);
    uint8_t mask = 0;
    if (!const_pose_block && config_.HasConstantTvec(id))
      for (int k : config_.ConstantTvec(id)) mask |= (uint8_t)(1u << k);   // SetSubsetManifold(3, idxs) :912-915
    image_const_tvec_.push_back(mask);
    image_used_.push_back(0);
    return idx;
  }
  int PointIndex(point3D_t id, const Reconstruction* rec) {
    auto it = point_index_.find(id);
    if (it != point_index_.end()) return it->second;
    const Point3D& p = rec->points3D.at(id);
    const int idx = (int)point_const_.size();
    point_index_[id] = idx;
    point_ids_.push_back(id);
    points_.insert(points_.end(), p.xyz, p.xyz + 3);
    point_const_.push_back(0);
    return idx;
  }
  void AddObservation(int image_idx, int point_idx, const double xy[2]) {
    obs_image_.push_back(image_idx);
    obs_point_.push_back(point_idx);
    obs_xy_.push_back(xy[0]);
    obs_xy_.push_back(xy[1]);
    image_used_[image_idx] = 1;
  }
  // :814-919 and :694-806 (in_sphere_only skips points with !IfInSphere(), :734)
  void AddImageToProblem(image_t image_id, Reconstruction* rec, bool in_sphere_only) {
    Image& image = rec->images.at(image_id);
    image.NormalizeQvec();                                                                     // :823
    const bool constant_pose = !options_.refine_extrinsics || config_.HasConstantPose(image_id);  // :831
    const int ii = ImageIndex(image_id, rec, constant_pose);
    for (const Point2D& p2 : image.points2D) {
      if (!p2.HasPoint3D()) continue;
      if (in_sphere_only && !rec->points3D.at(p2.point3D_id).if_in_sphere) continue;
      point3D_num_observations_[p2.point3D_id] += 1;
      AddObservation(ii, PointIndex(p2.point3D_id, rec), p2.xy);
    }
  }
  // :927-985: observations of the point from images outside the config -> constant-pose blocks
  void AddPointToProblem(point3D_t pid, Reconstruction* rec) {
    const Point3D& p = rec->points3D.at(pid);
    if (point3D_num_observations_[pid] == p.track.size()) return;
    for (const TrackElement& te : p.track) {
      if (config_.HasImage(te.image_id)) continue;
      point3D_num_observations_[pid] += 1;
      Image& image = rec->images.at(te.image_id);
      if (cam_index_.count(image.camera_id) == 0) config_.SetConstantCamera(image.camera_id);   // :951-954
      const int ii = ImageIndex(te.image_id, rec, /*const_pose_block=*/true);
      AddObservation(ii, PointIndex(pid, rec), image.points2D.at(te.point2D_idx).xy);
    }
  }
  // :993-1040
  void AddLidarToProblem(point3D_t pid, const LidarPoint& lp, const Reconstruction* rec) {
    for (int i = 0; i < 4; ++i)
      if (std::isnan(lp.abcd[i])) return;                                                       // :1005-1009
    double w;
    if (lp.type == LidarPointType::Proj) w = options_.proj_lidar_constraint_weight;            // :1013-1028
    else if (lp.type == LidarPointType::Icp) w = options_.icp_lidar_constraint_weight;
    else w = options_.icp_ground_lidar_constraint_weight;
    lidar_point_.push_back(PointIndex(pid, rec));
    lidar_abcd_.insert(lidar_abcd_.end(), lp.abcd.begin(), lp.abcd.end());
    lidar_w_.push_back(w);
  }
  // :1047-1100: which camera parameters are optimised
  void ParameterizeCameras() {
    const bool constant_camera =
        !options_.refine_focal_length && !options_.refine_principal_point && !options_.refine_extra_params;
    cam_refine_.assign(cam_params_.size(), 0);
    for (size_t c = 0; c < cam_model_.size(); ++c) {
      if (constant_camera || config_.IsConstantCamera(camera_ids_[c])) continue;   // SetParameterBlockConstant
      uint8_t group[PCD_CAM_JAC_STRIDE];
      if (pcd_camera_param_groups(cam_model_[c], group) != PCD_OK) continue;
      const int k = pcd_camera_num_params(cam_model_[c]);
      for (int j = 0; j < k; ++j) {                                                // SubsetManifold for the rest
        const bool refine = group[j] == 0 ? options_.refine_focal_length
                          : group[j] == 1 ? options_.refine_principal_point : options_.refine_extra_params;
        cam_refine_[cam_off_[c] + j] = refine ? 1 : 0;
      }
    }
  }
  // :1107-1131
  void ParameterizePoints(const Reconstruction* rec) {
    for (const auto& kv : point3D_num_observations_) {
      auto it = point_index_.find(kv.first);
      if (it == point_index_.end()) continue;
      if (rec->points3D.at(kv.first).track.size() > kv.second) point_const_[it->second] = 1;
    }
    for (point3D_t pid : config_.ConstantPoints()) {
      auto it = point_index_.find(pid);
      if (it != point_index_.end()) point_const_[it->second] = 1;
    }
  }

  const BundleAdjustmentOptions options_;
  BundleAdjustmentConfig config_;
  pcd_ba* ba_ = nullptr;
  std::unordered_map<camera_t, int> cam_index_;
  std::unordered_map<image_t, int> image_index_;
  std::unordered_map<point3D_t, int> point_index_;
  std::unordered_map<point3D_t, size_t> point3D_num_observations_;
};

}  // namespace colmap_hip
