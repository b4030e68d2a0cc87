// ceres_adapter.h -- Ceres keeps the solve, the GPU evaluates: the problem BundleAdjuster::SetUp*ByLidar builds
// (optim/bundle_adjustment.cc:601-682) is handed to Ceres as residual blocks whose CostFunction::Evaluate only COPIES
// what one batched pcd_ba_evaluate computed for all blocks in PrepareForEvaluation.
//
// Replaces, per residual block, the reference's
//   ceres::AutoDiffCostFunction<BundleAdjustmentCostFunction<Model>, 2, 4, 3, 3, K>              (optim/bundle_adjustment.cc:880-893)
//   ceres::AutoDiffCostFunction<BundleAdjustmentConstantPoseCostFunction<Model>, 2, 3, K>        (:858-876, :967-983)
//   ceres::AutoDiffCostFunction<BundleAdjustmentLidarCostFunction, 1, 3>                         (:1031-1037)
// with the same parameter-block order and sizes, so problem_->AddResidualBlock(cost, loss, qvec, tvec, xyz, params)
// and everything after it (loss functions, SetManifold / SetParameterBlockConstant, ceres::Solve, the iteration
// callback of controllers/bundle_adjustment.cc:43-61) stays as it is.  Wiring (controllers keep their code):
//
//   BundleAdjusterHip hip(options, config);  hip.SetUp(reconstruction, phrase);  hip.Create(device);
//   HipEvaluation<Source> cb(&hip, Source(reconstruction));
//   problem_options.evaluation_callback = &cb;                      // ceres::Problem::Options
//   for (o : observations)  problem.AddResidualBlock(cb.ReprojectionBlock(o), loss, <blocks as before>);
//   for (l : lidar terms)   problem.AddResidualBlock(cb.LidarBlock(l), loss, xyz);
//
// Ceres calls PrepareForEvaluation(jacobians, new_point) once per evaluation with the parameter memory (owned by
// Reconstruction, updated in place) holding the evaluation point; Evaluate may then run concurrently from Ceres'
// threads -- it only reads the result buffers.  jacobians == NULL and jacobians[i] == NULL (constant block) are
// honoured as Ceres requires.  COLMAP_PCD_HIP=0 (HipBackendEnabled) keeps the reference's own autodiff blocks.
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <vector>

#if __has_include(<ceres/ceres.h>)
#include <ceres/ceres.h>
#else
#include "ceres_interface_stub.h"
#endif

#include "ba_problem.h"

namespace colmap_hip {

// Run-time switch of the whole drop-in (SURVEY section 5 "Config / flags"): COLMAP_PCD_HIP=0 keeps the reference's CPU
// path (PCL/FLANN KD-tree loops, Ceres autodiff); unset or any other value uses libpcdhip when a gfx950 device exists.
// The call sites test it once: `if (colmap_hip::HipBackendEnabled()) { batched path } else { original loop }`.
inline bool HipBackendEnabled() {
  const char* e = std::getenv("COLMAP_PCD_HIP");
  if (e && e[0] == '0' && e[1] == '\0') return false;
  return pcd_device_count() > 0;
}

// Where the parameter blocks live.  With COLMAP: Qvec = reconstruction->Image(id).Qvec().data(), Tvec likewise,
// XYZ = reconstruction->Point3D(id).XYZ().data(), Params = reconstruction->Camera(id).ParamsData()
// (the pointers handed to AddResidualBlock at optim/bundle_adjustment.cc:825-828, :894).
struct ShimParameterSource {
  Reconstruction* rec;
  explicit ShimParameterSource(Reconstruction* r) : rec(r) {}
  double* Qvec(image_t id) const { return rec->images.at(id).qvec; }
  double* Tvec(image_t id) const { return rec->images.at(id).tvec; }
  double* XYZ(point3D_t id) const { return rec->points3D.at(id).xyz; }
  double* Params(camera_t id) const { return rec->cameras.at(id).params.data(); }
};

struct HipBlockBuffers {   // results of the last PrepareForEvaluation, read by every block: views of the pinned host
  pcd_ba_blocks b{};       // buffers the pcd_ba handle owns (pcd_ba_evaluate_blocks)
  bool have_jacobians = false;
  size_t num_obs = 0;      // reprojection blocks of the problem: the lidar residuals follow their 2 * num_obs entries
};

// one reprojection residual block (variable or constant pose), optim/bundle_adjustment.cc:858-893, :967-983
class HipReprojectionBlock : public ceres::CostFunction {
 public:
  HipReprojectionBlock(const HipBlockBuffers* buf, size_t obs, bool constant_pose, int num_camera_params)
      : buf_(buf), o_(obs), cpose_(constant_pose), K_(num_camera_params) {
    set_num_residuals(2);
    if (!cpose_) { mutable_parameter_block_sizes()->push_back(4); mutable_parameter_block_sizes()->push_back(3); }
    mutable_parameter_block_sizes()->push_back(3);
    mutable_parameter_block_sizes()->push_back(K_);
  }
  bool Evaluate(double const* const*, double* residuals, double** jacobians) const override {
    const pcd_ba_blocks& r = buf_->b;
    residuals[0] = r.residuals[2 * o_];
    residuals[1] = r.residuals[2 * o_ + 1];
    if (!jacobians) return true;
    if (!buf_->have_jacobians) return false;   // Ceres asked for Jacobians the callback was not told to prepare
    int b = 0;
    if (!cpose_) {
      const size_t row = r.pose_row[o_];       // constant-pose blocks have no pose rows (and no pose parameter blocks)
      if (jacobians[b]) std::memcpy(jacobians[b], r.jac_q + 8 * row, 8 * sizeof(double));
      ++b;
      if (jacobians[b]) std::memcpy(jacobians[b], r.jac_t + 6 * row, 6 * sizeof(double));
      ++b;
    }
    if (jacobians[b]) std::memcpy(jacobians[b], r.jac_X + 6 * o_, 6 * sizeof(double));
    ++b;
    if (jacobians[b]) {
      if (!r.jac_cam) return false;
      for (int k = 0; k < 2; ++k)   // device rows have PCD_CAM_JAC_STRIDE columns, Ceres wants 2 x K row-major
        std::memcpy(jacobians[b] + k * K_, r.jac_cam + (2 * o_ + k) * PCD_CAM_JAC_STRIDE, K_ * sizeof(double));
    }
    return true;
  }

 private:
  const HipBlockBuffers* buf_;
  size_t o_;
  bool cpose_;
  int K_;
};

// one point-to-plane residual block, optim/bundle_adjustment.cc:1031-1037
class HipLidarBlock : public ceres::CostFunction {
 public:
  HipLidarBlock(const HipBlockBuffers* buf, size_t l) : buf_(buf), l_(l) {
    set_num_residuals(1);
    mutable_parameter_block_sizes()->push_back(3);
  }
  bool Evaluate(double const* const*, double* residuals, double** jacobians) const override {
    residuals[0] = buf_->b.residuals[2 * buf_->num_obs + l_];
    if (jacobians && jacobians[0]) {
      if (!buf_->have_jacobians) return false;
      std::memcpy(jacobians[0], buf_->b.jac_lidar + 3 * l_, 3 * sizeof(double));
    }
    return true;
  }

 private:
  const HipBlockBuffers* buf_;
  size_t l_;
};

template <typename Source = ShimParameterSource>
class HipEvaluation : public ceres::EvaluationCallback {
 public:
  // `ba` must have been SetUp() and Create()d; `src` gives the parameter memory Ceres optimises in place
  HipEvaluation(BundleAdjusterHip* ba, const Source& src) : ba_(ba), src_(src) {
    for (uint8_t v : ba_->cam_refine_) cameras_variable_ |= v != 0;
    buf_.num_obs = ba_->obs_image_.size();
  }

  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    ok_ = true;
    if (new_evaluation_point) {
      // gather the evaluation point from the solver's parameter blocks (flat order of ba_problem.h)
      for (size_t i = 0; i < ba_->image_ids_.size(); ++i) {
        std::memcpy(&ba_->poses_[7 * i], src_.Qvec(ba_->image_ids_[i]), 4 * sizeof(double));
        std::memcpy(&ba_->poses_[7 * i + 4], src_.Tvec(ba_->image_ids_[i]), 3 * sizeof(double));
      }
      for (size_t p = 0; p < ba_->point_ids_.size(); ++p)
        std::memcpy(&ba_->points_[3 * p], src_.XYZ(ba_->point_ids_[p]), 3 * sizeof(double));
      ok_ &= pcd_ba_set_parameters(ba_->handle(), ba_->poses_.data(), ba_->points_.data()) == PCD_OK;
      if (cameras_variable_) {
        for (size_t c = 0; c < ba_->camera_ids_.size(); ++c)
          std::memcpy(&ba_->cam_params_[ba_->cam_off_[c]], src_.Params(ba_->camera_ids_[c]),
                      pcd_camera_num_params(ba_->cam_model_[c]) * sizeof(double));
        ok_ &= pcd_ba_set_camera_parameters(ba_->handle(), ba_->cam_params_.data()) == PCD_OK;
      }
    }
    // ONE batch for every residual block; results in pinned memory of the handle (constant cameras: Ceres passes
    // NULL for the camera block, so its Jacobian is not even computed)
    ok_ &= pcd_ba_evaluate_blocks(ba_->handle(), evaluate_jacobians ? 1 : 0, cameras_variable_ ? 1 : 0, &buf_.b) == PCD_OK;
    bytes_d2h_ += buf_.b.bytes_d2h;
    buf_.have_jacobians = evaluate_jacobians && ok_;
    ++num_evaluations_;
  }

  // blocks in the order of ba_problem.h's flat arrays (= the order of the reference's AddResidualBlock calls);
  // ownership passes to the caller (Ceres takes it in AddResidualBlock)
  ceres::CostFunction* ReprojectionBlock(size_t o) const {
    const int im = ba_->obs_image_[o];
    return new HipReprojectionBlock(&buf_, o, ba_->image_const_pose_[im] != 0,
                                    pcd_camera_num_params(ba_->cam_model_[ba_->image_cam_[im]]));
  }
  ceres::CostFunction* LidarBlock(size_t l) const { return new HipLidarBlock(&buf_, l); }

  bool ok() const { return ok_; }
  size_t num_evaluations() const { return num_evaluations_; }
  uint64_t bytes_d2h() const { return bytes_d2h_; }
  const HipBlockBuffers& buffers() const { return buf_; }

 private:
  BundleAdjusterHip* ba_;
  Source src_;
  HipBlockBuffers buf_;
  bool cameras_variable_ = false, ok_ = true;
  size_t num_evaluations_ = 0;
  uint64_t bytes_d2h_ = 0;
};

// The same route WITHOUT a mirror of the scene: the reference's own SetUp* / AddImageToProblem / AddPointToProblem /
// AddLidarToProblem keep walking colmap::Reconstruction and call AddResidualBlock exactly as they do today; only the
// cost-function object they create comes from this recorder (integration/colmap-pcd-hip.patch), which notes, in creation
// order, the addresses of the parameter blocks the block is added with.  Addresses identify images (qvec), points (xyz)
// and cameras (params) -- they are what Ceres itself keys parameter blocks on.  Finalize() (after SetUp*, before
// ceres::Solve) builds the flat pcd_ba from those notes; from then on it is HipEvaluation's loop: gather the evaluation
// point through the addresses, ONE pcd_ba_evaluate_blocks, every block's Evaluate copies.  Which blocks are constant is
// Ceres' knowledge (SetParameterBlockConstant / manifolds): it passes jacobians[i] == NULL for them.
class HipBlockRecorder : public ceres::EvaluationCallback {
 public:
  ~HipBlockRecorder() override { pcd_ba_destroy(ba_); }

  // optim/bundle_adjustment.cc:858-893 / :967-983.  constant_pose: the block is added WITHOUT qvec / tvec (the
  // reference's BundleAdjustmentConstantPoseCostFunction copies the pose at creation; here it is read at Finalize()).
  ceres::CostFunction* AddReprojection(int camera_model_id, double* qvec, double* tvec, double* xyz, double* params,
                                       const double* xy, bool constant_pose) {
    const int cam = Index(&cam_index_, params, &cam_ptr_);
    if (cam == (int)cam_model_.size()) {
      cam_model_.push_back(camera_model_id);
      cam_off_.push_back((int32_t)cam_len_);
      cam_len_ += (size_t)pcd_camera_num_params(camera_model_id);
    }
    const int im = Index(&image_index_, qvec, &qvec_ptr_);
    if (im == (int)image_cam_.size()) {
      tvec_ptr_.push_back(tvec);
      image_cam_.push_back(cam);
      image_const_pose_.push_back(constant_pose ? 1 : 0);
    }
    const int pt = Index(&point_index_, xyz, &xyz_ptr_);
    obs_image_.push_back(im);
    obs_point_.push_back(pt);
    obs_xy_.push_back(xy[0]);
    obs_xy_.push_back(xy[1]);
    return new HipReprojectionBlock(&buf_, obs_image_.size() - 1, constant_pose, pcd_camera_num_params(camera_model_id));
  }
  // optim/bundle_adjustment.cc:1031-1037
  ceres::CostFunction* AddLidar(double* xyz, const double* abcd, double weight) {
    lidar_point_.push_back(Index(&point_index_, xyz, &xyz_ptr_));
    lidar_abcd_.insert(lidar_abcd_.end(), abcd, abcd + 4);
    lidar_w_.push_back(weight);
    return new HipLidarBlock(&buf_, lidar_point_.size() - 1);
  }
  size_t NumResiduals() const { return 2 * obs_image_.size() + lidar_point_.size(); }

  // cameras_variable: some camera parameter block is optimised (ParameterizeCameras, :1047-1100): its Jacobians are
  // computed and copied only then.  Returns false when there is nothing to evaluate or the device refuses.
  bool Finalize(int device, bool cameras_variable) {
    if (NumResiduals() == 0) return false;
    cameras_variable_ = cameras_variable;
    buf_.num_obs = obs_image_.size();
    poses_.resize(7 * qvec_ptr_.size());
    points_.resize(3 * xyz_ptr_.size());
    cam_params_.resize(cam_len_);
    Gather(true);
    std::vector<uint8_t> const_tvec(qvec_ptr_.size(), 0), point_const(xyz_ptr_.size(), 0);
    std::vector<uint8_t> refine(cam_len_, cameras_variable ? 1 : 0);
    pcd_ba_desc d{};
    d.device = device;
    d.num_cameras = (int32_t)cam_model_.size(); d.cam_model = cam_model_.data(); d.cam_param_offset = cam_off_.data();
    d.cam_params = cam_params_.data(); d.cam_params_len = cam_params_.size();
    d.num_images = (int32_t)qvec_ptr_.size(); d.poses = poses_.data(); d.image_camera = image_cam_.data();
    d.image_const_pose = image_const_pose_.data(); d.image_const_tvec = const_tvec.data();
    d.num_points = (int32_t)xyz_ptr_.size(); d.points = points_.data(); d.point_const = point_const.data();
    d.num_obs = obs_image_.size(); d.obs_image = obs_image_.data(); d.obs_point = obs_point_.data(); d.obs_xy = obs_xy_.data();
    d.num_lidar = lidar_point_.size(); d.lidar_point = lidar_point_.data(); d.lidar_abcd = lidar_abcd_.data();
    d.lidar_weight = lidar_w_.data();
    d.loss_type = PCD_LOSS_TRIVIAL; d.loss_scale = 1.0;    // Ceres applies the loss function to the raw blocks itself
    d.camera_refine = cameras_variable ? refine.data() : nullptr;
    pcd_ba_destroy(ba_);
    ba_ = nullptr;
    return pcd_ba_create(&d, &ba_) == PCD_OK;
  }

  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    ok_ = ba_ != nullptr;
    if (!ok_) return;
    if (new_evaluation_point) {
      Gather(cameras_variable_);
      ok_ &= pcd_ba_set_parameters(ba_, poses_.data(), points_.data()) == PCD_OK;
      if (cameras_variable_) ok_ &= pcd_ba_set_camera_parameters(ba_, cam_params_.data()) == PCD_OK;
    }
    ok_ &= pcd_ba_evaluate_blocks(ba_, evaluate_jacobians ? 1 : 0, cameras_variable_ ? 1 : 0, &buf_.b) == PCD_OK;
    buf_.have_jacobians = evaluate_jacobians && ok_;
    ++num_evaluations_;
  }
  bool ok() const { return ok_; }
  size_t num_evaluations() const { return num_evaluations_; }
  pcd_ba* handle() const { return ba_; }

 private:
  static int Index(std::unordered_map<const double*, int>* index, double* ptr, std::vector<double*>* ptrs) {
    const auto it = index->find(ptr);
    if (it != index->end()) return it->second;
    const int idx = (int)ptrs->size();
    (*index)[ptr] = idx;
    ptrs->push_back(ptr);
    return idx;
  }
  void Gather(bool cameras) {
    for (size_t i = 0; i < qvec_ptr_.size(); ++i) {
      std::memcpy(&poses_[7 * i], qvec_ptr_[i], 4 * sizeof(double));
      std::memcpy(&poses_[7 * i + 4], tvec_ptr_[i], 3 * sizeof(double));
    }
    for (size_t p = 0; p < xyz_ptr_.size(); ++p) std::memcpy(&points_[3 * p], xyz_ptr_[p], 3 * sizeof(double));
    if (cameras)
      for (size_t c = 0; c < cam_ptr_.size(); ++c)
        std::memcpy(&cam_params_[cam_off_[c]], cam_ptr_[c], pcd_camera_num_params(cam_model_[c]) * sizeof(double));
  }

  std::unordered_map<const double*, int> cam_index_, image_index_, point_index_;
  std::vector<double*> cam_ptr_, qvec_ptr_, tvec_ptr_, xyz_ptr_;
  std::vector<int32_t> cam_model_, cam_off_, image_cam_, obs_image_, obs_point_, lidar_point_;
  std::vector<uint8_t> image_const_pose_;
  std::vector<double> cam_params_, poses_, points_, obs_xy_, lidar_abcd_, lidar_w_;
  size_t cam_len_ = 0, num_evaluations_ = 0;
  HipBlockBuffers buf_;
  pcd_ba* ba_ = nullptr;
  bool cameras_variable_ = false, ok_ = true;
};

}  // namespace colmap_hip
