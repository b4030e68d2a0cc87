"""Regenerates integration/colmap-pcd-hip.patch: copies the touched files of the reference tree (argv[1], default
/root/reference) to a scratch directory, applies the edits below to the copies and writes `diff -u` of the two trees.
The edits are anchored on exact source lines; an anchor that no longer matches stops the script.  Nothing of the
reference is kept in this repository besides the context lines a unified diff carries."""
import os
import shutil
import subprocess
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "colmap-pcd-hip.patch")
FILES = ["CMakeLists.txt", "src/lidar/lidar_point.h", "src/lidar/ply.h", "src/lidar/ply.cc",
         "src/optim/bundle_adjustment.h", "src/optim/bundle_adjustment.cc", "src/sfm/incremental_mapper.cc",
         "src/controllers/bundle_adjustment.cc", "src/util/option_manager.cc"]
NEW_FILES = {}


def edit(root, rel, pairs):
    p = os.path.join(root, rel)
    s = open(p, encoding="utf-8").read()
    for old, new in pairs:
        assert s.count(old) == 1, (rel, s.count(old), old[:80])
        s = s.replace(old, new)
    open(p, "w", encoding="utf-8").write(s)


# --------------------------------------------------------------------------------------------------------- new file
NEW_FILES["src/lidar/hip_backend.h"] = r'''// hip_backend.h -- glue between colmap-pcd and libpcdhip (MI355X / gfx950), added by colmap-pcd-hip.patch.
// The adapters themselves are the headers of <pcdhip>/colmap-pcd_amd/shim (on the include path, CMakeLists.txt).
#ifndef COLMAP_LIDAR_HIP_BACKEND_H
#define COLMAP_LIDAR_HIP_BACKEND_H

#include <string>

#include "ceres_adapter.h"   // colmap_hip::HipBackendEnabled, HipBlockRecorder; pulls in lidar_hip.h
#include "lidar/lidar_point.h"

namespace colmap {

// Mapper.lidar_backend (util/option_manager.cc): "hip" = libpcdhip when a gfx950 device is visible, "cpu" = the
// PCL / FLANN KD-tree loops and Ceres autodiff of the stock build.  COLMAP_PCD_HIP=0 in the environment forces "cpu".
inline std::string& LidarBackendOption() {
  static std::string backend = "hip";
  return backend;
}
inline bool HipLidarBackend() { return LidarBackendOption() != "cpu" && colmap_hip::HipBackendEnabled(); }

// one accepted association of pcd_associate_staged as the LidarPoint the call site of `gate_mode` builds
// (type, colour, normalised plane; dist / angle as BundleAdjustmentConfig::MatchClosestLidarPoint stores them)
inline LidarPoint ToColmapLidarPoint(const pcd_assoc_hit& h, int gate_mode) {
  const colmap_hip::LidarPoint p = colmap_hip::ToLidarPoint(h, gate_mode);
  Eigen::Vector3d xyz(p.xyz[0], p.xyz[1], p.xyz[2]);
  Eigen::Vector4d abcd(p.abcd[0], p.abcd[1], p.abcd[2], p.abcd[3]);
  LidarPoint lidar_point(xyz, abcd);
  lidar_point.SetNormalizedPlane(abcd);   // the device ran Normalize(); keep its bits
  lidar_point.SetType(p.type == colmap_hip::LidarPointType::IcpGround ? LidarPointType::IcpGround : LidarPointType::Icp);
  Eigen::Vector3ub color(p.color[0], p.color[1], p.color[2]);
  lidar_point.SetColor(color);
  if (gate_mode == PCD_GATE_MAPPER_LOCAL) {
    lidar_point.SetDist(p.dist);
    lidar_point.SetAngle(p.angle);
  }
  return lidar_point;
}

}  // namespace colmap

#endif  // COLMAP_LIDAR_HIP_BACKEND_H
'''


def apply_edits(b):
    edit(b, "CMakeLists.txt", [(
        "find_package(PCL REQUIRED)\n",
        "find_package(PCL REQUIRED)\n"
        "# libpcdhip (MI355X / gfx950): -DPCDHIP_ROOT=<checkout of the pcd-hip repository>, built with make -C colmap-pcd_amd\n"
        "find_library(PCDHIP_LIBRARY pcdhip HINTS ${PCDHIP_ROOT}/colmap-pcd_amd REQUIRED)\n"
        "include_directories(${PCDHIP_ROOT}/include ${PCDHIP_ROOT}/colmap-pcd_amd/shim)\n"),
        ("set(COLMAP_EXTERNAL_LIBRARIES\n", "set(COLMAP_EXTERNAL_LIBRARIES\n    ${PCDHIP_LIBRARY}\n")])

    edit(b, "src/lidar/lidar_point.h", [(
        "    inline void SetAngle(const double& angle) {angle_ = angle;} \n",
        "    inline void SetAngle(const double& angle) {angle_ = angle;} \n"
        "    // plane that is normalised already (lidar/hip_backend.h): no second Normalize()\n"
        "    inline void SetNormalizedPlane(const Eigen::Vector4d& abcd) { abcd_ = abcd; }\n")])

    edit(b, "src/lidar/ply.h", [
        ('#include "kdtree.h"\n', '#include "kdtree.h"\n#include "lidar_hip.h"   // <pcdhip>/colmap-pcd_amd/shim\n'),
        ("    std::shared_ptr<Kdtree> kdtree_ptr_;\n",
         "    std::shared_ptr<Kdtree> kdtree_ptr_;\n"
         "    // device index of the same cloud (libpcdhip); null on the \"cpu\" backend\n"
         "    colmap_hip::lidar::PointCloudProcess* Hip() const { return hip_.get(); }\n"),
        ("    std::string path_;\n", "    std::string path_;\n    std::shared_ptr<colmap_hip::lidar::PointCloudProcess> hip_;\n")])
    edit(b, "src/lidar/ply.cc", [
        ('#include "lidar/ply.h"\n', '#include "lidar/ply.h"\n#include "lidar/hip_backend.h"\n'),
        ("    global_pcd_ptr_ = PointCloudDirectionTrans(ptr);\n    // Cut point cloud to nodes\n",
         "    if (HipLidarBackend()) {\n"
         "        // the rows as PCL loaded them (lidarpt::Point = 32-byte AoS): the axis swap and the NaN filter of\n"
         "        // PointCloudDirectionTrans run on the device, indices are the post-filter row numbers as below\n"
         "        hip_ = std::make_shared<colmap_hip::lidar::PointCloudProcess>(path_);\n"
         "        if (!hip_->InitializeFromRawCloud(reinterpret_cast<const float*>(ptr->points.data()), nullptr,\n"
         "                                          ptr->points.size(), /*aos32=*/true)) {\n"
         "            std::cout << \"libpcdhip: \" << pcd_last_error() << \" -- staying on the KD-tree\" << std::endl;\n"
         "            hip_.reset();\n"
         "        }\n"
         "    }\n"
         "    global_pcd_ptr_ = PointCloudDirectionTrans(ptr);\n    // Cut point cloud to nodes\n")])

    # ------------------------------------------------------------------ optim/bundle_adjustment.{h,cc}
    edit(b, "src/optim/bundle_adjustment.h", [
        ('#include "lidar/ply.h"\n', '#include "lidar/ply.h"\n#include "lidar/hip_backend.h"\n'),
        ("  void MatchClosestLidarPoint(Reconstruction* reconstruction,const point3D_t& point3D_id, double& max_search_range);\n",
         "  void MatchClosestLidarPoint(Reconstruction* reconstruction,const point3D_t& point3D_id, double& max_search_range);\n"
         "  // the same for many points in one device call (libpcdhip); false: not available, use the loop above\n"
         "  bool MatchClosestLidarPoints(Reconstruction* reconstruction, const std::vector<point3D_t>& point3D_ids,\n"
         "                               const std::vector<double>& max_search_ranges);\n"),
        ("  std::unique_ptr<ceres::Problem> problem_;\n",
         "  std::unique_ptr<ceres::Problem> problem_;\n"
         "  // residuals and Jacobians of every block from ONE device evaluation per Ceres evaluation (libpcdhip)\n"
         "  std::unique_ptr<colmap_hip::HipBlockRecorder> hip_blocks_;\n")])

    edit(b, "src/optim/bundle_adjustment.cc", [
        # batched MatchClosestLidarPoint
        ("void BundleAdjustmentConfig::AddConstantPoint(const point3D_t point3D_id) {\n  CHECK(!HasVariablePoint(point3D_id));\n",
         "bool BundleAdjustmentConfig::MatchClosestLidarPoints(Reconstruction* reconstruction,\n"
         "                                                     const std::vector<point3D_t>& point3D_ids,\n"
         "                                                     const std::vector<double>& max_search_ranges) {\n"
         "  if (!HipLidarBackend() || !point_cloud_process_ || !point_cloud_process_->Hip()) return false;\n"
         "  const pcd_assoc_hit* hits = nullptr;\n"
         "  uint64_t num_hits = 0;\n"
         "  if (!colmap_hip::MatchClosestLidarPointsFlat(\n"
         "          *point_cloud_process_->Hip(), point3D_ids.size(), /*per_point_range=*/true, PCD_GATE_MAPPER_LOCAL,\n"
         "          [&](uint64_t i, double* xyz, double* range) {\n"
         "            const Eigen::Vector3d& pt_xyz = reconstruction->Point3D(point3D_ids[i]).XYZ();\n"
         "            xyz[0] = pt_xyz(0); xyz[1] = pt_xyz(1); xyz[2] = pt_xyz(2);\n"
         "            *range = max_search_ranges[i];\n"
         "          },\n"
         "          &hits, &num_hits)) {\n"
         "    return false;\n"
         "  }\n"
         "  // the accepted associations, in the order of point3D_ids: what MatchClosestLidarPoint adds one by one\n"
         "  for (uint64_t k = 0; k < num_hits; ++k) {\n"
         "    LidarPoint lidar_point = ToColmapLidarPoint(hits[k], PCD_GATE_MAPPER_LOCAL);\n"
         "    AddLidarPoint(point3D_ids[hits[k].query], lidar_point);\n"
         "    reconstruction->AddLidarPoint(point3D_ids[hits[k].query], lidar_point);\n"
         "  }\n"
         "  return true;\n"
         "}\n\n"
         "void BundleAdjustmentConfig::AddConstantPoint(const point3D_t point3D_id) {\n  CHECK(!HasVariablePoint(point3D_id));\n"),
        # Solve: problem options + finalize
        ("  // 创建一个新的Ceres优化问题\n  problem_ = std::make_unique<ceres::Problem>();\n",
         "  // 创建一个新的Ceres优化问题\n"
         "  ceres::Problem::Options problem_options;\n"
         "  if (HipLidarBackend()) {\n"
         "    // the blocks the SetUp* functions create below only copy what one pcd_ba_evaluate_blocks per Ceres\n"
         "    // evaluation computed for all of them (ceres_adapter.h)\n"
         "    hip_blocks_ = std::make_unique<colmap_hip::HipBlockRecorder>();\n"
         "    problem_options.evaluation_callback = hip_blocks_.get();\n"
         "  }\n"
         "  problem_ = std::make_unique<ceres::Problem>(problem_options);\n"),
        ("  // 检查优化问题是否包含有效的残差项，如果没有则终止优化\n  if (problem_->NumResiduals() == 0) {\n    return false;\n  }\n",
         "  // 检查优化问题是否包含有效的残差项，如果没有则终止优化\n  if (problem_->NumResiduals() == 0) {\n    return false;\n  }\n\n"
         "  if (hip_blocks_) {\n"
         "    const bool cameras_variable = options_.refine_focal_length || options_.refine_principal_point ||\n"
         "                                  options_.refine_extra_params;\n"
         "    CHECK(hip_blocks_->Finalize(/*device=*/0, cameras_variable)) << pcd_last_error();\n"
         "  }\n"),
    ])
    # the cost functions: every place an AutoDiffCostFunction is created
    p = os.path.join(b, "src/optim/bundle_adjustment.cc")
    s = open(p, encoding="utf-8").read()
    # BundleAdjuster only: RigBundleAdjuster (further down in the file) keeps its autodiff blocks
    cut = s.index("// ParallelBundleAdjuster")
    s, tail = s[:cut], s[cut:]
    const_pose_switch = ("      switch (camera.ModelId()) {\n"
                         "#define CAMERA_MODEL_CASE(CameraModel)                                 \\\n"
                         "  case CameraModel::kModelId:                                          \\\n"
                         "    cost_function =                                                    \\\n"
                         "        BundleAdjustmentConstantPoseCostFunction<CameraModel>::Create( \\\n"
                         "            image.Qvec(), image.Tvec(), point2D.XY());                 \\\n"
                         "    break;\n")
    n_const = s.count(const_pose_switch)
    assert n_const == 2, n_const      # AddImageInSphereToProblem, AddImageToProblem
    s = s.replace(const_pose_switch,
                  "      if (hip_blocks_) {\n"
                  "        cost_function = hip_blocks_->AddReprojection(camera.ModelId(), qvec_data, tvec_data, point3D.XYZ().data(),\n"
                  "                                                     camera_params_data, point2D.XY().data(), /*constant_pose=*/true);\n"
                  "      } else\n" + const_pose_switch)
    var_pose_switch = ("      switch (camera.ModelId()) {\n"
                       "#define CAMERA_MODEL_CASE(CameraModel)                                   \\\n"
                       "  case CameraModel::kModelId:                                            \\\n"
                       "    cost_function =                                                      \\\n"
                       "        BundleAdjustmentCostFunction<CameraModel>::Create(point2D.XY()); \\\n"
                       "    break;\n")
    n_var = s.count(var_pose_switch)
    assert n_var == 2, n_var
    s = s.replace(var_pose_switch,
                  "      if (hip_blocks_) {\n"
                  "        cost_function = hip_blocks_->AddReprojection(camera.ModelId(), qvec_data, tvec_data, point3D.XYZ().data(),\n"
                  "                                                     camera_params_data, point2D.XY().data(), /*constant_pose=*/false);\n"
                  "      } else\n" + var_pose_switch)
    point_switch = ("\n    switch (camera.ModelId()) {\n"
                    "#define CAMERA_MODEL_CASE(CameraModel)                                 \\\n"
                    "  case CameraModel::kModelId:                                          \\\n"
                    "    cost_function =                                                    \\\n"
                    "        BundleAdjustmentConstantPoseCostFunction<CameraModel>::Create( \\\n"
                    "            image.Qvec(), image.Tvec(), point2D.XY());                 \\\n"
                    "    break;\n")
    assert s.count(point_switch) == 1, s.count(point_switch)   # AddPointToProblem (4 spaces less indentation)
    s = s.replace(point_switch,
                  "\n    if (hip_blocks_) {\n"
                  "      cost_function = hip_blocks_->AddReprojection(camera.ModelId(), image.Qvec().data(), image.Tvec().data(),\n"
                  "                                                   point3D.XYZ().data(), camera.ParamsData(), point2D.XY().data(),\n"
                  "                                                   /*constant_pose=*/true);\n"
                  "    } else" + point_switch)
    lidar_create = ("    cost_function =BundleAdjustmentLidarCostFunction::Create( \n")
    assert s.count(lidar_create) == 1
    s = s.replace(lidar_create,
                  "    if (hip_blocks_) {\n"
                  "      cost_function = hip_blocks_->AddLidar(point3D.XYZ().data(), abcd.data(), w);\n"
                  "    } else\n" + lidar_create)
    open(p, "w", encoding="utf-8").write(s + tail)

    # ------------------------------------------------------------------ the three association loops
    edit(b, "src/sfm/incremental_mapper.cc", [
        # (a) AdjustLocalBundle: the long-track points, per-point range schedule
        ("      for (auto iter = search_closest_point3D_ids.begin(); iter != search_closest_point3D_ids.end(); iter++){\n"
         "        const point3D_t point3D_id = *iter;\n",
         "      // libpcdhip: one device call for all of them (the range schedule is the same expression as below)\n"
         "      std::vector<point3D_t> closest_ids(search_closest_point3D_ids.begin(), search_closest_point3D_ids.end());\n"
         "      std::vector<double> closest_ranges(closest_ids.size());\n"
         "      for (size_t i = 0; i < closest_ids.size(); ++i) {\n"
         "        const int opt_num = reconstruction_->Point3D(closest_ids[i]).GlobalOptNum();\n"
         "        closest_ranges[i] = std::max(options.kdtree_min_search_range,\n"
         "                                     options.kdtree_max_search_range - opt_num * options.search_range_drop_speed);\n"
         "      }\n"
         "      const bool closest_done = ba_config.MatchClosestLidarPoints(reconstruction_, closest_ids, closest_ranges);\n"
         "      for (auto iter = search_closest_point3D_ids.begin(); !closest_done && iter != search_closest_point3D_ids.end(); iter++){\n"
         "        const point3D_t point3D_id = *iter;\n"),
        # (b) AdjustGlobalBundleByLidar
        ("    for (auto iter = variable_point3D_ids.begin(); iter != variable_point3D_ids.end(); iter++){\n"
         "      point3D_t point3D_id = *iter;\n"
         "      Point3D& point3D = reconstruction_->Point3D(point3D_id);\n"
         "      // 标记该点在优化球内\n"
         "      point3D.IfInSphere() = true;\n",
         "    bool global_done = false;\n"
         "    if (HipLidarBackend() && lidar_pointcloud_process_ && lidar_pointcloud_process_->Hip()) {\n"
         "      // libpcdhip: the whole loop below as one device call (PCD_GATE_MAPPER_GLOBAL: blue / yellow, the gate is\n"
         "      // the point-to-point distance against the per-point range)\n"
         "      std::vector<point3D_t> ids(variable_point3D_ids.begin(), variable_point3D_ids.end());\n"
         "      const pcd_assoc_hit* hits = nullptr;\n"
         "      uint64_t num_hits = 0;\n"
         "      global_done = colmap_hip::MatchClosestLidarPointsFlat(\n"
         "          *lidar_pointcloud_process_->Hip(), ids.size(), /*per_point_range=*/true, PCD_GATE_MAPPER_GLOBAL,\n"
         "          [&](uint64_t i, double* xyz, double* range) {\n"
         "            Point3D& point3D = reconstruction_->Point3D(ids[i]);\n"
         "            point3D.IfInSphere() = true;\n"
         "            xyz[0] = point3D.XYZ()(0); xyz[1] = point3D.XYZ()(1); xyz[2] = point3D.XYZ()(2);\n"
         "            *range = std::max(options.kdtree_min_search_range,\n"
         "                              options.kdtree_max_search_range - point3D.GlobalOptNum() * options.search_range_drop_speed);\n"
         "          },\n"
         "          &hits, &num_hits);\n"
         "      for (uint64_t k = 0; global_done && k < num_hits; ++k) {\n"
         "        LidarPoint lidar_point = ToColmapLidarPoint(hits[k], PCD_GATE_MAPPER_GLOBAL);\n"
         "        ba_config.AddLidarPoint(ids[hits[k].query], lidar_point);\n"
         "        reconstruction_->AddLidarPointInGlobal(ids[hits[k].query], lidar_point);\n"
         "      }\n"
         "    }\n"
         "    for (auto iter = variable_point3D_ids.begin(); !global_done && iter != variable_point3D_ids.end(); iter++){\n"
         "      point3D_t point3D_id = *iter;\n"
         "      Point3D& point3D = reconstruction_->Point3D(point3D_id);\n"
         "      // 标记该点在优化球内\n"
         "      point3D.IfInSphere() = true;\n")])

    edit(b, "src/controllers/bundle_adjustment.cc", [
        ("    // 遍历每个3D点，为其建立与激光雷达点云的对应关系\n"
         "    for (point3D_t point3d_id : reg_point3D_ids) {\n",
         "    bool associated = false;\n"
         "    if (HipLidarBackend() && lidar_pointcloud_process_ && lidar_pointcloud_process_->Hip()) {\n"
         "      // libpcdhip: every point of the map in one device call (PCD_GATE_CONTROLLER: dist2plane > 1 ||\n"
         "      // dist2point > 2 rejects, before the ground / non-ground classification)\n"
         "      std::vector<point3D_t> ids(reg_point3D_ids.begin(), reg_point3D_ids.end());\n"
         "      const pcd_assoc_hit* hits = nullptr;\n"
         "      uint64_t num_hits = 0;\n"
         "      associated = colmap_hip::MatchClosestLidarPointsFlat(\n"
         "          *lidar_pointcloud_process_->Hip(), ids.size(), /*per_point_range=*/false, PCD_GATE_CONTROLLER,\n"
         "          [&](uint64_t i, double* xyz, double* range) {\n"
         "            ba_config.AddVariablePoint(ids[i]);\n"
         "            const Eigen::Vector3d& pt_xyz = reconstruction_->Point3D(ids[i]).XYZ();\n"
         "            xyz[0] = pt_xyz(0); xyz[1] = pt_xyz(1); xyz[2] = pt_xyz(2);\n"
         "            *range = 0.0;   // unused by this gate\n"
         "          },\n"
         "          &hits, &num_hits);\n"
         "      for (uint64_t k = 0; associated && k < num_hits; ++k) {\n"
         "        LidarPoint lidar_point = ToColmapLidarPoint(hits[k], PCD_GATE_CONTROLLER);\n"
         "        ba_config.AddLidarPoint(ids[hits[k].query], lidar_point);\n"
         "        reconstruction_->AddLidarPointInGlobal(ids[hits[k].query], lidar_point);\n"
         "      }\n"
         "    }\n"
         "    // 遍历每个3D点，为其建立与激光雷达点云的对应关系\n"
         "    for (point3D_t point3d_id : reg_point3D_ids) {\n"
         "      if (associated) break;\n")])

    edit(b, "src/util/option_manager.cc", [
        ('#include "feature/sift.h"\n', '#include "feature/sift.h"\n#include "lidar/hip_backend.h"\n'),
        ('  AddAndRegisterDefaultOption("Mapper.lidar_pointcloud_path",\n',
         '  // "hip": association and residual / Jacobian evaluation on the GPU (libpcdhip); "cpu": the stock path\n'
         '  AddAndRegisterDefaultOption("Mapper.lidar_backend", &LidarBackendOption());\n'
         '  AddAndRegisterDefaultOption("Mapper.lidar_pointcloud_path",\n')])


def main():
    tmp = tempfile.mkdtemp(prefix="colmap_patch_")
    a, b = os.path.join(tmp, "a"), os.path.join(tmp, "b")
    for f in FILES:
        for root in (a, b):
            os.makedirs(os.path.dirname(os.path.join(root, f)), exist_ok=True)
            shutil.copy(os.path.join(REF, f), os.path.join(root, f))
    apply_edits(b)
    for f, text in NEW_FILES.items():
        open(os.path.join(b, f), "w", encoding="utf-8").write(text)
    r = subprocess.run(["diff", "-urN", "a", "b"], cwd=tmp, capture_output=True, text=True)
    assert r.returncode == 1, r.stderr
    header = ("colmap-pcd on MI355X: the registration hot path through libpcdhip (pcd-hip repository).\n"
              "Apply in the root of Wangshihu12/colmap-pcd:  git apply colmap-pcd-hip.patch   (or patch -p1 < ...)\n"
              "Configure with -DPCDHIP_ROOT=<pcd-hip checkout>; run-time switch Mapper.lidar_backend = hip | cpu,\n"
              "COLMAP_PCD_HIP=0 forces cpu.  Generated by integration/make_patch.py.\n\n")
    # drop the timestamps of the ---/+++ lines: the file must not change from one generation to the next
    lines = []
    for ln in r.stdout.splitlines(keepends=True):
        if ln.startswith("--- ") or ln.startswith("+++ "):
            ln = ln.split("\t")[0] + "\n"
        lines.append(ln)
    open(OUT, "w", encoding="utf-8").write(header + "".join(lines))
    shutil.rmtree(tmp)
    print("wrote", OUT, sum(1 for ln in lines if ln.startswith("@@")), "hunks")


if __name__ == "__main__":
    main()
