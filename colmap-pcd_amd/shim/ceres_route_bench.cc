// ceres_route_bench.cc -- what one Ceres evaluation costs END TO END on the drop-in route (shim/ceres_adapter.h):
//   HipEvaluation::PrepareForEvaluation(jacobians = true, new_point = true)
//       = gather of the evaluation point from the solver's parameter blocks + H2D + the raw kernels + D2H of every
//         residual / Jacobian block into the handle's pinned buffers (pcd_ba_evaluate_blocks), and
//   one sweep of CostFunction::Evaluate over ALL residual blocks (what ceres::Problem::Evaluate / the LM loop does,
//   optim/bundle_adjustment.cc:537 -> Ceres -> the blocks added at :858-893, :967-983, :1031-1037), single-threaded and
//   on all host threads.
// Next to it: the raw pinned device->host rate for the same number of bytes (the PCIe ceiling of this route).
// Prints one JSON object.  Usage: ceres_route_bench <images> <points> [const_pose_fraction] [threads]
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "ceres_adapter.h"

using namespace colmap_hip;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

// parameter memory of the "solver": flat arrays indexed by the ids the problem was built with
struct FlatSource {
  double *poses, *points, *cams;
  double* Qvec(image_t i) const { return poses + 7 * (size_t)i; }
  double* Tvec(image_t i) const { return poses + 7 * (size_t)i + 4; }
  double* XYZ(point3D_t p) const { return points + 3 * (size_t)p; }
  double* Params(camera_t) const { return cams; }
};

static void opencv_project(const double* k, double u, double v, double* x, double* y) {
  const double r2 = u * u + v * v, rad = k[4] * r2 + k[5] * r2 * r2;
  const double du = u * rad + 2 * k[6] * u * v + k[7] * (r2 + 2 * u * u), dv = v * rad + 2 * k[7] * u * v + k[6] * (r2 + 2 * v * v);
  *x = k[0] * (u + du) + k[2];
  *y = k[1] * (v + dv) + k[3];
}

int main(int argc, char** argv) {
  const int I = argc > 1 ? std::atoi(argv[1]) : 1000;
  const int P = argc > 2 ? std::atoi(argv[2]) : 1000000;
  const double cfrac = argc > 3 ? std::atof(argv[3]) : 0.0;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned T = argc > 4 ? (unsigned)std::atoi(argv[4]) : std::min(hw, 16u);
  if (pcd_device_count() < 1) { std::printf("{\"error\": \"no gfx950 device\"}\n"); return 1; }

  // ---- synthetic scene (shape of SURVEY 8d: cameras along x looking down +z, OPENCV intrinsics, tracks of mean
  // length ~4.7, one shared constant camera, 90 % of the points with a LiDAR plane), observations image-major ----
  BundleAdjustmentOptions options;
  BundleAdjustmentConfig config;
  BundleAdjusterHip ba(options, config);
  std::mt19937_64 rng(11);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const double cam[8] = {3039, 3039, 2016, 1512, -0.05, 0.01, 1e-4, 1e-4};
  ba.cam_model_ = {PCD_CAM_OPENCV}; ba.cam_off_ = {0}; ba.cam_params_.assign(cam, cam + 8);
  ba.cam_refine_.assign(8, 0); ba.camera_ids_ = {0};
  const double span = 110.0;
  for (int i = 0; i < I; ++i) {
    const double q[4] = {1, 0, 0, 0}, c[3] = {5 + span * i / std::max(I - 1, 1), 12.5 + 0.2 * (U(rng) - 0.5), 0.5 * (U(rng) - 0.5)};
    ba.poses_.insert(ba.poses_.end(), q, q + 4);
    for (int k = 0; k < 3; ++k) ba.poses_.push_back(-c[k]);   // R = identity: t = -C
    ba.image_cam_.push_back(0);
    ba.image_const_pose_.push_back(U(rng) < cfrac ? 1 : 0);
    ba.image_const_tvec_.push_back(0); ba.image_used_.push_back(1); ba.image_ids_.push_back((image_t)i);
  }
  std::vector<std::vector<std::pair<int, std::pair<double, double>>>> per_image(I);
  for (int p = 0; p < P; ++p) {
    const int anchor = (int)(U(rng) * I) % I;
    const double z = 4 + 26 * U(rng), x = ba.poses_[7 * (size_t)anchor + 4] * -1 + (U(rng) - 0.5) * z, y = 12.5 + (U(rng) - 0.5) * 0.7 * z;
    const double X[3] = {x, y, z};
    ba.points_.insert(ba.points_.end(), X, X + 3);
    ba.point_const_.push_back(0); ba.point_ids_.push_back((point3D_t)p);
    int len = 2;
    while (len < 30 && U(rng) > 0.27) ++len;
    for (int j = 0; j < len; ++j) {
      const int im = anchor + (j - len / 2);
      if (im < 0 || im >= I) continue;
      const double* t = &ba.poses_[7 * (size_t)im + 4];
      const double pc[3] = {X[0] + t[0], X[1] + t[1], X[2] + t[2]};
      double ox, oy;
      opencv_project(cam, pc[0] / pc[2], pc[1] / pc[2], &ox, &oy);
      if (std::fabs(ox - cam[2]) > 4032 || std::fabs(oy - cam[3]) > 3024) continue;
      per_image[im].push_back({p, {ox + 4 * (U(rng) - 0.5), oy + 4 * (U(rng) - 0.5)}});
    }
    if (U(rng) < 0.9) {
      ba.lidar_point_.push_back(p);
      const double abcd[4] = {0, 1, 0, -(y + 0.05 * (U(rng) - 0.5))};
      ba.lidar_abcd_.insert(ba.lidar_abcd_.end(), abcd, abcd + 4);
      ba.lidar_w_.push_back(100.0);
    }
  }
  for (int im = 0; im < I; ++im)
    for (auto& ob : per_image[im]) {
      ba.obs_image_.push_back(im); ba.obs_point_.push_back(ob.first);
      ba.obs_xy_.push_back(ob.second.first); ba.obs_xy_.push_back(ob.second.second);
    }
  per_image.clear(); per_image.shrink_to_fit();
  const size_t O = ba.obs_image_.size(), L = ba.lidar_point_.size();
  auto t0 = Clock::now();
  if (!ba.Create(0)) { std::printf("{\"error\": \"pcd_ba_create failed: %s\"}\n", pcd_last_error()); return 1; }
  const double create_ms = ms_since(t0);

  // the solver's parameter memory (here: copies of the flat arrays; Ceres would move them in place)
  std::vector<double> s_poses = ba.poses_, s_points = ba.points_, s_cams = ba.cam_params_;
  FlatSource src{s_poses.data(), s_points.data(), s_cams.data()};
  HipEvaluation<FlatSource> cb(&ba, src);
  std::vector<std::unique_ptr<ceres::CostFunction>> blocks;
  blocks.reserve(O + L);
  for (size_t o = 0; o < O; ++o) blocks.emplace_back(cb.ReprojectionBlock(o));
  for (size_t l = 0; l < L; ++l) blocks.emplace_back(cb.LidarBlock(l));

  auto sweep = [&](unsigned threads, bool with_jac) {
    std::vector<std::thread> th;
    std::vector<double> sums(threads, 0.0);
    auto t1 = Clock::now();
    for (unsigned t = 0; t < threads; ++t)
      th.emplace_back([&, t] {
        // per-thread destination (Ceres copies each block into its own Jacobian storage)
        double r[2], jq[8], jt[6], jx[6];
        double* jp[4];
        double acc = 0;
        const size_t n = blocks.size(), b0 = n * t / threads, b1 = n * (t + 1) / threads;
        for (size_t b = b0; b < b1; ++b) {
          const auto& sz = blocks[b]->parameter_block_sizes();
          if (sz.size() == 4) { jp[0] = jq; jp[1] = jt; jp[2] = jx; jp[3] = nullptr; }        // q, t, X, camera (constant)
          else if (sz.size() == 2) { jp[0] = jx; jp[1] = nullptr; }                           // X, camera
          else { jp[0] = jx; }                                                                 // lidar: X
          blocks[b]->Evaluate(nullptr, r, with_jac ? jp : nullptr);
          acc += r[0] + (with_jac ? jx[0] : 0.0);
        }
        sums[t] = acc;
      });
    for (auto& x : th) x.join();
    double a = 0; for (double v : sums) a += v;
    return std::make_pair(ms_since(t1), a);
  };

  // warm-up (allocations of the pinned / device result buffers)
  cb.PrepareForEvaluation(true, true);
  cb.PrepareForEvaluation(false, true);
  const int reps = 5;
  double prep_j = 0, prep_r = 0, sw1 = 0, swT = 0, swr = 0, chk = 0;
  uint64_t bytes_j = 0;
  for (int k = 0; k < reps; ++k) {
    s_poses[7 * (size_t)(k % I) + 4] += 1e-6;                   // the solver moved: new evaluation point
    const uint64_t b0 = cb.bytes_d2h();
    t0 = Clock::now(); cb.PrepareForEvaluation(true, true); prep_j += ms_since(t0);
    bytes_j = cb.bytes_d2h() - b0;
    auto a = sweep(1, true); sw1 += a.first; chk += a.second;
    auto b = sweep(T, true); swT += b.first; chk += b.second;
    t0 = Clock::now(); cb.PrepareForEvaluation(false, true); prep_r += ms_since(t0);
    auto c = sweep(T, false); swr += c.first; chk += c.second;
  }
  prep_j /= reps; prep_r /= reps; sw1 /= reps; swT /= reps; swr /= reps;

  // the PCIe ceiling: the same number of bytes, device -> pinned host, one copy
  double pcie_ms = 0;
  {
    void *d = nullptr, *h = nullptr;
    if (hipMalloc(&d, bytes_j) == hipSuccess && hipHostMalloc(&h, bytes_j, hipHostMallocDefault) == hipSuccess) {
      (void)hipMemcpy(h, d, bytes_j, hipMemcpyDeviceToHost);
      t0 = Clock::now();
      for (int k = 0; k < 3; ++k) (void)hipMemcpy(h, d, bytes_j, hipMemcpyDeviceToHost);
      pcie_ms = ms_since(t0) / 3;
    }
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
  }
  const uint64_t h2d = (ba.poses_.size() + ba.points_.size()) * sizeof(double);
  std::printf("{\"images\": %d, \"points\": %d, \"observations\": %zu, \"lidar_terms\": %zu, \"const_pose_fraction\": %.2f, "
              "\"pose_rows\": %llu, \"create_ms\": %.2f, "
              "\"prepare_jacobians_ms\": %.3f, \"prepare_residuals_ms\": %.3f, \"bytes_d2h_jacobians\": %llu, \"bytes_h2d\": %llu, "
              "\"d2h_GBps_in_prepare\": %.1f, \"pinned_d2h_same_bytes_ms\": %.3f, \"pinned_d2h_GBps\": %.1f, "
              "\"prepare_over_pcie_ceiling\": %.2f, "
              "\"block_sweep_jacobians_ms_1_thread\": %.2f, \"block_sweep_jacobians_ms\": %.2f, \"block_sweep_residuals_ms\": %.2f, "
              "\"threads\": %u, \"ceres_route_e2e_ms\": %.2f, \"ceres_route_residual_pass_ms\": %.2f, \"checksum\": %.6g}\n",
              I, P, O, L, cfrac, (unsigned long long)cb.buffers().b.num_pose_rows, create_ms, prep_j, prep_r,
              (unsigned long long)bytes_j, (unsigned long long)h2d, bytes_j / prep_j * 1e-6, pcie_ms,
              pcie_ms > 0 ? bytes_j / pcie_ms * 1e-6 : 0.0, pcie_ms > 0 ? prep_j / pcie_ms : 0.0, sw1, swT, swr, T,
              prep_j + swT, prep_r + swr, chk);
  return 0;
}
