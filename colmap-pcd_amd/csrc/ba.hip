// ba.hip -- bundle-adjustment residual / Jacobian / normal-equation evaluation.
//
// Replaces the work ceres::Solve performs per iteration on the problem assembled by
// optim/bundle_adjustment.cc:694-1131 (AddImageToProblem, AddImageInSphereToProblem,
// AddPointToProblem, AddLidarToProblem, ParameterizeCameras/Points): every residual block's
// CostFunction::Evaluate (base/cost_functions.h:49-141, :150-241, :256-370), the loss
// correction (optim/bundle_adjustment.cc:53-68) and the manifold projection
// (base/cost_functions.h:610-627), then J^T J / J^T r block accumulation.
//
// Kernels (all fp64, memory/latency-bound: ~300 flop per ~60-200 B per observation):
//   k_ba_points  thread = 3D point (track).  Tracks are processed in order of track length and their
//                observations are stored in a sliced-ELL layout (64 tracks per slice, observation j of
//                lane l at slice_base + 64 j + l): the lanes of a wavefront run the same number of
//                iterations and every load instruction reads 64 consecutive records.  Accumulates the
//                point's 3x3 block, gradient and the cost.  No atomics: point blocks are complete inside
//                one thread, the cost goes through a fixed-order two-stage sum.
//   k_ba_images  workgroup = image; its observations are stored image-major (contiguous), strided over
//                256 lanes; each lane recomputes the 2x6 pose-tangent Jacobian and accumulates 21 + 6
//                unique entries; fixed-order block reduction -> deterministic 6x6 block + gradient.
//   k_ba_raw     thread = observation / LiDAR term: the raw ambient blocks exactly as
//                CostFunction::Evaluate returns them (for the Ceres EvaluationCallback adapter).
// Jacobians are recomputed in each kernel instead of being staged through HBM (160 B/obs of traffic
// would cost more than the ~300 flops).  MODEL >= 0 compiles one camera model in (all cameras of the
// problem share it -- the usual case); MODEL = -1 switches per observation.
#include <algorithm>
#include <numeric>

#include "ba_cam_jac.h"
#include "ba_math.h"
#include "common.h"

namespace pcd {

struct BaDev {
  // problem (device)
  const int* cam_model; const int* cam_off; const double* cam_params;
  const double* poses; const int* image_cam; const uint8_t* image_const_pose; const uint8_t* image_const_tvec;
  const double* points; const uint8_t* point_const;
  const int* obs_image; const int* obs_point; const double* obs_xy;
  const int* lidar_point; const double* lidar_abcd; const double* lidar_w;
  // per-track (sliced ELL, length-sorted) and per-image (contiguous) copies of the observations
  const int* pt_order;            // [nslices*64]  thread -> point id (-1 = padding)
  const uint32_t* slice_start;    // [nslices+1]   first slot of each 64-track slice
  const int* sell_img;            // [nslots]      image of the observation, -1 = padding
  const double* sell_xy;          // [nslots][2]
  const uint32_t* pt_lidar_start; const uint32_t* pt_lidar_list;
  const uint32_t* img_obs_start;  // [I+1]
  const int* img_pt;              // [O] point of the e-th observation of the image-major order
  const uint32_t* img_obs;        // [O] its index in the caller's observation order (W is written there)
  const uint32_t* seg_img;        // [nseg] image of each segment of <= kImgSeg observations (image-major order)
  const uint32_t* seg_begin;      // [nseg+1] first observation (image-major position) of each segment
  const uint32_t* img_seg_start;  // [I+1] segments of each image
  const uint8_t* cam_refine;      // [cam_params_len] 1 = parameter optimised (nullptr: all constant)
  const uint32_t* cam_img_start;  // [C+1] images of each camera (CSR, ascending image index)
  const uint32_t* cam_img_list;
  int C;
  int cam_k;                      // K of the camera accumulation: the model's when one of the compiled-in models is
                                  // used by every camera, PCD_CAM_JAC_STRIDE for the generic (per-observation switch) path
  const double* img_xy;           // [O][2]
  int I, P, nslices; uint64_t O, L;
  int loss_type; double loss_scale;
};

template <int MODEL>
__device__ __forceinline__ void eval_block(const BaDev& d, int im, const double X[3], double ox, double oy,
                                           ReprojBlock& b, double q[4]) {
  const double* pose = d.poses + 7 * (size_t)im;
  double t[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = pose[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = pose[4 + k];
  const int cm = d.image_cam[im];
  const int model = MODEL >= 0 ? MODEL : d.cam_model[cm];
  reproj_eval(model, d.cam_params + d.cam_off[cm], q, t, X, ox, oy, b);
}

// Coalesced store of one row of N doubles per lane, rows of consecutive lanes adjacent in memory
// (out[(row0 + lane) * N + k] = v[k] for lane < cnt): the rows are transposed through a per-wavefront LDS
// scratch of 64 * N doubles and leave as 16-B-per-lane stores of consecutive addresses (1 KiB per instruction)
// instead of N stores that each touch 64 different cache lines.  Every lane of the wavefront must call it.
template <int N>
__device__ __forceinline__ void wave_store_rows(double* __restrict__ out, uint64_t row0, int cnt, const double (&v)[N],
                                                double* __restrict__ lds) {
  static_assert(N % 2 == 0, "rows are moved in 16-byte units");
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < N; ++k) lds[lane * N + k] = v[k];
  // one wavefront: LDS operations complete in order; the fences keep the compiler from moving the reads up
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const double2* src = reinterpret_cast<const double2*>(lds);
  double2* dst = reinterpret_cast<double2*>(out + row0 * N);
  const int units = cnt * (N / 2);
#pragma unroll
  for (int j = 0; j < N / 2; ++j) {
    const int u = j * 64 + lane;
    if (u < units) dst[u] = src[u];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();   // the scratch may be rewritten after this
}

// ------------------------------------------------------------- points ------
// BLOCKS = false: residual-only pass (cost), what Ceres asks for when it evaluates a trial step
// Two threads per track (the even and the odd observations of its sliced-ELL column): the kernel is bound by the
// latency of each track's serial chain of observations, not by arithmetic or bytes, so halving the chain and doubling
// the wavefronts in flight took it from 0.18 to ~0.1 ms on the bench scene.  The halves live in different wavefronts
// of the workgroup (threads 0-63 / 64-127 = halves 0 / 1 of slice 2b, 128-255 of slice 2b + 1); half 1 hands its sums
// over through LDS and half 0 adds them in a fixed order and also takes the track's LiDAR terms.
template <int MODEL, bool BLOCKS>
__global__ __launch_bounds__(256) void k_ba_points(BaDev d, double* __restrict__ Hpt, double* __restrict__ gpt,
                                                   double* __restrict__ cost_partial) {
  __shared__ double s_half[2][64][9];
  const int sl = threadIdx.x >> 7, half = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
  const int slice = blockIdx.x * 2 + sl;
  const int t = slice * 64 + lane;
  double cost = 0.0;
  double H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  int p = -1;
  bool cpt = true;
  double X[3] = {0, 0, 0};
  if (slice < d.nslices) {
    p = d.pt_order[t];
    const uint32_t s0 = d.slice_start[slice], s1 = d.slice_start[slice + 1];
    if (p >= 0) {
      X[0] = d.points[3 * (size_t)p]; X[1] = d.points[3 * (size_t)p + 1]; X[2] = d.points[3 * (size_t)p + 2];
      cpt = d.point_const && d.point_const[p];
    }
    for (uint32_t s = s0 + lane + 64u * half; s < s1; s += 128) {
      const int im = d.sell_img[s];
      if (im < 0) continue;  // padding of a shorter track
      ReprojBlock b;
      double q[4];
      eval_block<MODEL>(d, im, X, d.sell_xy[2 * (size_t)s], d.sell_xy[2 * (size_t)s + 1], b, q);
      double rho0, rho1;
      loss_eval(d.loss_type, d.loss_scale, b.r[0] * b.r[0] + b.r[1] * b.r[1], rho0, rho1);
      cost += 0.5 * rho0;
      if (BLOCKS && !cpt) {
        const double sr = sqrt(rho1);
        double J[6];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int k = 0; k < 3; ++k)
            J[3 * r + k] = sr * (b.M[3 * r] * b.D[k] + b.M[3 * r + 1] * b.D[3 + k] + b.M[3 * r + 2] * b.D[6 + k]);
        const double r0 = sr * b.r[0], r1 = sr * b.r[1];
        H[0] += J[0] * J[0] + J[3] * J[3]; H[1] += J[0] * J[1] + J[3] * J[4]; H[2] += J[0] * J[2] + J[3] * J[5];
        H[3] += J[1] * J[1] + J[4] * J[4]; H[4] += J[1] * J[2] + J[4] * J[5]; H[5] += J[2] * J[2] + J[5] * J[5];
        g[0] += J[0] * r0 + J[3] * r1; g[1] += J[1] * r0 + J[4] * r1; g[2] += J[2] * r0 + J[5] * r1;
      }
    }
  }
  if (BLOCKS && half == 1) {
#pragma unroll
    for (int k = 0; k < 6; ++k) s_half[sl][lane][k] = H[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) s_half[sl][lane][6 + k] = g[k];
  }
  __syncthreads();
  if (half == 0 && slice < d.nslices) {
    if (BLOCKS) {
#pragma unroll
      for (int k = 0; k < 6; ++k) H[k] += s_half[sl][lane][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] += s_half[sl][lane][6 + k];
    }
    if (p >= 0) {
      for (uint32_t e = d.pt_lidar_start[p]; e < d.pt_lidar_start[p + 1]; ++e) {
        const uint32_t l = d.pt_lidar_list[e];
        const double abcd[4] = {d.lidar_abcd[4 * (size_t)l], d.lidar_abcd[4 * (size_t)l + 1],
                                d.lidar_abcd[4 * (size_t)l + 2], d.lidar_abcd[4 * (size_t)l + 3]};
        double r, J[3];
        lidar_eval(X, abcd, d.lidar_w[l], 0, r, J);
        double rho0, rho1;
        loss_eval(d.loss_type, d.loss_scale, r * r, rho0, rho1);
        cost += 0.5 * rho0;
        if (BLOCKS && !cpt) {
          const double sr = sqrt(rho1);
          const double rc = sr * r;
          J[0] *= sr; J[1] *= sr; J[2] *= sr;
          H[0] += J[0] * J[0]; H[1] += J[0] * J[1]; H[2] += J[0] * J[2];
          H[3] += J[1] * J[1]; H[4] += J[1] * J[2]; H[5] += J[2] * J[2];
          g[0] += J[0] * rc; g[1] += J[1] * rc; g[2] += J[2] * rc;
        }
      }
      if (BLOCKS && Hpt) {
        double* h = Hpt + 9 * (size_t)p;
        h[0] = H[0]; h[1] = H[1]; h[2] = H[2]; h[3] = H[1]; h[4] = H[3]; h[5] = H[4]; h[6] = H[2]; h[7] = H[4]; h[8] = H[5];
      }
      if (BLOCKS && gpt) { gpt[3 * (size_t)p] = g[0]; gpt[3 * (size_t)p + 1] = g[1]; gpt[3 * (size_t)p + 2] = g[2]; }
    }
  }
  __syncthreads();
  // fixed-order block sum of the cost
  __shared__ double s_c[256];
  s_c[threadIdx.x] = cost;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_c[threadIdx.x] += s_c[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) cost_partial[blockIdx.x] = s_c[0];
}

// Cost only (the LM trial-step evaluation): no per-point accumulation is needed, so the sum runs over
// observations and LiDAR terms in their given order, one thread each -- 5.9 M independent threads instead of
// 1 M tracks with serial inner loops.  Same fixed-order two-stage sum (bitwise reproducible run to run; the
// summation order, hence the last bits, differ from the cost the Jacobian pass reports).
template <int MODEL>
__global__ __launch_bounds__(256) void k_ba_cost(BaDev d, double* __restrict__ cost_partial) {
  // one residual block per thread (a fixed grid of 2048 workgroups looping over them measured 0.109 ms against
  // 0.096 ms: fewer wavefronts in flight for a latency-bound gather); the loop form is kept for grids that are
  // capped.  The assignment of blocks to threads and the order of the sums depend on the problem size only.
  double cost = 0.0;
  for (uint64_t i = blockIdx.x * (uint64_t)256 + threadIdx.x; i < d.O + d.L; i += (uint64_t)gridDim.x * 256) {
    if (i < d.O) {
      const int im = d.obs_image[i], pt = d.obs_point[i];
      const double X[3] = {d.points[3 * (size_t)pt], d.points[3 * (size_t)pt + 1], d.points[3 * (size_t)pt + 2]};
      ReprojBlock b;
      double q[4];
      eval_block<MODEL>(d, im, X, d.obs_xy[2 * i], d.obs_xy[2 * i + 1], b, q);
      double rho0, rho1;
      loss_eval(d.loss_type, d.loss_scale, b.r[0] * b.r[0] + b.r[1] * b.r[1], rho0, rho1);
      cost += 0.5 * rho0;
    } else {
      const uint64_t l = i - d.O;
      const int pt = d.lidar_point[l];
      const double X[3] = {d.points[3 * (size_t)pt], d.points[3 * (size_t)pt + 1], d.points[3 * (size_t)pt + 2]};
      const double abcd[4] = {d.lidar_abcd[4 * l], d.lidar_abcd[4 * l + 1], d.lidar_abcd[4 * l + 2], d.lidar_abcd[4 * l + 3]};
      double r, J[3];
      lidar_eval(X, abcd, d.lidar_w[l], 0, r, J);
      double rho0, rho1;
      loss_eval(d.loss_type, d.loss_scale, r * r, rho0, rho1);
      cost += 0.5 * rho0;
    }
  }
  __shared__ double s_c[256];
  s_c[threadIdx.x] = cost;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_c[threadIdx.x] += s_c[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) cost_partial[blockIdx.x] = s_c[0];
}
constexpr unsigned kCostBlocks = 1u << 20;   // cap of the cost pass's grid (256 M residual blocks in one sweep)

__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ partial, int n, double* __restrict__ out) {
  __shared__ double s_c[256];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four loads in flight per thread; the order is fixed
  int i = threadIdx.x;
  for (; i + 768 < n; i += 1024) { a0 += partial[i]; a1 += partial[i + 256]; a2 += partial[i + 512]; a3 += partial[i + 768]; }
  for (; i < n; i += 256) a0 += partial[i];
  s_c[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_c[threadIdx.x] += s_c[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s_c[0];
}

// ------------------------------------------------------------- images ------
// WANT_W: also write the pose-point coupling block W[o] = Jp^T JX (6x3, loss-corrected, manifold-projected) of
// every observation -- what a Schur complement needs next to H_img / H_pt -- from the Jacobians this pass has
// in registers anyway (no second sweep).  When the caller's observations are image-major (the order
// AddImageToProblem creates them in, optim/bundle_adjustment.cc:814-919) the 144-B blocks of consecutive
// lanes are adjacent in memory.
// Work item = one SEGMENT of <= kImgSeg observations of an image (an image per workgroup left a quarter of the chip
// idle on a 1000-image scene and nearly all of it on a 25-image one); the 27 partial sums of a segment go to
// `partial` and k_ba_images_reduce adds the segments of an image in ascending order: still no atomics, still
// bitwise reproducible.
constexpr uint32_t kImgSeg = 1024;
template <int MODEL, bool WANT_W>
__global__ __launch_bounds__(256) void k_ba_images(BaDev d, double* __restrict__ partial, double* __restrict__ W_o) {
  __shared__ __attribute__((aligned(16))) double s_w[WANT_W ? 4 : 1][WANT_W ? 64 * 18 : 2];
  const int im = (int)d.seg_img[blockIdx.x];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform for the compiler
  const bool cpose = d.image_const_pose && d.image_const_pose[im];
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  const unsigned tmask = d.image_const_tvec ? d.image_const_tvec[im] : 0u;
  const uint32_t e_beg = d.seg_begin[blockIdx.x];
  const uint32_t e_end = min(d.seg_begin[blockIdx.x + 1], d.img_obs_start[im + 1]);   // segments never span images
  if (!cpose || WANT_W) {
    // every lane runs every iteration (the W store below is a whole-wavefront operation)
    for (uint32_t e0 = e_beg; e0 < e_end; e0 += 256) {
      const uint32_t e = e0 + threadIdx.x;
      const bool active = e < e_end;
      double w[18];
#pragma unroll
      for (int k = 0; k < 18; ++k) w[k] = 0.0;   // constant pose / constant point: the coupling is zero
      if (active && !cpose) {
        const int pt = d.img_pt[e];
        const double X[3] = {d.points[3 * (size_t)pt], d.points[3 * (size_t)pt + 1], d.points[3 * (size_t)pt + 2]};
        ReprojBlock b;
        double q[4];
        eval_block<MODEL>(d, im, X, d.img_xy[2 * (size_t)e], d.img_xy[2 * (size_t)e + 1], b, q);
        double rho0, rho1;
        loss_eval(d.loss_type, d.loss_scale, b.r[0] * b.r[0] + b.r[1] * b.r[1], rho0, rho1);
        const double sr = sqrt(rho1);
        double Jq[8], Jt[6], JX[6], Jqt[6];
        reproj_jacobians(b, Jq, Jt, JX);
        quat_tangent(q, Jq, Jqt);
        double J[12];  // 2 x 6
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            J[6 * r + k] = sr * Jqt[3 * r + k];
            J[6 * r + 3 + k] = ((tmask >> k) & 1u) ? 0.0 : sr * Jt[3 * r + k];
          }
        const double r0 = sr * b.r[0], r1 = sr * b.r[1];
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = a; c < 6; ++c) acc[idx++] += J[a] * J[c] + J[6 + a] * J[6 + c];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r0 + J[6 + a] * r1;
        if (WANT_W && !(d.point_const && d.point_const[pt])) {
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) w[3 * a + c] = J[a] * (sr * JX[c]) + J[6 + a] * (sr * JX[3 + c]);
        }
      }
      if (WANT_W) {
        // rows of this wavefront: observation indices in the caller's order
        const uint32_t o = active ? d.img_obs[e] : 0u;
        const uint32_t o_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
        const int cnt = (int)min(64u, e_end > e0 + wave * 64u ? e_end - (e0 + wave * 64u) : 0u);
        const bool contiguous = __all(!active || o == o_first + (uint32_t)lane);
        if (contiguous) {
          wave_store_rows<18>(W_o, o_first, cnt, w, s_w[wave]);
        } else if (active) {
          double* dst = W_o + 18 * (size_t)o;
#pragma unroll
          for (int k = 0; k < 18; ++k) dst[k] = w[k];
        }
      }
    }
  }
  // fixed-order reduction: wave butterfly, then the 4 waves through LDS
  __shared__ double s_a[4][27];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) s_a[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27)
    partial[27 * (size_t)blockIdx.x + threadIdx.x] =
        (s_a[0][threadIdx.x] + s_a[1][threadIdx.x]) + (s_a[2][threadIdx.x] + s_a[3][threadIdx.x]);
}

// segments of an image added in ascending order -> 6x6 block (symmetric) + gradient
__global__ __launch_bounds__(64) void k_ba_images_reduce(BaDev d, const double* __restrict__ partial,
                                                         double* __restrict__ Himg, double* __restrict__ gimg) {
  const int im = blockIdx.x * 2 + (threadIdx.x >> 5), k = threadIdx.x & 31;
  if (im >= d.I || k >= 27) return;
  double v = 0.0;
  for (uint32_t sgm = d.img_seg_start[im]; sgm < d.img_seg_start[im + 1]; ++sgm) v += partial[27 * (size_t)sgm + k];
  if (k >= 21) {
    if (gimg) gimg[6 * (size_t)im + (k - 21)] = v;
  } else if (Himg) {
    int a = 0, kk = k;  // unpack the upper-triangle index
    while (kk >= 6 - a) { kk -= 6 - a; ++a; }
    const int c = a + kk;
    Himg[36 * (size_t)im + 6 * a + c] = v;
    Himg[36 * (size_t)im + 6 * c + a] = v;
  }
}

// ------------------------------------------------------------ cameras ------
// Camera blocks of the normal equations (refined intrinsics; ParameterizeCameras, optim/bundle_adjustment.cc:
// 1047-1100).  Per image the accumulation has NE = K(K+1)/2 + 6K + K entries -- upper triangle of Jc^T Jc,
// Jc^T Jp (K x 6), Jc^T r -- too many to keep per lane next to the Jacobians, so the four wavefronts of the
// workgroup each own a quarter of the entries and every wavefront sweeps all observations of the image
// (Jacobians recomputed four times; this pass only runs when intrinsics are refined, which the fork's defaults
// switch off).  Entry -> operand columns is resolved at compile time (K and the chunk are template constants).
// Fixed-order reductions, no atomics: bitwise reproducible.  k_ba_cameras_reduce then sums the images of a camera.
__host__ __device__ constexpr int cam_ne(int K) { return K * (K + 1) / 2 + 6 * K + K; }
// operand columns of entry e in A = [Jc (K) | Jp (6) | r (1)]
__host__ __device__ constexpr int cam_ent_l(int K, int e) {
  const int ncc = K * (K + 1) / 2;
  if (e < ncc) { int a = 0; while (e >= K - a) { e -= K - a; ++a; } return a; }
  e -= ncc;
  if (e < 6 * K) return e / 6;
  return e - 6 * K;
}
__host__ __device__ constexpr int cam_ent_r(int K, int e) {
  const int ncc = K * (K + 1) / 2;
  if (e < ncc) { int a = 0; while (e >= K - a) { e -= K - a; ++a; } return a + e; }
  e -= ncc;
  if (e < 6 * K) return K + e % 6;
  return K + 6;
}

template <int MODEL, int CH>
__device__ __forceinline__ void cam_accumulate(const BaDev& d, int im, double* __restrict__ partial) {
  constexpr int K = MODEL >= 0 ? cam_num_params(MODEL >= 0 ? MODEL : 0) : PCD_CAM_JAC_STRIDE;
  constexpr int NE = cam_ne(K), PER = (NE + 3) / 4, E0 = CH * PER, E1 = E0 + PER < NE ? E0 + PER : NE;
  constexpr int NA = K + 7;
  const int lane = threadIdx.x & 63;
  const bool cpose = d.image_const_pose && d.image_const_pose[im];
  const unsigned tmask = d.image_const_tvec ? d.image_const_tvec[im] : 0u;
  const int cm = d.image_cam[im];
  const double* cam = d.cam_params + d.cam_off[cm];
  const uint8_t* refine = d.cam_refine ? d.cam_refine + d.cam_off[cm] : nullptr;
  const int model = MODEL >= 0 ? MODEL : d.cam_model[cm];
  double acc[PER > 0 ? PER : 1];
#pragma unroll
  for (int t = 0; t < PER; ++t) acc[t] = 0.0;
  for (uint32_t e = d.img_obs_start[im] + lane; e < d.img_obs_start[im + 1]; e += 64) {
    const int pt = d.img_pt[e];
    const double X[3] = {d.points[3 * (size_t)pt], d.points[3 * (size_t)pt + 1], d.points[3 * (size_t)pt + 2]};
    ReprojBlock b;
    double q[4];
    eval_block<MODEL>(d, im, X, d.img_xy[2 * (size_t)e], d.img_xy[2 * (size_t)e + 1], b, q);
    double rho0, rho1;
    loss_eval(d.loss_type, d.loss_scale, b.r[0] * b.r[0] + b.r[1] * b.r[1], rho0, rho1);
    const double sr = sqrt(rho1);
    double A[2][NA];
    {  // camera columns: same normalised coordinates as reproj_eval
      const double* pose = d.poses + 7 * (size_t)im;
      const double w = pose[0], a = pose[1], bq = pose[2], c = pose[3];
      const double cx = bq * X[2] - c * X[1], cy = c * X[0] - a * X[2], cz = a * X[1] - bq * X[0];
      const double ux = 2.0 * cx, uy = 2.0 * cy, uz = 2.0 * cz;
      const double Px = X[0] + w * ux + (bq * uz - c * uy) + pose[4];
      const double Py = X[1] + w * uy + (c * ux - a * uz) + pose[5];
      const double Pz = X[2] + w * uz + (a * uy - bq * ux) + pose[6];
      const double iz = 1.0 / Pz;
      double Jc[2 * K];
#pragma unroll
      for (int k = 0; k < 2 * K; ++k) Jc[k] = 0.0;
      if (MODEL >= 0) cam_param_jacobian<(MODEL >= 0 ? MODEL : 0)>(cam, Px * iz, Py * iz, Jc, K);
      else cam_param_jacobian_any(model, cam, Px * iz, Py * iz, Jc, K);
      const int kn = MODEL >= 0 ? K : cam_num_params(model);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool var = refine && k < kn && refine[k < kn ? k : 0];
        A[0][k] = var ? sr * Jc[k] : 0.0;
        A[1][k] = var ? sr * Jc[K + k] : 0.0;
      }
    }
    double Jq[8], Jt[6], JX[6], Jqt[6];
    reproj_jacobians(b, Jq, Jt, JX);
    quat_tangent(q, Jq, Jqt);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        A[r][K + k] = cpose ? 0.0 : sr * Jqt[3 * r + k];
        A[r][K + 3 + k] = (cpose || ((tmask >> k) & 1u)) ? 0.0 : sr * Jt[3 * r + k];
      }
      A[r][K + 6] = sr * b.r[r];
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      if (E0 + t < E1) {
        const int l = cam_ent_l(K, E0 + t), r = cam_ent_r(K, E0 + t);
        acc[t] += A[0][l] * A[0][r] + A[1][l] * A[1][r];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < PER; ++t) {
    if (E0 + t < E1) {
      double v = acc[t];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) partial[(size_t)im * cam_ne(PCD_CAM_JAC_STRIDE) + E0 + t] = v;
    }
  }
}

template <int MODEL>
__global__ __launch_bounds__(256) void k_ba_cameras(BaDev d, double* __restrict__ partial) {
  const int im = blockIdx.x;
  switch (threadIdx.x >> 6) {
    case 0: cam_accumulate<MODEL, 0>(d, im, partial); break;
    case 1: cam_accumulate<MODEL, 1>(d, im, partial); break;
    case 2: cam_accumulate<MODEL, 2>(d, im, partial); break;
    default: cam_accumulate<MODEL, 3>(d, im, partial); break;
  }
}

// partial [I][NE(12)] (entries laid out for the image's own K) -> H_cam / g_cam per camera (images summed in
// ascending order) and E_cam per image
__global__ __launch_bounds__(256) void k_ba_cameras_reduce(BaDev d, const double* __restrict__ partial,
                                                           double* __restrict__ Hcam, double* __restrict__ gcam,
                                                           double* __restrict__ Ecam) {
  constexpr int S = PCD_CAM_JAC_STRIDE, NEMAX = cam_ne(S);
  const int c = blockIdx.x;
  const int model = d.cam_model[c];
  const int K = d.cam_k;
  (void)model;
  const int ncc = K * (K + 1) / 2, NE = ncc + 7 * K;
  for (int e = threadIdx.x; e < S * S + S; e += 256) {   // zero-fill, then the K x K / K part
    if (e < S * S) { if (Hcam) Hcam[(size_t)c * S * S + e] = 0.0; }
    else if (gcam) gcam[(size_t)c * S + (e - S * S)] = 0.0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NE; e += 256) {
    if (e >= ncc && e < ncc + 6 * K) continue;   // coupling entries are per image
    double v = 0.0;
    for (uint32_t j = d.cam_img_start[c]; j < d.cam_img_start[c + 1]; ++j)
      v += partial[(size_t)d.cam_img_list[j] * NEMAX + e];
    if (e < ncc) {
      int a = 0, k = e;
      while (k >= K - a) { k -= K - a; ++a; }
      const int b = a + k;
      if (Hcam) { Hcam[(size_t)c * S * S + a * S + b] = v; Hcam[(size_t)c * S * S + b * S + a] = v; }
    } else if (gcam) {
      gcam[(size_t)c * S + (e - ncc - 6 * K)] = v;
    }
  }
  if (Ecam) {
    for (uint32_t j = d.cam_img_start[c]; j < d.cam_img_start[c + 1]; ++j) {
      const uint32_t im = d.cam_img_list[j];
      for (int e = threadIdx.x; e < S * 6; e += 256) {
        const int a = e / 6;
        Ecam[(size_t)im * S * 6 + e] = a < K ? partial[(size_t)im * NEMAX + ncc + e] : 0.0;
      }
    }
  }
}

// camera x point coupling of every observation: W_cam[o] = Jc^T JX (S x 3, loss-corrected, masks applied)
template <int MODEL>
__global__ __launch_bounds__(256) void k_ba_cam_w(BaDev d, double* __restrict__ Wc_o) {
  constexpr int S = PCD_CAM_JAC_STRIDE;
  __shared__ __attribute__((aligned(16))) double s_rows[4][64 * 3 * S];
  const int wave = threadIdx.x >> 6;
  const uint64_t o = blockIdx.x * (uint64_t)256 + threadIdx.x;
  const uint64_t o_wave = blockIdx.x * (uint64_t)256 + wave * 64;
  if (o_wave >= d.O) return;
  const int cnt = (int)min((uint64_t)64, d.O - o_wave);
  double Wc[3 * S];
#pragma unroll
  for (int k = 0; k < 3 * S; ++k) Wc[k] = 0.0;
  if (o < d.O) {
    const int im = d.obs_image[o], pt = d.obs_point[o];
    const bool cpt = d.point_const && d.point_const[pt];
    const int cm = d.image_cam[im];
    const uint8_t* refine = d.cam_refine ? d.cam_refine + d.cam_off[cm] : nullptr;
    if (!cpt && refine) {
      const double X[3] = {d.points[3 * (size_t)pt], d.points[3 * (size_t)pt + 1], d.points[3 * (size_t)pt + 2]};
      ReprojBlock b;
      double q[4];
      eval_block<MODEL>(d, im, X, d.obs_xy[2 * o], d.obs_xy[2 * o + 1], b, q);
      double rho0, rho1;
      loss_eval(d.loss_type, d.loss_scale, b.r[0] * b.r[0] + b.r[1] * b.r[1], rho0, rho1);
      double Jq[8], Jt[6], JX[6];
      reproj_jacobians(b, Jq, Jt, JX);
      const double* pose = d.poses + 7 * (size_t)im;
      const double w = pose[0], a = pose[1], bq = pose[2], c = pose[3];
      const double cx = bq * X[2] - c * X[1], cy = c * X[0] - a * X[2], cz = a * X[1] - bq * X[0];
      const double ux = 2.0 * cx, uy = 2.0 * cy, uz = 2.0 * cz;
      const double Px = X[0] + w * ux + (bq * uz - c * uy) + pose[4];
      const double Py = X[1] + w * uy + (c * ux - a * uz) + pose[5];
      const double Pz = X[2] + w * uz + (a * uy - bq * ux) + pose[6];
      const double iz = 1.0 / Pz;
      const int model = MODEL >= 0 ? MODEL : d.cam_model[cm];
      double Jc[2 * S];
#pragma unroll
      for (int k = 0; k < 2 * S; ++k) Jc[k] = 0.0;
      if (MODEL >= 0) cam_param_jacobian<(MODEL >= 0 ? MODEL : 0)>(d.cam_params + d.cam_off[cm], Px * iz, Py * iz, Jc, S);
      else cam_param_jacobian_any(model, d.cam_params + d.cam_off[cm], Px * iz, Py * iz, Jc, S);
      const int kn = cam_num_params(model);
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const bool var = k < kn && refine[k < kn ? k : 0];
#pragma unroll
        for (int j = 0; j < 3; ++j)
          Wc[3 * k + j] = var ? rho1 * (Jc[k] * JX[j] + Jc[S + k] * JX[3 + j]) : 0.0;
      }
    }
  }
  wave_store_rows<3 * S>(Wc_o, o_wave, cnt, Wc, s_rows[wave]);
}

// ---------------------------------------------------------------- raw ------
template <int MODEL>
__global__ __launch_bounds__(256) void k_ba_raw(BaDev d, double* __restrict__ residuals, double* __restrict__ Jq_o,
                                                double* __restrict__ Jt_o, double* __restrict__ JX_o,
                                                double* __restrict__ W_o) {
  // thread = observation in the caller's order: the blocks of a wavefront are adjacent in every output array,
  // so each array is written through wave_store_rows (coalesced 16-B stores)
  __shared__ __attribute__((aligned(16))) double s_rows[4][64 * 18];
  const int wave = threadIdx.x >> 6;
  const uint64_t o = blockIdx.x * (uint64_t)256 + threadIdx.x;
  const uint64_t o_wave = blockIdx.x * (uint64_t)256 + wave * 64;
  if (o_wave >= d.O) return;   // whole wavefront past the end
  const int cnt = (int)min((uint64_t)64, d.O - o_wave);
  const bool active = o < d.O;
  double res[2] = {0, 0}, Jq[8], Jt[6], JX[6], Wb[18];
#pragma unroll
  for (int k = 0; k < 8; ++k) Jq[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) { Jt[k] = 0.0; JX[k] = 0.0; }
#pragma unroll
  for (int k = 0; k < 18; ++k) Wb[k] = 0.0;
  if (active) {
    const int im = d.obs_image[o], pt = d.obs_point[o];
    const double X[3] = {d.points[3 * (size_t)pt], d.points[3 * (size_t)pt + 1], d.points[3 * (size_t)pt + 2]};
    ReprojBlock b;
    double q[4];
    eval_block<MODEL>(d, im, X, d.obs_xy[2 * o], d.obs_xy[2 * o + 1], b, q);
    reproj_jacobians(b, Jq, Jt, JX);
    const bool cpose = d.image_const_pose && d.image_const_pose[im];
    res[0] = b.r[0]; res[1] = b.r[1];
    if (W_o) {
      double rho0, rho1;
      loss_eval(d.loss_type, d.loss_scale, b.r[0] * b.r[0] + b.r[1] * b.r[1], rho0, rho1);
      const double sr = sqrt(rho1);
      const unsigned tmask = d.image_const_tvec ? d.image_const_tvec[im] : 0u;
      const bool cpt = d.point_const && d.point_const[pt];
      double Jqt[6], J[12], Jx[6];
      quat_tangent(q, Jq, Jqt);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          J[6 * r + k] = cpose ? 0.0 : sr * Jqt[3 * r + k];
          J[6 * r + 3 + k] = (cpose || ((tmask >> k) & 1u)) ? 0.0 : sr * Jt[3 * r + k];
          Jx[3 * r + k] = cpt ? 0.0 : sr * JX[3 * r + k];
        }
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Wb[3 * a + c] = J[a] * Jx[c] + J[6 + a] * Jx[3 + c];
    }
    if (cpose) {   // the constant-pose functor has no pose blocks: zero rows
#pragma unroll
      for (int k = 0; k < 8; ++k) Jq[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) Jt[k] = 0.0;
    }
  }
  if (residuals) wave_store_rows<2>(residuals, o_wave, cnt, res, s_rows[wave]);
  if (Jq_o) wave_store_rows<8>(Jq_o, o_wave, cnt, Jq, s_rows[wave]);
  if (Jt_o) wave_store_rows<6>(Jt_o, o_wave, cnt, Jt, s_rows[wave]);
  if (JX_o) wave_store_rows<6>(JX_o, o_wave, cnt, JX, s_rows[wave]);
  if (W_o) wave_store_rows<18>(W_o, o_wave, cnt, Wb, s_rows[wave]);
}

// Camera-parameter block of every reprojection residual (2 x K, row stride PCD_CAM_JAC_STRIDE).
template <int MODEL>
__global__ __launch_bounds__(256) void k_ba_cam_jac(BaDev d, double* __restrict__ Jc_o) {
  const uint64_t o = blockIdx.x * (uint64_t)256 + threadIdx.x;
  if (o >= d.O) return;
  const int im = d.obs_image[o], pt = d.obs_point[o];
  const double* pose = d.poses + 7 * (size_t)im;
  const double* Xp = d.points + 3 * (size_t)pt;
  const double X[3] = {Xp[0], Xp[1], Xp[2]};
  const double w = pose[0], a = pose[1], bq = pose[2], c = pose[3];
  // same rotation polynomial as reproj_eval
  const double cx = bq * X[2] - c * X[1], cy = c * X[0] - a * X[2], cz = a * X[1] - bq * X[0];
  const double ux = 2.0 * cx, uy = 2.0 * cy, uz = 2.0 * cz;
  const double Px = X[0] + w * ux + (bq * uz - c * uy) + pose[4];
  const double Py = X[1] + w * uy + (c * ux - a * uz) + pose[5];
  const double Pz = X[2] + w * uz + (a * uy - bq * ux) + pose[6];
  const double iz = 1.0 / Pz;
  const int cm = d.image_cam[im];
  double J[2 * PCD_CAM_JAC_STRIDE];
#pragma unroll
  for (int k = 0; k < 2 * PCD_CAM_JAC_STRIDE; ++k) J[k] = 0.0;
  if (MODEL >= 0) cam_param_jacobian<(MODEL >= 0 ? MODEL : 0)>(d.cam_params + d.cam_off[cm], Px * iz, Py * iz, J, PCD_CAM_JAC_STRIDE);
  else cam_param_jacobian_any(d.cam_model[cm], d.cam_params + d.cam_off[cm], Px * iz, Py * iz, J, PCD_CAM_JAC_STRIDE);
  double* out = Jc_o + 2 * PCD_CAM_JAC_STRIDE * o;
#pragma unroll
  for (int k = 0; k < 2 * PCD_CAM_JAC_STRIDE; ++k) out[k] = J[k];
}

// Inputs of the post-BA filters, per observation:
//   sq_err = CalculateSquaredReprojectionError (base/projection.cc:104-117; quaternion normalised first as
//            base/pose.cc QuaternionRotatePoint does; DBL_MAX when the point is not in front of the camera)
//   depth  = P.z (FilterObservationsWithNegativeDepth, base/reconstruction.cc:837-855, tests it against eps)
template <int MODEL>
__global__ __launch_bounds__(256) void k_ba_obs_errors(BaDev d, double* __restrict__ sq_err, double* __restrict__ depth) {
  const uint64_t o = blockIdx.x * (uint64_t)256 + threadIdx.x;
  if (o >= d.O) return;
  const int im = d.obs_image[o], pt = d.obs_point[o];
  const double* pose = d.poses + 7 * (size_t)im;
  double q[4] = {pose[0], pose[1], pose[2], pose[3]};
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n == 0.0) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
  const double X[3] = {d.points[3 * (size_t)pt], d.points[3 * (size_t)pt + 1], d.points[3 * (size_t)pt + 2]};
  const double cx = q[2] * X[2] - q[3] * X[1], cy = q[3] * X[0] - q[1] * X[2], cz = q[1] * X[1] - q[2] * X[0];
  const double ux = 2.0 * cx, uy = 2.0 * cy, uz = 2.0 * cz;
  const double Px = X[0] + q[0] * ux + (q[2] * uz - q[3] * uy) + pose[4];
  const double Py = X[1] + q[0] * uy + (q[3] * ux - q[1] * uz) + pose[5];
  const double Pz = X[2] + q[0] * uz + (q[1] * uy - q[2] * ux) + pose[6];
  if (depth) depth[o] = Pz;
  if (!sq_err) return;
  if (Pz < 2.220446049250313e-16) { sq_err[o] = 1.7976931348623157e308; return; }
  const int cm = d.image_cam[im];
  const int model = MODEL >= 0 ? MODEL : d.cam_model[cm];
  D2 x, y;
  world_to_image_d2(model, d.cam_params + d.cam_off[cm], Px / Pz, Py / Pz, x, y);
  const double dx = x.a - d.obs_xy[2 * o], dy = y.a - d.obs_xy[2 * o + 1];
  sq_err[o] = dx * dx + dy * dy;
}

__global__ __launch_bounds__(256) void k_ba_lidar_raw(BaDev d, double* __restrict__ residuals, double* __restrict__ JL) {
  const uint64_t l = blockIdx.x * (uint64_t)256 + threadIdx.x;
  if (l >= d.L) return;
  const int p = d.lidar_point[l];
  const double X[3] = {d.points[3 * (size_t)p], d.points[3 * (size_t)p + 1], d.points[3 * (size_t)p + 2]};
  const double abcd[4] = {d.lidar_abcd[4 * l], d.lidar_abcd[4 * l + 1], d.lidar_abcd[4 * l + 2], d.lidar_abcd[4 * l + 3]};
  double r, J[3];
  lidar_eval(X, abcd, d.lidar_w[l], 0, r, J);
  if (residuals) residuals[2 * d.O + l] = r;
  if (JL) { JL[3 * l] = J[0]; JL[3 * l + 1] = J[1]; JL[3 * l + 2] = J[2]; }
}

// ---- post-BA filters, reduced per track (base/reconstruction.cc:1662-1712, :837-855, :906-921) -----------------
// thread = point; its observations in ascending observation index (pt_obs_list)
__global__ __launch_bounds__(256) void k_ba_filter_tracks(int P, const uint32_t* __restrict__ pt_start,
                                                          const uint32_t* __restrict__ pt_list,
                                                          const double* __restrict__ sq_err, double max_sq,
                                                          uint8_t* __restrict__ obs_erase, uint8_t* __restrict__ point_delete,
                                                          double* __restrict__ point_error, double* __restrict__ partial) {
  __shared__ double s_f[4], s_e[4], s_n[4];
  const int p = blockIdx.x * 256 + threadIdx.x;
  double filtered = 0.0, esum = 0.0, valid = 0.0;
  if (p < P) {
    const uint32_t b = pt_start[p], len = pt_start[p + 1] - b;
    uint32_t ndel = 0;
    double sum = 0.0;
    for (uint32_t j = 0; j < len; ++j) {
      const double e = sq_err[pt_list[b + j]];
      if (e > max_sq) ++ndel; else sum += sqrt(e);
    }
    // reconstruction.cc:1677-1681 (length < 2) and :1700-1702 (at most one element survives): DeletePoint3D
    const bool del = len < 2 || ndel + 1 >= len;
    if (obs_erase)
      for (uint32_t j = 0; j < len; ++j) {
        const uint32_t o = pt_list[b + j];
        obs_erase[o] = (del || sq_err[o] > max_sq) ? 1 : 0;
      }
    filtered = del ? (double)len : (double)ndel;
    const double err = del ? -1.0 : sum / (double)(len - ndel);   // Track().Length() after the deletions (:1708)
    if (point_delete) point_delete[p] = del ? 1 : 0;
    if (point_error) point_error[p] = err;
    if (!del) { esum = err; valid = 1.0; }
  }
  // fixed-order reduction: lanes by butterfly, the 4 wavefronts in order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    filtered += __shfl_xor(filtered, off); esum += __shfl_xor(esum, off); valid += __shfl_xor(valid, off);
  }
  if (lane == 0) { s_f[wave] = filtered; s_e[wave] = esum; s_n[wave] = valid; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[3 * (size_t)blockIdx.x] = (s_f[0] + s_f[1]) + (s_f[2] + s_f[3]);
    partial[3 * (size_t)blockIdx.x + 1] = (s_e[0] + s_e[1]) + (s_e[2] + s_e[3]);
    partial[3 * (size_t)blockIdx.x + 2] = (s_n[0] + s_n[1]) + (s_n[2] + s_n[3]);
  }
}
__global__ __launch_bounds__(256) void k_ba_negative_depth(uint64_t O, const double* __restrict__ depth,
                                                           uint8_t* __restrict__ flag, double* __restrict__ partial) {
  __shared__ double s_c[4];
  const uint64_t o = blockIdx.x * (uint64_t)256 + threadIdx.x;
  const bool neg = o < O && depth[o] < 2.220446049250313e-16;   // !HasPointPositiveDepth (base/projection.cc:191-195)
  if (o < O && flag) flag[o] = neg ? 1 : 0;
  double c = neg ? 1.0 : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
}
// one workgroup: partials added in block order (strided per thread, then the threads in order)
__global__ __launch_bounds__(256) void k_ba_filter_summary(const double* __restrict__ part3, int nb3,
                                                           const double* __restrict__ part1, int nb1,
                                                           double* __restrict__ summary) {
  __shared__ double s_v[4][256];
  double a[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < nb3; b += 256) { a[0] += part3[3 * (size_t)b]; a[1] += part3[3 * (size_t)b + 1]; a[2] += part3[3 * (size_t)b + 2]; }
  for (int b = threadIdx.x; b < nb1; b += 256) a[3] += part1[b];
  for (int k = 0; k < 4; ++k) s_v[k][threadIdx.x] = a[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4] = {0, 0, 0, 0};
    for (int i = 0; i < 256; ++i) for (int k = 0; k < 4; ++k) t[k] += s_v[k][i];
    summary[0] = t[0];
    summary[1] = t[2] > 0.0 ? t[1] / t[2] : 0.0;   // ComputeMeanReprojectionError: 0 when no point has an error
    summary[2] = t[2];
    summary[3] = t[3];
  }
}

// rows of the variable-pose observations, packed: thread = 16-byte unit of an output row
template <int N>   // doubles per row
__global__ void k_pack_rows(const double* __restrict__ in, const uint32_t* __restrict__ vobs, uint64_t nrows,
                            double* __restrict__ out) {
  const uint64_t u = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  constexpr int UPR = N / 2;
  if (u >= nrows * UPR) return;
  const uint64_t r = u / UPR;
  const int k = (int)(u - r * UPR);
  reinterpret_cast<double2*>(out)[u] = reinterpret_cast<const double2*>(in + (size_t)vobs[r] * N)[k];
}

}  // namespace pcd

using namespace pcd;

struct pcd_ba {
  int device = 0;
  int C = 0, I = 0, P = 0, nslices = 0;
  uint64_t O = 0, L = 0, cam_params_len = 0;
  int loss_type = 0;
  double loss_scale = 1.0;
  int uniform_model = -1;  // >= 0: every camera has this model
  DevBuf<int> cam_model, cam_off, image_cam, obs_image, obs_point, lidar_point, pt_order, sell_img, img_pt;
  DevBuf<double> cam_params, poses, points, obs_xy, lidar_abcd, lidar_w, sell_xy, img_xy;
  DevBuf<uint8_t> image_const_pose, image_const_tvec, point_const;
  bool has_cpose = false, has_ctvec = false, has_cpt = false;
  DevBuf<uint32_t> slice_start, pt_lidar_start, pt_lidar_list, img_obs_start, img_obs, cam_img_start, cam_img_list;
  DevBuf<uint32_t> seg_img, seg_begin, img_seg_start;
  uint32_t nseg = 0;
  DevBuf<double> img_partial;
  DevBuf<uint8_t> cam_refine;
  bool has_refine = false;
  DevBuf<double> cam_partial;
  DevBuf<double> cost_partial, cost;
  // host-API staging
  DevBuf<double> o_res, o_jq, o_jt, o_jx, o_jl, o_himg, o_gimg, o_hpt, o_gpt, o_w, o_jc, o_hcam, o_gcam, o_ecam, o_wcam;
  // pcd_ba_evaluate_blocks: rows of the variable-pose observations, packed pose Jacobians, pinned results
  std::vector<uint32_t> h_pose_row;      // [O] row of observation o in the packed jac_q / jac_t (0xFFFFFFFF: constant pose)
  uint64_t n_pose_rows = 0;
  DevBuf<uint32_t> vobs;                 // [n_pose_rows] observation of every packed row
  DevBuf<double> p_jq, p_jt;             // packed pose Jacobians (only when some pose is constant)
  PinnedBuf<double> h_blocks;            // residuals | jac_q | jac_t | jac_X | jac_lidar | jac_cam
  // pcd_ba_filter_tracks: the track CSR (point -> its observations, ascending), scratch
  DevBuf<uint32_t> pt_obs_start, pt_obs_list;
  DevBuf<double> f_sq, f_depth, f_part, f_summary;
  DevBuf<uint8_t> f_u8;
  BaDev dev() const {
    BaDev d;
    d.cam_model = cam_model.p; d.cam_off = cam_off.p; d.cam_params = cam_params.p;
    d.poses = poses.p; d.image_cam = image_cam.p;
    d.image_const_pose = has_cpose ? image_const_pose.p : nullptr;
    d.image_const_tvec = has_ctvec ? image_const_tvec.p : nullptr;
    d.points = points.p; d.point_const = has_cpt ? point_const.p : nullptr;
    d.obs_image = obs_image.p; d.obs_point = obs_point.p; d.obs_xy = obs_xy.p;
    d.lidar_point = lidar_point.p; d.lidar_abcd = lidar_abcd.p; d.lidar_w = lidar_w.p;
    d.pt_order = pt_order.p; d.slice_start = slice_start.p; d.sell_img = sell_img.p; d.sell_xy = sell_xy.p;
    d.pt_lidar_start = pt_lidar_start.p; d.pt_lidar_list = pt_lidar_list.p;
    d.img_obs_start = img_obs_start.p; d.img_pt = img_pt.p; d.img_xy = img_xy.p; d.img_obs = img_obs.p;
    d.seg_img = seg_img.p; d.seg_begin = seg_begin.p; d.img_seg_start = img_seg_start.p;
    d.cam_refine = has_refine ? cam_refine.p : nullptr; d.cam_img_start = cam_img_start.p; d.cam_img_list = cam_img_list.p;
    d.C = C; d.cam_k = (uniform_model >= 0 && uniform_model <= 4) ? cam_num_params(uniform_model) : PCD_CAM_JAC_STRIDE;
    d.I = I; d.P = P; d.nslices = nslices; d.O = O; d.L = L; d.loss_type = loss_type; d.loss_scale = loss_scale;
    return d;
  }
};

template <typename T>
static pcd_status upload(DevBuf<T>& b, const T* src, size_t n) {
  PCD_TRY(b.reserve(std::max<size_t>(n, 1)));
  if (n) PCD_HIP_TRY(hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice));
  return PCD_OK;
}

// ---- derived layouts, filled on the device from the uploaded observation arrays -----------------------------
// The host only sorts indices (counting sorts); the 16-byte observation payloads never make a second trip over PCIe
// and are never gathered by a host loop (pcd_ba_create: 220 -> see DESIGN 4.3 for the bench scene).
// sliced ELL: thread = (slice, lane): track p = order[slice*64 + lane], its j-th observation goes to slot
// slice_start[slice] + 64 j + lane
__global__ void k_ba_fill_sell(const int* __restrict__ order, const uint32_t* __restrict__ slice_start,
                               const uint32_t* __restrict__ pt_start, const uint32_t* __restrict__ pt_list,
                               const int* __restrict__ obs_image, const double* __restrict__ obs_xy, int nslices,
                               int* __restrict__ sell_img, double* __restrict__ sell_xy) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int slice = t >> 6, lane = t & 63;
  if (slice >= nslices) return;
  const uint32_t s0 = slice_start[slice], width = (slice_start[slice + 1] - s0) >> 6;
  const int p = order[t];
  const uint32_t b = p >= 0 ? pt_start[p] : 0u, len = p >= 0 ? pt_start[p + 1] - b : 0u;
  for (uint32_t j = 0; j < width; ++j) {
    const size_t slot = (size_t)s0 + 64 * (size_t)j + lane;
    if (j < len) {
      const uint32_t o = pt_list[b + j];
      sell_img[slot] = obs_image[o];
      const double2 xy = *reinterpret_cast<const double2*>(obs_xy + 2 * (size_t)o);
      *reinterpret_cast<double2*>(sell_xy + 2 * slot) = xy;
    } else {
      sell_img[slot] = -1;   // padding of a shorter track
      *reinterpret_cast<double2*>(sell_xy + 2 * slot) = make_double2(0.0, 0.0);
    }
  }
}
// image-major copies: e-th observation of the image-major order = observation img_obs[e] of the caller's order
__global__ void k_ba_fill_image_major(const uint32_t* __restrict__ img_obs, const int* __restrict__ obs_point,
                                      const double* __restrict__ obs_xy, uint64_t O, int* __restrict__ img_pt,
                                      double* __restrict__ img_xy) {
  const uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (e >= O) return;
  const uint32_t o = img_obs[e];
  img_pt[e] = obs_point[o];
  *reinterpret_cast<double2*>(img_xy + 2 * e) = *reinterpret_cast<const double2*>(obs_xy + 2 * (size_t)o);
}

// stable counting sort of element ids by key -> CSR (start[nkeys+1], list[n])
static void build_csr(const int32_t* key, uint64_t n, int nkeys, std::vector<uint32_t>& start,
                      std::vector<uint32_t>& list) {
  start.assign((size_t)nkeys + 1, 0);
  for (uint64_t i = 0; i < n; ++i) start[(size_t)key[i] + 1]++;
  for (int k = 0; k < nkeys; ++k) start[k + 1] += start[k];
  list.resize(n);
  std::vector<uint32_t> cur(start.begin(), start.end() - 1);
  for (uint64_t i = 0; i < n; ++i) list[cur[key[i]]++] = (uint32_t)i;
}

// kernel dispatch on the (uniform) camera model; the five common models are compiled in
#define PCD_BA_DISPATCH(MODELVAR, ...)                            \
  switch (MODELVAR) {                                             \
    case 0: { constexpr int M = 0; __VA_ARGS__; } break;          \
    case 1: { constexpr int M = 1; __VA_ARGS__; } break;          \
    case 2: { constexpr int M = 2; __VA_ARGS__; } break;          \
    case 3: { constexpr int M = 3; __VA_ARGS__; } break;          \
    case 4: { constexpr int M = 4; __VA_ARGS__; } break;          \
    default: { constexpr int M = -1; __VA_ARGS__; } break;        \
  }

// Grid of the kernel that writes cost_partial[blockIdx.x]: k_ba_points takes two 64-track slices per workgroup,
// k_ba_cost sweeps the residual blocks with a capped grid.  pcd_ba_create sizes cost_partial from the SAME function
// (round 2 sized it for an older k_ba_points grid: scenes with O + L < ~2 P wrote past the end).
static unsigned cost_blocks(const pcd_ba* b, bool want_blocks) {
  return want_blocks ? std::max(1u, div_up((size_t)b->nslices, 2))
                     : std::max(1u, std::min(kCostBlocks, div_up(b->O + b->L, 256)));
}

extern "C" {

pcd_status pcd_ba_create(const pcd_ba_desc* d, pcd_ba** out) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(d && out, "null pointer");
  *out = nullptr;
  PCD_REQUIRE(d->num_cameras > 0 && d->cam_model && d->cam_param_offset && d->cam_params, "cameras");
  PCD_REQUIRE(d->num_images > 0 && d->poses && d->image_camera, "images");
  PCD_REQUIRE(d->num_points > 0 && d->points, "points");
  PCD_REQUIRE(d->num_obs == 0 || (d->obs_image && d->obs_point && d->obs_xy), "observations");
  PCD_REQUIRE(d->num_lidar == 0 || (d->lidar_point && d->lidar_abcd && d->lidar_weight), "lidar terms");
  PCD_REQUIRE(d->loss_type >= 0 && d->loss_type <= 2, "loss_type");
  PCD_REQUIRE(d->loss_type == PCD_LOSS_TRIVIAL || d->loss_scale > 0, "loss_scale");
  PCD_REQUIRE(d->num_obs < 0xFFFFFFF0ull && d->num_lidar < 0xFFFFFFF0ull, "too many residual blocks");
  for (int c = 0; c < d->num_cameras; ++c) {
    const int k = pcd_camera_num_params(d->cam_model[c]);
    PCD_REQUIRE(k > 0, "unknown camera model id");  // reference: std::domain_error, camera_models.h:140
    PCD_REQUIRE(d->cam_param_offset[c] >= 0 && (uint64_t)d->cam_param_offset[c] + k <= d->cam_params_len,
                "camera parameter offsets");
  }
  for (int i = 0; i < d->num_images; ++i)
    PCD_REQUIRE(d->image_camera[i] >= 0 && d->image_camera[i] < d->num_cameras, "image_camera out of range");
  for (uint64_t o = 0; o < d->num_obs; ++o) {
    PCD_REQUIRE(d->obs_image[o] >= 0 && d->obs_image[o] < d->num_images, "obs_image out of range");
    PCD_REQUIRE(d->obs_point[o] >= 0 && d->obs_point[o] < d->num_points, "obs_point out of range");
  }
  for (uint64_t l = 0; l < d->num_lidar; ++l)
    PCD_REQUIRE(d->lidar_point[l] >= 0 && d->lidar_point[l] < d->num_points, "lidar_point out of range");
  PCD_TRY(require_device(d->device));

  pcd_ba* b = new pcd_ba();
  b->device = d->device;
  b->C = d->num_cameras; b->I = d->num_images; b->P = d->num_points; b->O = d->num_obs; b->L = d->num_lidar;
  b->loss_type = d->loss_type; b->loss_scale = d->loss_scale;
  b->cam_params_len = d->cam_params_len;
  b->uniform_model = d->cam_model[0];
  for (int c = 1; c < d->num_cameras; ++c)
    if (d->cam_model[c] != b->uniform_model) b->uniform_model = -1;
  auto fail = [&](pcd_status st) { pcd_ba_destroy(b); return st; };
#define UP(buf, src, n) do { pcd_status _st = upload(b->buf, src, (size_t)(n)); if (_st != PCD_OK) return fail(_st); } while (0)
  UP(cam_model, d->cam_model, b->C); UP(cam_off, d->cam_param_offset, b->C); UP(cam_params, d->cam_params, d->cam_params_len);
  UP(poses, d->poses, 7 * (size_t)b->I); UP(image_cam, d->image_camera, b->I);
  UP(points, d->points, 3 * (size_t)b->P);
  UP(obs_image, d->obs_image, b->O); UP(obs_point, d->obs_point, b->O); UP(obs_xy, d->obs_xy, 2 * b->O);
  UP(lidar_point, d->lidar_point, b->L); UP(lidar_abcd, d->lidar_abcd, 4 * b->L); UP(lidar_w, d->lidar_weight, b->L);
  if (d->image_const_pose) { b->has_cpose = true; UP(image_const_pose, d->image_const_pose, b->I); }
  if (d->image_const_tvec) { b->has_ctvec = true; UP(image_const_tvec, d->image_const_tvec, b->I); }
  if (d->point_const) { b->has_cpt = true; UP(point_const, d->point_const, b->P); }
  if (d->camera_refine) { b->has_refine = true; UP(cam_refine, d->camera_refine, d->cam_params_len); }

  std::vector<uint32_t> st, li;
  // ---- per-track sliced ELL in order of track length ----
  build_csr(d->obs_point, b->O, b->P, st, li);
  {
    // tracks in ascending order of length, ties in ascending point id: a counting sort (lengths are small numbers)
    uint32_t maxlen = 0;
    for (int p = 0; p < b->P; ++p) maxlen = std::max(maxlen, st[p + 1] - st[p]);
    std::vector<uint32_t> bucket((size_t)maxlen + 2, 0);
    for (int p = 0; p < b->P; ++p) bucket[(size_t)(st[p + 1] - st[p]) + 1]++;
    for (size_t k = 1; k < bucket.size(); ++k) bucket[k] += bucket[k - 1];
    const int nslices = (b->P + 63) / 64;
    b->nslices = nslices;
    std::vector<int> order((size_t)nslices * 64, -1);
    for (int p = 0; p < b->P; ++p) order[bucket[st[p + 1] - st[p]]++] = p;
    std::vector<uint32_t> slice_start(nslices + 1, 0);
    for (int s = 0; s < nslices; ++s) {
      // ascending lengths: the longest track of a slice is its last real one
      uint32_t mx = 0;
      for (int l = 63; l >= 0; --l) {
        const int p = order[(size_t)s * 64 + l];
        if (p >= 0) { mx = st[p + 1] - st[p]; break; }
      }
      slice_start[s + 1] = slice_start[s] + mx * 64;
    }
    const size_t nslots = slice_start[nslices];
    UP(pt_order, order.data(), order.size());
    UP(slice_start, slice_start.data(), slice_start.size());
    UP(pt_obs_start, st.data(), st.size()); UP(pt_obs_list, li.data(), li.size());   // kept: pcd_ba_filter_tracks
    // the ELL fill borrows the buffers of the lidar CSR uploaded right after
    UP(pt_lidar_start, st.data(), st.size()); UP(pt_lidar_list, li.data(), li.size());
    pcd_status sa = b->sell_img.reserve(std::max<size_t>(nslots, 1));
    if (sa == PCD_OK) sa = b->sell_xy.reserve(std::max<size_t>(2 * nslots, 2));
    if (sa != PCD_OK) return fail(sa);
    if (nslots)
      hipLaunchKernelGGL(k_ba_fill_sell, dim3(div_up((size_t)nslices * 64, 256)), dim3(256), 0, nullptr, b->pt_order.p,
                         b->slice_start.p, b->pt_lidar_start.p, b->pt_lidar_list.p, b->obs_image.p, b->obs_xy.p, nslices,
                         b->sell_img.p, b->sell_xy.p);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
      set_error("pcd_ba_create: filling the per-track layout failed");
      return fail(PCD_ERR_HIP);
    }
  }
  build_csr(d->lidar_point, b->L, b->P, st, li);
  UP(pt_lidar_start, st.data(), st.size()); UP(pt_lidar_list, li.data(), li.size());
  // ---- per-image contiguous copies ----
  build_csr(d->obs_image, b->O, b->I, st, li);
  {
    UP(img_obs_start, st.data(), st.size());
    {  // segments of <= kImgSeg observations, never spanning two images (an image without observations has none)
      std::vector<uint32_t> seg_img, seg_begin, img_seg_start(b->I + 1, 0);
      for (int i = 0; i < b->I; ++i) {
        img_seg_start[i] = (uint32_t)seg_img.size();
        for (uint32_t e = st[i]; e < st[i + 1]; e += kImgSeg) { seg_img.push_back((uint32_t)i); seg_begin.push_back(e); }
      }
      img_seg_start[b->I] = (uint32_t)seg_img.size();
      b->nseg = (uint32_t)seg_img.size();
      seg_begin.push_back((uint32_t)b->O);
      UP(seg_img, seg_img.data(), seg_img.size());
      UP(seg_begin, seg_begin.data(), seg_begin.size());
      UP(img_seg_start, img_seg_start.data(), img_seg_start.size());
    }
    {  // images of each camera, ascending
      std::vector<uint32_t> cst, cli;
      build_csr(d->image_camera, (uint64_t)b->I, b->C, cst, cli);
      UP(cam_img_start, cst.data(), cst.size());
      UP(cam_img_list, cli.data(), cli.size());
    }
    UP(img_obs, li.data(), li.size());
    pcd_status sa = b->img_pt.reserve(std::max<size_t>(b->O, 1));
    if (sa == PCD_OK) sa = b->img_xy.reserve(std::max<size_t>(2 * b->O, 2));
    if (sa != PCD_OK) return fail(sa);
    if (b->O)
      hipLaunchKernelGGL(k_ba_fill_image_major, dim3(div_up(b->O, 256)), dim3(256), 0, nullptr, b->img_obs.p,
                         b->obs_point.p, b->obs_xy.p, b->O, b->img_pt.p, b->img_xy.p);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
      set_error("pcd_ba_create: filling the image-major layout failed");
      return fail(PCD_ERR_HIP);
    }
  }
#undef UP
  {  // rows of the packed pose Jacobians (pcd_ba_evaluate_blocks)
    b->h_pose_row.assign(b->O, 0xFFFFFFFFu);
    std::vector<uint32_t> vobs;
    vobs.reserve(b->O);
    for (uint64_t o = 0; o < b->O; ++o)
      if (!(d->image_const_pose && d->image_const_pose[d->obs_image[o]])) {
        b->h_pose_row[o] = (uint32_t)vobs.size();
        vobs.push_back((uint32_t)o);
      }
    b->n_pose_rows = vobs.size();
    if (b->n_pose_rows != b->O) {
      pcd_status sv = upload(b->vobs, vobs.data(), vobs.size());
      if (sv != PCD_OK) return fail(sv);
    }
  }
  // one partial per workgroup of whichever of the two cost-producing kernels has the larger grid (cost_blocks)
  pcd_status s1 = b->cost_partial.reserve((size_t)std::max(cost_blocks(b, true), cost_blocks(b, false)) + 1);
  if (s1 != PCD_OK) return fail(s1);
  if ((s1 = b->cost.reserve(1)) != PCD_OK) return fail(s1);
  *out = b;
  return PCD_OK;
  });
}

void pcd_ba_destroy(pcd_ba* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  delete b;
}

pcd_status pcd_ba_set_parameters(pcd_ba* b, const double* poses, const double* points) {
  PCD_REQUIRE(b, "null handle");
  PCD_HIP_TRY(hipSetDevice(b->device));
  if (poses) PCD_HIP_TRY(hipMemcpy(b->poses.p, poses, 7 * (size_t)b->I * sizeof(double), hipMemcpyHostToDevice));
  if (points) PCD_HIP_TRY(hipMemcpy(b->points.p, points, 3 * (size_t)b->P * sizeof(double), hipMemcpyHostToDevice));
  return PCD_OK;
}

pcd_status pcd_ba_set_camera_parameters(pcd_ba* b, const double* cam_params) {
  PCD_REQUIRE(b && cam_params, "null pointer");
  PCD_HIP_TRY(hipSetDevice(b->device));
  PCD_HIP_TRY(hipMemcpy(b->cam_params.p, cam_params, b->cam_params_len * sizeof(double), hipMemcpyHostToDevice));
  return PCD_OK;
}

pcd_status pcd_ba_device_parameters(pcd_ba* b, double** d_poses, double** d_points) {
  PCD_REQUIRE(b, "null handle");
  if (d_poses) *d_poses = b->poses.p;
  if (d_points) *d_points = b->points.p;
  return PCD_OK;
}

pcd_status pcd_ba_evaluate_device(pcd_ba* b, const pcd_ba_out* o, void* stream) {
  PCD_REQUIRE(b && o, "null pointer");
  PCD_REFUSE_CAPTURE(stream);
  PCD_HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  const BaDev d = b->dev();
  const int model = b->uniform_model;
  if (o->cost || o->H_pt || o->g_pt) {
    const bool want_blocks = o->H_pt || o->g_pt;
    const unsigned blocks = cost_blocks(b, want_blocks);
    {
      ScopedKernelTimer t(want_blocks ? "ba_points" : "ba_points_cost", s);
      if (want_blocks) {
        PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_points<M, true>), dim3(blocks), dim3(256), 0, s, d, o->H_pt,
                                                   o->g_pt, b->cost_partial.p));
      } else {
        PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_cost<M>), dim3(blocks), dim3(256), 0, s, d, b->cost_partial.p));
      }
    }
    if (o->cost) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, b->cost_partial.p, (int)blocks, o->cost);
  }
  // W rides on the image pass when that runs anyway (normal-equation mode); otherwise the raw kernel fills it
  const bool w_fused = o->W && (o->H_img || o->g_img) && b->O;
  if (o->H_img || o->g_img) {
    PCD_TRY(b->img_partial.reserve(27 * (size_t)std::max(b->nseg, 1u)));
    ScopedKernelTimer t(w_fused ? "ba_images_w" : "ba_images", s);
    if (b->nseg) {
      if (w_fused) {
        PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_images<M, true>), dim3(b->nseg), dim3(256), 0, s, d,
                                                   b->img_partial.p, o->W));
      } else {
        PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_images<M, false>), dim3(b->nseg), dim3(256), 0, s, d,
                                                   b->img_partial.p, (double*)nullptr));
      }
    }
    hipLaunchKernelGGL(k_ba_images_reduce, dim3(div_up(b->I, 2)), dim3(64), 0, s, d, b->img_partial.p, o->H_img, o->g_img);
  }
  double* const W_raw = w_fused ? nullptr : o->W;
  if ((o->residuals || o->jac_q || o->jac_t || o->jac_X || W_raw) && b->O) {
    ScopedKernelTimer t("ba_raw", s);
    PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_raw<M>), dim3(div_up(b->O, 256)), dim3(256), 0, s, d, o->residuals,
                                               o->jac_q, o->jac_t, o->jac_X, W_raw));
  }
  if (o->jac_cam && b->O) {
    ScopedKernelTimer t("ba_cam_jac", s);
    PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_cam_jac<M>), dim3(div_up(b->O, 256)), dim3(256), 0, s, d, o->jac_cam));
  }
  if (o->H_cam || o->g_cam || o->E_cam) {
    PCD_TRY(b->cam_partial.reserve((size_t)b->I * cam_ne(PCD_CAM_JAC_STRIDE)));
    ScopedKernelTimer t("ba_cameras", s);
    PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_cameras<M>), dim3(b->I), dim3(256), 0, s, d, b->cam_partial.p));
    hipLaunchKernelGGL(k_ba_cameras_reduce, dim3(b->C), dim3(256), 0, s, d, b->cam_partial.p, o->H_cam, o->g_cam,
                       o->E_cam);
  }
  if (o->W_cam && b->O) {
    ScopedKernelTimer t("ba_cam_w", s);
    PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_cam_w<M>), dim3(div_up(b->O, 256)), dim3(256), 0, s, d, o->W_cam));
  }
  if ((o->residuals || o->jac_lidar) && b->L) {
    ScopedKernelTimer t("ba_lidar_raw", s);
    hipLaunchKernelGGL(k_ba_lidar_raw, dim3(div_up(b->L, 256)), dim3(256), 0, s, d, o->residuals, o->jac_lidar);
  }
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

pcd_status pcd_ba_observation_errors_device(pcd_ba* b, double* d_sq_err, double* d_depth, void* stream) {
  PCD_REQUIRE(b, "null handle");
  PCD_REFUSE_CAPTURE(stream);
  if (!b->O || (!d_sq_err && !d_depth)) return PCD_OK;
  PCD_HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  const BaDev d = b->dev();
  const int model = b->uniform_model;
  ScopedKernelTimer t("ba_obs_errors", s);
  PCD_BA_DISPATCH(model, hipLaunchKernelGGL((k_ba_obs_errors<M>), dim3(div_up(b->O, 256)), dim3(256), 0, s, d, d_sq_err,
                                            d_depth));
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

pcd_status pcd_ba_observation_errors(pcd_ba* b, double* sq_err, double* depth) {
  PCD_REQUIRE(b, "null handle");
  if (!b->O) return PCD_OK;
  PCD_HIP_TRY(hipSetDevice(b->device));
  PCD_TRY(b->o_jq.reserve(b->O));
  PCD_TRY(b->o_jt.reserve(b->O));
  PCD_TRY(pcd_ba_observation_errors_device(b, sq_err ? b->o_jq.p : nullptr, depth ? b->o_jt.p : nullptr, nullptr));
  if (sq_err) PCD_HIP_TRY(hipMemcpy(sq_err, b->o_jq.p, b->O * sizeof(double), hipMemcpyDeviceToHost));
  if (depth) PCD_HIP_TRY(hipMemcpy(depth, b->o_jt.p, b->O * sizeof(double), hipMemcpyDeviceToHost));
  return PCD_OK;
}

pcd_status pcd_ba_filter_tracks_device(pcd_ba* b, double max_reproj_error, const pcd_ba_filter_out* o, void* stream) {
  PCD_REQUIRE(b && o, "null pointer");
  PCD_REFUSE_CAPTURE(stream);
  PCD_HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  const uint64_t O = b->O;
  const int nbp = (int)div_up((uint64_t)b->P, 256), nbo = (int)div_up(std::max<uint64_t>(O, 1), 256);
  PCD_TRY(b->f_sq.reserve(std::max<uint64_t>(O, 1))); PCD_TRY(b->f_depth.reserve(std::max<uint64_t>(O, 1)));
  PCD_TRY(b->f_part.reserve(3 * (size_t)nbp + nbo + 4));
  if (O) PCD_TRY(pcd_ba_observation_errors_device(b, b->f_sq.p, b->f_depth.p, s));
  ScopedKernelTimer t("ba_filter_tracks", s);
  double* part3 = b->f_part.p, *part1 = b->f_part.p + 3 * (size_t)nbp;
  hipLaunchKernelGGL(k_ba_filter_tracks, dim3(nbp), dim3(256), 0, s, b->P, b->pt_obs_start.p, b->pt_obs_list.p, b->f_sq.p,
                     max_reproj_error * max_reproj_error, o->obs_erase, o->point_delete, o->point_error, part3);
  hipLaunchKernelGGL(k_ba_negative_depth, dim3(nbo), dim3(256), 0, s, O, b->f_depth.p, o->obs_negative_depth, part1);
  if (o->summary) hipLaunchKernelGGL(k_ba_filter_summary, dim3(1), dim3(256), 0, s, part3, nbp, part1, nbo, o->summary);
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

pcd_status pcd_ba_filter_tracks(pcd_ba* b, double max_reproj_error, const pcd_ba_filter_out* o) {
  PCD_REQUIRE(b && o, "null pointer");
  PCD_HIP_TRY(hipSetDevice(b->device));
  const size_t O = b->O, P = (size_t)b->P;
  PCD_TRY(b->f_u8.reserve(2 * O + P + 1)); PCD_TRY(b->f_summary.reserve(P + 4));
  pcd_ba_filter_out d{};
  d.obs_erase = o->obs_erase ? b->f_u8.p : nullptr;
  d.obs_negative_depth = o->obs_negative_depth ? b->f_u8.p + O : nullptr;
  d.point_delete = o->point_delete ? b->f_u8.p + 2 * O : nullptr;
  d.point_error = o->point_error ? b->f_summary.p + 4 : nullptr;
  d.summary = o->summary ? b->f_summary.p : nullptr;
  PCD_TRY(pcd_ba_filter_tracks_device(b, max_reproj_error, &d, nullptr));
  if (o->obs_erase && O) PCD_HIP_TRY(hipMemcpy(o->obs_erase, d.obs_erase, O, hipMemcpyDeviceToHost));
  if (o->obs_negative_depth && O) PCD_HIP_TRY(hipMemcpy(o->obs_negative_depth, d.obs_negative_depth, O, hipMemcpyDeviceToHost));
  if (o->point_delete) PCD_HIP_TRY(hipMemcpy(o->point_delete, d.point_delete, P, hipMemcpyDeviceToHost));
  if (o->point_error) PCD_HIP_TRY(hipMemcpy(o->point_error, d.point_error, P * sizeof(double), hipMemcpyDeviceToHost));
  if (o->summary) PCD_HIP_TRY(hipMemcpy(o->summary, d.summary, 4 * sizeof(double), hipMemcpyDeviceToHost));
  PCD_HIP_TRY(hipDeviceSynchronize());
  return PCD_OK;
}

pcd_status pcd_ba_evaluate_blocks(pcd_ba* b, int want_jacobians, int want_jac_cam, pcd_ba_blocks* out) {
  PCD_REQUIRE(b && out, "null pointer");
  PCD_HIP_TRY(hipSetDevice(b->device));
  std::memset(out, 0, sizeof *out);
  const uint64_t O = b->O, L = b->L, V = b->n_pose_rows;
  const bool packed = V != O;
  const size_t n_res = 2 * O + L, n_jq = want_jacobians ? 8 * V : 0, n_jt = want_jacobians ? 6 * V : 0,
               n_jx = want_jacobians ? 6 * O : 0, n_jl = want_jacobians ? 3 * L : 0,
               n_jc = (want_jacobians && want_jac_cam) ? 2 * (size_t)PCD_CAM_JAC_STRIDE * O : 0;
  PCD_TRY(b->h_blocks.reserve(n_res + n_jq + n_jt + n_jx + n_jl + n_jc + 2));
  PCD_TRY(b->o_res.reserve(std::max<size_t>(n_res, 1)));
  pcd_ba_out d{};
  d.residuals = b->o_res.p;
  if (want_jacobians) {
    PCD_TRY(b->o_jq.reserve(std::max<size_t>(8 * O, 1))); PCD_TRY(b->o_jt.reserve(std::max<size_t>(6 * O, 1)));
    PCD_TRY(b->o_jx.reserve(std::max<size_t>(6 * O, 1))); PCD_TRY(b->o_jl.reserve(std::max<size_t>(3 * L, 1)));
    d.jac_q = b->o_jq.p; d.jac_t = b->o_jt.p; d.jac_X = b->o_jx.p; d.jac_lidar = b->o_jl.p;
    if (n_jc) { PCD_TRY(b->o_jc.reserve(n_jc)); d.jac_cam = b->o_jc.p; }
    if (packed) { PCD_TRY(b->p_jq.reserve(std::max<size_t>(8 * V, 1))); PCD_TRY(b->p_jt.reserve(std::max<size_t>(6 * V, 1))); }
  }
  hipStream_t s = nullptr;
  PCD_TRY(pcd_ba_evaluate_device(b, &d, s));
  const double *src_jq = b->o_jq.p, *src_jt = b->o_jt.p;
  if (want_jacobians && packed && V) {
    hipLaunchKernelGGL(k_pack_rows<8>, dim3(div_up(V * 4, 256)), dim3(256), 0, s, b->o_jq.p, b->vobs.p, V, b->p_jq.p);
    hipLaunchKernelGGL(k_pack_rows<6>, dim3(div_up(V * 3, 256)), dim3(256), 0, s, b->o_jt.p, b->vobs.p, V, b->p_jt.p);
    src_jq = b->p_jq.p; src_jt = b->p_jt.p;
  }
  double* h = b->h_blocks.p;
  auto down = [&](const double*& slot, const double* src, size_t n) -> hipError_t {
    slot = n ? h : nullptr;
    const hipError_t e = n ? hipMemcpyAsync(h, src, n * sizeof(double), hipMemcpyDeviceToHost, s) : hipSuccess;
    h += n;
    out->bytes_d2h += n * sizeof(double);
    return e;
  };
  PCD_HIP_TRY(down(out->residuals, b->o_res.p, n_res));
  PCD_HIP_TRY(down(out->jac_q, src_jq, n_jq));
  PCD_HIP_TRY(down(out->jac_t, src_jt, n_jt));
  PCD_HIP_TRY(down(out->jac_X, b->o_jx.p, n_jx));
  PCD_HIP_TRY(down(out->jac_lidar, b->o_jl.p, n_jl));
  PCD_HIP_TRY(down(out->jac_cam, b->o_jc.p, n_jc));
  PCD_HIP_TRY(hipStreamSynchronize(s));
  out->pose_row = b->h_pose_row.data();
  out->num_pose_rows = V;
  return PCD_OK;
}

pcd_status pcd_ba_evaluate(pcd_ba* b, const pcd_ba_out* o) {
  PCD_REQUIRE(b && o, "null pointer");
  PCD_HIP_TRY(hipSetDevice(b->device));
  pcd_ba_out d{};
  struct Item { double* host; DevBuf<double>* buf; size_t n; double** slot; };
  Item items[] = {
      {o->cost, &b->cost, 1, &d.cost},
      {o->residuals, &b->o_res, 2 * b->O + b->L, &d.residuals},
      {o->jac_q, &b->o_jq, 8 * b->O, &d.jac_q},
      {o->jac_t, &b->o_jt, 6 * b->O, &d.jac_t},
      {o->jac_X, &b->o_jx, 6 * b->O, &d.jac_X},
      {o->jac_lidar, &b->o_jl, 3 * b->L, &d.jac_lidar},
      {o->H_img, &b->o_himg, 36 * (size_t)b->I, &d.H_img},
      {o->g_img, &b->o_gimg, 6 * (size_t)b->I, &d.g_img},
      {o->H_pt, &b->o_hpt, 9 * (size_t)b->P, &d.H_pt},
      {o->g_pt, &b->o_gpt, 3 * (size_t)b->P, &d.g_pt},
      {o->W, &b->o_w, 18 * b->O, &d.W},
      {o->jac_cam, &b->o_jc, 2 * PCD_CAM_JAC_STRIDE * b->O, &d.jac_cam},
      {o->H_cam, &b->o_hcam, (size_t)PCD_CAM_JAC_STRIDE * PCD_CAM_JAC_STRIDE * b->C, &d.H_cam},
      {o->g_cam, &b->o_gcam, (size_t)PCD_CAM_JAC_STRIDE * b->C, &d.g_cam},
      {o->E_cam, &b->o_ecam, (size_t)PCD_CAM_JAC_STRIDE * 6 * b->I, &d.E_cam},
      {o->W_cam, &b->o_wcam, (size_t)PCD_CAM_JAC_STRIDE * 3 * b->O, &d.W_cam},
  };
  for (auto& it : items)
    if (it.host) {
      PCD_TRY(it.buf->reserve(std::max<size_t>(it.n, 1)));
      *it.slot = it.buf->p;
    }
  PCD_TRY(pcd_ba_evaluate_device(b, &d, nullptr));
  for (auto& it : items)
    if (it.host && it.n) PCD_HIP_TRY(hipMemcpy(it.host, it.buf->p, it.n * sizeof(double), hipMemcpyDeviceToHost));
  PCD_HIP_TRY(hipDeviceSynchronize());
  return PCD_OK;
}

}  // extern "C"
