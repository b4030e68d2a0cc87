// assoc.hip -- fused plane-association epilogue (double precision).
//
// Replaces, per 3D feature point, what the reference does right after the
// KD-tree lookup in its three serial loops:
//   lidar/ply.cc:95-101            winner -> Vector6d (float->double), NaN / ||n|| < 1e-6 reject
//   lidar/lidar_point.cc:5-16,39-50 LidarPoint(l_pt, plane) -> Normalize(): n/|n|, d = 0 - a*x - b*y - c*z
//   lidar/lidar_point.cc:21-37     ComputeDist, ComputePointToPointDist, ComputeAngle
//   optim/bundle_adjustment.cc:381-404   classify on the raw normal, gate on max_search_range   (mode 0)
//   sfm/incremental_mapper.cc:1444-1463  same gate, dist/angle not stored by the reference      (mode 1)
//   controllers/bundle_adjustment.cc:156-176 gate dist2plane > 1 || dist2point > 2              (mode 2)
// Compiled with -ffp-contract=off so the operation order below is what runs.
#include <cstring>  // rocprim's texture_cache_iterator.hpp needs memset declared first

#include <rocprim/rocprim.hpp>

#include "cloud.h"
#include "scratch.h"

namespace pcd {

struct AssocOut {
  double* lidar_xyz; double* abcd; uint8_t* type; double* dist; double* angle; double* dist2plane;
  uint32_t* nn_idx; float* nn_sqdist;
};

__device__ __forceinline__ void assoc_core(const double X[3], const float p[3], const float nv[3], bool found,
                                           double max_range, int mode, uint64_t i, const AssocOut& o) {
  double l[3] = {0, 0, 0}, n[3] = {0, 0, 0}, abcd[4] = {0, 0, 0, 0};
  double dist = 0, ang = 0, d2p = 0;
  int type = PCD_LIDAR_NONE;
  bool ok = found;
  if (ok) {
    for (int k = 0; k < 3; ++k) { l[k] = (double)p[k]; n[k] = (double)nv[k]; }
    // ply.cc:100-101
    bool bad = isnan(l[0]) || isnan(l[1]) || isnan(l[2]) || isnan(n[0]) || isnan(n[1]) || isnan(n[2]);
    const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    ok = !bad && !(nn < 1e-6);
  }
  if (ok) {
    // lidar_point.cc:39-50
    double a = n[0], b = n[1], c = n[2];
    const double norm = sqrt(a * a + b * b + c * c);
    a = a / norm;
    b = b / norm;
    c = c / norm;
    const double d = 0 - a * l[0] - b * l[1] - c * l[2];
    abcd[0] = a; abcd[1] = b; abcd[2] = c; abcd[3] = d;
    const double vx = X[0] - l[0], vy = X[1] - l[1], vz = X[2] - l[2];
    const double p2p = sqrt(vx * vx + vy * vy + vz * vz);              // lidar_point.cc:27-30
    d2p = fabs((X[0] * a + X[1] * b + X[2] * c) + d);                  // lidar_point.cc:21-25
    const double an = fabs((a * vx + b * vy + c * vz) / p2p);          // lidar_point.cc:32-37
    // bundle_adjustment.cc:381: raw normal, IEEE division
    const bool ground = (fabs(n[1] / n[0]) > 10) && (fabs(n[1] / n[2]) > 10);
    type = ground ? PCD_LIDAR_ICP_GROUND : PCD_LIDAR_ICP;
    bool reject;
    if (mode == PCD_GATE_CONTROLLER) reject = (d2p > 1) || (p2p > 2);
    else reject = p2p > max_range;
    if (reject) type = PCD_LIDAR_NONE;
    else { dist = p2p; ang = an; }
  }
  if (o.lidar_xyz) { o.lidar_xyz[3 * i] = l[0]; o.lidar_xyz[3 * i + 1] = l[1]; o.lidar_xyz[3 * i + 2] = l[2]; }
  if (o.abcd) { o.abcd[4 * i] = abcd[0]; o.abcd[4 * i + 1] = abcd[1]; o.abcd[4 * i + 2] = abcd[2]; o.abcd[4 * i + 3] = abcd[3]; }
  if (o.type) o.type[i] = (uint8_t)type;
  if (o.dist) o.dist[i] = dist;
  if (o.angle) o.angle[i] = ang;
  if (o.dist2plane) o.dist2plane[i] = d2p;
}

__global__ void k_associate(const float4* __restrict__ pn8, ShardIndex si,
                            const double* __restrict__ q, uint64_t Q,
                            const uint64_t* __restrict__ keys, const double* __restrict__ max_range,
                            uint64_t mr_count, int mode, AssocOut o) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const uint64_t key = keys[i];
  bool found = key != PCD_KEY_NONE;
  float p[3] = {0, 0, 0}, nv[3] = {0, 0, 0};
  uint32_t gi = (uint32_t)key;
  if (found) {
    // same ownership test as k_winner_payload: a key another shard won (e.g. after a cross-rank MIN)
    // must not be mapped onto one of this shard's rows
    const uint64_t li = shard_local_row(si, gi);
    if (li != ~0ull) {
      const float4 a = pn8[2 * li], b = pn8[2 * li + 1];   // one 32-byte record: point + normal
      p[0] = a.x; p[1] = a.y; p[2] = a.z;
      nv[0] = b.x; nv[1] = b.y; nv[2] = b.z;
    } else {
      found = false;  // key of another shard: no association here, use the payload path instead
    }
  }
  const double X[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]};
  const double mr = max_range ? max_range[mr_count == 1 ? 0 : i] : 0.0;
  if (o.nn_idx) o.nn_idx[i] = key != PCD_KEY_NONE ? gi : 0xFFFFFFFFu;
  if (o.nn_sqdist) o.nn_sqdist[i] = key != PCD_KEY_NONE ? __uint_as_float((uint32_t)(key >> 32)) : 3.402823466e+38f;
  assoc_core(X, p, nv, found, mr, mode, i, o);
}

// winner (xyz, normal) of the keys this shard owns, as int32 bit patterns; zeros elsewhere
__global__ void k_winner_payload(const float4* __restrict__ pn8, ShardIndex si,
                                 const uint64_t* __restrict__ keys, uint64_t Q, int32_t* __restrict__ payload) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const uint64_t key = keys[i];
  int32_t out[6] = {0, 0, 0, 0, 0, 0};
  if (key != PCD_KEY_NONE) {
    const uint64_t li = shard_local_row(si, (uint32_t)key);
    if (li != ~0ull) {
      const float4 a = pn8[2 * li], b = pn8[2 * li + 1];
      out[0] = __float_as_int(a.x); out[1] = __float_as_int(a.y); out[2] = __float_as_int(a.z);
      out[3] = __float_as_int(b.x); out[4] = __float_as_int(b.y); out[5] = __float_as_int(b.z);
    }
  }
  for (int k = 0; k < 6; ++k) payload[6 * i + k] = out[k];
}

__global__ void k_associate_payload(const double* __restrict__ q, uint64_t Q, const uint64_t* __restrict__ keys,
                                    const int32_t* __restrict__ payload, const double* __restrict__ max_range,
                                    uint64_t mr_count, int mode, AssocOut o) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const uint64_t key = keys[i];
  const bool found = key != PCD_KEY_NONE;
  float p[3], nv[3];
  for (int k = 0; k < 3; ++k) {
    p[k] = __int_as_float(payload[6 * i + k]);
    nv[k] = __int_as_float(payload[6 * i + 3 + k]);
  }
  const double X[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]};
  const double mr = max_range ? max_range[mr_count == 1 ? 0 : i] : 0.0;
  if (o.nn_idx) o.nn_idx[i] = found ? (uint32_t)key : 0xFFFFFFFFu;
  if (o.nn_sqdist) o.nn_sqdist[i] = found ? __uint_as_float((uint32_t)(key >> 32)) : 3.402823466e+38f;
  assoc_core(X, p, nv, found, mr, mode, i, o);
}

// base/reconstruction.cc:771-805 FilterLidarOutlier: erase the association when the point-to-point distance
// between the (re-optimised) 3D point and its LiDAR point exceeds the bound of its type
__global__ void k_filter_lidar_outlier(const double* __restrict__ X, const double* __restrict__ lxyz,
                                       const uint8_t* __restrict__ type, uint64_t n, double max_proj,
                                       double max_icp, uint8_t* __restrict__ erase) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t e = 0;
  const uint8_t t = type[i];
  if (t != PCD_LIDAR_NONE) {
    const double vx = lxyz[3 * i] - X[3 * i], vy = lxyz[3 * i + 1] - X[3 * i + 1], vz = lxyz[3 * i + 2] - X[3 * i + 2];
    const double dist = sqrt(vx * vx + vy * vy + vz * vz);
    e = dist > (t == PCD_LIDAR_PROJ ? max_proj : max_icp) ? 1 : 0;
  }
  erase[i] = e;
}

// accepted associations (type != 0) -> 80-byte records at their scanned position, ascending query order
struct HitFlag {
  const uint8_t* type;
  __device__ uint32_t operator()(uint32_t i) const { return type[i] != PCD_LIDAR_NONE ? 1u : 0u; }
};
__global__ void k_pack_hits(const uint8_t* __restrict__ type, const uint32_t* __restrict__ pos, uint32_t Q,
                            const double* __restrict__ xyz, const double* __restrict__ abcd,
                            const double* __restrict__ dist, const double* __restrict__ angle,
                            pcd_assoc_hit* __restrict__ hits, uint32_t* __restrict__ count, uint32_t q0) {
  // q0: index of the chunk's first query in the caller's batch (all pointers are already offset by it)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const uint8_t t = type[i];
  if (t != PCD_LIDAR_NONE) {
    pcd_assoc_hit h;
    for (int k = 0; k < 3; ++k) h.lidar_xyz[k] = xyz[3 * (size_t)i + k];
    for (int k = 0; k < 4; ++k) h.abcd[k] = abcd[4 * (size_t)i + k];
    h.dist = dist[i]; h.angle = angle[i];
    h.query = q0 + i; h.type = t; h.pad[0] = h.pad[1] = h.pad[2] = 0;
    hits[pos[i]] = h;
  }
  if (i == Q - 1) *count = pos[i] + (t != PCD_LIDAR_NONE ? 1u : 0u);
}

pcd_status nn_query_device_internal(pcd_cloud* c, const double* d_q, uint64_t Q, int algo, uint64_t* d_keys,
                                    hipStream_t s);
pcd_status nn_query_bounded_internal(pcd_cloud* c, const double* d_q, uint64_t Q, const double* d_max_range,
                                     uint64_t mr_count, double fixed_range, uint64_t* d_keys, hipStream_t s);

static AssocOut to_dev(const pcd_assoc_out* o) {
  AssocOut a{o->lidar_xyz, o->abcd, o->type, o->dist, o->angle, o->dist2plane, o->nn_idx, o->nn_sqdist};
  return a;
}

}  // namespace pcd

using namespace pcd;

extern "C" {

pcd_status pcd_associate_device(pcd_cloud* c, const double* d_q_xyz, uint64_t Q, const double* d_max_range,
                                uint64_t max_range_count, int gate_mode, const uint64_t* d_keys_in,
                                const pcd_assoc_out* d_out, void* stream) {
  PCD_REQUIRE(c && d_out, "null pointer");
  PCD_REFUSE_CAPTURE(stream);
  const bool bounded = (gate_mode & PCD_GATE_BOUNDED_SEARCH) != 0;
  gate_mode &= ~PCD_GATE_BOUNDED_SEARCH;
  PCD_REQUIRE(gate_mode >= 0 && gate_mode <= 2, "gate_mode");
  PCD_REQUIRE(gate_mode == PCD_GATE_CONTROLLER || (d_max_range && (max_range_count == 1 || max_range_count == Q)),
              "max_range must have 1 or Q entries");
  if (Q == 0) return PCD_OK;
  PCD_REQUIRE(d_q_xyz, "null queries");
  PCD_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  const uint64_t* keys = d_keys_in;
  if (!keys) {
    QueryScratch* sc = scratch_of(c);
    PCD_TRY(sc->a_keys.reserve(Q));
    if (bounded) {   // controllers/bundle_adjustment.cc:160: dist2point > 2 rejects
      PCD_TRY(nn_query_bounded_internal(c, d_q_xyz, Q, gate_mode == PCD_GATE_CONTROLLER ? nullptr : d_max_range,
                                        max_range_count, 2.0, sc->a_keys.p, s));
    } else {
      PCD_TRY(nn_query_device_internal(c, d_q_xyz, Q, PCD_NN_AUTO, sc->a_keys.p, s));
    }
    keys = sc->a_keys.p;
  }
  {
    ScopedKernelTimer t("associate", s);
    hipLaunchKernelGGL(k_associate, dim3(div_up(Q, 256)), dim3(256), 0, s, c->pn8.p, c->shard_index(),
                       d_q_xyz, Q, keys, gate_mode == PCD_GATE_CONTROLLER ? nullptr : d_max_range,
                       max_range_count, gate_mode, to_dev(d_out));
  }
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

pcd_status pcd_nn_winner_payload_device(pcd_cloud* c, const uint64_t* d_keys, uint64_t Q, int32_t* d_payload,
                                        void* stream) {
  PCD_REQUIRE(c && (Q == 0 || (d_keys && d_payload)), "null pointer");
  PCD_REFUSE_CAPTURE(stream);
  if (Q == 0) return PCD_OK;
  PCD_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  ScopedKernelTimer t("winner_payload", s);
  hipLaunchKernelGGL(k_winner_payload, dim3(div_up(Q, 256)), dim3(256), 0, s, c->pn8.p, c->shard_index(),
                     d_keys, Q, d_payload);
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

pcd_status pcd_associate_from_payload_device(int device, const double* d_q_xyz, uint64_t Q, const double* d_max_range,
                                             uint64_t max_range_count, int gate_mode, const uint64_t* d_keys,
                                             const int32_t* d_payload, const pcd_assoc_out* d_out, void* stream) {
  PCD_REQUIRE(d_out && (Q == 0 || (d_q_xyz && d_keys && d_payload)), "null pointer");
  PCD_REFUSE_CAPTURE(stream);
  PCD_REQUIRE(gate_mode >= 0 && gate_mode <= 2, "gate_mode");
  PCD_REQUIRE(gate_mode == PCD_GATE_CONTROLLER || (d_max_range && (max_range_count == 1 || max_range_count == Q)),
              "max_range must have 1 or Q entries");
  if (Q == 0) return PCD_OK;
  PCD_TRY(require_device(device));
  hipStream_t s = (hipStream_t)stream;
  ScopedKernelTimer t("associate_payload", s);
  hipLaunchKernelGGL(k_associate_payload, dim3(div_up(Q, 256)), dim3(256), 0, s, d_q_xyz, Q, d_keys, d_payload,
                     gate_mode == PCD_GATE_CONTROLLER ? nullptr : d_max_range, max_range_count, gate_mode,
                     to_dev(d_out));
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

pcd_status pcd_filter_lidar_outlier_device(int device, const double* d_points_xyz, const double* d_lidar_xyz,
                                           const uint8_t* d_type, uint64_t n, double max_proj_dist_error,
                                           double max_icp_dist_error, uint8_t* d_erase, void* stream) {
  PCD_REQUIRE(n == 0 || (d_points_xyz && d_lidar_xyz && d_type && d_erase), "null pointer");
  PCD_REFUSE_CAPTURE(stream);
  if (n == 0) return PCD_OK;
  PCD_TRY(require_device(device));
  hipStream_t s = (hipStream_t)stream;
  ScopedKernelTimer t("filter_lidar_outlier", s);
  hipLaunchKernelGGL(k_filter_lidar_outlier, dim3(div_up(n, 256)), dim3(256), 0, s, d_points_xyz, d_lidar_xyz, d_type,
                     n, max_proj_dist_error, max_icp_dist_error, d_erase);
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

pcd_status pcd_assoc_staging(pcd_cloud* c, uint64_t Q, double** q_xyz, double** max_range) {
  PCD_REQUIRE(c && q_xyz && max_range, "null pointer");
  PCD_HIP_TRY(hipSetDevice(c->device));
  QueryScratch& a = *scratch_of(c);
  PCD_TRY(a.h_q.reserve(3 * std::max<uint64_t>(Q, 1)));
  PCD_TRY(a.h_mr.reserve(std::max<uint64_t>(Q, 1)));
  *q_xyz = a.h_q.p;
  *max_range = a.h_mr.p;
  return PCD_OK;
}

pcd_status pcd_associate_staged(pcd_cloud* c, uint64_t Q, uint64_t max_range_count, int gate_mode,
                                const pcd_assoc_hit** hits, uint64_t* num_hits) {
  PCD_REQUIRE(c && hits && num_hits, "null pointer");
  gate_mode &= ~PCD_GATE_BOUNDED_SEARCH;   // always searched gate-bounded here; the flag is accepted like everywhere else
  PCD_REQUIRE(gate_mode >= 0 && gate_mode <= 2, "gate_mode");
  PCD_REQUIRE(gate_mode == PCD_GATE_CONTROLLER || max_range_count == 1 || max_range_count == Q,
              "max_range must have 1 or Q entries");
  *hits = nullptr;
  *num_hits = 0;
  if (Q == 0) return PCD_OK;
  PCD_REQUIRE(Q < 0xFFFFFFF0ull, "more than 2^32 queries in one call");
  PCD_HIP_TRY(hipSetDevice(c->device));
  QueryScratch& a = *scratch_of(c);
  PCD_REQUIRE(a.h_q.n >= 3 * Q && a.h_mr.n >= std::max<uint64_t>(max_range_count, 1),
              "call pcd_assoc_staging(Q) and fill the buffers first");
  PCD_TRY(a.d_q.reserve(3 * Q));
  PCD_TRY(a.a_mr.reserve(std::max<uint64_t>(max_range_count, 1)));
  PCD_TRY(a.a_xyz.reserve(3 * Q)); PCD_TRY(a.a_abcd.reserve(4 * Q)); PCD_TRY(a.a_dist.reserve(Q));
  PCD_TRY(a.a_angle.reserve(Q)); PCD_TRY(a.a_type.reserve(Q));
  PCD_TRY(a.hit_pos.reserve(Q)); PCD_TRY(a.d_hits.reserve(Q)); PCD_TRY(a.h_hits.reserve(Q));
  constexpr int kC = QueryScratch::kStageChunks;
  PCD_TRY(a.h_count.reserve(kC)); PCD_TRY(a.hit_count.reserve(kC));
  if (!a.st_in) {
    PCD_HIP_TRY(hipStreamCreateWithFlags(&a.st_in, hipStreamNonBlocking));
    PCD_HIP_TRY(hipStreamCreateWithFlags(&a.st_comp, hipStreamNonBlocking));
    PCD_HIP_TRY(hipStreamCreateWithFlags(&a.st_out, hipStreamNonBlocking));
    PCD_HIP_TRY(hipStreamCreateWithFlags(&a.st_cnt, hipStreamNonBlocking));
    for (int k = 0; k < kC; ++k) {
      PCD_HIP_TRY(hipEventCreateWithFlags(&a.ev_in[k], hipEventDisableTiming));
      PCD_HIP_TRY(hipEventCreateWithFlags(&a.ev_comp[k], hipEventDisableTiming));
      PCD_HIP_TRY(hipEventCreateWithFlags(&a.ev_cnt[k], hipEventDisableTiming));
    }
  }
  // A large batch is cut into chunks that move through three streams: chunk k+1's queries cross PCIe and chunk k-1's
  // accepted records go back while chunk k is searched.  One chunk = the whole path of DESIGN 4.1-4.2 on its own
  // (the per-call fixed costs are paid per chunk, hidden under the copies: the copies are the longer leg,
  // 32 B in + 80 B per accepted association out against ~1 us per thousand queries on the device).
  // Measured, 1 M queries in the caller's (spatially random) order -> 0.9 M records: 1 chunk 2.99 ms, 2 chunks
  // 2.52 ms, 4 chunks 2.70 ms -- a chunk of randomly placed queries shares fewer bricks, so the device time grows
  // with the number of chunks (0.5 ms per quarter against 1.0 ms for the whole batch).  Small batches stay in one
  // piece on one stream (the cross-stream events cost ~30 us per call).
  const int nchunk = Q >= 200000 ? 2 : 1;
  static_assert(kC >= 2, "two chunks");
  hipStream_t const s_in = nchunk > 1 ? a.st_in : a.st_comp, s_cnt = nchunk > 1 ? a.st_cnt : a.st_comp,
                    s_out = nchunk > 1 ? a.st_out : a.st_comp;
  const uint64_t per = (Q + nchunk - 1) / nchunk;
  if (gate_mode != PCD_GATE_CONTROLLER)
    PCD_HIP_TRY(hipMemcpyAsync(a.a_mr.p, a.h_mr.p, max_range_count * sizeof(double), hipMemcpyHostToDevice, s_in));
  uint64_t q0s[kC], qns[kC];
  for (int k = 0; k < nchunk; ++k) {
    const uint64_t q0 = std::min<uint64_t>(Q, k * per), qn = std::min<uint64_t>(Q, q0 + per) - q0;
    q0s[k] = q0; qns[k] = qn;
    if (qn) PCD_HIP_TRY(hipMemcpyAsync(a.d_q.p + 3 * q0, a.h_q.p + 3 * q0, 3 * qn * sizeof(double), hipMemcpyHostToDevice, s_in));
    if (nchunk > 1) PCD_HIP_TRY(hipEventRecord(a.ev_in[k], s_in));
  }
  for (int k = 0; k < nchunk; ++k) {
    const uint64_t q0 = q0s[k], qn = qns[k];
    hipStream_t s = a.st_comp;
    if (nchunk > 1) PCD_HIP_TRY(hipStreamWaitEvent(s, a.ev_in[k], 0));
    uint32_t* d_count = a.hit_count.p + k;
    if (qn) {
      pcd_assoc_out d{a.a_xyz.p + 3 * q0, a.a_abcd.p + 4 * q0, a.a_type.p + q0, a.a_dist.p + q0, a.a_angle.p + q0,
                      nullptr, nullptr, nullptr};
      const bool per_query = max_range_count == Q && Q > 1;
      // only accepted associations leave this entry point: the search is bounded by the gate
      PCD_TRY(pcd_associate_device(c, a.d_q.p + 3 * q0, qn, per_query ? a.a_mr.p + q0 : a.a_mr.p,
                                   per_query ? qn : max_range_count, gate_mode | PCD_GATE_BOUNDED_SEARCH, nullptr, &d, s));
      ScopedKernelTimer t("associate_compact", s);
      const auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u),
                                                          HitFlag{a.a_type.p + q0});
      size_t tb = 0;
      PCD_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, flags, a.hit_pos.p + q0, 0u, (size_t)qn, rocprim::plus<uint32_t>(), s));
      PCD_TRY(a.tmp.reserve(tb));
      PCD_HIP_TRY(rocprim::exclusive_scan(a.tmp.p, tb, flags, a.hit_pos.p + q0, 0u, (size_t)qn, rocprim::plus<uint32_t>(), s));
      // chunk k's records are packed at the start of its own region of d_hits (first record = slot q0)
      hipLaunchKernelGGL(k_pack_hits, dim3(div_up(qn, 256)), dim3(256), 0, s, a.a_type.p + q0, a.hit_pos.p + q0, (uint32_t)qn,
                         a.a_xyz.p + 3 * q0, a.a_abcd.p + 4 * q0, a.a_dist.p + q0, a.a_angle.p + q0, a.d_hits.p + q0,
                         d_count, (uint32_t)q0);
    } else {
      PCD_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
    }
    // the counts travel on their own stream: st_out is in order, and a count copy of a later chunk queued ahead of
    // an earlier chunk's records would hold those back until all chunks are searched
    if (nchunk > 1) {
      PCD_HIP_TRY(hipEventRecord(a.ev_comp[k], s));
      PCD_HIP_TRY(hipStreamWaitEvent(s_cnt, a.ev_comp[k], 0));
    }
    PCD_HIP_TRY(hipMemcpyAsync(a.h_count.p + k, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s_cnt));
    if (nchunk > 1) PCD_HIP_TRY(hipEventRecord(a.ev_cnt[k], s_cnt));
  }
  // the host learns chunk k's count, then asks for exactly its records, appended to the previous chunks' in pinned
  // memory: ascending query order, contiguous
  uint64_t total = 0;
  for (int k = 0; k < nchunk; ++k) {
    if (nchunk > 1) PCD_HIP_TRY(hipEventSynchronize(a.ev_cnt[k]));
    else PCD_HIP_TRY(hipStreamSynchronize(s_cnt));
    const uint32_t m = a.h_count.p[k];
    if (m)
      PCD_HIP_TRY(hipMemcpyAsync(a.h_hits.p + total, a.d_hits.p + q0s[k], (size_t)m * sizeof(pcd_assoc_hit),
                                 hipMemcpyDeviceToHost, s_out));
    total += m;
  }
  PCD_HIP_TRY(hipStreamSynchronize(s_out));
  *hits = a.h_hits.p;
  *num_hits = total;
  return PCD_OK;
}

pcd_status pcd_associate(pcd_cloud* c, const double* q_xyz, uint64_t Q, const double* max_range,
                         uint64_t max_range_count, int gate_mode, const pcd_assoc_out* out) {
  PCD_REQUIRE(c && out, "null pointer");
  const int gate_only = gate_mode & ~PCD_GATE_BOUNDED_SEARCH;
  PCD_REQUIRE(gate_only >= 0 && gate_only <= 2, "gate_mode");
  PCD_REQUIRE(gate_only == PCD_GATE_CONTROLLER || (max_range && (max_range_count == 1 || max_range_count == Q)),
              "max_range must have 1 or Q entries");
  if (Q == 0) return PCD_OK;
  PCD_REQUIRE(q_xyz, "null queries");
  PCD_HIP_TRY(hipSetDevice(c->device));
  QueryScratch& a = *scratch_of(c);
  hipStream_t s = nullptr;
  PCD_TRY(a.d_q.reserve(3 * Q));
  PCD_TRY(a.a_mr.reserve(std::max<uint64_t>(max_range_count, 1)));
  PCD_TRY(a.a_xyz.reserve(3 * Q)); PCD_TRY(a.a_abcd.reserve(4 * Q)); PCD_TRY(a.a_dist.reserve(Q));
  PCD_TRY(a.a_angle.reserve(Q)); PCD_TRY(a.a_d2p.reserve(Q)); PCD_TRY(a.a_type.reserve(Q));
  PCD_TRY(a.d_idx.reserve(Q)); PCD_TRY(a.d_sq.reserve(Q));
  PCD_HIP_TRY(hipMemcpyAsync(a.d_q.p, q_xyz, 3 * Q * sizeof(double), hipMemcpyHostToDevice, s));
  if (gate_only != PCD_GATE_CONTROLLER)
    PCD_HIP_TRY(hipMemcpyAsync(a.a_mr.p, max_range, max_range_count * sizeof(double), hipMemcpyHostToDevice, s));
  pcd_assoc_out d{a.a_xyz.p, a.a_abcd.p, a.a_type.p, a.a_dist.p, a.a_angle.p, a.a_d2p.p, a.d_idx.p, a.d_sq.p};
  PCD_TRY(pcd_associate_device(c, a.d_q.p, Q, a.a_mr.p, max_range_count, gate_mode, nullptr, &d, s));
  auto back = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s) : hipSuccess;
  };
  PCD_HIP_TRY(back(out->lidar_xyz, a.a_xyz.p, 3 * Q * sizeof(double)));
  PCD_HIP_TRY(back(out->abcd, a.a_abcd.p, 4 * Q * sizeof(double)));
  PCD_HIP_TRY(back(out->type, a.a_type.p, Q));
  PCD_HIP_TRY(back(out->dist, a.a_dist.p, Q * sizeof(double)));
  PCD_HIP_TRY(back(out->angle, a.a_angle.p, Q * sizeof(double)));
  PCD_HIP_TRY(back(out->dist2plane, a.a_d2p.p, Q * sizeof(double)));
  PCD_HIP_TRY(back(out->nn_idx, a.d_idx.p, Q * sizeof(uint32_t)));
  PCD_HIP_TRY(back(out->nn_sqdist, a.d_sq.p, Q * sizeof(float)));
  PCD_HIP_TRY(hipStreamSynchronize(s));
  return PCD_OK;
}

}  // extern "C"
