// shards.hip -- one LiDAR cloud over several devices of ONE host process (pcd_cloud_create(...,
// const int* devices, int ndev, ...); section 8e: cloud shards + a MIN over the per-query keys).
//
// The reference's call sites are single-process C++ (controllers/bundle_adjustment.cc:130-185,
// sfm/incremental_mapper.cc:1413-1469): this is the form in which THEY reach the sharded search -- the process that
// owns the Reconstruction drives ndev devices; the one-process-per-GPU form (pcdhip/dist.py) is for launchers.
//
//   pcd_cloud_create_sharded   ply.cc:38-54 transform / NaN filter on the host (indices = post-filter file order, as
//                              everywhere), rows put in a spatially compact order (coarse 1 m cells, z-major) and
//                              cut into ndev equal-count ranges: a shard is a slab of space with its own grid.  Every
//                              shard row keeps its ORIGINAL index (cloud.h row_index), so the packed keys of all
//                              shards live in one index space and the element-wise MIN over shards is the
//                              single-cloud result bit for bit, ties (lowest original index) included.
//   pcd_nn_query_sharded       two phases (pcdhip/dist.py two_phase_search): every query is searched in its HOME shard
//                              (nearest bounding box), MIN over shards, then pcd_nn_refine_device on every shard for
//                              the foreign queries whose distance does not rule its bounding box out, MIN again.
//   pcd_associate_sharded      + the winners' (xyz, normal) from their owners (bit patterns, SUM over shards) and the
//                              association epilogue on the first shard's device.
// The reductions are the caller's (pcd_shard_reduce: e.g. ncclAllReduce inside ncclGroupStart / End over the ndev
// buffers) or, with NULL callbacks, the library's own: peer copies to the first shard's device, one kernel, copies back.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "cloud.h"

using namespace pcd;

struct pcd_cloud_shards {
  std::vector<int> devices;
  std::vector<pcd_cloud*> shard;
  uint64_t n_total = 0;
  struct Dev {
    DevBuf<double> q, mr;
    DevBuf<uint64_t> keys;
    DevBuf<uint8_t> skip_foreign, skip_home;
    DevBuf<int32_t> payload;
    DevBuf<float> boxes;
    // device 0 only: gathers of the other shards' buffers, association outputs
    DevBuf<uint64_t> gather_keys;
    DevBuf<int32_t> gather_payload;
    DevBuf<double> o_xyz, o_abcd, o_dist, o_angle, o_d2p;
    DevBuf<uint8_t> o_type;
    DevBuf<uint32_t> o_idx;
    DevBuf<float> o_sq;
  };
  std::vector<Dev*> dev;
  std::vector<float> boxes;   // [nsh][6] lo, hi
};

namespace {

// home shard of every query = nearest bounding box (ties: lowest shard); the two skip masks of shard `me`
__global__ void k_home_masks(const double* __restrict__ q, uint64_t Q, const float* __restrict__ boxes, int nsh, int me,
                             uint8_t* __restrict__ skip_foreign, uint8_t* __restrict__ skip_home) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const double x = q[3 * i], y = q[3 * i + 1], z = q[3 * i + 2];
  int best = 0;
  double bd = INFINITY;
  for (int s = 0; s < nsh; ++s) {
    const float* b = boxes + 6 * s;
    const double dx = x - fmin(fmax(x, (double)b[0]), (double)b[3]), dy = y - fmin(fmax(y, (double)b[1]), (double)b[4]),
                 dz = z - fmin(fmax(z, (double)b[2]), (double)b[5]);
    double d = dx * dx + dy * dy + dz * dz;
    if (!(d == d)) d = INFINITY;   // non-finite query: shard 0 (it is skipped by the search anyway)
    if (d < bd) { bd = d; best = s; }
  }
  skip_foreign[i] = best != me;   // phase 1 searches the home queries only
  skip_home[i] = best == me;      // phase 2 the foreign ones
}

__global__ void k_fill_keys(uint64_t* __restrict__ keys, uint64_t Q, uint64_t v) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < Q) keys[i] = v;
}
__global__ void k_min_keys(uint64_t* __restrict__ acc, const uint64_t* __restrict__ other, uint64_t Q) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < Q) { const uint64_t a = acc[i], b = other[i]; acc[i] = b < a ? b : a; }
}
__global__ void k_sum_i32(int32_t* __restrict__ acc, const int32_t* __restrict__ other, uint64_t n) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < n) acc[i] += other[i];
}

// the library's own reductions: everything to the first shard's device, reduce, copy back
template <typename T, typename K>
pcd_status builtin_reduce(pcd_cloud_shards* sh, T* const* buf, uint64_t count, DevBuf<T>& gather, K kernel) {
  const int nsh = (int)sh->shard.size();
  if (nsh == 1) return PCD_OK;
  const int d0 = sh->devices[0];
  for (int s = 0; s < nsh; ++s) { PCD_HIP_TRY(hipSetDevice(sh->devices[s])); PCD_HIP_TRY(hipDeviceSynchronize()); }
  PCD_HIP_TRY(hipSetDevice(d0));
  PCD_TRY(gather.reserve(std::max<uint64_t>(count, 1)));
  for (int s = 1; s < nsh; ++s) {
    PCD_HIP_TRY(hipMemcpyPeer(gather.p, d0, buf[s], sh->devices[s], count * sizeof(T)));
    hipLaunchKernelGGL(kernel, dim3(div_up(count, 256)), dim3(256), 0, nullptr, buf[0], gather.p, count);
    PCD_HIP_TRY(hipDeviceSynchronize());
  }
  for (int s = 1; s < nsh; ++s) PCD_HIP_TRY(hipMemcpyPeer(buf[s], sh->devices[s], buf[0], d0, count * sizeof(T)));
  return PCD_OK;
}

pcd_status reduce_min(pcd_cloud_shards* sh, const pcd_shard_reduce* red, uint64_t Q) {
  std::vector<uint64_t*> buf;
  for (auto* d : sh->dev) buf.push_back(d->keys.p);
  if (red && red->min_u64) {
    for (size_t s = 0; s < sh->shard.size(); ++s) { PCD_HIP_TRY(hipSetDevice(sh->devices[s])); PCD_HIP_TRY(hipDeviceSynchronize()); }
    if (red->min_u64(red->user, buf.data(), sh->devices.data(), (int)buf.size(), Q) != 0) {
      set_error("pcd_shard_reduce.min_u64 failed");
      return PCD_ERR_INVALID;
    }
    return PCD_OK;
  }
  return builtin_reduce<uint64_t>(sh, buf.data(), Q, sh->dev[0]->gather_keys, k_min_keys);
}
pcd_status reduce_sum(pcd_cloud_shards* sh, const pcd_shard_reduce* red, uint64_t count) {
  std::vector<int32_t*> buf;
  for (auto* d : sh->dev) buf.push_back(d->payload.p);
  if (red && red->sum_i32) {
    for (size_t s = 0; s < sh->shard.size(); ++s) { PCD_HIP_TRY(hipSetDevice(sh->devices[s])); PCD_HIP_TRY(hipDeviceSynchronize()); }
    if (red->sum_i32(red->user, buf.data(), sh->devices.data(), (int)buf.size(), count) != 0) {
      set_error("pcd_shard_reduce.sum_i32 failed");
      return PCD_ERR_INVALID;
    }
    return PCD_OK;
  }
  return builtin_reduce<int32_t>(sh, buf.data(), count, sh->dev[0]->gather_payload, k_sum_i32);
}

// the two-phase search: leaves the final keys in every shard's Dev::keys
pcd_status search(pcd_cloud_shards* sh, const double* q_xyz, uint64_t Q, const pcd_shard_reduce* red) {
  const int nsh = (int)sh->shard.size();
  for (int s = 0; s < nsh; ++s) {
    PCD_HIP_TRY(hipSetDevice(sh->devices[s]));
    auto* d = sh->dev[s];
    PCD_TRY(d->q.reserve(3 * Q)); PCD_TRY(d->keys.reserve(Q));
    PCD_TRY(d->skip_foreign.reserve(Q)); PCD_TRY(d->skip_home.reserve(Q));
    PCD_HIP_TRY(hipMemcpyAsync(d->q.p, q_xyz, 3 * Q * sizeof(double), hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_home_masks, dim3(div_up(Q, 256)), dim3(256), 0, nullptr, d->q.p, Q, d->boxes.p, nsh, s,
                       d->skip_foreign.p, d->skip_home.p);
    hipLaunchKernelGGL(k_fill_keys, dim3(div_up(Q, 256)), dim3(256), 0, nullptr, d->keys.p, Q, (uint64_t)PCD_KEY_NONE);
    // phase 1: home queries only (incoming key NONE: nothing rules the shard out)
    PCD_TRY(pcd_nn_refine_device(sh->shard[s], d->q.p, Q, d->skip_foreign.p, d->keys.p, nullptr));
  }
  PCD_TRY(reduce_min(sh, red, Q));
  if (nsh > 1) {
    for (int s = 0; s < nsh; ++s) {
      PCD_HIP_TRY(hipSetDevice(sh->devices[s]));
      PCD_TRY(pcd_nn_refine_device(sh->shard[s], sh->dev[s]->q.p, Q, sh->dev[s]->skip_home.p, sh->dev[s]->keys.p, nullptr));
    }
    PCD_TRY(reduce_min(sh, red, Q));
  }
  return PCD_OK;
}

}  // namespace

extern "C" {

pcd_status pcd_cloud_create_sharded(const float* xyz, const float* nrm, uint64_t n, const pcd_cloud_options* opts,
                                    const int* devices, int ndev, pcd_cloud_shards** out) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(out, "out is null");
  *out = nullptr;
  PCD_REQUIRE(devices && ndev >= 1 && ndev <= 64, "devices / ndev");
  pcd_cloud_options o;
  if (opts) o = *opts; else pcd_cloud_options_default(&o);
  PCD_REQUIRE(o.layout == PCD_LAYOUT_XYZ_NRM || o.layout == PCD_LAYOUT_AOS32, "unknown layout");
  PCD_REQUIRE(o.index_base == 0 && (o.index_stride == 0 || o.index_stride == 1), "a sharded cloud is given whole");
  PCD_REQUIRE(n == 0 || xyz, "xyz is null");
  PCD_REQUIRE(n == 0 || o.layout == PCD_LAYOUT_AOS32 || nrm, "nrm is null");
  PCD_REQUIRE(n < 0xFFFFFFF0ull, "more than 2^32 rows");
  for (int s = 0; s < ndev; ++s) PCD_TRY(require_device(devices[s]));
  // ---- ply.cc:38-54 on the host: (x,y,z) -> (-y,-z,x) for position and normal, rows with a NaN dropped, order kept ----
  const size_t row = o.layout == PCD_LAYOUT_AOS32 ? 8 : 3;
  std::vector<float> P, N;
  P.reserve(3 * n); N.reserve(3 * n);
  for (uint64_t i = 0; i < n; ++i) {
    const float* p = xyz + row * i;
    const float* v = o.layout == PCD_LAYOUT_AOS32 ? xyz + row * i + 4 : nrm + 3 * i;
    float a[3] = {p[0], p[1], p[2]}, b[3] = {v[0], v[1], v[2]};
    if (o.raw_lidar_frame) {
      const float t[3] = {-a[1], -a[2], a[0]}, u[3] = {-b[1], -b[2], b[0]};
      bool nan = false;
      for (int k = 0; k < 3; ++k) nan = nan || std::isnan(t[k]) || std::isnan(u[k]);
      if (nan) continue;
      for (int k = 0; k < 3; ++k) { a[k] = t[k]; b[k] = u[k]; }
    }
    P.insert(P.end(), a, a + 3); N.insert(N.end(), b, b + 3);
  }
  const uint64_t m = P.size() / 3;
  // ---- spatially compact order: coarse cells of 1 m, z-major (stable: original order inside a cell) ----
  double lo[3] = {0, 0, 0};
  bool any = false;
  for (uint64_t i = 0; i < m; ++i) {
    const float* p = &P[3 * i];
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
    for (int k = 0; k < 3; ++k) lo[k] = any ? std::min(lo[k], (double)p[k]) : (double)p[k];
    any = true;
  }
  std::vector<uint64_t> key(m);
  for (uint64_t i = 0; i < m; ++i) {
    const float* p = &P[3 * i];
    uint64_t c[3] = {0, 0, 0};
    if (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))
      for (int k = 0; k < 3; ++k) c[k] = (uint64_t)std::min(std::floor((double)p[k] - lo[k]), 2097151.0);
    key[i] = (c[2] << 42) | (c[1] << 21) | c[0];
  }
  std::vector<uint32_t> order(m);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
  key.clear(); key.shrink_to_fit();

  pcd_cloud_shards* sh = new pcd_cloud_shards();
  sh->n_total = m;
  auto fail = [&](pcd_status st) { pcd_cloud_shards_destroy(sh); return st; };
  std::vector<float> sx, sn;
  for (int s = 0; s < ndev; ++s) {
    const uint64_t r0 = m * (uint64_t)s / ndev, r1 = m * (uint64_t)(s + 1) / ndev, cnt = r1 - r0;
    sx.resize(3 * cnt); sn.resize(3 * cnt);
    for (uint64_t i = 0; i < cnt; ++i)
      for (int k = 0; k < 3; ++k) { sx[3 * i + k] = P[3 * (size_t)order[r0 + i] + k]; sn[3 * i + k] = N[3 * (size_t)order[r0 + i] + k]; }
    pcd_cloud_options so = o;
    so.layout = PCD_LAYOUT_XYZ_NRM; so.raw_lidar_frame = 0; so.device = devices[s]; so.index_base = 0; so.index_stride = 1;
    pcd_cloud* c = nullptr;
    const pcd_status st = cloud_create_indexed(sx.data(), sn.data(), cnt, &so, order.data() + r0, m, &c);
    if (st != PCD_OK) return fail(st);
    sh->devices.push_back(devices[s]);
    sh->shard.push_back(c);
    sh->dev.push_back(new pcd_cloud_shards::Dev());
    pcd_cloud_info info;
    (void)pcd_cloud_get_info(c, &info);
    // an empty shard (or one without a finite row) must never be anybody's home and never be refined: inverted box
    const bool has = info.num_indexed > 0;
    for (int k = 0; k < 3; ++k) sh->boxes.push_back(has ? info.bbox_lo[k] : INFINITY);
    for (int k = 0; k < 3; ++k) sh->boxes.push_back(has ? info.bbox_hi[k] : -INFINITY);
  }
  for (int s = 0; s < ndev; ++s) {
    if (hipSetDevice(devices[s]) != hipSuccess) return fail(PCD_ERR_HIP);
    const pcd_status st = sh->dev[s]->boxes.reserve(sh->boxes.size());
    if (st != PCD_OK) return fail(st);
    if (hipMemcpy(sh->dev[s]->boxes.p, sh->boxes.data(), sh->boxes.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return fail(PCD_ERR_HIP);
  }
  *out = sh;
  return PCD_OK;
  });
}

void pcd_cloud_shards_destroy(pcd_cloud_shards* sh) {
  if (!sh) return;
  for (size_t s = 0; s < sh->shard.size(); ++s) {
    (void)hipSetDevice(sh->devices[s]);
    delete sh->dev[s];
    pcd_cloud_destroy(sh->shard[s]);
  }
  delete sh;
}

int pcd_cloud_shards_count(const pcd_cloud_shards* sh) { return sh ? (int)sh->shard.size() : 0; }
uint64_t pcd_cloud_shards_size(const pcd_cloud_shards* sh) { return sh ? sh->n_total : 0; }
pcd_cloud* pcd_cloud_shards_get(pcd_cloud_shards* sh, int s) {
  return (sh && s >= 0 && s < (int)sh->shard.size()) ? sh->shard[s] : nullptr;
}

pcd_status pcd_nn_query_sharded(pcd_cloud_shards* sh, const double* q_xyz, uint64_t Q, const pcd_shard_reduce* red,
                                uint32_t* idx, float* sqdist, uint8_t* found) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(sh && !sh->shard.empty(), "null shards");
  PCD_REQUIRE(Q == 0 || (q_xyz && idx && sqdist && found), "null pointer");
  if (Q == 0) return PCD_OK;
  PCD_TRY(search(sh, q_xyz, Q, red));
  PCD_HIP_TRY(hipSetDevice(sh->devices[0]));
  std::vector<uint64_t> k(Q);
  PCD_HIP_TRY(hipMemcpy(k.data(), sh->dev[0]->keys.p, Q * sizeof(uint64_t), hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < Q; ++i) {
    const bool f = k[i] != PCD_KEY_NONE;
    const uint32_t hb = (uint32_t)(k[i] >> 32);
    float d;
    std::memcpy(&d, &hb, 4);
    idx[i] = f ? (uint32_t)k[i] : 0xFFFFFFFFu;
    sqdist[i] = f ? d : 3.402823466e+38f;
    found[i] = f ? 1 : 0;
  }
  return PCD_OK;
  });
}

pcd_status pcd_associate_sharded(pcd_cloud_shards* sh, const double* q_xyz, uint64_t Q, const double* max_range,
                                 uint64_t max_range_count, int gate_mode, const pcd_shard_reduce* red,
                                 const pcd_assoc_out* out) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(sh && !sh->shard.empty() && out, "null pointer");
  gate_mode &= ~PCD_GATE_BOUNDED_SEARCH;   // the sharded search is the exact unbounded one
  PCD_REQUIRE(gate_mode >= 0 && gate_mode <= 2, "gate_mode");
  PCD_REQUIRE(gate_mode == PCD_GATE_CONTROLLER || (max_range && (max_range_count == 1 || max_range_count == Q)),
              "max_range must have 1 or Q entries");
  if (Q == 0) return PCD_OK;
  PCD_REQUIRE(q_xyz, "null queries");
  PCD_TRY(search(sh, q_xyz, Q, red));
  // the winners' (xyz, normal) from their owners
  for (size_t s = 0; s < sh->shard.size(); ++s) {
    PCD_HIP_TRY(hipSetDevice(sh->devices[s]));
    PCD_TRY(sh->dev[s]->payload.reserve(6 * Q));
    PCD_TRY(pcd_nn_winner_payload_device(sh->shard[s], sh->dev[s]->keys.p, Q, sh->dev[s]->payload.p, nullptr));
  }
  PCD_TRY(reduce_sum(sh, red, 6 * Q));
  // epilogue on the first shard's device
  PCD_HIP_TRY(hipSetDevice(sh->devices[0]));
  auto* d = sh->dev[0];
  PCD_TRY(d->mr.reserve(std::max<uint64_t>(max_range_count, 1)));
  PCD_TRY(d->o_xyz.reserve(3 * Q)); PCD_TRY(d->o_abcd.reserve(4 * Q)); PCD_TRY(d->o_dist.reserve(Q));
  PCD_TRY(d->o_angle.reserve(Q)); PCD_TRY(d->o_d2p.reserve(Q)); PCD_TRY(d->o_type.reserve(Q));
  if (gate_mode != PCD_GATE_CONTROLLER)
    PCD_HIP_TRY(hipMemcpy(d->mr.p, max_range, max_range_count * sizeof(double), hipMemcpyHostToDevice));
  pcd_assoc_out dv{d->o_xyz.p, d->o_abcd.p, d->o_type.p, d->o_dist.p, d->o_angle.p, d->o_d2p.p, nullptr, nullptr};
  PCD_TRY(pcd_associate_from_payload_device(sh->devices[0], d->q.p, Q, d->mr.p, max_range_count, gate_mode, d->keys.p,
                                            d->payload.p, &dv, nullptr));
  auto back = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    return dst ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) : hipSuccess;
  };
  PCD_HIP_TRY(back(out->lidar_xyz, d->o_xyz.p, 3 * Q * sizeof(double)));
  PCD_HIP_TRY(back(out->abcd, d->o_abcd.p, 4 * Q * sizeof(double)));
  PCD_HIP_TRY(back(out->type, d->o_type.p, Q));
  PCD_HIP_TRY(back(out->dist, d->o_dist.p, Q * sizeof(double)));
  PCD_HIP_TRY(back(out->angle, d->o_angle.p, Q * sizeof(double)));
  PCD_HIP_TRY(back(out->dist2plane, d->o_d2p.p, Q * sizeof(double)));
  if (out->nn_idx || out->nn_sqdist) {
    std::vector<uint64_t> k(Q);
    PCD_HIP_TRY(hipMemcpy(k.data(), d->keys.p, Q * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < Q; ++i) {
      const bool f = k[i] != PCD_KEY_NONE;
      const uint32_t hb = (uint32_t)(k[i] >> 32);
      float dd;
      std::memcpy(&dd, &hb, 4);
      if (out->nn_idx) out->nn_idx[i] = f ? (uint32_t)k[i] : 0xFFFFFFFFu;
      if (out->nn_sqdist) out->nn_sqdist[i] = f ? dd : 3.402823466e+38f;
    }
  }
  return PCD_OK;
  });
}

}  // extern "C"
