// grid.h -- cell arithmetic shared by the build and the search kernels.
// Every kernel bins with this one expression; the library is compiled with
// -ffp-contract=off, so cloud rows and queries are binned identically.
#pragma once
#include "cloud.h"

namespace pcd {

// unclamped cell coordinate, limited to [-1, n] so the int conversion is defined
__device__ __forceinline__ int cell_coord_raw(float p, float o, float inv_h, int n) {
  float t = floorf((p - o) * inv_h);
  t = fminf(fmaxf(t, -1.0f), (float)n);
  return (int)t;
}
__device__ __forceinline__ int cell_coord(float p, float o, float inv_h, int n) {
  int c = cell_coord_raw(p, o, inv_h, n);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
// Cell order of the sorted cloud ("quad rows"): cells are grouped in 2x2 quads in (y,z); along x the four
// cells of a quad are adjacent:  index = ((zq * nyq + yq) * nx + x) * 4 + (z&1)*2 + (y&1).
// Any x-range of a quad row is ONE contiguous point range, and every 2-aligned box in (y,z) -- the halo
// regions of the bricks (B = 2, R = 2) and the 4^3 pyramid blocks -- is made of whole quad rows: 9 ranges
// of ~110 points per brick region instead of 36 rows of ~27, 4 ranges per block instead of 16.
__device__ __forceinline__ uint64_t quad_row_base(const GridParams& g, int yq, int zq) {
  return ((uint64_t)zq * g.qdims[0] + yq) * (uint64_t)g.dims[0] * 4u;
}
__device__ __forceinline__ uint32_t cell_index(const GridParams& g, int cx, int cy, int cz) {
  return (uint32_t)(quad_row_base(g, cy >> 1, cz >> 1) + (uint64_t)cx * 4u + (uint32_t)(((cz & 1) << 1) | (cy & 1)));
}
__device__ __forceinline__ uint32_t cell_of(const GridParams& g, float x, float y, float z) {
  int cx = cell_coord(x, g.origin[0], g.inv_h, g.dims[0]);
  int cy = cell_coord(y, g.origin[1], g.inv_h, g.dims[1]);
  int cz = cell_coord(z, g.origin[2], g.inv_h, g.dims[2]);
  return cell_index(g, cx, cy, cz);
}

// FLANN L2_Simple<float> over x,y,z: ((dx*dx) + dy*dy) + dz*dz, separate mul / add
// (the .hip files are built with -ffp-contract=off; the ISA is checked for v_fma in tests).
__device__ __forceinline__ float l2_simple3(float qx, float qy, float qz, float px, float py, float pz) {
  float dx = qx - px, dy = qy - py, dz = qz - pz;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

__device__ __forceinline__ uint64_t make_key(float d, uint32_t idx) {
  return ((uint64_t)__float_as_uint(d) << 32) | idx;
}

// ---- wave64 scans / reductions on the VALU (DPP), not through the LDS crossbar ----------------
// gfx9 recipe: row_shr 1,2,4,8 inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3.  Lanes without a source keep the identity (bound_ctrl = false).
#define PCD_DPP_STEP(OP, V, ID, CTRL, ROWMASK) \
  V = OP(V, (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(V), CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ uint32_t op_min_u32(uint32_t a, uint32_t b) { return b < a ? b : a; }
__device__ __forceinline__ uint32_t op_add_u32(uint32_t a, uint32_t b) { return a + b; }

// inclusive prefix minimum; lane 63 holds the minimum of the wave
__device__ __forceinline__ uint32_t wave_scan_min_u32(uint32_t v) {
  PCD_DPP_STEP(op_min_u32, v, 0xFFFFFFFFu, 0x111, 0xf);
  PCD_DPP_STEP(op_min_u32, v, 0xFFFFFFFFu, 0x112, 0xf);
  PCD_DPP_STEP(op_min_u32, v, 0xFFFFFFFFu, 0x114, 0xf);
  PCD_DPP_STEP(op_min_u32, v, 0xFFFFFFFFu, 0x118, 0xf);
  PCD_DPP_STEP(op_min_u32, v, 0xFFFFFFFFu, 0x142, 0xa);
  PCD_DPP_STEP(op_min_u32, v, 0xFFFFFFFFu, 0x143, 0xc);
  return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_min_u32(v), 63);
}
// inclusive prefix sum
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
  PCD_DPP_STEP(op_add_u32, v, 0u, 0x111, 0xf);
  PCD_DPP_STEP(op_add_u32, v, 0u, 0x112, 0xf);
  PCD_DPP_STEP(op_add_u32, v, 0u, 0x114, 0xf);
  PCD_DPP_STEP(op_add_u32, v, 0u, 0x118, 0xf);
  PCD_DPP_STEP(op_add_u32, v, 0u, 0x142, 0xa);
  PCD_DPP_STEP(op_add_u32, v, 0u, 0x143, 0xc);
  return v;
}

// wave64 min of a u64 key (distance bits high, index low): two 32-bit DPP reductions
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t k) {
  const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
  const uint32_t mh = wave_min_u32(hi);
  const uint32_t ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
  return ((uint64_t)mh << 32) | ml;
}

// lane holding the smallest value among the lanes of `mask` (ties: lowest lane); mask must be != 0
__device__ __forceinline__ int wave_argmin_u32(uint32_t v, unsigned long long mask) {
  const bool in = (mask >> (threadIdx.x & 63)) & 1ull;
  const uint32_t m = wave_min_u32(in ? v : 0xFFFFFFFFu);
  const unsigned long long hit = __builtin_amdgcn_ballot_w64(v == m) & mask;   // (one v_cmp + s_and: no select round trip)
  return __ffsll((long long)(hit ? hit : mask)) - 1;
}

}  // namespace pcd
