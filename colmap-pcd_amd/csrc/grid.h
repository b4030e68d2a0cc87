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
__device__ __forceinline__ uint32_t cell_of(const GridParams& g, float x, float y, float z) {
  int cx = cell_coord(x, g.origin[0], g.inv_h, g.dims[0]);
  int cy = cell_coord(y, g.origin[1], g.inv_h, g.dims[1]);
  int cz = cell_coord(z, g.origin[2], g.inv_h, g.dims[2]);
  return (uint32_t)(((uint64_t)cz * g.dims[1] + cy) * g.dims[0] + cx);
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

// wave64 min of a u64 key
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t k) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t lo = __shfl_xor((uint32_t)k, off);
    uint32_t hi = __shfl_xor((uint32_t)(k >> 32), off);
    uint64_t o = ((uint64_t)hi << 32) | lo;
    k = o < k ? o : k;
  }
  return k;
}

}  // namespace pcd
