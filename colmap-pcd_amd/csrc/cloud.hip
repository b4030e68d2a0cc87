// cloud.hip -- build of the device-resident cloud index.
// Reference behaviour restated: lidar/ply.cc:33-57 (axis swap, NaN-row filter,
// order preserved); lidar/kdtree.cc:5-8 (index over x,y,z of every kept row).
#include <cstring>  // rocprim's texture_cache_iterator.hpp needs memset declared first

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <cmath>

#include "cloud.h"
#include "grid.h"

namespace pcd {

// ------------------------------------------------------------- kernels ----
// ply.cc:38-54: p' = (-y,-z,x), n' = (-ny,-nz,nx); row dropped if any of the six is NaN.
__global__ void k_flag_rows(const float* __restrict__ xyz, const float* __restrict__ nrm, uint64_t n, int layout,
                            uint32_t* __restrict__ keep) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = layout == PCD_LAYOUT_AOS32 ? xyz + 8 * i : xyz + 3 * i;
  const float* q = layout == PCD_LAYOUT_AOS32 ? xyz + 8 * i + 4 : nrm + 3 * i;
  bool bad = isnan(p[0]) || isnan(p[1]) || isnan(p[2]) || isnan(q[0]) || isnan(q[1]) || isnan(q[2]);
  keep[i] = bad ? 0u : 1u;
}

__global__ void k_compact_rows(const float* __restrict__ xyz, const float* __restrict__ nrm, uint64_t n, int layout,
                               int raw_frame, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
                               float4* __restrict__ pts4, float4* __restrict__ pn8) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t dst = (uint32_t)i;
  if (raw_frame) {
    if (!keep[i]) return;
    dst = pos[i];
  }
  const float* p = layout == PCD_LAYOUT_AOS32 ? xyz + 8 * i : xyz + 3 * i;
  const float* q = layout == PCD_LAYOUT_AOS32 ? xyz + 8 * i + 4 : nrm + 3 * i;
  float4 a, b;
  if (raw_frame) {
    a = make_float4(-p[1], -p[2], p[0], 0.f);
    b = make_float4(-q[1], -q[2], q[0], 0.f);
  } else {
    a = make_float4(p[0], p[1], p[2], 0.f);
    b = make_float4(q[0], q[1], q[2], 0.f);
  }
  pts4[dst] = a;
  pn8[2 * (size_t)dst] = a;
  pn8[2 * (size_t)dst + 1] = b;
}

__device__ inline int f2ord(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ inline float ord2f(int i) {
  int j = i >= 0 ? i : i ^ 0x7fffffff;
#ifdef __HIP_DEVICE_COMPILE__
  return __int_as_float(j);
#else
  float f;
  std::memcpy(&f, &j, 4);
  return f;
#endif
}

// bbox of rows with three finite coordinates; bb[0..2] = min (ordered ints), bb[3..5] = max, bb[6] = count
__global__ void k_bbox(const float4* __restrict__ pts4, uint64_t n, int* __restrict__ bb,
                       unsigned long long* __restrict__ cnt) {
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  unsigned c = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    float4 p = pts4[i];
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
      int o[3] = {f2ord(p.x), f2ord(p.y), f2ord(p.z)};
      for (int d = 0; d < 3; ++d) {
        lo[d] = min(lo[d], o[d]);
        hi[d] = max(hi[d], o[d]);
      }
      ++c;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    for (int d = 0; d < 3; ++d) {
      lo[d] = min(lo[d], __shfl_xor(lo[d], off));
      hi[d] = max(hi[d], __shfl_xor(hi[d], off));
    }
    c += __shfl_xor(c, off);
  }
  if ((threadIdx.x & 63) == 0) {
    for (int d = 0; d < 3; ++d) {
      atomicMin(&bb[d], lo[d]);
      atomicMax(&bb[3 + d], hi[d]);
    }
    atomicAdd(cnt, (unsigned long long)c);
  }
}

// per-cell histogram; non-finite rows are skipped (they keep an index but can never win, see nn.hip)
__global__ void k_cell_hist(const float4* __restrict__ pts4, uint64_t n, GridParams g, uint32_t* __restrict__ count) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts4[i];
  if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return;
  atomicAdd(&count[cell_of(g, p.x, p.y, p.z)], 1u);
}

__global__ void k_count_nonzero(const uint32_t* __restrict__ count, uint64_t ncells, unsigned long long* __restrict__ out) {
  unsigned c = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < ncells; i += (uint64_t)gridDim.x * blockDim.x)
    c += count[i] != 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

// sort keys: cell id for finite rows, 0xFFFFFFFF for the rest (sorted to the tail, never referenced)
__global__ void k_cell_keys(const float4* __restrict__ pts4, uint64_t n, GridParams g, uint32_t* __restrict__ keys,
                            uint32_t* __restrict__ vals) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts4[i];
  bool fin = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
  keys[i] = fin ? cell_of(g, p.x, p.y, p.z) : 0xFFFFFFFFu;
  vals[i] = (uint32_t)i;
}

__global__ void k_gather_sorted(const float4* __restrict__ pts4, const uint32_t* __restrict__ order, uint64_t m,
                                uint32_t index_base, uint32_t index_stride, const uint32_t* __restrict__ row_index,
                                float4* __restrict__ sorted) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= m + kSortedSpare) return;
  // rows m .. m+15 repeat the last record: the brick kernel reads ranges in groups of 4 records and may run up to
  // 3 records past the end of the last range (brick_kernel.h), the stencil kernel in steps of 16 records
  // (stencil_kernel.h); a repeated real point cannot change a minimum
  uint32_t src = order[i < m ? i : m - 1];
  float4 p = pts4[src];
  p.w = __uint_as_float(row_index ? row_index[src] : index_base + src * index_stride);
  sorted[i] = p;
}

// Pyramid level 0 = LEAVES: every sub-block of 2x2x2 cells (half a quad row of two x-cells: ONE contiguous point
// range) with its tight box and its range, {lo.xyz, hi.x | hi.y, hi.z, bits(first point), bits(count)}.  One wavefront
// per leaf.  (Round 2 had 4x4x4-cell blocks as level 0 with 8 sub-block records each: a far query then paid one walk
// step per BLOCK of the shell between its true distance and the blocks' looser bounds; with the sub-blocks as the
// children of a 64-ary node one step tests 64 of them.)
__global__ void k_leaf_aabb(const float4* __restrict__ sorted, const uint32_t* __restrict__ cell_start, GridParams g,
                            int sdx, int sdy, int sdz, float* __restrict__ aabb) {
  const int lane = threadIdx.x & 63;
  const uint64_t id = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
  if (id >= (uint64_t)sdx * sdy * sdz) return;
  const int sx = (int)(id % sdx), sy = (int)((id / sdx) % sdy), sz = (int)(id / ((uint64_t)sdx * sdy));
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  const int cx0 = min(2 * sx, g.dims[0]), cx1 = min(cx0 + 2, g.dims[0]);
  const uint64_t rowbase = quad_row_base(g, sy, sz);
  const uint32_t s = cell_start[rowbase + 4 * cx0], e = cell_start[rowbase + 4 * cx1];
  for (uint32_t i = s + lane; i < e; i += 64) {
    const float4 p = sorted[i];
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
  if (s != e)
    for (int off = 32; off > 0; off >>= 1)
      for (int d = 0; d < 3; ++d) {
        lo[d] = fminf(lo[d], __shfl_xor(lo[d], off));
        hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off));
      }
  if (lane == 0) {
    float* o = aabb + 8 * id;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2];
    o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
    o[6] = __uint_as_float(s); o[7] = __uint_as_float(e - s);
  }
}

// pyramid level k+1 from level k: union of the <= 64 children boxes (empty = {+inf, -inf})
__global__ void k_pyramid_level(float* __restrict__ aabb, uint32_t off_child, int cdx, int cdy, int cdz,
                                uint32_t off_parent, int pdx, int pdy, int pdz) {
  const uint64_t id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (id >= (uint64_t)pdx * pdy * pdz) return;
  const int px = (int)(id % pdx), py = (int)((id / pdx) % pdy), pz = (int)(id / ((uint64_t)pdx * pdy));
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 4; ++i) {
        const int cx = 4 * px + i, cy = 4 * py + j, cz = 4 * pz + k;
        if (cx >= cdx || cy >= cdy || cz >= cdz) continue;
        const float* a = aabb + 8 * ((uint64_t)off_child + ((uint64_t)cz * cdy + cy) * cdx + cx);
        for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], a[d]); hi[d] = fmaxf(hi[d], a[3 + d]); }
      }
  float* o = aabb + 8 * ((uint64_t)off_parent + id);
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.f; o[7] = 0.f;
}

// ------------------------------------------------------------ host side ----
static void set_dims(GridParams& g, const float lo[3], const float hi[3], float h) {
  g.h = h;
  g.inv_h = 1.0f / h;
  float ext = 0.f;
  for (int d = 0; d < 3; ++d) {
    g.origin[d] = lo[d];
    double e = (double)hi[d] - (double)lo[d];
    long c = (long)std::floor(e / h) + 1;
    if (c < 1) c = 1;
    g.dims[d] = (int)c;
    g.bdims[d] = (g.dims[d] + kBlockCells - 1) / kBlockCells;
    ext = std::fmax(ext, (float)e);
    ext = std::fmax(ext, std::fmax(std::fabs(lo[d]), std::fabs(hi[d])));
  }
  g.qdims[0] = (g.dims[1] + 1) / 2;
  g.qdims[1] = (g.dims[2] + 1) / 2;
  // bound on |true coordinate - nominal cell face| caused by float binning (see nn.hip "slack")
  g.slack = ext * 9.6e-7f + 1e-30f;
}

// cell table entries (quad-row order pads odd y / z extents with empty cells)
static uint64_t num_cells(const GridParams& g) { return (uint64_t)g.dims[0] * g.qdims[0] * g.qdims[1] * 4u; }

constexpr uint64_t kMaxCells = 1ull << 26;
constexpr double kTargetOcc = 24.0;  // mean points per occupied cell

static pcd_status build_grid(pcd_cloud* c, float user_h, hipStream_t s) {
  const uint64_t n = c->n;
  GridParams& g = c->grid;
  DevBuf<int> bb;
  DevBuf<unsigned long long> cnt;
  PCD_TRY(bb.reserve(6));
  PCD_TRY(cnt.reserve(2));
  int init[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
  PCD_HIP_TRY(hipMemcpyAsync(bb.p, init, sizeof init, hipMemcpyHostToDevice, s));
  PCD_HIP_TRY(hipMemsetAsync(cnt.p, 0, 2 * sizeof(unsigned long long), s));
  if (n) {
    unsigned blocks = std::min<unsigned>(div_up(n, 256), 2048);
    hipLaunchKernelGGL(k_bbox, dim3(blocks), dim3(256), 0, s, c->pts4.p, n, bb.p, cnt.p);
  }
  int hb[6];
  unsigned long long hc[2];
  PCD_HIP_TRY(hipMemcpyAsync(hb, bb.p, sizeof hb, hipMemcpyDeviceToHost, s));
  PCD_HIP_TRY(hipMemcpyAsync(hc, cnt.p, sizeof hc, hipMemcpyDeviceToHost, s));
  PCD_HIP_TRY(hipStreamSynchronize(s));
  c->m = hc[0];
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  if (c->m) for (int d = 0; d < 3; ++d) { lo[d] = ord2f(hb[d]); hi[d] = ord2f(hb[3 + d]); }
  for (int d = 0; d < 3; ++d) { c->bb_lo[d] = lo[d]; c->bb_hi[d] = hi[d]; }

  // --- cell size: user value, or iterate towards kTargetOcc points per occupied cell ---
  double ext[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
  double maxext = std::fmax(ext[0], std::fmax(ext[1], ext[2]));
  auto min_h_for_budget = [&]() {
    double h = std::cbrt((ext[0] + 1e-3) * (ext[1] + 1e-3) * (ext[2] + 1e-3) / (double)kMaxCells);
    for (int it = 0; it < 64; ++it) {
      double cells = (std::floor(ext[0] / h) + 1) * (std::floor(ext[1] / h) + 1) * (std::floor(ext[2] / h) + 1);
      if (cells <= (double)kMaxCells) break;
      h *= 1.05;
    }
    return h;
  };
  float h;
  const double hmin = std::fmax(min_h_for_budget(), 1e-6 * std::fmax(maxext, 1e-3));
  if (c->m == 0 || maxext <= 0) {
    h = user_h > 0 ? user_h : 1.0f;
  } else if (user_h > 0) {
    h = (float)std::fmax((double)user_h, hmin);
  } else {
    // start from a surface-like guess and refine with measured occupancy
    double hguess = std::sqrt(kTargetOcc * (ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2] + 1e-6) / (double)c->m);
    h = (float)std::fmin(std::fmax(hguess, hmin), std::fmax(maxext, hmin));
    for (int it = 0; it < 6; ++it) {
      set_dims(g, lo, hi, h);
      uint64_t nc = num_cells(g);
      PCD_TRY(c->cell_start.reserve(nc + 1));
      PCD_HIP_TRY(hipMemsetAsync(c->cell_start.p, 0, (nc + 1) * sizeof(uint32_t), s));
      PCD_HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s));
      hipLaunchKernelGGL(k_cell_hist, dim3(div_up(n, 256)), dim3(256), 0, s, c->pts4.p, n, g, c->cell_start.p);
      hipLaunchKernelGGL(k_count_nonzero, dim3(std::min<unsigned>(div_up(nc, 256), 4096)), dim3(256), 0, s,
                         c->cell_start.p, nc, cnt.p);
      PCD_HIP_TRY(hipMemcpyAsync(hc, cnt.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      PCD_HIP_TRY(hipStreamSynchronize(s));
      double occ = (double)c->m / (double)std::max<unsigned long long>(hc[0], 1);
      if (occ > 0.75 * kTargetOcc && occ < 1.5 * kTargetOcc) break;
      // occupancy ~ h^2 on surfaces, h^3 in volumes: use exponent 2.5
      double hn = h * std::pow(kTargetOcc / occ, 1.0 / 2.5);
      hn = std::fmin(std::fmax(hn, hmin), std::fmax(maxext, hmin));
      if (std::fabs(hn - h) < 1e-3 * h) break;
      h = (float)hn;
    }
  }
  set_dims(g, lo, hi, h);
  c->ncells = num_cells(g);
  c->nblocks = (uint64_t)g.bdims[0] * g.bdims[1] * g.bdims[2];
  if (c->ncells > kMaxCells * 2) {
    set_error("grid of %llu cells exceeds the budget", (unsigned long long)c->ncells);
    return PCD_ERR_UNSUPPORTED;
  }

  // --- final histogram -> cell_start (exclusive scan) ---
  PCD_TRY(c->cell_start.reserve(c->ncells + 1));
  {
    DevBuf<uint32_t> counts;
    PCD_TRY(counts.reserve(c->ncells + 1));
    PCD_HIP_TRY(hipMemsetAsync(counts.p, 0, (c->ncells + 1) * sizeof(uint32_t), s));
    PCD_HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s));
    if (n) hipLaunchKernelGGL(k_cell_hist, dim3(div_up(n, 256)), dim3(256), 0, s, c->pts4.p, n, g, counts.p);
    hipLaunchKernelGGL(k_count_nonzero, dim3(std::min<unsigned>(div_up(c->ncells, 256), 4096)), dim3(256), 0, s,
                       counts.p, c->ncells, cnt.p);
    PCD_HIP_TRY(hipMemcpyAsync(hc, cnt.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    size_t tb = 0;
    PCD_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, counts.p, c->cell_start.p, 0u, c->ncells + 1,
                                        rocprim::plus<uint32_t>(), s));
    DevBuf<char> tmp;
    PCD_TRY(tmp.reserve(tb));
    PCD_HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, counts.p, c->cell_start.p, 0u, c->ncells + 1,
                                        rocprim::plus<uint32_t>(), s));
    PCD_HIP_TRY(hipStreamSynchronize(s));
  }
  c->occupied = hc[0];

  // --- sort rows by cell (stable radix sort keeps original order inside a cell) ---
  PCD_TRY(c->sorted.reserve(c->m + kSortedSpare));   // spare records, see k_gather_sorted
  if (n) {
    DevBuf<uint32_t> k0, k1, v0, v1;
    PCD_TRY(k0.reserve(n)); PCD_TRY(k1.reserve(n)); PCD_TRY(v0.reserve(n)); PCD_TRY(v1.reserve(n));
    hipLaunchKernelGGL(k_cell_keys, dim3(div_up(n, 256)), dim3(256), 0, s, c->pts4.p, n, g, k0.p, v0.p);
    size_t tb = 0;
    PCD_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, k0.p, k1.p, v0.p, v1.p, n, 0, 32, s));
    DevBuf<char> tmp;
    PCD_TRY(tmp.reserve(tb));
    PCD_HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, k0.p, k1.p, v0.p, v1.p, n, 0, 32, s));
    if (c->m)
      hipLaunchKernelGGL(k_gather_sorted, dim3(div_up(c->m + kSortedSpare, 256)), dim3(256), 0, s, c->pts4.p, v1.p, c->m,
                         c->index_base, c->index_stride, c->row_index.p, c->sorted.p);
    PCD_HIP_TRY(hipStreamSynchronize(s));
  }

  // --- leaves (2x2x2-cell sub-blocks: tight box + point range) + coarser pyramid levels ---
  PyramidParams& py = c->pyr;
  py.nlev = 1;
  py.dims[0][0] = (g.dims[0] + 1) / 2; py.dims[0][1] = g.qdims[0]; py.dims[0][2] = g.qdims[1];
  py.off[0] = 0;
  const uint64_t nleaves = (uint64_t)py.dims[0][0] * py.dims[0][1] * py.dims[0][2];
  uint64_t total_nodes = nleaves;
  // real levels until one has at most 64 nodes; above it sits a VIRTUAL top whose children are ALL nodes of that level,
  // one per lane (k_nn_fallback): for workload M's grid that is 5 x 2 x 5 = 50 nodes -- the walk starts there instead of
  // expanding a root of 1 and a level of 4 nodes first (two expansions and two pops less per query)
  while (py.nlev < kMaxPyrLevels - 1) {
    const int* pd = py.dims[py.nlev - 1];
    if ((uint64_t)pd[0] * pd[1] * pd[2] <= 64) break;
    for (int d = 0; d < 3; ++d) py.dims[py.nlev][d] = (pd[d] + 3) / 4;
    py.off[py.nlev] = (uint32_t)total_nodes;
    total_nodes += (uint64_t)py.dims[py.nlev][0] * py.dims[py.nlev][1] * py.dims[py.nlev][2];
    py.nlev++;
  }
  if ((uint64_t)py.dims[py.nlev - 1][0] * py.dims[py.nlev - 1][1] * py.dims[py.nlev - 1][2] > 64) {
    set_error("grid too large for %d pyramid levels", kMaxPyrLevels);
    return PCD_ERR_UNSUPPORTED;
  }
  py.dims[py.nlev][0] = py.dims[py.nlev][1] = py.dims[py.nlev][2] = 1;   // the virtual top
  py.off[py.nlev] = (uint32_t)total_nodes;
  py.nlev++;
  PCD_TRY(c->blk_aabb.reserve(8 * total_nodes));
  hipLaunchKernelGGL(k_leaf_aabb, dim3(div_up(nleaves * 64, 256)), dim3(256), 0, s, c->sorted.p, c->cell_start.p, g,
                     py.dims[0][0], py.dims[0][1], py.dims[0][2], c->blk_aabb.p);
  for (int l = 1; l < py.nlev - 1; ++l) {   // (the last level is the virtual top: no boxes)
    const int* cd = py.dims[l - 1];
    const int* pd = py.dims[l];
    const uint64_t np = (uint64_t)pd[0] * pd[1] * pd[2];
    hipLaunchKernelGGL(k_pyramid_level, dim3(div_up(np, 256)), dim3(256), 0, s, c->blk_aabb.p, py.off[l - 1], cd[0],
                       cd[1], cd[2], py.off[l], pd[0], pd[1], pd[2]);
  }
  PCD_HIP_TRY(hipStreamSynchronize(s));
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

void free_query_scratch(QueryScratch* s);  // nn.hip

}  // namespace pcd

using namespace pcd;

extern "C" {

void pcd_cloud_options_default(pcd_cloud_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof *o);
  o->layout = PCD_LAYOUT_XYZ_NRM;
  o->raw_lidar_frame = 1;
  o->index_stride = 1;
}

pcd_status pcd_cloud_create(const float* xyz, const float* nrm, uint64_t n, const pcd_cloud_options* opts,
                            pcd_cloud** out) {
  return pcd::guard([&]() -> pcd_status {
  return pcd::cloud_create_indexed(xyz, nrm, n, opts, nullptr, 0, out);
  });
}

}  // extern "C"

__global__ static void k_fill_g2l(const uint32_t* __restrict__ row_index, uint64_t n, uint32_t* __restrict__ g2l) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i < n) g2l[row_index[i]] = (uint32_t)i;
}

pcd_status pcd::cloud_create_indexed(const float* xyz, const float* nrm, uint64_t n, const pcd_cloud_options* opts,
                                     const uint32_t* row_index, uint64_t global_n, pcd_cloud** out) {
  PCD_REQUIRE(out, "out is null");
  *out = nullptr;
  pcd_cloud_options o;
  if (opts) o = *opts; else pcd_cloud_options_default(&o);
  PCD_REQUIRE(o.layout == PCD_LAYOUT_XYZ_NRM || o.layout == PCD_LAYOUT_AOS32, "unknown layout");
  PCD_REQUIRE(n == 0 || xyz, "xyz is null");
  PCD_REQUIRE(n == 0 || o.layout == PCD_LAYOUT_AOS32 || nrm, "nrm is null");
  PCD_REQUIRE(n < 0xFFFFFFF0ull, "more than 2^32 rows");
  if (o.index_stride == 0) o.index_stride = 1;
  PCD_REQUIRE(!(o.raw_lidar_frame && (o.index_stride != 1 || o.index_base != 0)),
              "sharded clouds (index_stride/base) need raw_lidar_frame = 0");
  PCD_REQUIRE(o.cell_size >= 0 && std::isfinite(o.cell_size), "cell_size");
  PCD_TRY(require_device(o.device));

  auto t0 = std::chrono::steady_clock::now();
  pcd_cloud* c = new pcd_cloud();
  c->device = o.device;
  c->index_base = o.index_base;
  c->index_stride = o.index_stride;
  hipStream_t s = nullptr;
  auto fail = [&](pcd_status st) { pcd_cloud_destroy(c); return st; };
  if (row_index) {
    if (o.raw_lidar_frame) { set_error("indexed shards need raw_lidar_frame = 0"); return fail(PCD_ERR_INVALID); }
    pcd_status sr;
    if ((sr = c->row_index.reserve(std::max<uint64_t>(n, 1))) != PCD_OK) return fail(sr);
    if ((sr = c->g2l.reserve(std::max<uint64_t>(global_n, 1))) != PCD_OK) return fail(sr);
    c->g2l_n = global_n;
    if (n && hipMemcpy(c->row_index.p, row_index, n * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return fail(PCD_ERR_HIP);
    if (hipMemset(c->g2l.p, 0xFF, std::max<uint64_t>(global_n, 1) * sizeof(uint32_t)) != hipSuccess) return fail(PCD_ERR_HIP);
    if (n) hipLaunchKernelGGL(k_fill_g2l, dim3(div_up(n, 256)), dim3(256), 0, s, c->row_index.p, n, c->g2l.p);
  }

  const size_t row = o.layout == PCD_LAYOUT_AOS32 ? 8 : 3;
  DevBuf<float> d_xyz, d_nrm;
  DevBuf<uint32_t> keep, pos;
  pcd_status st;
  if ((st = d_xyz.reserve(std::max<size_t>(n * row, 1))) != PCD_OK) return fail(st);
  if ((st = d_nrm.reserve(std::max<size_t>(n * 3, 1))) != PCD_OK) return fail(st);
  if (n) {
    if (hipMemcpy(d_xyz.p, xyz, n * row * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(PCD_ERR_HIP);
    if (o.layout == PCD_LAYOUT_XYZ_NRM &&
        hipMemcpy(d_nrm.p, nrm, n * 3 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(PCD_ERR_HIP);
  }
  uint64_t kept = n;
  if (o.raw_lidar_frame && n) {
    if ((st = keep.reserve(n)) != PCD_OK) return fail(st);
    if ((st = pos.reserve(n)) != PCD_OK) return fail(st);
    hipLaunchKernelGGL(k_flag_rows, dim3(div_up(n, 256)), dim3(256), 0, s, d_xyz.p, d_nrm.p, n, o.layout, keep.p);
    size_t tb = 0;
    if (rocprim::exclusive_scan(nullptr, tb, keep.p, pos.p, 0u, n, rocprim::plus<uint32_t>(), s) != hipSuccess)
      return fail(PCD_ERR_HIP);
    DevBuf<char> tmp;
    if ((st = tmp.reserve(tb)) != PCD_OK) return fail(st);
    if (rocprim::exclusive_scan(tmp.p, tb, keep.p, pos.p, 0u, n, rocprim::plus<uint32_t>(), s) != hipSuccess)
      return fail(PCD_ERR_HIP);
    uint32_t lastp = 0, lastk = 0;
    if (hipMemcpy(&lastp, pos.p + (n - 1), 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(PCD_ERR_HIP);
    if (hipMemcpy(&lastk, keep.p + (n - 1), 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(PCD_ERR_HIP);
    kept = (uint64_t)lastp + lastk;
  }
  c->n = kept;
  if ((st = c->pts4.reserve(std::max<uint64_t>(kept, 1))) != PCD_OK) return fail(st);
  if ((st = c->pn8.reserve(2 * std::max<uint64_t>(kept, 1))) != PCD_OK) return fail(st);
  if (n)
    hipLaunchKernelGGL(k_compact_rows, dim3(div_up(n, 256)), dim3(256), 0, s, d_xyz.p, d_nrm.p, n, o.layout,
                       o.raw_lidar_frame, keep.p, pos.p, c->pts4.p, c->pn8.p);
  if (hipStreamSynchronize(s) != hipSuccess) { set_error("row transform failed"); return fail(PCD_ERR_HIP); }
  if ((st = build_grid(c, o.cell_size, s)) != PCD_OK) return fail(st);
  c->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *out = c;
  return PCD_OK;
}

extern "C" {

void pcd_cloud_destroy(pcd_cloud* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  pcd::free_query_scratch(c->scratch);
  delete c;
}

uint64_t pcd_cloud_size(const pcd_cloud* c) { return c ? c->n : 0; }

pcd_status pcd_cloud_get_info(const pcd_cloud* c, pcd_cloud_info* info) {
  PCD_REQUIRE(c && info, "null pointer");
  info->cell_size = c->grid.h;
  for (int d = 0; d < 3; ++d) {
    info->origin[d] = c->grid.origin[d];
    info->dims[d] = c->grid.dims[d];
    info->block_dims[d] = c->grid.bdims[d];
  }
  info->num_indexed = c->m;
  info->occupied_cells = c->occupied;
  info->build_ms = c->build_ms;
  for (int d = 0; d < 3; ++d) { info->bbox_lo[d] = c->bb_lo[d]; info->bbox_hi[d] = c->bb_hi[d]; }
  return PCD_OK;
}

pcd_status pcd_cloud_download(const pcd_cloud* c, float* xyz, float* nrm) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(c, "null cloud");
  PCD_HIP_TRY(hipSetDevice(c->device));
  std::vector<float4> h(2 * c->n);
  if (xyz && c->n) {
    PCD_HIP_TRY(hipMemcpy(h.data(), c->pts4.p, c->n * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < c->n; ++i) { xyz[3 * i] = h[i].x; xyz[3 * i + 1] = h[i].y; xyz[3 * i + 2] = h[i].z; }
  }
  if (nrm && c->n) {
    PCD_HIP_TRY(hipMemcpy(h.data(), c->pn8.p, 2 * c->n * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < c->n; ++i) { nrm[3 * i] = h[2 * i + 1].x; nrm[3 * i + 1] = h[2 * i + 1].y; nrm[3 * i + 2] = h[2 * i + 1].z; }
  }
  return PCD_OK;
  });
}

}  // extern "C"
