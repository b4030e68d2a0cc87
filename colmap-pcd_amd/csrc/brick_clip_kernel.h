// brick_clip_kernel.h -- the default first stage of the NN grid path (included by nn.hip behind brick_kernel.h).
//
// k_nn_brick (brick_kernel.h) compares every query of a brick with every point of the brick's 6x6x6-cell region:
// 1 293 point-query pairs per query on workload M, 278 M staged points for a 10 M-point cloud.  Most of them cannot
// matter: a query that sits on a sampled surface has its neighbour a few centimetres away, the region reaches 0.49 m.
// This kernel keeps the brick kernel's machinery (one wavefront per work item of <= 8 queries of one brick, wave-uniform
// queries, LDS-DMA tiles, FLANN's float arithmetic + v_min_f64 on the packed key, one transposed reduction per item)
// and clips the region INSIDE the item, wave-uniformly:
//
//   stage A   the brick's own 2x2x2 cells (one contiguous range of the centre quad row) are staged and compared,
//             64 points per step; one reduction gives every query its tentative distance d_k (or the bound it came
//             with: gate-bounded search, refining another shard's result);
//   clip      a point that can still beat query k lies in the ball of radius sqrt(d_k) around it.  Lane k turns its
//             ball's bounding box into a mask of (quad row, x cell) pairs of the region -- 9 rows x 6 cells = 54 bits;
//             the masks of the item's queries are OR-ed (three DPP steps) and become wave-uniform scalars;
//   stage B   of every quad row only the hull of its masked x cells is staged (the centre row without the cells of
//             stage A); rows the balls do not touch are dropped.  The ranges come from ONE table per item -- lane
//             7 row + k holds cell_start at x boundary k of quad row `row` (63 loads in one instruction, prefetched
//             an item ahead) -- so clipping costs no dependent memory access: v_readlane with a scalar lane index.
//
// Exactness.  Every point that is not compared lies (a) outside the region -- the region bound of nn.hip's header
// decides as before whether the query is final -- or (b) inside the region but outside the box of every query's ball:
// then its float distance to query k exceeds d_k >= the final distance (strictly; an equal distance lies inside the
// closed ball).  For (b): FLANN's sum is monotone in every term, so fl_dist(q, p) <= d implies fl((qx - px)^2) <= d,
// i.e. |qx - px| <= sqrt(d) (1 + 2^-23); the radius is widened by 1e-5 relative, the box by 2e-7 relative outwards, and
// cells are compared through the very expression the build bins points with (grid.h cell_coord: monotone in the
// coordinate), so a point's cell lies between the cells of the box's corners.
#pragma once

namespace pcd {

constexpr int kClipNk = 7;      // x boundaries of a region row: cells 2 bx - 2 .. 2 bx + 4
constexpr int kClipRows = 9;    // quad rows of the region: (by - 1 .. by + 1) x (bz - 1 .. bz + 1)
constexpr int kATile = 128;     // points stage A stages at most (2 DMA instructions); the rest of the brick's range joins stage B
constexpr int kClipRanges = 10; // 8 halo rows + the centre row's left and right parts
#ifdef PCD_ABLATE   // timing-only ablations (tools/nn_ablate.py; results are wrong): brick_kernel.h
constexpr int kAblateClipMath = 0x10000, kAblateBound = 0x20000, kAblateStageA = 0x40000, kAblateSelect = 0x80000;
#else
constexpr int kAblateClipMath = 0, kAblateBound = 0, kAblateStageA = 0, kAblateSelect = 0;
#endif

// one staged point against ONE query (the odd query of a group when a step holds one point per lane)
__device__ __forceinline__ void compare_q1p1(const f32x4 p, const float qx, const float qy, const float qz, double& best) {
  asm("v_mov_b32 v120, %[pw]\n\t"
      "v_sub_f32 v122, %[qx], %[px]\n\tv_sub_f32 v123, %[qy], %[py]\n\tv_sub_f32 v124, %[qz], %[pz]\n\t"
      "v_mul_f32 v122, v122, v122\n\tv_mul_f32 v123, v123, v123\n\tv_mul_f32 v124, v124, v124\n\t"
      "v_add_f32 v122, v122, v123\n\tv_add_f32 v121, v122, v124\n\t"
      "v_min_f64 %[b], %[b], v[120:121]\n\t"
      : [b] "+v"(best)
      : [px] "v"(p.x), [py] "v"(p.y), [pz] "v"(p.z), [pw] "v"(p.w), [qx] "v"(qx), [qy] "v"(qy), [qz] "v"(qz)
      : "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

// NP (1 or 2) staged points per lane against the first NQ queries of the group
template <int NQ, int NP>
__device__ __forceinline__ void compare_step(const f32x4 (&p)[2], const float (&qx)[8], const float (&qy)[8],
                                             const float (&qz)[8], double (&best)[8]) {
  static_assert(NQ >= 1 && NQ <= 8 && (NP == 1 || NP == 2), "1..8 queries, 1 or 2 points");
  constexpr int E = NQ & ~1;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (E >= 4) compare_point4(p[k], qx, qy, qz, best);
    if (E == 8) compare_point4(p[k], qx + 4, qy + 4, qz + 4, best + 4);
    if (E == 2 || E == 6) compare_point2(p[k], qx + (E - 2), qy + (E - 2), qz + (E - 2), best + (E - 2));
  }
  if (NQ & 1) {
    if (NP == 2) compare_query1(p[0], p[1], qx[NQ - 1], qy[NQ - 1], qz[NQ - 1], best[NQ - 1]);
    else compare_q1p1(p[0], qx[NQ - 1], qy[NQ - 1], qz[NQ - 1], best[NQ - 1]);
  }
}

// nn.hip proven_bound in float, rounded to the safe side: squared safe radius around q inside the cells [c0, c1); < 0:
// nothing is proven.  The face positions and the differences carry at most 2 ulp of max(|face|, |q|) = 2.4e-7 max |coord|
// of rounding, which a second grid.slack (9.6e-7 max(extent, |coord|)) covers; the square is shrunk by 1e-5 instead of
// 1e-6.  The bound never exceeds the double one, so a query it proves is proven there as well.
__device__ __forceinline__ float proven_bound_f(const GridParams& g, float qx, float qy, float qz, const int c0[3],
                                                const int c1[3]) {
  const float q[3] = {qx, qy, qz};
  float margin = 3e38f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (c0[d] > 0) margin = fminf(margin, q[d] - (g.origin[d] + (float)c0[d] * g.h));
    if (c1[d] < g.dims[d]) margin = fminf(margin, (g.origin[d] + (float)c1[d] * g.h) - q[d]);
  }
  if (margin >= 3e38f) return 3e38f;   // the range covers the whole grid
  margin -= 2.0f * g.slack;
  if (!(margin > 0.0f)) return -1.0f;
  return margin * margin * (1.0f - 1e-5f);
}

struct ClipMeta {    // per-item loads issued one item ahead
  float4 q;          // lane < cnt: query (x, y, z, bits(query id))
  uint64_t prior;    // lane < cnt: the key the query came with
  uint32_t bnd;      // lane 7 row + k: cell_start at x boundary k of region row `row` (0 for rows outside the grid)
};

// lpos: the lane's place in the boundary table, k | ry << 8 | rz << 16 (rz == 3: no such row)
__device__ __forceinline__ ClipMeta clip_load_meta(const GridParams& g, const uint4 it, const float4* __restrict__ qsorted,
                                                   const uint64_t* __restrict__ ksorted,
                                                   const uint32_t* __restrict__ cell_start, uint32_t lpos) {
  const int lane = threadIdx.x & 63;
  const int lk = (int)(lpos & 0xFFu), lry = (int)((lpos >> 8) & 0xFFu), lrz = (int)(lpos >> 16);
  ClipMeta m;
  // every lane issues every load (clamped indices, results masked afterwards): brick_kernel.h brick_load_meta
  const int cnt = item_count(it);   // >= 1
  m.q = qsorted[it.x + (lane < cnt ? lane : cnt - 1)];
  m.prior = ksorted[it.x + (lane < cnt ? lane : cnt - 1)];
  const int bx = (int)it.y, by = (int)it.z, bz = (int)(it.w & 0x0FFFFFFFu);
  const int yq = by - 1 + lry, zq = bz - 1 + lrz;
  const bool ok = lrz < 3 && yq >= 0 && yq < g.qdims[0] && zq >= 0 && zq < g.qdims[1];
  const int x = min(max(2 * bx - 2 + lk, 0), g.dims[0]);
  const uint32_t v = cell_start[quad_row_base(g, ok ? yq : 0, ok ? zq : 0) + 4u * (uint64_t)x];
  m.bnd = ok ? v : 0u;
  return m;
}

__global__ __launch_bounds__(256, 4) void k_nn_brick_clip(GridParams g, const float4* __restrict__ sorted,
                                                          const uint32_t* __restrict__ cell_start,
                                                          const float4* __restrict__ qsorted,
                                                          const uint64_t* __restrict__ ksorted,
                                                          const uint4* __restrict__ items, NnCounters* __restrict__ ctr,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ fb_list,
                                                          uint32_t* __restrict__ fb_count, int flags) {
  constexpr int G = 8;
  constexpr int NK = kClipNk, NC = kClipNk - 1;   // x boundaries / cells of a region row
  const int collect_stats = flags & 1;
  const bool no_clip = (flags & 2) != 0;   // A/B switch: stage the whole region in stage B (results identical)
  __shared__ __attribute__((aligned(16))) float4 s_tile[4][2][kTile];
  __shared__ __attribute__((aligned(16))) float4 s_a[4][kATile];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t nitems = ctr->nitems;
  const uint32_t nwaves = gridDim.x * 4;
  unsigned long long st_staged = 0, st_pairs = 0, st_groups = 0;
  uint32_t fb_base = 0, fb_left = 0;
  // the lane's place in the boundary table
  uint32_t lpos;
  {
    const int lrow = (lane * 9363) >> 16, lk = lane - 7 * lrow;   // lane / 7 for lane < 64
    const int lrz = (lrow * 21846) >> 16, lry = lrow - 3 * lrz;   // lrow / 3 (lane 63: row 9 = (0, 3), not a row)
    lpos = (uint32_t)lk | ((uint32_t)lry << 8) | ((uint32_t)lrz << 16);
    asm volatile("" : "+v"(lpos));   // one register; not three hoisted ones
  }
  // entry e of the table of item metadata m (e wave-uniform)
#define PCD_BND(m, e) PCD_RL((m).bnd, (e))

  // XCD-aware work split (brick_kernel.h): blocks b, b+8, ... walk their own contiguous eighth of the item list
  uint32_t item, item_end, stride;
  if ((gridDim.x & 7u) == 0) {
    const uint32_t cls = blockIdx.x & 7u, per = (nitems + 7u) / 8u;
    stride = (gridDim.x >> 3) * 4;
    item = cls * per + (blockIdx.x >> 3) * 4 + wave;
    item_end = min(nitems, (cls + 1) * per);
  } else {
    stride = nwaves;
    item = blockIdx.x * 4 + wave;
    item_end = nitems;
  }
  if (item >= item_end) return;
  const char* __restrict__ src_bytes = reinterpret_cast<const char*>(sorted);
  float4* const bufA = s_a[wave];
#define PCD_RL(v, r) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(r)))
  // stage A of an item: the brick's own cells = x boundaries 2 .. 4 of the centre row (row 4); always exactly two
  // DMA instructions (lanes past the range re-read its last record; an empty range reads the record at its start,
  // which exists -- `sorted` ends with spare records -- and is not compared)
  auto issue_a = [&](const ClipMeta& m) {
    const uint32_t sA = PCD_BND(m, 4 * NK + 2), eA = PCD_BND(m, 4 * NK + 4);
    const uint32_t nA = min(eA - sA, (uint32_t)kATile);
    const uint32_t last = nA ? nA - 1u : 0u;
    const uint32_t i0 = sA + min((uint32_t)lane, last), i1 = sA + min((uint32_t)lane + 64u, last);
    lds_dma16(reinterpret_cast<const float4*>(src_bytes + ((uint64_t)i0 << 4)), bufA);
    lds_dma16(reinterpret_cast<const float4*>(src_bytes + ((uint64_t)i1 << 4)), bufA + 64);
  };

  uint4 it0 = items[PCD_UNI(item)];
  ClipMeta m0 = clip_load_meta(g, it0, qsorted, ksorted, cell_start, lpos);
  uint4 it1 = items[PCD_UNI(min(item + stride, item_end - 1))];
  issue_a(m0);

  for (; item < item_end; item += stride) {
    // stage A of this item (issued while the previous item was compared) has landed; nothing else is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- prefetch: metadata of the next item, item record of the one after (clamped to the last item) ----
    const ClipMeta m1 = clip_load_meta(g, it1, qsorted, ksorted, cell_start, lpos);
    const uint4 it2 = items[PCD_UNI(min(item + 2 * stride, item_end - 1))];

    const uint32_t cnt = (uint32_t)item_count(it0);
    // The item's queries as wave-uniform values in VGPRs (an SGPR operand halves the VALU rate: brick_kernel.h).  Through
    // LDS: every lane writes its record (lanes >= cnt hold a copy of the last query) into the first tile buffer -- free
    // between two items' tiles -- and all lanes read the first 8 records back, 12 bytes each, same address for the whole
    // wavefront (a broadcast).  One ds_write_b128 + 8 ds_read_b96 on the LDS port instead of 24 v_readlane + 24 v_mov
    // on the VALU and a branch per query slot.
    float qx[G], qy[G], qz[G];
    {
      typedef float f32x3 __attribute__((ext_vector_type(3)));
      const uint32_t qb = lds_addr(s_tile[wave][0]);
      const f32x4 qrec = {m0.q.x, m0.q.y, m0.q.z, m0.q.w};
      f32x3 qv[G];
      asm volatile("ds_write_b128 %9, %8\n\ts_waitcnt lgkmcnt(0)\n\t"
                   "ds_read_b96 %0, %10\n\tds_read_b96 %1, %10 offset:16\n\tds_read_b96 %2, %10 offset:32\n\t"
                   "ds_read_b96 %3, %10 offset:48\n\tds_read_b96 %4, %10 offset:64\n\tds_read_b96 %5, %10 offset:80\n\t"
                   "ds_read_b96 %6, %10 offset:96\n\tds_read_b96 %7, %10 offset:112\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(qv[0]), "=&v"(qv[1]), "=&v"(qv[2]), "=&v"(qv[3]), "=&v"(qv[4]), "=&v"(qv[5]), "=&v"(qv[6]), "=&v"(qv[7])
                   : "v"(qrec), "v"(qb + (uint32_t)lane * 16u), "v"(qb)
                   : "memory");
#pragma unroll
      for (int k = 0; k < G; ++k) { qx[k] = qv[k].x; qy[k] = qv[k].y; qz[k] = qv[k].z; }
    }
    const int bx = (int)it0.y, by = (int)it0.z, bz = (int)(it0.w & 0x0FFFFFFFu);
    double best[G];
#pragma unroll
    for (int k = 0; k < G; ++k) best[k] = __builtin_bit_cast(double, kKeyInit);

    // ---- stage A: the brick's own cells ----------------------------------------------------------------------
    const uint32_t sA = PCD_BND(m0, 4 * NK + 2), eA = PCD_BND(m0, 4 * NK + 4);
    const uint32_t nA = min(eA - sA, (uint32_t)kATile);
    if (nA > 0 && !(flags & kAblateStageA)) {
      f32x4 pa[2];
      const uint32_t rd = lds_addr(bufA) + lane * 16;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(pa[0]), "=&v"(pa[1]) : "v"(rd) : "memory");
      if (flags & kAblateCompare) {
        asm volatile("" ::"v"(pa[0]), "v"(pa[1]));
      } else if (nA > 64) {
        switch (cnt) {   // wave-uniform
          case 1: compare_step<1, 2>(pa, qx, qy, qz, best); break;
          case 2: compare_step<2, 2>(pa, qx, qy, qz, best); break;
          case 3: compare_step<3, 2>(pa, qx, qy, qz, best); break;
          case 4: compare_step<4, 2>(pa, qx, qy, qz, best); break;
          case 5: compare_step<5, 2>(pa, qx, qy, qz, best); break;
          case 6: compare_step<6, 2>(pa, qx, qy, qz, best); break;
          case 7: compare_step<7, 2>(pa, qx, qy, qz, best); break;
          default: compare_step<8, 2>(pa, qx, qy, qz, best); break;
        }
      } else {
        switch (cnt) {
          case 1: compare_step<1, 1>(pa, qx, qy, qz, best); break;
          case 2: compare_step<2, 1>(pa, qx, qy, qz, best); break;
          case 3: compare_step<3, 1>(pa, qx, qy, qz, best); break;
          case 4: compare_step<4, 1>(pa, qx, qy, qz, best); break;
          case 5: compare_step<5, 1>(pa, qx, qy, qz, best); break;
          case 6: compare_step<6, 1>(pa, qx, qy, qz, best); break;
          case 7: compare_step<7, 1>(pa, qx, qy, qz, best); break;
          default: compare_step<8, 1>(pa, qx, qy, qz, best); break;
        }
      }
    }
    // lane k < cnt: the tentative key of query k (value v of the reduction sits in the lanes with bits (5,4,3) = v)
    const int holder = ((lane & 4) ? 32 : 0) | ((lane & 2) ? 16 : 0) | ((lane & 1) ? 8 : 0);
    uint64_t mine;
    {
      const uint64_t red = __builtin_bit_cast(uint64_t, (flags & kAblateReduce) ? best[0] : wave_min8_key(best));
      mine = ((uint64_t)__shfl((uint32_t)(red >> 32), holder) << 32) | __shfl((uint32_t)red, holder);
      mine = min_u64(mine, m0.prior);
    }
    // ---- clip: mask of the (row, x cell) pairs the balls of the item's queries touch --------------------------
    uint32_t w0 = 0, w1 = 0, w2 = 0;   // z slab 0 / 1 / 2: bit 6 ry + k
    if (flags & kAblateClipMath) {
      w0 = w1 = w2 = lane < (int)cnt ? 0x3FFFFu : 0u;
    } else if (lane < (int)cnt) {
      // radius: sqrt(d) widened by 1e-5 (v_sqrt_f32's 1 ulp, the roundings of FLANN's sum) + the rounding of q -+ r
      // (half an ulp of |q| + r: 2.4e-7 |q| covers it four times over).  Cells in float: t = (p - origin) / h as the
      // build computes it (grid.h cell_coord_raw); floor(t) - first cell of the region clamped to the region is what the
      // clamped integer cell coordinate gives or wider (only the clamp to the grid is missing, and cells outside the
      // grid are empty ranges of the boundary table); quad rows: floor(floor(t) / 2) = floor(t / 2), exact in float.
      const float d = __uint_as_float((uint32_t)(mine >> 32));
      const float r = __builtin_amdgcn_sqrtf(d) * 1.00001f + ((fabsf(m0.q.x) + fabsf(m0.q.y)) + fabsf(m0.q.z)) * 2.4e-7f + 1e-18f;
      const float tx0 = floorf((m0.q.x - r - g.origin[0]) * g.inv_h), tx1 = floorf((m0.q.x + r - g.origin[0]) * g.inv_h);
      const float ty0 = floorf((m0.q.y - r - g.origin[1]) * g.inv_h * 0.5f), ty1 = floorf((m0.q.y + r - g.origin[1]) * g.inv_h * 0.5f);
      const float tz0 = floorf((m0.q.z - r - g.origin[2]) * g.inv_h * 0.5f), tz1 = floorf((m0.q.z + r - g.origin[2]) * g.inv_h * 0.5f);
      const float fx = (float)(2 * bx - 2), fy = (float)(by - 1), fz = (float)(bz - 1);
      const uint32_t kx0 = (uint32_t)__builtin_amdgcn_fmed3f(tx0 - fx, 0.f, (float)(NC - 1)), kx1 = (uint32_t)__builtin_amdgcn_fmed3f(tx1 - fx, 0.f, (float)(NC - 1));
      const uint32_t ry0 = (uint32_t)__builtin_amdgcn_fmed3f(ty0 - fy, 0.f, 2.f), ry1 = (uint32_t)__builtin_amdgcn_fmed3f(ty1 - fy, 0.f, 2.f);
      const uint32_t rz0 = (uint32_t)__builtin_amdgcn_fmed3f(tz0 - fz, 0.f, 2.f), rz1 = (uint32_t)__builtin_amdgcn_fmed3f(tz1 - fz, 0.f, 2.f);
      const uint32_t xm = (2u << kx1) - (1u << kx0);                     // bits kx0 .. kx1
      const uint32_t ym = (2u << ry1) - (1u << ry0), zm = (2u << rz1) - (1u << rz0);
      const uint32_t pat = xm * ((ym * 0x421u) & 0x1041u);               // the x mask in the 6-bit fields of the rows ry0 .. ry1
      w0 = (zm & 1u) ? pat : 0u; w1 = (zm & 2u) ? pat : 0u; w2 = (zm & 4u) ? pat : 0u;
      if (no_clip) w0 = w1 = w2 = 0x3FFFFu;
    }
    // OR over lanes 0..7 (row_shr 1, 2, 4; lanes without a source contribute 0): lane 7 holds the item's mask
#define PCD_OR_SHR(v, ctrl) v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, true)
    PCD_OR_SHR(w0, 0x111); PCD_OR_SHR(w1, 0x111); PCD_OR_SHR(w2, 0x111);
    PCD_OR_SHR(w0, 0x112); PCD_OR_SHR(w1, 0x112); PCD_OR_SHR(w2, 0x112);
    PCD_OR_SHR(w0, 0x114); PCD_OR_SHR(w1, 0x114); PCD_OR_SHR(w2, 0x114);
#undef PCD_OR_SHR
    const uint32_t M[3] = {PCD_RL(w0, 7), PCD_RL(w1, 7), PCD_RL(w2, 7)};
    // ---- stage B ranges (all wave-uniform scalars): start and length of the 10 ranges -------------------------
    uint32_t rs[kClipRanges], rl[kClipRanges];
#pragma unroll
    for (int r = 0; r < kClipRows; ++r) {
      const uint32_t xm = (M[r / 3] >> (6 * (r % 3))) & 63u;
      const int k0 = xm ? __builtin_ctz(xm) : 0, k1 = xm ? 32 - __builtin_clz(xm) : 0;   // cells k0 .. k1 - 1
      const uint32_t s = PCD_BND(m0, r * NK + k0), e = PCD_BND(m0, r * NK + k1);
      if (r == 4) {
        // the centre row without what stage A staged: [s, sA) and [sA + nA, e)
        const uint32_t rgt = sA + nA;
        rs[4] = s; rl[4] = (xm && sA > s) ? sA - s : 0u;
        rs[9] = rgt; rl[9] = (xm && e > rgt) ? e - rgt : 0u;
      } else {
        rs[r] = s; rl[r] = e - s;
      }
    }
    uint32_t ro[kClipRanges + 1];   // start of range i in the concatenation padded to groups of 4 slots
    ro[0] = 0;
#pragma unroll
    for (int i = 0; i < kClipRanges; ++i) ro[i + 1] = ro[i] + ((rl[i] + 3u) & ~3u);
    const uint32_t T = (flags & kAblateTiles) ? 0u : ro[kClipRanges];
    const int ntiles = (int)((T + kTile - 1) / kTile);

    if (T > 0) {
      uint32_t dv[kClipRanges];   // source - start deltas in VGPRs (v_cndmask cannot read an SGPR next to VCC)
#pragma unroll
      for (int i = 0; i < kClipRanges; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(dv[i]) : "s"(rs[i] - ro[i]));
      auto issue_tile = [&](int t) {
        float4* buf = s_tile[wave][t & 1];
        const uint32_t s4 = min((uint32_t)t * kTile + 4u * (uint32_t)lane, T - 4u);
        uint32_t dl = dv[0];   // empty ranges share their start with the next one: the last one wins
        if (!(flags & kAblateSelect)) {
#define PCD_SEL(o, d) asm("v_cmp_le_u32_e32 vcc, %2, %1\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(dl) : "v"(s4), "s"(o), "v"(d) : "vcc")
        PCD_SEL(ro[1], dv[1]); PCD_SEL(ro[2], dv[2]); PCD_SEL(ro[3], dv[3]); PCD_SEL(ro[4], dv[4]); PCD_SEL(ro[5], dv[5]);
        PCD_SEL(ro[6], dv[6]); PCD_SEL(ro[7], dv[7]); PCD_SEL(ro[8], dv[8]); PCD_SEL(ro[9], dv[9]);
#undef PCD_SEL
        }
        const float4* gp = reinterpret_cast<const float4*>(src_bytes + ((uint64_t)(s4 + dl) << 4));
        // (the instruction offset is added to the LDS address as well as to the source address: brick_kernel.h)
        if (flags & kAblateNoDma) { asm volatile("" ::"v"(gp)); return; }
        lds_dma16_off<0>(gp, buf);
        lds_dma16_off<16>(gp, buf + 64 - 1);
        lds_dma16_off<32>(gp, buf + 128 - 2);
        lds_dma16_off<48>(gp, buf + 192 - 3);
      };
      issue_tile(0);
      issue_a(m1);   // stage A of the NEXT item: lands under this item's tiles
      // One tile loop per group size (cnt is wave-uniform): the dispatch on the number of queries happens once per item,
      // not once per tile -- the compare-and-branch chain in front of every tile's compare block was 7 % of the kernel
      // (0.488 -> 0.453 ms), more than any memory instruction of the loop.
      auto run_tiles = [&](auto nq_tag) {
        constexpr int NQ = decltype(nq_tag)::value;
        for (int t = 0; t < ntiles; ++t) {
          // in flight, oldest first: tile 0 (4), stage A of the next item (2), tile 1 (4), ...
          if (t + 1 < ntiles) {
            issue_tile(t + 1);
            if (t == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          } else {
            if (t == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_wave_barrier();
          f32x4 p[4];
          const uint32_t rd = lds_addr(s_tile[wave][t & 1]) + lane * 16;
          if (flags & kAblateNoLdsRead) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = f32x4{(float)lane, (float)t, (float)k, 0.f};
          } else
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                       "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3])
                       : "v"(rd)
                       : "memory");
          if (flags & kAblateCompare) {
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(p[k]));
          } else {
            compare_tile<NQ>(p, qx, qy, qz, best);
          }
        }
      };
      switch (cnt) {   // wave-uniform
        case 1: run_tiles(std::integral_constant<int, 1>{}); break;
        case 2: run_tiles(std::integral_constant<int, 2>{}); break;
        case 3: run_tiles(std::integral_constant<int, 3>{}); break;
        case 4: run_tiles(std::integral_constant<int, 4>{}); break;
        case 5: run_tiles(std::integral_constant<int, 5>{}); break;
        case 6: run_tiles(std::integral_constant<int, 6>{}); break;
        case 7: run_tiles(std::integral_constant<int, 7>{}); break;
        default: run_tiles(std::integral_constant<int, 8>{}); break;
      }
      // the second reduction: stage A's result and the key the query came with are in `mine` already
      const uint64_t red = __builtin_bit_cast(uint64_t, (flags & kAblateReduce) ? best[0] : wave_min8_key(best));
      mine = min_u64(mine, ((uint64_t)__shfl((uint32_t)(red >> 32), holder) << 32) | __shfl((uint32_t)red, holder));
    } else {
      issue_a(m1);
    }
    // ---- epilogue: final, or on to the exact fallback with the tentative key as starting bound ----------------
    bool unproven = false;
    if (lane < (int)cnt) {
      const uint32_t my_qi = __float_as_uint(m0.q.w);
      const int c0[3] = {max(2 * bx - 2, 0), max(2 * by - 2, 0), max(2 * bz - 2, 0)};
      const int c1[3] = {min(2 * bx + 4, g.dims[0]), min(2 * by + 4, g.dims[1]), min(2 * bz + 4, g.dims[2])};
      const float bound = (flags & kAblateBound) ? 3e38f : proven_bound_f(g, m0.q.x, m0.q.y, m0.q.z, c0, c1);
      unproven = !(__uint_as_float((uint32_t)(mine >> 32)) < bound) && !(flags & kAblateFallback);
      keys[my_qi] = unproven ? mine : finalized_key(mine);   // final, or the starting bound of the exact fallback
    }
    const unsigned long long um = __ballot(unproven);
    if (um) {   // chunked fallback list: brick_kernel.h
      const uint32_t k = (uint32_t)__popcll(um);
      if (k > fb_left) {
        if (lane < (int)fb_left) fb_list[fb_base + lane] = 0xFFFFFFFFu;
        uint32_t nb = 0;
        if (lane == 0) nb = atomicAdd(fb_count, (uint32_t)kFbChunk);
        fb_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
        fb_left = kFbChunk;
      }
      if (unproven) fb_list[fb_base + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u))] = __float_as_uint(m0.q.w);
      fb_base += k;
      fb_left -= k;
    }
    if (collect_stats) {
      const uint32_t slots = (nA > 64 ? 128u : nA ? 64u : 0u) + (uint32_t)ntiles * kTile;
      st_staged += nA + T; st_pairs += (unsigned long long)slots * cnt; st_groups += 1;
    }
    it0 = it1; it1 = it2; m0 = m1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (redundant) stage-A prefetch must not outlive the wavefront's LDS
  if (lane < (int)fb_left) fb_list[fb_base + lane] = 0xFFFFFFFFu;
  if (collect_stats && lane == 0) {
    atomicAdd(&ctr->staged_points, st_staged);
    atomicAdd(&ctr->pair_evals, st_pairs);
    atomicAdd(&ctr->brick_groups, st_groups);
  }
#undef PCD_BND
#undef PCD_RL
}

}  // namespace pcd
