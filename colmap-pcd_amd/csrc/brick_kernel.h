// brick_kernel.h -- the LDS-tiled brick kernel of the NN search (included by nn.hip).
//
// One wavefront per work item = (brick of B^3 cells, <= G queries whose home cell lies in it).
// The quad rows (grid.h: 2x2 cells in (y,z), contiguous along x) of the brick grown by R cells are contiguous
// ranges of the cell-sorted cloud -- 9 ranges for B = R = 2; their concatenation is streamed through two
// 256-point LDS tiles per wavefront:
//   * slot -> source address: the concatenation is addressed in groups of 4 consecutive slots (every range padded to
//     a multiple of 4 records), lane l of tile t owns group 64 t + l and finds its range with one v_cmp + v_cndmask
//     per range start (starts wave-uniform in SGPRs, source - start deltas in VGPRs); the tile's four DMA
//     instructions are that one address + 0 / 16 / 32 / 48 bytes of immediate offset;
//   * staging is LDS-DMA (global_load_lds_dwordx4: one 16-B record per lane, no VGPR round trip);
//     the 4 DMAs of tile t+1 are in flight while tile t is compared (counted s_waitcnt vmcnt(4));
//   * compare: lanes = staged points (ds_read_b128), the G queries are wave-uniform (copied from SGPRs to VGPRs once
//     per item: an SGPR operand halves the VALU rate on gfx950), every point is tested against every query with
//     FLANN's float arithmetic, per-lane running minima of the packed (distance, index) keys (one v_min_f64 each);
//   * one transposed butterfly reduces all G per-lane minima at once (reduce-scatter over
//     xor 32/16/8, then xor 4/2/1), instead of G separate wavefront reductions;
//   * work items are software-pipelined: the item record of group k+2 and the query / row-range
//     loads of group k+1 are issued before group k is processed.
// A query is final when best < (distance to the staged region's boundary)^2 (nn.hip header); the
// others go to the exact fallback with their tentative key as starting bound.
#pragma once
#include <type_traits>

namespace pcd {

// build-time knobs (tools/nn_tune.sh sweeps them): points per LDS tile buffer (two buffers per
// wavefront) and the wavefronts per SIMD the register allocator must leave room for
#ifndef PCD_KTILE
#define PCD_KTILE 256
#endif
#ifndef PCD_BRICK_MINWAVES
#define PCD_BRICK_MINWAVES 4
#endif
constexpr int kTile = PCD_KTILE;
constexpr int kFbChunk = 64;   // fallback-list slots a wavefront reserves per atomic (>= queries per item)
static_assert(kTile == 256, "a tile is 4 DMA instructions (128/192-point tiles were measured slower and removed)");
// timing-only ablations (results are then wrong): compiled in only with -DPCD_ABLATE (tools/nn_ablate.py builds
// such a variant); in the shipped library the masks are 0 and every `flags & kAblate*` folds away.
#ifdef PCD_ABLATE
constexpr int kAblateCompare = 0x100, kAblateReduce = 0x400, kAblateFallback = 0x800, kAblateTiles = 0x1000,
              kAblateQuarter = 0x2000, kAblateNoDma = 0x4000, kAblateNoLdsRead = 0x8000;
#else
constexpr int kAblateCompare = 0, kAblateReduce = 0, kAblateFallback = 0, kAblateTiles = 0, kAblateQuarter = 0,
              kAblateNoDma = 0, kAblateNoLdsRead = 0;
#endif

__device__ __forceinline__ void lds_dma16(const float4* gsrc, float4* lds_wave_base) {
  // LDS destination = wave-uniform base + lane * 16 (hardware adds the lane offset)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same with a compile-time byte offset on the source address (instruction k of a 4-slot group)
template <int OFF>
__device__ __forceinline__ void lds_dma16_off(const float4* gsrc, float4* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  const uint32_t lo = __shfl_xor((uint32_t)v, m), hi = __shfl_xor((uint32_t)(v >> 32), m);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t min_u64(uint64_t a, uint64_t b) { return b < a ? b : a; }

// All-lanes minimum of 8 per-lane packed keys at once (the keys as f64, see compare_point: v_min_f64 on them
// is the lexicographic (distance, index) minimum; none of them is a NaN pattern here).  Returns, in every lane,
// the minimum of value index ((lane>>5)&1)*4 + ((lane>>4)&1)*2 + ((lane>>3)&1).
// A reduce-scatter: each stage halves the number of values a lane still carries.  The first two stages are the
// gfx950 lane-swap instructions: v_permlane32_swap exchanges lanes 32..63 of one register with lanes 0..31 of
// another, so after swapping the registers of values i and 4+i the lane-wise minimum of the pair IS "value i over
// both halves" in lanes 0..31 and "value 4+i over both halves" in lanes 32..63 -- 2 swaps + 1 v_min_f64 per pair
// where select + ds_bpermute + 64-bit compare/select took 13 instructions; v_permlane16_swap does the same for
// the 16-lane rows.  The remaining 8-lane groups are folded with DPP moves (row_ror:8 with a select, then
// row_half_mirror, quad_perm xor 2, quad_perm xor 1).
__device__ __forceinline__ double min_key_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = __builtin_amdgcn_update_dpp(0u, (uint32_t)u, CTRL, 0xf, 0xf, false);
  const uint32_t hi = __builtin_amdgcn_update_dpp(0u, (uint32_t)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
template <int ROW>  // 32 or 16: min over the swapped pair (x keeps the even rows' value, y the odd rows')
__device__ __forceinline__ double swap_min_f64(double x, double y) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const uint64_t ux = __builtin_bit_cast(uint64_t, x), uy = __builtin_bit_cast(uint64_t, y);
  u32x2 lo, hi;
  if (ROW == 32) {
    lo = __builtin_amdgcn_permlane32_swap((uint32_t)ux, (uint32_t)uy, false, false);
    hi = __builtin_amdgcn_permlane32_swap((uint32_t)(ux >> 32), (uint32_t)(uy >> 32), false, false);
  } else {
    lo = __builtin_amdgcn_permlane16_swap((uint32_t)ux, (uint32_t)uy, false, false);
    hi = __builtin_amdgcn_permlane16_swap((uint32_t)(ux >> 32), (uint32_t)(uy >> 32), false, false);
  }
  return min_key_f64(__builtin_bit_cast(double, ((uint64_t)hi.x << 32) | lo.x),
                     __builtin_bit_cast(double, ((uint64_t)hi.y << 32) | lo.y));
}
__device__ __forceinline__ double wave_min8_key(const double (&v)[8]) {
  const int lane = threadIdx.x & 63;
  double a[4], b2[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = swap_min_f64<32>(v[i], v[4 + i]);   // lanes 0..31: value i, 32..63: value 4+i
#pragma unroll
  for (int i = 0; i < 2; ++i) b2[i] = swap_min_f64<16>(a[i], a[2 + i]);  // even rows: a[i], odd rows: a[2+i]
  const bool up = lane & 8;
  const double keep = up ? b2[1] : b2[0], send = up ? b2[0] : b2[1];
  double c = min_key_f64(keep, dpp_f64<0x128>(send));   // row_ror:8 = lane ^ 8 inside a row of 16
  c = min_key_f64(c, dpp_f64<0x141>(c));                // row_half_mirror: lane <- 7 - lane inside each 8
  c = min_key_f64(c, dpp_f64<0x4E>(c));                 // quad_perm [2,3,0,1]
  c = min_key_f64(c, dpp_f64<0xB1>(c));                 // quad_perm [1,0,3,2]
  return c;
}

// One staged point against NQ wave-uniform queries: 8 scalar f32 ops (FLANN's ((dx*dx) + dy*dy) + dz*dz, no FMA)
// + ONE v_min_f64 on the packed (distance, index) key per query.  The key float_bits(d) << 32 | index of a
// non-negative, non-NaN float d is a positive finite (or denormal: d == 0) double whose numeric order is the
// unsigned order of its bits, so the f64 minimum is exactly the lexicographic (distance, index) minimum that
// v_cmp_lt_u64 + 2 v_cndmask computed before -- in one half-rate instruction instead of three (measured on
// gfx950, tools/ubench/valu_rate.hip: v_cmp_lt_u64 4.2, v_min_f64 4.2, v_mul/v_fma_f32 2.4 cycles per wave64
// instruction).  Inline asm: llvm's fmin would canonicalise both operands first (IEEE mode), two more f64 ops.
// f64 denormals are never flushed on gfx9 compute kernels (only the f32 mode is configurable).
// Measured alternatives that did NOT help: packed v_pk_add/mul_f32 (half rate on gfx950), 32-bit lexicographic
// compares instead of the 64-bit one, more wavefronts per SIMD (5, 6, 8: within 3 %).
// Hand-scheduled form.  hipcc builds each 64-bit key with an extra v_mov (the index into the low half of a fresh
// register pair) and folds the wave-uniform queries into the subtractions as SGPR operands, which run at HALF rate
// on gfx950 (tools/ubench/valu_rate2.hip: v_sub_f32 v,v 1.03 ns, s,v 1.75 ns per wave64 instruction); so the
// queries are copied to VGPRs once per item and one staged point is compared with 4 queries per asm block:
// per (point, query) pair 3 v_sub + 3 v_mul + 2 v_add + 1 v_min_f64 = 9 VALU instructions, plus one v_mov per
// block that parks the point's index in the low half of the key pair v[120:121]; the last v_add writes the
// distance straight into its high half.  Two queries are interleaved so that no instruction waits for its
// predecessor.  Same IEEE operations in the same order as l2_simple3 (grid.h): results are bit-identical.
// v118..v127 are scratch of the block (clobbers): the kernel stays within 128 VGPRs = 4 wavefronts per SIMD.
#define PCD_CMP_PAIR(A, B)                                                                         \
  "v_sub_f32 v122, %[qx" #A "], %[px]\n\tv_sub_f32 v125, %[qx" #B "], %[px]\n\t"                       \
  "v_sub_f32 v123, %[qy" #A "], %[py]\n\tv_sub_f32 v126, %[qy" #B "], %[py]\n\t"                       \
  "v_sub_f32 v124, %[qz" #A "], %[pz]\n\tv_sub_f32 v127, %[qz" #B "], %[pz]\n\t"                       \
  "v_mul_f32 v122, v122, v122\n\tv_mul_f32 v125, v125, v125\n\t"                                     \
  "v_mul_f32 v123, v123, v123\n\tv_mul_f32 v126, v126, v126\n\t"                                     \
  "v_mul_f32 v124, v124, v124\n\tv_mul_f32 v127, v127, v127\n\t"                                     \
  "v_add_f32 v122, v122, v123\n\tv_add_f32 v125, v125, v126\n\t"                                     \
  "v_add_f32 v121, v122, v124\n\tv_add_f32 v119, v125, v127\n\t"                                     \
  "v_min_f64 %[b" #A "], %[b" #A "], v[120:121]\n\tv_min_f64 %[b" #B "], %[b" #B "], v[118:119]\n\t"

// one staged point against queries Q0..Q0+3 (whose coordinates are in VGPRs)
__device__ __forceinline__ void compare_point4(const f32x4 p, const float* qx, const float* qy, const float* qz,
                                               double* best) {
  asm("v_mov_b32 v120, %[pw]\n\tv_mov_b32 v118, %[pw]\n\t"
      PCD_CMP_PAIR(0, 1) PCD_CMP_PAIR(2, 3)
      : [b0] "+v"(best[0]), [b1] "+v"(best[1]), [b2] "+v"(best[2]), [b3] "+v"(best[3])
      : [px] "v"(p.x), [py] "v"(p.y), [pz] "v"(p.z), [pw] "v"(p.w),
        [qx0] "v"(qx[0]), [qy0] "v"(qy[0]), [qz0] "v"(qz[0]), [qx1] "v"(qx[1]), [qy1] "v"(qy[1]), [qz1] "v"(qz[1]),
        [qx2] "v"(qx[2]), [qy2] "v"(qy[2]), [qz2] "v"(qz[2]), [qx3] "v"(qx[3]), [qy3] "v"(qy[3]), [qz3] "v"(qz[3])
      : "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}
__device__ __forceinline__ void compare_point2(const f32x4 p, const float* qx, const float* qy, const float* qz,
                                               double* best) {
  asm("v_mov_b32 v120, %[pw]\n\tv_mov_b32 v118, %[pw]\n\t"
      PCD_CMP_PAIR(0, 1)
      : [b0] "+v"(best[0]), [b1] "+v"(best[1])
      : [px] "v"(p.x), [py] "v"(p.y), [pz] "v"(p.z), [pw] "v"(p.w),
        [qx0] "v"(qx[0]), [qy0] "v"(qy[0]), [qz0] "v"(qz[0]), [qx1] "v"(qx[1]), [qy1] "v"(qy[1]), [qz1] "v"(qz[1])
      : "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

// ONE query against TWO staged points (the odd query of a group): the two points' chains interleave the way the two
// queries of PCD_CMP_PAIR do.
__device__ __forceinline__ void compare_query1(const f32x4 pa, const f32x4 pb, const float qx, const float qy,
                                               const float qz, double& best) {
  asm("v_mov_b32 v120, %[aw]\n\tv_mov_b32 v118, %[bw]\n\t"
      "v_sub_f32 v122, %[qx], %[ax]\n\tv_sub_f32 v125, %[qx], %[bx]\n\t"
      "v_sub_f32 v123, %[qy], %[ay]\n\tv_sub_f32 v126, %[qy], %[by]\n\t"
      "v_sub_f32 v124, %[qz], %[az]\n\tv_sub_f32 v127, %[qz], %[bz]\n\t"
      "v_mul_f32 v122, v122, v122\n\tv_mul_f32 v125, v125, v125\n\t"
      "v_mul_f32 v123, v123, v123\n\tv_mul_f32 v126, v126, v126\n\t"
      "v_mul_f32 v124, v124, v124\n\tv_mul_f32 v127, v127, v127\n\t"
      "v_add_f32 v122, v122, v123\n\tv_add_f32 v125, v125, v126\n\t"
      "v_add_f32 v121, v122, v124\n\tv_add_f32 v119, v125, v127\n\t"
      "v_min_f64 %[b], %[b], v[120:121]\n\tv_min_f64 %[b], %[b], v[118:119]\n\t"
      : [b] "+v"(best)
      : [ax] "v"(pa.x), [ay] "v"(pa.y), [az] "v"(pa.z), [aw] "v"(pa.w), [bx] "v"(pb.x), [by] "v"(pb.y), [bz] "v"(pb.z),
        [bw] "v"(pb.w), [qx] "v"(qx), [qy] "v"(qy), [qz] "v"(qz)
      : "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

// the lane's 4 staged points of a tile against the first NQ queries of the group: pairs of queries per point, the
// odd query (if any) against pairs of points -- exactly NQ compares per point, no padded query slots
template <int NQ>
__device__ __forceinline__ void compare_tile(const f32x4 (&p)[4], const float (&qx)[8], const float (&qy)[8],
                                             const float (&qz)[8], double (&best)[8]) {
  static_assert(NQ >= 1 && NQ <= 8, "1..8 queries per group");
  constexpr int E = NQ & ~1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (E >= 4) compare_point4(p[k], qx, qy, qz, best);
    if (E == 8) compare_point4(p[k], qx + 4, qy + 4, qz + 4, best + 4);
    if (E == 2 || E == 6) compare_point2(p[k], qx + (E - 2), qy + (E - 2), qz + (E - 2), best + (E - 2));
  }
  if (NQ & 1) {
    compare_query1(p[0], p[1], qx[NQ - 1], qy[NQ - 1], qz[NQ - 1], best[NQ - 1]);
    compare_query1(p[2], p[3], qx[NQ - 1], qy[NQ - 1], qz[NQ - 1], best[NQ - 1]);
  }
}

struct BrickMeta {   // per-group loads issued one group ahead
  float4 q;          // lane < cnt: query (x,y,z, bits(query id))
  uint64_t prior;    // lane < cnt: the key the query came with (kKeyInit, or another shard's result when refining)
  uint32_t s, e;     // lane < nrows: point range of the lane's cell row
};

// item record: {first query, brick x, brick y, brick z | count << 28}
__device__ __forceinline__ int item_count(const uint4 it) { return (int)(it.w >> 28); }

__device__ __forceinline__ void brick_region(const GridParams& g, const BrickParams& b, const uint4 it, int c0[3],
                                             int c1[3]) {
  const int bx = (int)it.y, by = (int)it.z, bz = (int)(it.w & 0x0FFFFFFFu);
  c0[0] = max(bx * b.Bx - b.R, 0); c0[1] = max(by * b.B - b.R, 0); c0[2] = max(bz * b.B - b.R, 0);
  c1[0] = min(bx * b.Bx + b.Bx + b.R, g.dims[0]); c1[1] = min(by * b.B + b.B + b.R, g.dims[1]);
  c1[2] = min(bz * b.B + b.B + b.R, g.dims[2]);
}

// Lane r < nrows gets the point range of quad row r of the region: the cells [c0,c1) of brick_region, grown in
// y and z to whole quads (a superset; proven_bound keeps using the un-grown box, which is conservative).
__device__ __forceinline__ BrickMeta brick_load_meta(const GridParams& g, const BrickParams& b, const uint4 it,
                                                     const float4* __restrict__ qsorted,
                                                     const uint64_t* __restrict__ ksorted,
                                                     const uint32_t* __restrict__ cell_start) {
  const int lane = threadIdx.x & 63;
  BrickMeta m;
  // Every lane issues every load (indices clamped, results masked afterwards): a load under a
  // divergent `if` may be branched around, and then hipcc can no longer count the loads in flight and
  // falls back to s_waitcnt vmcnt(0) at the first use -- which would serialise the prefetch.
  const int cnt = item_count(it);  // >= 1
  m.q = qsorted[it.x + (lane < cnt ? lane : cnt - 1)];
  m.prior = ksorted[it.x + (lane < cnt ? lane : cnt - 1)];   // brick-sorted copy: no dependent gather at the item's end
  int c0[3], c1[3];
  brick_region(g, b, it, c0, c1);
  const int yq0 = c0[1] >> 1, zq0 = c0[2] >> 1;
  const int ny = ((c1[1] + 1) >> 1) - yq0, nrows = ny * (((c1[2] + 1) >> 1) - zq0);  // 1 <= ny <= 8, nrows <= 64
  const int row = lane < nrows ? lane : nrows - 1;
  // row / ny without an integer division: ceil(2^16 / ny) is exact for row < 64, ny <= 8
  const uint32_t inv = ny == 1 ? 65536u : ny == 2 ? 32768u : ny == 3 ? 21846u : ny == 4 ? 16384u
                     : ny == 5 ? 13108u : ny == 6 ? 10923u : ny == 7 ? 9363u : 8192u;
  const int rz = (int)(((uint32_t)row * inv) >> 16), ry = row - rz * ny;
  const uint64_t rowbase = quad_row_base(g, yq0 + ry, zq0 + rz);
  m.s = cell_start[rowbase + 4 * c0[0]];
  m.e = cell_start[rowbase + 4 * c1[0]];
  if (lane >= nrows) m.e = m.s;  // empty row
  return m;
}

template <int G>
__global__ __launch_bounds__(256, PCD_BRICK_MINWAVES) void k_nn_brick(GridParams g, BrickParams b, const float4* __restrict__ sorted,
                                                  const uint32_t* __restrict__ cell_start,
                                                  const float4* __restrict__ qsorted,
                                                  const uint64_t* __restrict__ ksorted,
                                                  const uint4* __restrict__ items, NnCounters* __restrict__ ctr,
                                                  uint64_t* __restrict__ keys, uint32_t* __restrict__ fb_list,
                                                  uint32_t* __restrict__ fb_count, int flags) {
  // flags: bit 0 = collect statistics (bits 8.. = ablations, only in -DPCD_ABLATE builds)
  const int collect_stats = flags & 1;
  static_assert(G == 8, "the transposed reduction is written for 8 queries per group");
  __shared__ __attribute__((aligned(16))) float4 s_tile[4][2][kTile];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform for the compiler too: LDS tile addresses and M0 values stay in SGPRs
  const uint32_t nitems = ctr->nitems;
  const uint32_t nwaves = gridDim.x * 4;
  unsigned long long st_staged = 0, st_pairs = 0, st_groups = 0;
  uint32_t fb_base = 0, fb_left = 0;   // this wavefront's current chunk of the fallback list

  // XCD-aware work split: blocks b, b+8, b+16, ... share an XCD (and its 4 MiB L2), so each of the 8
  // block classes walks its own contiguous eighth of the item list (items are in brick order, x fastest):
  // bricks that are neighbours in space -- and share most of their staged rows -- meet in one L2.
  // (speed only: any mapping gives the same results.)
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
  // item records through uniform (scalar) indices: they land in SGPRs instead of 12 VGPRs
#define PCD_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
  uint4 it0 = items[PCD_UNI(item)];
  BrickMeta m0 = brick_load_meta(g, b, it0, qsorted, ksorted, cell_start);
  uint4 it1 = items[PCD_UNI(min(item + stride, item_end - 1))];

  for (; item < item_end; item += stride) {
    // ---- prefetch: metadata of the next group, item record of the one after ----
    // (unconditional, clamped to the last item: see brick_load_meta)
    const BrickMeta m1 = brick_load_meta(g, b, it1, qsorted, ksorted, cell_start);
    const uint4 it2 = items[PCD_UNI(min(item + 2 * stride, item_end - 1))];

    // ---- current group ----
    const uint32_t cnt = (uint32_t)item_count(it0);
    float qx[G], qy[G], qz[G];   // wave-uniform, but held in VGPRs: SGPR operands halve the VALU rate (compare_point)
#pragma unroll
    for (int k = 0; k < G; ++k) {
      // slots >= cnt stay unset: the compare variant of this group size never reads them (wave-uniform skip)
      if (k == 0 || k < (int)cnt) {
        asm volatile("v_mov_b32 %0, %1" : "=v"(qx[k]) : "s"(readlane_f(m0.q.x, k)));
        asm volatile("v_mov_b32 %0, %1" : "=v"(qy[k]) : "s"(readlane_f(m0.q.y, k)));
        asm volatile("v_mov_b32 %0, %1" : "=v"(qz[k]) : "s"(readlane_f(m0.q.z, k)));
      } else {
        qx[k] = qy[k] = qz[k] = 0.f;
      }
    }
    int c0[3], c1[3];
    brick_region(g, b, it0, c0, c1);
    // ---- slot -> source index ------------------------------------------------------------------
    // The concatenation of the region's ranges is addressed in GROUPS of 4 consecutive slots: every range is
    // padded to a multiple of 4 slots (the <= 3 padding slots read the records that follow the range in memory:
    // real cloud points of the neighbouring cells, harmless extra candidates; the `sorted` buffer ends with 4
    // spare records), so the 4 slots of a group are 4 consecutive records.  Lane l of tile t takes group
    // t * 64 + l: ONE source address per lane per tile, and the tile's 4 DMA instructions are that address +
    // 0 / 16 / 32 / 48 bytes (instruction k fills LDS slot k * 64 + l -- a permutation of the tile, and the
    // compare does not care about the order).  The range of a group is found with one compare + select per
    // range start (the starts are wave-uniform): no per-window tables, no v_readlane inside the tile loop,
    // any number of range starts inside a tile.  Round 1-2a computed an address per 64-slot DMA window (three
    // v_readlane + ~10 VALU each, plus a generic path for windows with two starts); timing-only ablations put
    // that address generation at 0.34 ms of the kernel's 0.84.
    uint32_t T;
    const uint32_t len = m0.e - m0.s;
    const uint32_t len4 = (len + 3u) & ~3u;
    // lane r: start of range r in the padded concatenation (lanes >= nrows: T) and source - start
    const uint32_t off = wave_excl_scan_u32(len4, T);
    const uint32_t delta = m0.s - off;
    const int yq0 = c0[1] >> 1, zq0 = c0[2] >> 1;
    const int nrows = (((c1[1] + 1) >> 1) - yq0) * (((c1[2] + 1) >> 1) - zq0);   // wave-uniform, <= 64

    double best[G];   // packed keys, minimised as doubles (compare_point)
#pragma unroll
    for (int k = 0; k < G; ++k) best[k] = __builtin_bit_cast(double, kKeyInit);

    if (T > 0 && !(flags & kAblateTiles)) {
      const int ntiles = (int)((T + kTile - 1) / kTile);
      const char* __restrict__ src_bytes = reinterpret_cast<const char*>(sorted);
      // issue the 4 DMAs of tile t (always exactly 4 instructions: groups past T re-read the last group, which
      // cannot change a minimum -- the compare needs no tail mask)
      // range starts / deltas of the first 9 ranges (all of them for B = R = 2) once per item, wave-uniform;
      // lanes >= nrows hold start = T, which no group reaches.  Scalars, not arrays: captured arrays stay in
      // memory and hipcc then turns the select chain into an indexed scratch load inside the tile loop.
      // starts stay in SGPRs (a VOPC compare may read one), deltas are copied to VGPRs (v_cndmask cannot read an
      // SGPR next to VCC on gfx9): 8 live VGPRs per item instead of 12 v_readlane + hazards per tile
#define PCD_RL(v, r) ((uint32_t)__builtin_amdgcn_readlane((int)(v), r))
      const uint32_t o1 = PCD_RL(off, 1), o2 = PCD_RL(off, 2), o3 = PCD_RL(off, 3), o4 = PCD_RL(off, 4),
                     o5 = PCD_RL(off, 5), o6 = PCD_RL(off, 6), o7 = PCD_RL(off, 7), o8 = PCD_RL(off, 8);
      uint32_t d0, d1, d2, d3, d4, d5, d6, d7, d8;
#define PCD_VB(dst, r) asm volatile("v_mov_b32 %0, %1" : "=v"(dst) : "s"(PCD_RL(delta, r)))
      PCD_VB(d0, 0); PCD_VB(d1, 1); PCD_VB(d2, 2); PCD_VB(d3, 3); PCD_VB(d4, 4); PCD_VB(d5, 5); PCD_VB(d6, 6);
      PCD_VB(d7, 7); PCD_VB(d8, 8);
#undef PCD_VB
#undef PCD_RL
      auto issue_tile = [&](int t) {
        float4* buf = s_tile[wave][t & 1];
        const uint32_t s4 = min((uint32_t)t * kTile + 4u * (uint32_t)lane, T - 4u);
        uint32_t dl = d0;   // empty ranges share their start with the next one: the last one wins
#define PCD_SEL(o, d) asm("v_cmp_le_u32_e32 vcc, %2, %1\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(dl) : "v"(s4), "s"(o), "v"(d) : "vcc")
        PCD_SEL(o1, d1); PCD_SEL(o2, d2); PCD_SEL(o3, d3); PCD_SEL(o4, d4);
        PCD_SEL(o5, d5); PCD_SEL(o6, d6); PCD_SEL(o7, d7); PCD_SEL(o8, d8);
#undef PCD_SEL
        for (int r = 9; r < nrows; ++r) {   // other brick / halo settings have more rows
          const uint32_t o_r = (uint32_t)__builtin_amdgcn_readlane((int)off, r);
          const uint32_t d_r = (uint32_t)__builtin_amdgcn_readlane((int)delta, r);
          dl = s4 >= o_r ? d_r : dl;
        }
        // uniform base + byte offset (64-bit: a cloud may exceed 2^28 points = 4 GiB of records)
        const float4* gp = reinterpret_cast<const float4*>(src_bytes + ((uint64_t)(s4 + dl) << 4));
        if (!(flags & kAblateNoDma)) {
          // the instruction offset is added to the LDS address as well as to the source address
          // (LDS address = M0 base + instruction offset + lane * 16): take it off the base again
          lds_dma16_off<0>(gp, buf);
          lds_dma16_off<16>(gp, buf + 64 - 1);
          lds_dma16_off<32>(gp, buf + 128 - 2);
          lds_dma16_off<48>(gp, buf + 192 - 3);
        } else {
          asm volatile("" ::"v"(gp));
        }
      };
      issue_tile(0);
      for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) {
          issue_tile(t + 1);
          // tile t landed, the 4 DMAs of tile t+1 still in flight
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_wave_barrier();
        // The tile is read with inline-asm ds_read_b128: for an ordinary LDS load hipcc would insert
        // s_waitcnt vmcnt(0) (it cannot tell the two buffers apart) and drain tile t+1's DMAs.
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
        // one compare variant per group size (cnt is wave-uniform): no padded query slots
        if (flags & kAblateCompare) {
#pragma unroll
          for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(p[k]));
        } else if (flags & kAblateQuarter) {   // a quarter of the compare work, everything else unchanged
          compare_point4(p[0], qx, qy, qz, best);
          compare_point4(p[0], qx + 4, qy + 4, qz + 4, best + 4);
#pragma unroll
          for (int k = 1; k < 4; ++k) asm volatile("" ::"v"(p[k]));
        } else {
          switch (cnt) {   // wave-uniform
            case 1: compare_tile<1>(p, qx, qy, qz, best); break;
            case 2: compare_tile<2>(p, qx, qy, qz, best); break;
            case 3: compare_tile<3>(p, qx, qy, qz, best); break;
            case 4: compare_tile<4>(p, qx, qy, qz, best); break;
            case 5: compare_tile<5>(p, qx, qy, qz, best); break;
            case 6: compare_tile<6>(p, qx, qy, qz, best); break;
            case 7: compare_tile<7>(p, qx, qy, qz, best); break;
            default: compare_tile<8>(p, qx, qy, qz, best); break;
          }
        }
        // (the reads of this buffer have returned -- waited inside the asm block -- before tile t+2's DMAs)
      }
    }
    // ---- one transposed reduction for the 8 queries; lane k fetches result k ----
    const uint64_t red = __builtin_bit_cast(uint64_t, (flags & kAblateReduce) ? best[0] : wave_min8_key(best));
    // value index v sits in lanes with bits (5,4,3) = v  ->  lane 8*bitrev... v = b5*4 + b4*2 + b3
    const int holder = ((lane & 4) ? 32 : 0) | ((lane & 2) ? 16 : 0) | ((lane & 1) ? 8 : 0);
    uint64_t mine = ((uint64_t)__shfl((uint32_t)(red >> 32), holder) << 32) | __shfl((uint32_t)red, holder);
    bool unproven = false;
    if (lane < (int)cnt) {
      const uint32_t my_qi = __float_as_uint(m0.q.w);
      // the key the query came with: kKeyInit for a plain query, another shard's result for pcd_nn_refine_device
      mine = min_u64(mine, m0.prior);
      const double bound = proven_bound(g, m0.q.x, m0.q.y, m0.q.z, c0, c1);
      const double bd = (double)__uint_as_float((uint32_t)(mine >> 32));
      unproven = !(bd < bound) && !(flags & kAblateFallback);
      keys[my_qi] = unproven ? mine : finalized_key(mine);  // final, or the starting bound of the fallback
    }
    const unsigned long long um = __ballot(unproven);
    if (um) {
      // the wavefront appends to the fallback list inside chunks of kFbChunk slots it reserves with ONE returning
      // atomic each (an atomic per item exposed its latency on every fourth item); the slots of a chunk it does not
      // use get 0xFFFFFFFF when the chunk is left (here and at the end of the kernel): the list needs no memset;
      // k_fb_compact squeezes the sentinels out
      const uint32_t k = (uint32_t)__popcll(um);
      if (k > fb_left) {   // wave-uniform
        if (lane < (int)fb_left) fb_list[fb_base + lane] = 0xFFFFFFFFu;
        uint32_t nb = 0;
        if (lane == 0) nb = atomicAdd(fb_count, (uint32_t)kFbChunk);
        fb_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
        fb_left = kFbChunk;
      }
      if (unproven) fb_list[fb_base + __popcll(um & ((1ull << lane) - 1))] = __float_as_uint(m0.q.w);
      fb_base += k;
      fb_left -= k;
    }
    if (collect_stats) { st_staged += T; st_pairs += (unsigned long long)T * cnt; st_groups += 1; }
    it0 = it1; it1 = it2; m0 = m1;
  }
  if (lane < (int)fb_left) fb_list[fb_base + lane] = 0xFFFFFFFFu;   // rest of the last chunk
  if (collect_stats && lane == 0) {
    atomicAdd(&ctr->staged_points, st_staged);
    atomicAdd(&ctr->pair_evals, st_pairs);
    atomicAdd(&ctr->brick_groups, st_groups);
  }
}

}  // namespace pcd
