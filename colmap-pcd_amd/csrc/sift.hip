// sift.hip -- exact brute-force SIFT descriptor matching on the matrix cores.
//
// Replaces feature/sift.cc:171-204 (ComputeSiftDistanceMatrix: int32 dot products of uint8 x 128
// descriptors), :55-107 (FindBestMatchesOneWayBruteForce: best / second best per row, acos + distance
// and ratio tests) and :109-144 (FindBestMatchesBruteForce: cross check) -- the exact specification that
// SiftGPU's MultiplyDescriptor / RowMatch / ColMatch kernels (lib/SiftGPU/ProgramCU.cu:1408-1795) and the
// CPU brute-force matcher implement.  This is the one dense contraction of the path, hence MFMA:
//
//   S[i][j] = sum_k a[i][k] b[j][k]   (uint8 inputs, exact int32)
//           = sum_k (a-128)(b-128) + 128 (sum_k a[i][k] + sum_k b[j][k]) - 128*128*128
// so the bytes are re-centred to int8 (a ^ 0x80) and fed to v_mfma_i32_32x32x32_i8 (K = 32 per
// instruction, int32 accumulate: exact), the row-sum correction is added in the epilogue.
//
// k_sift_scores: one 128 x 128 score tile per workgroup (4 wavefronts, 64 x 64 each).  Both sets' tiles
// (128 rows x 128 B) are staged once in LDS (144-B row pitch: conflict-free ds_read_b128 fragments).  The
// tile is computed TWICE from the same fragments, as A.B^T and as B.A^T: in the MFMA result layout a lane
// owns one column and 16 rows per 32x32 block, so the best / second-best scan over the *other* set is a
// register-local loop in both directions (no cross-lane top-2 reductions, no score matrix in memory) --
// 32 extra MFMAs per wavefront are cheaper than 64 shuffled reductions.  Per tile and direction one
// (best, second, argbest) triple per descriptor goes to a partial buffer; k_sift_finalize merges the
// partials in ascending tile order (ties -> first index, as the reference's ascending strict-> scan),
// applies acos / max_distance / max_ratio; the cross check and the ordered compaction follow.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace pcd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kSiftTile = 128;
constexpr int kSiftPitch = 144;   // bytes per staged descriptor row (128 + 16 pad)
constexpr int kSiftConst = 128 * 128 * 128;

__global__ void k_sift_rowsum(const uint8_t* __restrict__ da, int na, int* __restrict__ suma,
                              const uint8_t* __restrict__ db, int nb, int* __restrict__ sumb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= na + nb) return;
  const uint8_t* d = i < na ? da : db;
  int* sum = i < na ? suma : sumb;
  if (i >= na) i -= na;
  const uint4* p = reinterpret_cast<const uint4*>(d + (size_t)i * 128);
  unsigned s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint4 v = p[c];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (w[k] & 0xFF) + ((w[k] >> 8) & 0xFF) + ((w[k] >> 16) & 0xFF) + (w[k] >> 24);
  }
  sum[i] = (int)s;
}

// running (best, second, argbest) with the reference's update rule: strictly greater replaces the best
__device__ __forceinline__ void top2_update(int v, int idx, int& best, int& second, int& arg) {
  second = max(second, min(v, best));
  const bool gt = v > best;
  arg = gt ? idx : arg;
  best = max(best, v);
}
// merge another triple; on equal best the lower index wins (= first in an ascending scan)
__device__ __forceinline__ void top2_merge(int b2, int s2, int a2, int& best, int& second, int& arg) {
  const int nsecond = max(max(second, s2), min(best, b2));
  const bool take = (b2 > best) || (b2 == best && (unsigned)a2 < (unsigned)arg);
  arg = take ? a2 : arg;
  best = max(best, b2);
  second = nsecond;
}

// part12 [n1][nbx] / part21 [n2][nby] int4 {best, second, arg, 0}
__global__ __launch_bounds__(256) void k_sift_scores(const uint8_t* __restrict__ d1, int n1, const uint8_t* __restrict__ d2,
                                                     int n2, const int* __restrict__ sum1, const int* __restrict__ sum2,
                                                     int4* __restrict__ part12, int4* __restrict__ part21, int nbx,
                                                     int nby) {
  __shared__ __attribute__((aligned(16))) uint8_t sA[kSiftTile * kSiftPitch];
  __shared__ __attribute__((aligned(16))) uint8_t sB[kSiftTile * kSiftPitch];
  __shared__ int sSumA[kSiftTile], sSumB[kSiftTile];
  __shared__ int2 sMerge[2][2][64];   // [direction][64-column half][column]: packed (best, second) of the upper row half
  const int bx = blockIdx.x, by = blockIdx.y;           // bx: tile of set 2 (columns), by: tile of set 1 (rows)
  const int row0 = by * kSiftTile, col0 = bx * kSiftTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // ---- stage both tiles: 1024 16-B chunks each, re-centred to int8 ----
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = tid + it * 256, r = c >> 3, q = c & 7;
    uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
    if (row0 + r < n1) va = *reinterpret_cast<const uint4*>(d1 + (size_t)(row0 + r) * 128 + q * 16);
    if (col0 + r < n2) vb = *reinterpret_cast<const uint4*>(d2 + (size_t)(col0 + r) * 128 + q * 16);
    va.x ^= 0x80808080u; va.y ^= 0x80808080u; va.z ^= 0x80808080u; va.w ^= 0x80808080u;
    vb.x ^= 0x80808080u; vb.y ^= 0x80808080u; vb.z ^= 0x80808080u; vb.w ^= 0x80808080u;
    *reinterpret_cast<uint4*>(sA + r * kSiftPitch + q * 16) = va;
    *reinterpret_cast<uint4*>(sB + r * kSiftPitch + q * 16) = vb;
  }
  if (tid < kSiftTile) {
    sSumA[tid] = row0 + tid < n1 ? sum1[row0 + tid] : 0;
    sSumB[tid] = col0 + tid < n2 ? sum2[col0 + tid] : 0;
  }
  __syncthreads();

  // ---- 64 x 64 per wavefront, both orientations from the same fragments ----
  // The accumulators start at 128 * rowsum(row) (exact int32), so the scan below needs no per-element add.
  v16i acc1[2][2], acc2[2][2];   // acc1[mt][nt]: rows = set-1, cols = set-2;  acc2[nt][mt]: rows = set-2, cols = set-1
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int rr = a * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
      const int ra = 128 * sSumA[wr * 64 + rr], rb = 128 * sSumB[wc * 64 + rr];
#pragma unroll
      for (int c = 0; c < 2; ++c) { acc1[a][c][k] = ra; acc2[a][c][k] = rb; }
    }
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    v4i fa[2], fb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fa[t] = *reinterpret_cast<const v4i*>(sA + (wr * 64 + t * 32 + lr) * kSiftPitch + kk * 32 + lh * 16);
      fb[t] = *reinterpret_cast<const v4i*>(sB + (wc * 64 + t * 32 + lr) * kSiftPitch + kk * 32 + lh * 16);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc1[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[mt], fb[nt], acc1[mt][nt], 0, 0, 0);
        acc2[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[nt], fa[mt], acc2[nt][mt], 0, 0, 0);
      }
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  // Direction 2 -> 1 (for every set-2 descriptor j the best set-1 rows): acc1, lane owns column j.
  // The column constant 128*sum2[j] - 128^3 does not change the order inside a column, so the scan runs on the
  // accumulators as they are and the constant is added back at the end.
  // Packed scan: value = score << 8 | (255 - row in the 128-row tile).  Scores are < 2^23 in magnitude and the
  // code is unique per row and larger for lower rows, so a signed max picks the higher score and, at equal
  // scores, the lower row -- the reference's ascending strict-> scan (sift.cc:72-84) -- and the running second
  // best is the median of (best, second, value): 3 VALU per score (shift-or, max, med3) and no index tracking.
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {               // 32-column tile inside the wavefront's 64 columns
      // dir 0: columns = set 2 (wc, nt = ct), rows = set 1 (wr, mt);  dir 1: columns = set 1 (wr, mt = ct), rows = set 2 (wc, nt)
      const int ccol = (dir == 0 ? wc : wr) * 64 + ct * 32 + lr;            // column inside the 128 tile
      const int cconst = 128 * (dir == 0 ? sSumB[ccol] : sSumA[ccol]) - kSiftConst;
      const int other = dir == 0 ? wr : wc;          // which of the two 64-row halves this wavefront holds
      const int colhalf = dir == 0 ? wc : wr;        // which 64-column half
      const int init = (int)((unsigned)(-cconst) << 8);   // true score 0, code 0 = "no row" (sift.cc:66-68)
      const unsigned code_base = 255u - (unsigned)(other * 64 + 4 * lh);
      int best = init, second = init;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const v16i& acc = dir == 0 ? acc1[rt][ct] : acc2[rt][ct];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const unsigned code = code_base - (unsigned)(rt * 32 + (reg & 3) + 8 * (reg >> 2));
          const int v = (int)(((unsigned)acc[reg] << 8) | code);
          int med;
          asm("v_med3_i32 %0, %1, %2, %3" : "=v"(med) : "v"(best), "v"(second), "v"(v));
          second = med;
          best = max(best, v);
        }
      }
      // the two lane halves hold interleaved rows of the same column
      {
        const int b2 = __shfl_xor(best, 32), s2 = __shfl_xor(second, 32);
        second = max(max(second, s2), min(best, b2));
        best = max(best, b2);
      }
      // merge the two wavefronts that share these columns (dir 0: wr = 0,1; dir 1: wc = 0,1) through LDS
      if (other == 1 && lh == 0) sMerge[dir][colhalf][ct * 32 + lr] = make_int2(best, second);
      __syncthreads();
      if (other == 0 && lh == 0) {
        const int2 o = sMerge[dir][colhalf][ct * 32 + lr];
        second = max(max(second, o.y), min(best, o.x));
        best = max(best, o.x);
        const int bs = (best >> 8) + cconst, ss = (second >> 8) + cconst;      // true scores (>= 0)
        const int arg = bs > 0 ? (dir == 0 ? row0 : col0) + 255 - (best & 255) : -1;   // strict >: a zero score never matches
        const int gcol = (dir == 0 ? col0 : row0) + ccol;
        // partials are [tile][descriptor]: coalesced here and in the merge kernel
        if (dir == 0) { if (gcol < n2) part21[(size_t)by * n2 + gcol] = make_int4(bs, ss, arg, 0); }
        else { if (gcol < n1) part12[(size_t)bx * n1 + gcol] = make_int4(bs, ss, arg, 0); }
      }
      __syncthreads();
    }
  }
}

// ---- persistent row-stripe variant -------------------------------------------------------------------
// Workgroup (by, chunk) keeps the 128-row tile `by` of set 1 in LDS
// and walks `ct` column tiles of set 2, whose next tile is fetched into registers while the current one is
// multiplied and scanned (one __syncthreads per tile).  The best-of-set-2 results of the stripe's rows stay
// in registers across the walk (one partial per chunk); the best-of-set-1 results of a column tile are
// written per 64-row half (no cross-wave merge inside the loop).  part12 [nchunk][n1], part21 [2 nby][n2].
__global__ __launch_bounds__(256, 2) void k_sift_scores_stripe(const uint8_t* __restrict__ d1, int n1,
                                                            const uint8_t* __restrict__ d2, int n2,
                                                            const int* __restrict__ sum1, const int* __restrict__ sum2,
                                                            int4* __restrict__ part12, int4* __restrict__ part21,
                                                            int nbx, int ct_per_chunk) {
  __shared__ __attribute__((aligned(16))) uint8_t sA[kSiftTile * kSiftPitch];
  __shared__ __attribute__((aligned(16))) uint8_t sB[2][kSiftTile * kSiftPitch];
  __shared__ int sSumA[kSiftTile], sSumB[2][kSiftTile];
  __shared__ int4 sMerge[2][64];   // [64-column half of the stripe][column]
  const int chunk = blockIdx.x, by = blockIdx.y;
  const int row0 = by * kSiftTile;
  const int bx0 = chunk * ct_per_chunk, bx1 = min(bx0 + ct_per_chunk, nbx);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  if (bx0 >= bx1) return;

  // ---- stage the A tile and the first B tile (re-centred to int8) ----
  uint4 pre[4];
  int presum = 0;
  auto fetch_b = [&](int bx) {
    const int col0 = bx * kSiftTile;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = tid + it * 256, r = c >> 3, q = c & 7;
      pre[it] = make_uint4(0, 0, 0, 0);
      if (col0 + r < n2) pre[it] = *reinterpret_cast<const uint4*>(d2 + (size_t)(col0 + r) * 128 + q * 16);
    }
    presum = (tid < kSiftTile && col0 + tid < n2) ? sum2[col0 + tid] : 0;
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = tid + it * 256, r = c >> 3, q = c & 7;
      uint4 v = pre[it];
      v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
      *reinterpret_cast<uint4*>(sB[buf] + r * kSiftPitch + q * 16) = v;
    }
    if (tid < kSiftTile) sSumB[buf][tid] = presum;
  };
  fetch_b(bx0);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = tid + it * 256, r = c >> 3, q = c & 7;
    uint4 va = make_uint4(0, 0, 0, 0);
    if (row0 + r < n1) va = *reinterpret_cast<const uint4*>(d1 + (size_t)(row0 + r) * 128 + q * 16);
    va.x ^= 0x80808080u; va.y ^= 0x80808080u; va.z ^= 0x80808080u; va.w ^= 0x80808080u;
    *reinterpret_cast<uint4*>(sA + r * kSiftPitch + q * 16) = va;
  }
  if (tid < kSiftTile) sSumA[tid] = row0 + tid < n1 ? sum1[row0 + tid] : 0;
  store_b(0);
  __syncthreads();

  // running best-of-set-2 for this wave's two 32-column groups of stripe rows (true scores, global index)
  int rb[2] = {0, 0}, rs[2] = {0, 0}, ra[2] = {-1, -1};

  for (int bx = bx0; bx < bx1; ++bx) {
    const int buf = (bx - bx0) & 1;
    const int col0 = bx * kSiftTile;
    if (bx + 1 < bx1) fetch_b(bx + 1);   // lands while this tile is multiplied and scanned

    // accumulators start at 128 * rowsum(row): the scans need no per-element add
    v16i acc1[2][2], acc2[2][2];   // acc1[mt][nt]: rows = set-1, cols = set-2;  acc2[nt][mt]: rows = set-2, cols = set-1
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int rr = a * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh;
        const int va = 128 * sSumA[wr * 64 + rr], vb = 128 * sSumB[buf][wc * 64 + rr];
#pragma unroll
        for (int c = 0; c < 2; ++c) { acc1[a][c][k] = va; acc2[a][c][k] = vb; }
      }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      v4i fa[2], fb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[t] = *reinterpret_cast<const v4i*>(sA + (wr * 64 + t * 32 + lr) * kSiftPitch + kk * 32 + lh * 16);
        fb[t] = *reinterpret_cast<const v4i*>(sB[buf] + (wc * 64 + t * 32 + lr) * kSiftPitch + kk * 32 + lh * 16);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc1[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[mt], fb[nt], acc1[mt][nt], 0, 0, 0);
          acc2[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[nt], fa[mt], acc2[nt][mt], 0, 0, 0);
        }
    }

    // packed top-2 scans (see k_sift_scores): value = score << 8 | (255 - row in the 128-row tile)
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int ccol = (dir == 0 ? wc : wr) * 64 + ct * 32 + lr;
        const int cconst = 128 * (dir == 0 ? sSumB[buf][ccol] : sSumA[ccol]) - kSiftConst;
        const int other = dir == 0 ? wr : wc;
        const int init = (int)((unsigned)(-cconst) << 8);
        const unsigned code_base = 255u - (unsigned)(other * 64 + 4 * lh);
        int best = init, second = init;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const v16i& acc = dir == 0 ? acc1[rt][ct] : acc2[rt][ct];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const unsigned code = code_base - (unsigned)(rt * 32 + (reg & 3) + 8 * (reg >> 2));
            const int v = (int)(((unsigned)acc[reg] << 8) | code);
            int med;
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(med) : "v"(best), "v"(second), "v"(v));
            second = med;
            best = max(best, v);
          }
        }
        {
          const int b2 = __shfl_xor(best, 32), s2 = __shfl_xor(second, 32);
          second = max(max(second, s2), min(best, b2));
          best = max(best, b2);
        }
        const int bs = (best >> 8) + cconst, ss = (second >> 8) + cconst;        // true scores (>= 0)
        const int arg = bs > 0 ? (dir == 0 ? row0 : col0) + 255 - (best & 255) : -1;
        if (dir == 0) {
          // best set-1 row (of this 64-row half of the stripe) for the tile's set-2 descriptors
          const int gcol = col0 + ccol;
          if (lh == 0 && gcol < n2) part21[(size_t)(by * 2 + wr) * n2 + gcol] = make_int4(bs, ss, arg, 0);
        } else {
          top2_merge(bs, ss, arg, rb[ct], rs[ct], ra[ct]);   // ascending tiles: ties keep the earlier index
        }
      }
    }
    if (bx + 1 < bx1) store_b(buf ^ 1);
    __syncthreads();
  }

  // the two waves that share stripe rows (wc = 0, 1: the two 64-row halves of every column tile) merge
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
    if (wc == 1 && lh == 0) sMerge[wr][ct * 32 + lr] = make_int4(rb[ct], rs[ct], ra[ct], 0);
  __syncthreads();
  if (wc == 0 && lh == 0) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int4 o = sMerge[wr][ct * 32 + lr];
      top2_merge(o.x, o.y, o.z, rb[ct], rs[ct], ra[ct]);
      const int grow = row0 + wr * 64 + ct * 32 + lr;
      if (grow < n1) part12[(size_t)chunk * n1 + grow] = make_int4(rb[ct], rs[ct], ra[ct], 0);
    }
  }
}

// sift.cc:72-104: merge the per-tile triples in ascending tile order, then the distance / ratio tests.
// One launch for both directions: threads [0, n1) finish set 1 -> 2, threads [n1, n1 + n2) set 2 -> 1.
__global__ __launch_bounds__(256) void k_sift_finalize(const int4* __restrict__ part12, int n1, int nbx,
                                                       const int4* __restrict__ part21, int n2, int nby,
                                                       float max_ratio, float max_distance, int* __restrict__ m12,
                                                       int* __restrict__ m21) {
  // 16 lanes per descriptor: lane p merges tiles p, p + 16, ... in ascending order, then a 4-step butterfly.
  // On equal best scores the lower index wins, which is the earlier tile (indices ascend with the tile) --
  // the same winner as the reference's single ascending scan.
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, p = threadIdx.x & 15;
  if (g >= n1 + n2) return;
  const bool first = g < n1;
  const int4* __restrict__ part = first ? part12 : part21;
  const int n = first ? n1 : n2, nb = first ? nbx : nby;
  const int i = first ? g : g - n1;
  int best = 0, second = 0, arg = -1;
  for (int b = p; b < nb; b += 16) {
    const int4 v = part[(size_t)b * n + i];
    top2_merge(v.x, v.y, v.z, best, second, arg);
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const int b2 = __shfl_xor(best, o), s2 = __shfl_xor(second, o), a2 = __shfl_xor(arg, o);
    top2_merge(b2, s2, a2, best, second, arg);
  }
  if (p != 0) return;
  int m = -1;
  if (arg != -1 && best > 0) {
    const float kDistNorm = 1.0f / (512.0f * 512.0f);
    const float bn = acosf(fminf(kDistNorm * (float)best, 1.0f));
    if (!(bn > max_distance)) {
      const float sn = acosf(fminf(kDistNorm * (float)second, 1.0f));
      if (!(bn >= max_ratio * sn)) m = arg;
    }
  }
  (first ? m12 : m21)[i] = m;
}

// sift.cc:118-143 for n1 <= 1024 * kCompactPer: cross check, ordered compaction and count in ONE workgroup
// (a launch costs more than the work: 8192 flags).  Larger sets take the three-kernel path below.
constexpr int kCompactPer = 16;
__global__ __launch_bounds__(1024) void k_sift_keep_compact(const int* __restrict__ m12, const int* __restrict__ m21,
                                                            int n1, int cross_check, uint32_t* __restrict__ matches,
                                                            int* __restrict__ count) {
  __shared__ uint32_t s_wave[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n1 + 1023) / 1024;            // <= kCompactPer consecutive rows per thread
  const int i0 = tid * per;
  uint32_t flags = 0, cnt = 0;
  for (int k = 0; k < per; ++k) {
    const int i = i0 + k;
    bool keep = false;
    if (i < n1) {
      const int j = m12[i];
      keep = j != -1;
      if (keep && cross_check) keep = (m21[j] != -1) && (m21[j] == i);
    }
    flags |= (keep ? 1u : 0u) << k;
    cnt += keep ? 1u : 0u;
  }
  // exclusive scan of cnt over the 1024 threads: DPP-free shuffle scan inside the wave, LDS across waves
  uint32_t inc = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    const uint32_t c = s_wave[w];
    if (w < wave) base += c;
    total += c;
  }
  uint32_t pos = base + inc - cnt;
  for (int k = 0; k < per; ++k)
    if ((flags >> k) & 1u) {
      matches[2 * pos] = (uint32_t)(i0 + k);
      matches[2 * pos + 1] = (uint32_t)m12[i0 + k];
      ++pos;
    }
  if (tid == 0) *count = (int)total;
}

// sift.cc:118-143: keep flags, in set-1 order
__global__ void k_sift_keep(const int* __restrict__ m12, const int* __restrict__ m21, int n1, int cross_check,
                            uint32_t* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n1) return;
  const int j = m12[i];
  bool k = j != -1;
  if (k && cross_check) k = (m21[j] != -1) && (m21[j] == i);
  keep[i] = k ? 1u : 0u;
}
__global__ void k_sift_compact(const int* __restrict__ m12, const uint32_t* __restrict__ keep,
                               const uint32_t* __restrict__ pos, int n1, uint32_t* __restrict__ matches,
                               int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n1) return;
  if (keep[i]) { matches[2 * pos[i]] = (uint32_t)i; matches[2 * pos[i] + 1] = (uint32_t)m12[i]; }
  if (i == n1 - 1) *count = (int)(pos[i] + keep[i]);
}

struct SiftScratch {
  DevBuf<uint8_t> d1, d2;
  DevBuf<int> sum1, sum2, m12, m21, count;
  DevBuf<int4> part12, part21;
  DevBuf<uint32_t> keep, pos, matches;
  DevBuf<char> tmp;
};
static SiftScratch* g_sift[64] = {nullptr};
static int g_sift_tile_kernel = 0;   // 1 = one 128x128 tile per workgroup (the first design; A/B timing via PCD_SIFT_TILE=1)
static std::mutex g_sift_mu;

static pcd_status sift_device(int device, const uint8_t* d_d1, int n1, const uint8_t* d_d2, int n2, float max_ratio,
                              float max_distance, int cross_check, int* d_m12, int* d_m21, uint32_t* d_matches,
                              int* d_count, SiftScratch& sc, hipStream_t s) {
  const int nby = (n1 + kSiftTile - 1) / kSiftTile, nbx = (n2 + kSiftTile - 1) / kSiftTile;
  static const int tile_env = std::getenv("PCD_SIFT_TILE") ? std::atoi(std::getenv("PCD_SIFT_TILE")) : 0;
  g_sift_tile_kernel = tile_env;
  // stripe walk: enough (row tile, chunk) workgroups to fill the chip twice over
  const int nchunk = std::max(1, std::min(nbx, (512 + nby - 1) / nby));
  const int ct_per_chunk = (nbx + nchunk - 1) / nchunk;
  const int nchunk_used = (nbx + ct_per_chunk - 1) / ct_per_chunk;
  PCD_TRY(sc.sum1.reserve(n1)); PCD_TRY(sc.sum2.reserve(n2));
  PCD_TRY(sc.part12.reserve((size_t)n1 * std::max(nbx, nchunk_used))); PCD_TRY(sc.part21.reserve((size_t)n2 * nby * 2));
  PCD_TRY(sc.keep.reserve(n1)); PCD_TRY(sc.pos.reserve(n1));
  {
    ScopedKernelTimer t("sift_rowsum", s);
    hipLaunchKernelGGL(k_sift_rowsum, dim3(div_up((uint64_t)n1 + n2, 256)), dim3(256), 0, s, d_d1, n1, sc.sum1.p, d_d2, n2,
                       sc.sum2.p);
  }
  {
    ScopedKernelTimer t("sift_scores", s);
    if (g_sift_tile_kernel)
      hipLaunchKernelGGL(k_sift_scores, dim3(nbx, nby), dim3(256), 0, s, d_d1, n1, d_d2, n2, sc.sum1.p, sc.sum2.p,
                         sc.part12.p, sc.part21.p, nbx, nby);
    else
      hipLaunchKernelGGL(k_sift_scores_stripe, dim3(nchunk_used, nby), dim3(256), 0, s, d_d1, n1, d_d2, n2, sc.sum1.p,
                         sc.sum2.p, sc.part12.p, sc.part21.p, nbx, ct_per_chunk);
  }
  {
    ScopedKernelTimer t("sift_finalize", s);
    hipLaunchKernelGGL(k_sift_finalize, dim3(div_up(((uint64_t)n1 + n2) * 16, 256)), dim3(256), 0, s, sc.part12.p, n1,
                       g_sift_tile_kernel ? nbx : nchunk_used, sc.part21.p, n2, g_sift_tile_kernel ? nby : 2 * nby,
                       max_ratio, max_distance, d_m12, d_m21);
  }
  {
    ScopedKernelTimer t("sift_compact", s);
    if (n1 <= 1024 * kCompactPer) {
      hipLaunchKernelGGL(k_sift_keep_compact, dim3(1), dim3(1024), 0, s, d_m12, d_m21, n1, cross_check, d_matches,
                         d_count);
    } else {
      hipLaunchKernelGGL(k_sift_keep, dim3(div_up(n1, 256)), dim3(256), 0, s, d_m12, d_m21, n1, cross_check, sc.keep.p);
      size_t tb = 0;
      PCD_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, sc.keep.p, sc.pos.p, 0u, (size_t)n1, rocprim::plus<uint32_t>(), s));
      PCD_TRY(sc.tmp.reserve(tb));
      PCD_HIP_TRY(rocprim::exclusive_scan(sc.tmp.p, tb, sc.keep.p, sc.pos.p, 0u, (size_t)n1, rocprim::plus<uint32_t>(), s));
      hipLaunchKernelGGL(k_sift_compact, dim3(div_up(n1, 256)), dim3(256), 0, s, d_m12, sc.keep.p, sc.pos.p, n1,
                         d_matches, d_count);
    }
  }
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

}  // namespace pcd

using namespace pcd;

extern "C" {

pcd_status pcd_sift_match_device(int device, const uint8_t* d_desc1, int n1, const uint8_t* d_desc2, int n2,
                                 float max_ratio, float max_distance, int cross_check, int32_t* d_m12,
                                 int32_t* d_m21, uint32_t* d_matches, int32_t* d_num_matches, void* stream) {
  PCD_REQUIRE(n1 >= 0 && n2 >= 0 && d_num_matches, "sizes / count pointer");
  PCD_TRY(require_device(device));
  hipStream_t s = (hipStream_t)stream;
  if (n1 == 0 || n2 == 0) {   // MatchSiftFeaturesCPU with an empty set: no matches (sift_test.cc:311-318)
    PCD_HIP_TRY(hipMemsetAsync(d_num_matches, 0, sizeof(int32_t), s));
    if (n1 && d_m12) PCD_HIP_TRY(hipMemsetAsync(d_m12, 0xFF, sizeof(int32_t) * n1, s));
    if (n2 && d_m21) PCD_HIP_TRY(hipMemsetAsync(d_m21, 0xFF, sizeof(int32_t) * n2, s));
    return PCD_OK;
  }
  PCD_REQUIRE(d_desc1 && d_desc2 && d_m12 && d_m21 && d_matches, "null pointer");
  PCD_REQUIRE(device < 64, "device ordinal");
  std::lock_guard<std::mutex> g(g_sift_mu);
  if (!g_sift[device]) g_sift[device] = new SiftScratch();
  return sift_device(device, d_desc1, n1, d_desc2, n2, max_ratio, max_distance, cross_check, d_m12, d_m21, d_matches,
                     d_num_matches, *g_sift[device], s);
}

pcd_status pcd_sift_match(int device, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float max_ratio,
                          float max_distance, int cross_check, uint32_t* matches, int32_t* num_matches) {
  PCD_REQUIRE(num_matches && n1 >= 0 && n2 >= 0, "sizes / count pointer");
  *num_matches = 0;
  if (n1 == 0 || n2 == 0) return PCD_OK;
  PCD_REQUIRE(desc1 && desc2 && matches, "null pointer");
  PCD_REQUIRE(device >= 0 && device < 64, "device ordinal");
  PCD_TRY(require_device(device));
  SiftScratch* sc;
  {
    std::lock_guard<std::mutex> g(g_sift_mu);
    if (!g_sift[device]) g_sift[device] = new SiftScratch();
    sc = g_sift[device];
  }
  PCD_TRY(sc->d1.reserve((size_t)n1 * 128)); PCD_TRY(sc->d2.reserve((size_t)n2 * 128));
  PCD_TRY(sc->m12.reserve(n1)); PCD_TRY(sc->m21.reserve(n2)); PCD_TRY(sc->matches.reserve(2 * (size_t)n1));
  PCD_TRY(sc->count.reserve(1));
  hipStream_t s = nullptr;
  PCD_HIP_TRY(hipMemcpyAsync(sc->d1.p, desc1, (size_t)n1 * 128, hipMemcpyHostToDevice, s));
  PCD_HIP_TRY(hipMemcpyAsync(sc->d2.p, desc2, (size_t)n2 * 128, hipMemcpyHostToDevice, s));
  PCD_TRY(pcd_sift_match_device(device, sc->d1.p, n1, sc->d2.p, n2, max_ratio, max_distance, cross_check, sc->m12.p,
                                sc->m21.p, sc->matches.p, sc->count.p, s));
  int cnt = 0;
  PCD_HIP_TRY(hipMemcpyAsync(&cnt, sc->count.p, sizeof(int), hipMemcpyDeviceToHost, s));
  PCD_HIP_TRY(hipStreamSynchronize(s));
  if (cnt) PCD_HIP_TRY(hipMemcpy(matches, sc->matches.p, 2 * (size_t)cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
  *num_matches = cnt;
  return PCD_OK;
}

// ---- matcher handle: two descriptor slots resident on the device (SiftMatchGPU's usage pattern) ----
}  // extern "C"

struct pcd_sift_matcher {
  int device = 0;
  int max_sift = 4096;
  int n[2] = {0, 0};
  pcd::DevBuf<uint8_t> d[2];
  pcd::DevBuf<int32_t> m12, m21, count;
  pcd::DevBuf<uint32_t> matches;
};

extern "C" {

pcd_status pcd_sift_matcher_create(int device, int max_sift, pcd_sift_matcher** out) {
  PCD_REQUIRE(out && max_sift > 0, "null pointer / max_sift");
  PCD_REQUIRE(device >= 0 && device < 64, "device ordinal");
  PCD_TRY(require_device(device));
  pcd_sift_matcher* m = new pcd_sift_matcher();
  m->device = device;
  m->max_sift = max_sift;
  *out = m;
  return PCD_OK;
}

void pcd_sift_matcher_destroy(pcd_sift_matcher* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  delete m;
}

pcd_status pcd_sift_matcher_set_max_sift(pcd_sift_matcher* m, int max_sift) {
  PCD_REQUIRE(m && max_sift > 0, "null pointer / max_sift");
  m->max_sift = max_sift;
  return PCD_OK;
}

// SiftMatchGPU::SetDescriptors(index, num, const unsigned char*): num is clipped to max_sift
pcd_status pcd_sift_matcher_set_descriptors(pcd_sift_matcher* m, int index, int num, const uint8_t* desc) {
  PCD_REQUIRE(m && (index == 0 || index == 1) && num >= 0, "index must be 0 or 1");
  PCD_REQUIRE(num == 0 || desc, "null descriptors");
  PCD_HIP_TRY(hipSetDevice(m->device));
  if (num > m->max_sift) num = m->max_sift;
  m->n[index] = num;
  if (num) {
    PCD_TRY(m->d[index].reserve((size_t)num * 128));
    PCD_HIP_TRY(hipMemcpy(m->d[index].p, desc, (size_t)num * 128, hipMemcpyHostToDevice));
  }
  return PCD_OK;
}

// SiftMatchGPU::GetSiftMatch: number of matches written (at most max_match, in ascending index of set 0)
pcd_status pcd_sift_matcher_match(pcd_sift_matcher* m, int max_match, uint32_t* matches, float distmax, float ratiomax,
                                  int mutual_best_match, int32_t* num_matches) {
  PCD_REQUIRE(m && num_matches && max_match >= 0, "null pointer");
  *num_matches = 0;
  const int n1 = m->n[0], n2 = m->n[1];
  if (n1 == 0 || n2 == 0 || max_match == 0) return PCD_OK;
  PCD_REQUIRE(matches, "null match buffer");
  PCD_HIP_TRY(hipSetDevice(m->device));
  PCD_TRY(m->m12.reserve(n1)); PCD_TRY(m->m21.reserve(n2)); PCD_TRY(m->matches.reserve(2 * (size_t)n1));
  PCD_TRY(m->count.reserve(1));
  hipStream_t s = nullptr;
  PCD_TRY(pcd_sift_match_device(m->device, m->d[0].p, n1, m->d[1].p, n2, ratiomax, distmax, mutual_best_match,
                                m->m12.p, m->m21.p, m->matches.p, m->count.p, s));
  int cnt = 0;
  PCD_HIP_TRY(hipMemcpyAsync(&cnt, m->count.p, sizeof(int), hipMemcpyDeviceToHost, s));
  PCD_HIP_TRY(hipStreamSynchronize(s));
  if (cnt > max_match) cnt = max_match;
  if (cnt) PCD_HIP_TRY(hipMemcpy(matches, m->matches.p, 2 * (size_t)cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
  *num_matches = cnt;
  return PCD_OK;
}

}  // extern "C"
