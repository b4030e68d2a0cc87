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
// Default scores kernel: sift_stripe (k_sift_scores_stripe / k_sift_scores_batch, below): a persistent row-stripe walk
// that multiplies every tile ONCE -- the column-direction top-2 is register-local in the MFMA result layout, the
// row-direction top-2 is a running per-lane state over the whole walk that is merged across lanes once at the end.
// (Rounds 1-2 multiplied every tile twice, A.B^T and B.A^T, so that both scans were register-local.)
// Per chunk / 64-row group and direction one (best, second, argbest) triple per descriptor goes to a partial buffer
// (column direction: 8 bytes, the two packed scan values as they are; k_sift_finalize decodes them);
// k_sift_finalize merges the partials in ascending order (ties -> first index, as the reference's ascending strict->
// scan), applies acos / max_distance / max_ratio; the cross check and the ordered compaction follow.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace pcd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// test / fuzzing knobs (pcd_sift_set_tuning; 0 = the library's choice): column chunks per stripe walk, bytes of partial
// results per sub-batch.  Atomics read once per call: no environment lookups on the call path.
static std::atomic<int> g_sift_nchunk{0};
static std::atomic<uint64_t> g_sift_batch_partials{0};

constexpr int kSiftTile = 128;
constexpr int kSiftPitch = 144;   // bytes per staged descriptor row (128 + 16 pad)
constexpr int kSiftConst = 128 * 128 * 128;
#ifndef PCD_SIFT_WGS
#define PCD_SIFT_WGS 2
#endif

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

// ---- persistent row-stripe variant (round 3: ONE orientation) ----------------------------------------------------
// Workgroup (by, chunk) owns the 128-row tile `by` of set 1 and walks `ct` column tiles of set 2 (staged through LDS,
// the next tile fetched into registers while the current one is used; one __syncthreads per tile).  A wavefront owns
// 64 rows (wr) x the 64-column half wc of every tile and works in blocks of 64 x 32: 8 MFMAs into one of TWO
// accumulator sets while the other set -- the previous block -- is scanned, so the matrix pipe and the VALU run side
// by side inside one wavefront (rounds 1-2 computed every tile in both orientations to keep both scans register-local
// and the two phases of the workgroup's wavefronts ran in step: MFMA time + scan time, 0.13 of the dense i8 peak).
//   * the set-1 fragments of the wavefront's 64 rows stay in registers for the whole walk (32 VGPRs);
//   * accumulators are preloaded with the ROW constant 128 rowsum1 (LDS, in C-layout order: no VALU);
//   * column direction (best set-1 row per set-2 column): register-local in the C layout (lane = column); packed
//     value = (acc << 8) + code, code = 64 - row (an inline constant per register), 3 VALU per score; one partial
//     per (64-row half, column) and block, as before;
//   * row direction (best set-2 column per set-1 row): every lane keeps a running (best, second) for each of its 32
//     (row, lane-column-class) slots over the WHOLE walk; packed value = (acc << 8) + K, K = (column constant << 8) +
//     (255 - block sequence number) is one VGPR per block, so the column constant costs nothing: 3 VALU per score.
//     The 32 lanes that share a row are merged ONCE at the end of the walk, through LDS.
// part12 [nchunk][n1] int4 {best, second, arg, 0}; part21 [2 nby][n2] int2 = the packed (best, second) of the column
// scan, relative to the column constant (sift_finalize decodes: 8 bytes per partial instead of 16 -- the finalize
// kernel is bound by reading them).  A chunk is at most 128 tiles (8-bit sequence code).
__device__ __forceinline__ void sift_stripe(const uint8_t* __restrict__ d1, int n1, const uint8_t* __restrict__ d2,
                                            int n2, const int* __restrict__ sum1, const int* __restrict__ sum2,
                                            int4* __restrict__ part12, int2* __restrict__ part21, int nbx,
                                            int ct_per_chunk, const int chunk, const int by) {
  __shared__ __attribute__((aligned(16))) uint8_t sB[2][kSiftTile * kSiftPitch];
  __shared__ int sSumB[2][kSiftTile];
  __shared__ __attribute__((aligned(16))) int sRc[2][2][32];   // [wr][lane half][mt * 16 + reg]: 128 rowsum1 in C-layout order
  __shared__ int4 sMerge[2][64];
  const int row0 = by * kSiftTile;
  const int bx0 = chunk * ct_per_chunk, bx1 = min(bx0 + ct_per_chunk, nbx);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  if (bx0 >= bx1) return;

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
  // the wavefront's set-1 fragments, straight from global memory (rows past n1: zero descriptors, score 0)
  v4i fa[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int r = row0 + wr * 64 + mt * 32 + lr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < n1) v = *reinterpret_cast<const uint4*>(d1 + (size_t)r * 128 + kk * 32 + lh * 16);
      fa[mt][kk] = v4i{(int)(v.x ^ 0x80808080u), (int)(v.y ^ 0x80808080u), (int)(v.z ^ 0x80808080u), (int)(v.w ^ 0x80808080u)};
    }
  if (tid < kSiftTile) {
    // row t of the tile sits in register k = (r & 3) + 4 (r >> 3) of lane half (r >> 2) & 1, r = t & 31
    const int r = tid & 31;
    sRc[tid >> 6][(r >> 2) & 1][((tid >> 5) & 1) * 16 + (r & 3) + 4 * (r >> 3)] = row0 + tid < n1 ? 128 * sum1[row0 + tid] : 0;
  }
  store_b(0);
  __syncthreads();
  if (bx0 + 1 < bx1) fetch_b(bx0 + 1);

  // running row-direction state: packed (true score << 8 | 255 - sequence number); 0 = score 0, no column
  int rbest[2][16], rsec[2][16];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int k = 0; k < 16; ++k) { rbest[mt][k] = 0; rsec[mt][k] = 0; }

  const v4i* rcp = reinterpret_cast<const v4i*>(&sRc[wr][lh][0]);
  auto mfma_block = [&](v16i (&acc)[2], int buf, int blk) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4i t = rcp[mt * 4 + q];
        acc[mt][4 * q] = t[0]; acc[mt][4 * q + 1] = t[1]; acc[mt][4 * q + 2] = t[2]; acc[mt][4 * q + 3] = t[3];
      }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const v4i fb = *reinterpret_cast<const v4i*>(sB[buf] + (wc * 64 + blk * 32 + lr) * kSiftPitch + kk * 32 + lh * 16);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[mt][kk], fb, acc[mt], 0, 0, 0);
    }
  };
  // cc = the block's column constant 128 rowsum2 - 128^3 of the lane's column (read while the tile's buffer is live)
  auto scan_block = [&](const v16i (&acc)[2], int cc, int bx, int blk, int seq) {
    const int init = (int)((unsigned)(-cc) << 8);   // true score 0, code 0 = "no row" (sift.cc:66-68)
    int cbest = init, csec = init;
    const int K = (int)((unsigned)cc << 8) + (255 - seq);
    int tv, tw;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        // 6 VALU per accumulator: v_lshl_add_u32 (pack), v_med3_i32, v_max_i32 for each direction.  Inline asm: hipcc
        // splits the packs into a shared shift + or + add + add3 (8 per accumulator), and on gfx950 every one of these
        // integer ops issues at half rate (tools/ubench/valu_rate3.hip: 4.4 cycles per wave64 instruction) -- the scan,
        // not the MFMAs (36 cycles per 32x32x32), bounds the kernel.
        // column direction: rows of one lane differ in mt / reg only, so the code is an inline constant
        asm("v_lshl_add_u32 %2, %3, 8, %4\n\tv_med3_i32 %1, %0, %1, %2\n\tv_max_i32 %0, %0, %2"
            : "+v"(cbest), "+v"(csec), "=&v"(tv)
            : "v"(acc[mt][reg]), "n"(64 - (mt * 32 + (reg & 3) + 8 * (reg >> 2))));
        // row direction
        asm("v_lshl_add_u32 %2, %3, 8, %4\n\tv_med3_i32 %1, %0, %1, %2\n\tv_max_i32 %0, %0, %2"
            : "+v"(rbest[mt][reg]), "+v"(rsec[mt][reg]), "=&v"(tw)
            : "v"(acc[mt][reg]), "v"(K));
      }
    // the two lane halves hold interleaved rows (row = ... + 4 lh) of the same column: code -> 68 - row in the
    // wavefront's 64 rows (the "no row" code stays below every real one)
    cbest += 4 * (1 - lh); csec += 4 * (1 - lh);
    {
      const int b2 = __shfl_xor(cbest, 32), s2 = __shfl_xor(csec, 32);
      csec = max(max(csec, s2), min(cbest, b2));
      cbest = max(cbest, b2);
    }
    // packed (score - column constant) << 8 | 68 - row in the 64-row group: decoded in sift_finalize
    const int gcol = bx * kSiftTile + wc * 64 + blk * 32 + lr;
    if (lh == 0 && gcol < n2) part21[(size_t)(by * 2 + wr) * n2 + gcol] = make_int2(cbest, csec);
  };

  v16i acc0[2], acc1[2];
  mfma_block(acc0, 0, 0);
  for (int bx = bx0; bx < bx1; ++bx) {
    const int buf = (bx - bx0) & 1;
    const int cc0 = 128 * sSumB[buf][wc * 64 + lr] - kSiftConst, cc1 = 128 * sSumB[buf][wc * 64 + 32 + lr] - kSiftConst;
    mfma_block(acc1, buf, 1);
    scan_block(acc0, cc0, bx, 0, 2 * (bx - bx0));
    if (bx + 1 < bx1) store_b(buf ^ 1);
    __syncthreads();
    if (bx + 2 < bx1) fetch_b(bx + 2);
    mfma_block(acc0, buf ^ 1, 0);   // (after the last tile: the previous tile once more, unused -- keeps the MFMAs and the scan in one block)
    scan_block(acc1, cc1, bx, 1, 2 * (bx - bx0) + 1);
  }

  // ---- end of the walk: merge the 32 lanes that share a row (through the tile buffers), then the two wavefronts that
  // share the stripe's rows
  __syncthreads();   // every wavefront is done with sB
  int2* tbuf = reinterpret_cast<int2*>(&sB[0][0]) + wave * (32 * 33);   // 32 rows x 33 (pitch) per wavefront: 8448 B of 9216
  int rb[2], rs[2], ra[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
      tbuf[((reg & 3) + 8 * (reg >> 2) + 4 * lh) * 33 + lr] = make_int2(rbest[mt][reg], rsec[mt][reg]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the region is this wavefront's own: program order is enough
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane (rho = lr, h = lh) merges columns 16 h .. 16 h + 15 of row rho in ascending order
    int best = 0, second = 0, arg = -1;
#pragma unroll 4
    for (int c = 16 * lh; c < 16 * lh + 16; ++c) {
      const int2 e = tbuf[lr * 33 + c];
      const int sc = e.x >> 8, seq = 255 - (e.x & 255);
      const int col = (bx0 + (seq >> 1)) * kSiftTile + wc * 64 + (seq & 1) * 32 + c;
      top2_merge(sc, e.y >> 8, sc > 0 ? col : -1, best, second, arg);
    }
    {
      const int b2 = __shfl_xor(best, 32), s2 = __shfl_xor(second, 32), a2 = __shfl_xor(arg, 32);
      top2_merge(b2, s2, a2, best, second, arg);
    }
    rb[mt] = best; rs[mt] = second; ra[mt] = arg;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    if (wc == 1 && lh == 0) sMerge[wr][mt * 32 + lr] = make_int4(rb[mt], rs[mt], ra[mt], 0);
  __syncthreads();
  if (wc == 0 && lh == 0) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int4 o = sMerge[wr][mt * 32 + lr];
      top2_merge(o.x, o.y, o.z, rb[mt], rs[mt], ra[mt]);
      const int grow = row0 + wr * 64 + mt * 32 + lr;
      if (grow < n1) part12[(size_t)chunk * n1 + grow] = make_int4(rb[mt], rs[mt], ra[mt], 0);
    }
  }
}

__global__ __launch_bounds__(256, PCD_SIFT_WGS) void k_sift_scores_stripe(const uint8_t* __restrict__ d1, int n1,
                                                            const uint8_t* __restrict__ d2, int n2,
                                                            const int* __restrict__ sum1, const int* __restrict__ sum2,
                                                            int4* __restrict__ part12, int2* __restrict__ part21,
                                                            int nbx, int ct_per_chunk) {
  sift_stripe(d1, n1, d2, n2, sum1, sum2, part12, part21, nbx, ct_per_chunk, blockIdx.x, blockIdx.y);
}

// ---- many image pairs in one launch set (pcd_sift_match_batch_device) ----------------------------------
// All descriptors live in one arena; a pair names two row ranges of it.  blockIdx.z = pair; the grid's x / y
// extents are sized for the largest pair of the batch, smaller pairs leave early.
struct SiftPairDev {
  uint32_t row1, n1, row2, n2;   // arena rows of the two images
  uint64_t part12, part21;       // int4 / int2 offsets of the pair's partial results
  uint64_t m12, m21;             // int offsets of the pair's best-match arrays
  uint64_t match;                // offset (in matches) of the pair's output list
};

__global__ __launch_bounds__(256, PCD_SIFT_WGS) void k_sift_scores_batch(const uint8_t* __restrict__ arena,
                                                           const int* __restrict__ sum,
                                                           const SiftPairDev* __restrict__ pairs,
                                                           int4* __restrict__ part12, int2* __restrict__ part21,
                                                           int nchunk) {
  const SiftPairDev pr = pairs[blockIdx.z];
  const int nbx = ((int)pr.n2 + kSiftTile - 1) / kSiftTile, nby = ((int)pr.n1 + kSiftTile - 1) / kSiftTile;
  if ((int)blockIdx.y >= nby) return;
  const int ct = (nbx + nchunk - 1) / nchunk;   // chunks past the pair's last column tile leave inside sift_stripe
  sift_stripe(arena + (size_t)pr.row1 * 128, (int)pr.n1, arena + (size_t)pr.row2 * 128, (int)pr.n2, sum + pr.row1,
              sum + pr.row2, part12 + pr.part12, part21 + pr.part21, nbx, ct, blockIdx.x, blockIdx.y);
}

// sift.cc:72-104: merge the per-tile triples in ascending tile order, then the distance / ratio tests.
// One launch for both directions: threads [0, n1) finish set 1 -> 2, threads [n1, n1 + n2) set 2 -> 1.
__device__ __forceinline__ void sift_finalize(const int4* __restrict__ part12, int n1, int nbx,
                                              const int2* __restrict__ part21, int n2, int nby,
                                              const int* __restrict__ sum2, float max_ratio,
                                              float max_distance, int* __restrict__ m12, int* __restrict__ m21) {
  // 16 lanes per descriptor: lane p merges tiles p, p + 16, ... in ascending order, then a 4-step butterfly.
  // On equal best scores the lower index wins, which is the earlier tile (indices ascend with the tile) --
  // the same winner as the reference's single ascending scan.
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, p = threadIdx.x & 15;
  if (g >= n1 + n2) return;
  const bool first = g < n1;
  const int i = first ? g : g - n1;
  int best = 0, second = 0, arg = -1;
  if (first) {
    for (int b = p; b < nbx; b += 16) {
      const int4 v = part12[(size_t)b * n1 + i];
      top2_merge(v.x, v.y, v.z, best, second, arg);
    }
  } else {
    // column direction: partial b = the 64-row group b (rows 64 b ..); packed (score - cc) << 8 | 68 - row in the group
    const int cc = 128 * sum2[i] - kSiftConst;
    for (int b = p; b < nby; b += 16) {
      const int2 v = part21[(size_t)b * n2 + i];
      const int bs = (v.x >> 8) + cc, ss = (v.y >> 8) + cc;   // true scores (>= 0)
      top2_merge(bs, ss, bs > 0 ? b * 64 + 68 - (v.x & 255) : -1, best, second, arg);
    }
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

__global__ __launch_bounds__(256) void k_sift_finalize(const int4* __restrict__ part12, int n1, int nbx,
                                                       const int2* __restrict__ part21, int n2, int nby,
                                                       const int* __restrict__ sum2, float max_ratio,
                                                       float max_distance, int* __restrict__ m12,
                                                       int* __restrict__ m21) {
  sift_finalize(part12, n1, nbx, part21, n2, nby, sum2, max_ratio, max_distance, m12, m21);
}

__global__ __launch_bounds__(256) void k_sift_finalize_batch(const SiftPairDev* __restrict__ pairs,
                                                             const int* __restrict__ sum,
                                                             const int4* __restrict__ part12,
                                                             const int2* __restrict__ part21, int nchunk,
                                                             float max_ratio, float max_distance, int* __restrict__ m12,
                                                             int* __restrict__ m21) {
  const SiftPairDev pr = pairs[blockIdx.z];
  const int nbx = ((int)pr.n2 + kSiftTile - 1) / kSiftTile, nby = ((int)pr.n1 + kSiftTile - 1) / kSiftTile;
  const int ct = max(1, (nbx + nchunk - 1) / nchunk), used = (nbx + ct - 1) / ct;   // chunks that wrote a partial
  sift_finalize(part12 + pr.part12, (int)pr.n1, used, part21 + pr.part21, (int)pr.n2, 2 * nby, sum + pr.row2, max_ratio,
                max_distance, m12 + pr.m12, m21 + pr.m21);
}

// sift.cc:118-143 for n1 <= 1024 * kCompactPer: cross check, ordered compaction and count in ONE workgroup
// (a launch costs more than the work: 8192 flags).  Larger sets take the three-kernel path below.
constexpr int kCompactPer = 16;
__device__ __forceinline__ void sift_keep_compact(const int* __restrict__ m12, const int* __restrict__ m21, int n1,
                                                  int cross_check, uint32_t* __restrict__ matches,
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

__global__ __launch_bounds__(1024) void k_sift_keep_compact(const int* __restrict__ m12, const int* __restrict__ m21,
                                                            int n1, int cross_check, uint32_t* __restrict__ matches,
                                                            int* __restrict__ count) {
  sift_keep_compact(m12, m21, n1, cross_check, matches, count);
}

// one workgroup per pair
__global__ __launch_bounds__(1024) void k_sift_keep_compact_batch(const SiftPairDev* __restrict__ pairs,
                                                                  const int* __restrict__ m12,
                                                                  const int* __restrict__ m21, int cross_check,
                                                                  uint32_t* __restrict__ matches,
                                                                  int* __restrict__ counts) {
  const SiftPairDev pr = pairs[blockIdx.x];
  sift_keep_compact(m12 + pr.m12, m21 + pr.m21, (int)pr.n1, cross_check, matches + 2 * pr.match, counts + blockIdx.x);
}

// dense packing of the per-pair lists (host entry): list p moves from matches_in + 2 * src_off[p] to
// matches_out + 2 * dense_off[p]
__global__ __launch_bounds__(256) void k_sift_pack_lists(const uint64_t* __restrict__ src_off,
                                                         const int* __restrict__ counts,
                                                         const uint64_t* __restrict__ dense_off,
                                                         const uint32_t* __restrict__ matches_in,
                                                         uint32_t* __restrict__ matches_out) {
  const int p = blockIdx.x;
  const uint2* __restrict__ src = reinterpret_cast<const uint2*>(matches_in) + src_off[p];
  uint2* __restrict__ dst = reinterpret_cast<uint2*>(matches_out) + dense_off[p];
  for (int i = threadIdx.x; i < counts[p]; i += blockDim.x) dst[i] = src[i];
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
  DevBuf<int4> part12;
  DevBuf<int2> part21;
  DevBuf<uint32_t> keep, pos, matches;
  DevBuf<char> tmp;
  // batch entry
  DevBuf<uint8_t> arena;
  DevBuf<SiftPairDev> pairs;
  DevBuf<int> counts;
  DevBuf<uint64_t> dense_off;
  DevBuf<uint32_t> dense;
  // The scratch is per DEVICE and shared by every call on it.  Host side: g_sift_mu (recursive) is held for the whole
  // of a host entry point and while a *_device entry point enqueues.  Device side: calls on different streams are
  // ordered through ev_done (the next call's stream waits for the previous call's last kernel before any of its own
  // work touches part12 / m12 / pairs ...; calls on one stream are ordered anyway).  The pair table goes up with an
  // asynchronous copy on the caller's stream from h_pairs; ev_tab guards the pinned buffer against being rewritten
  // while that copy is still in flight.
  PinnedBuf<SiftPairDev> h_pairs;
  hipEvent_t ev_done = nullptr, ev_tab = nullptr;
  hipStream_t last_stream = nullptr;
  bool used = false, tab_pending = false;
  ~SiftScratch() {
    if (ev_done) (void)hipEventDestroy(ev_done);
    if (ev_tab) (void)hipEventDestroy(ev_tab);
  }
};
static SiftScratch* g_sift[64] = {nullptr};
static std::recursive_mutex g_sift_mu;
typedef std::lock_guard<std::recursive_mutex> SiftLock;

// bracket of every use of the shared scratch on stream s (g_sift_mu held)
static pcd_status sift_begin_use(SiftScratch& sc, hipStream_t s) {
  if (!sc.ev_done) {
    PCD_HIP_TRY(hipEventCreateWithFlags(&sc.ev_done, hipEventDisableTiming));
    PCD_HIP_TRY(hipEventCreateWithFlags(&sc.ev_tab, hipEventDisableTiming));
  }
  if (sc.used && sc.last_stream != s) PCD_HIP_TRY(hipStreamWaitEvent(s, sc.ev_done, 0));
  return PCD_OK;
}
static pcd_status sift_end_use(SiftScratch& sc, hipStream_t s) {
  PCD_HIP_TRY(hipEventRecord(sc.ev_done, s));
  sc.last_stream = s;
  sc.used = true;
  return PCD_OK;
}

static pcd_status sift_device(int device, const uint8_t* d_d1, int n1, const uint8_t* d_d2, int n2, float max_ratio,
                              float max_distance, int cross_check, int* d_m12, int* d_m21, uint32_t* d_matches,
                              int* d_count, SiftScratch& sc, hipStream_t s) {
  const int nby = (n1 + kSiftTile - 1) / kSiftTile, nbx = (n2 + kSiftTile - 1) / kSiftTile;
  // stripe walk: enough (row tile, chunk) workgroups to fill the chip twice over
  // (a chunk is at most 128 column tiles: the stripe kernel's 8-bit sequence code)
  // pcd_sift_set_tuning (tests / fuzzing) can force the number of column chunks, e.g. 1 = every stripe walks all tiles
  const int nchunk_env = g_sift_nchunk.load(std::memory_order_relaxed);   // pcd_sift_set_tuning (tests / fuzzing)
  const int want = nchunk_env > 0 ? std::min(nchunk_env, nbx) : std::min(nbx, (512 + nby - 1) / nby);
  const int nchunk = std::max({1, want, (nbx + 127) / 128});
  const int ct_per_chunk = (nbx + nchunk - 1) / nchunk;
  const int nchunk_used = (nbx + ct_per_chunk - 1) / ct_per_chunk;
  PCD_TRY(sc.sum1.reserve(n1)); PCD_TRY(sc.sum2.reserve(n2));
  PCD_TRY(sc.part12.reserve((size_t)n1 * nchunk_used)); PCD_TRY(sc.part21.reserve((size_t)n2 * nby * 2));
  PCD_TRY(sc.keep.reserve(n1)); PCD_TRY(sc.pos.reserve(n1));
  {
    ScopedKernelTimer t("sift_rowsum", s);
    hipLaunchKernelGGL(k_sift_rowsum, dim3(div_up((uint64_t)n1 + n2, 256)), dim3(256), 0, s, d_d1, n1, sc.sum1.p, d_d2, n2,
                       sc.sum2.p);
  }
  {
    ScopedKernelTimer t("sift_scores", s);
    hipLaunchKernelGGL(k_sift_scores_stripe, dim3(nchunk_used, nby), dim3(256), 0, s, d_d1, n1, d_d2, n2, sc.sum1.p,
                       sc.sum2.p, sc.part12.p, sc.part21.p, nbx, ct_per_chunk);
  }
  {
    ScopedKernelTimer t("sift_finalize", s);
    hipLaunchKernelGGL(k_sift_finalize, dim3(div_up(((uint64_t)n1 + n2) * 16, 256)), dim3(256), 0, s, sc.part12.p, n1,
                       nchunk_used, sc.part21.p, n2, 2 * nby, sc.sum2.p, max_ratio, max_distance, d_m12, d_m21);
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


// Many pairs over one descriptor arena.  Sub-batches are cut so that the partial results of a sub-batch stay under
// kSiftBatchPartials int4 (2 GiB); each sub-batch is three launches (scores, finalize, cross check + compaction),
// the arena's row sums one launch in front.  Nothing synchronises with the host.
constexpr size_t kSiftBatchPartials = (size_t)1 << 27;   // PCD_SIFT_BATCH_PARTIALS overrides it (tests: forces the cuts)

static pcd_status sift_batch_device(int device, const uint8_t* d_arena, const uint64_t* first_row, int n_images,
                                    const uint32_t* pair_ids, int n_pairs, float max_ratio, float max_distance,
                                    int cross_check, uint32_t* d_matches, const uint64_t* match_offset, int* d_counts,
                                    SiftScratch& sc, hipStream_t s) {
  const uint64_t total_rows = first_row[n_images];
  PCD_REQUIRE(total_rows < (1ull << 32), "arena larger than 2^32 descriptors");
  const uint64_t budget_set = g_sift_batch_partials.load(std::memory_order_relaxed);   // pcd_sift_set_tuning
  const size_t budget = budget_set ? (size_t)budget_set : kSiftBatchPartials;
  for (int i = 0; i < n_images; ++i) PCD_REQUIRE(first_row[i] <= first_row[i + 1], "first_row must ascend");
  uint64_t max_nbx = 1;
  bool wide = false;   // a set too long for the one-workgroup compaction: those batches run pair by pair
  for (int p = 0; p < n_pairs; ++p) {
    PCD_REQUIRE(pair_ids[2 * p] < (uint32_t)n_images && pair_ids[2 * p + 1] < (uint32_t)n_images, "pair names an image outside the arena");
    const uint64_t n1 = first_row[pair_ids[2 * p] + 1] - first_row[pair_ids[2 * p]];
    const uint64_t n2 = first_row[pair_ids[2 * p + 1] + 1] - first_row[pair_ids[2 * p + 1]];
    PCD_REQUIRE(n1 < (1u << 30) && n2 < (1u << 30), "image too large");
    wide = wide || n1 > (uint64_t)1024 * kCompactPer;
    max_nbx = std::max(max_nbx, (n2 + kSiftTile - 1) / kSiftTile);
  }
  if (wide) {
    for (int p = 0; p < n_pairs; ++p) {
      const uint32_t a = pair_ids[2 * p], b = pair_ids[2 * p + 1];
      const int n1 = (int)(first_row[a + 1] - first_row[a]), n2 = (int)(first_row[b + 1] - first_row[b]);
      if (n1 == 0 || n2 == 0) { PCD_HIP_TRY(hipMemsetAsync(d_counts + p, 0, sizeof(int), s)); continue; }
      PCD_TRY(sc.m12.reserve(n1)); PCD_TRY(sc.m21.reserve(n2));
      PCD_TRY(sift_device(device, d_arena + first_row[a] * 128, n1, d_arena + first_row[b] * 128, n2, max_ratio,
                          max_distance, cross_check, sc.m12.p, sc.m21.p, d_matches + 2 * match_offset[p], d_counts + p,
                          sc, s));
    }
    return PCD_OK;
  }
  PCD_TRY(sc.sum1.reserve(total_rows));
  if (total_rows) {
    ScopedKernelTimer t("sift_rowsum", s);
    hipLaunchKernelGGL(k_sift_rowsum, dim3(div_up(total_rows, 256)), dim3(256), 0, s, d_arena, (int)total_rows, sc.sum1.p,
                       (const uint8_t*)nullptr, 0, (int*)nullptr);
  }
  // pair table for the whole call (uploaded once; sub-batches index into it)
  std::vector<SiftPairDev> tab((size_t)n_pairs);
  std::vector<int> cut;   // sub-batch boundaries
  std::vector<int> cut_nchunk;
  cut.push_back(0);
  size_t max12 = 0, max21 = 0, maxm12 = 0, maxm21 = 0;
  {
    int p0 = 0;
    while (p0 < n_pairs) {
      // how many column chunks per row stripe: enough workgroups to fill the chip twice when the batch is small
      // (decided on the first pair's size and the pairs left; any value gives the same results)
      const uint32_t a0 = pair_ids[2 * p0], b0 = pair_ids[2 * p0 + 1];
      const int nby0 = std::max<int>(1, (int)((first_row[a0 + 1] - first_row[a0] + kSiftTile - 1) / kSiftTile));
      const int nbx0 = std::max<int>(1, (int)((first_row[b0 + 1] - first_row[b0] + kSiftTile - 1) / kSiftTile));
      const long left = n_pairs - p0;
      const int nchunk_env = g_sift_nchunk.load(std::memory_order_relaxed);
      const long want = nchunk_env > 0 ? std::min<long>(nchunk_env, nbx0) : std::min<long>(nbx0, (512 + nby0 * left - 1) / (nby0 * left));
      const int nchunk = (int)std::max<long>({1, want, (long)((max_nbx + 127) / 128)});   // a chunk is at most 128 column tiles
      size_t o12 = 0, o21 = 0, om12 = 0, om21 = 0;
      int p = p0;
      for (; p < n_pairs; ++p) {
        const uint32_t a = pair_ids[2 * p], b = pair_ids[2 * p + 1];
        uint64_t n1 = first_row[a + 1] - first_row[a], n2 = first_row[b + 1] - first_row[b];
        if (n1 == 0 || n2 == 0) n1 = n2 = 0;   // an empty image: no matches (sift_test.cc:311-318); every kernel skips the pair
        const size_t nby = (n1 + kSiftTile - 1) / kSiftTile;
        const size_t need12 = (size_t)nchunk * n1, need21 = 2 * nby * n2;
        // (the budget counts 16-byte units: part12 entries are int4, part21 entries int2)
        if (p > p0 && (o12 + need12 + (o21 + need21 + 1) / 2 > budget || p - p0 >= 65535)) break;
        tab[p] = SiftPairDev{(uint32_t)first_row[a], (uint32_t)n1, (uint32_t)first_row[b], (uint32_t)n2, o12, o21, om12, om21,
                             match_offset[p]};
        o12 += need12; o21 += need21; om12 += n1; om21 += n2;
      }
      max12 = std::max(max12, o12); max21 = std::max(max21, o21);
      maxm12 = std::max(maxm12, om12); maxm21 = std::max(maxm21, om21);
      cut.push_back(p);
      cut_nchunk.push_back(nchunk);
      p0 = p;
    }
  }
  PCD_TRY(sc.pairs.reserve(n_pairs));
  PCD_TRY(sc.part12.reserve(max12)); PCD_TRY(sc.part21.reserve(max21));
  PCD_TRY(sc.m12.reserve(maxm12)); PCD_TRY(sc.m21.reserve(maxm21));
  // the table goes up ON THE CALLER'S STREAM (a null-stream copy is not ordered against a non-blocking stream: a second
  // call could overwrite sc.pairs under the first call's kernels); pinned staging, guarded by ev_tab
  if (sc.tab_pending) PCD_HIP_TRY(hipEventSynchronize(sc.ev_tab));
  PCD_TRY(sc.h_pairs.reserve(n_pairs));
  std::memcpy(sc.h_pairs.p, tab.data(), sizeof(SiftPairDev) * (size_t)n_pairs);
  PCD_HIP_TRY(hipMemcpyAsync(sc.pairs.p, sc.h_pairs.p, sizeof(SiftPairDev) * (size_t)n_pairs, hipMemcpyHostToDevice, s));
  PCD_HIP_TRY(hipEventRecord(sc.ev_tab, s));
  sc.tab_pending = true;
  for (size_t k = 0; k + 1 < cut.size(); ++k) {
    const int p0 = cut[k], np = cut[k + 1] - cut[k], nchunk = cut_nchunk[k];
    uint32_t mx1 = 0, mx2 = 0, mxsum = 0;
    for (int p = p0; p < p0 + np; ++p) {
      mx1 = std::max(mx1, tab[p].n1); mx2 = std::max(mx2, tab[p].n2); mxsum = std::max(mxsum, tab[p].n1 + tab[p].n2);
    }
    if (mx1 && mx2) {
      {
        ScopedKernelTimer t("sift_scores", s);
        hipLaunchKernelGGL(k_sift_scores_batch, dim3(nchunk, (mx1 + kSiftTile - 1) / kSiftTile, np), dim3(256), 0, s, d_arena,
                           sc.sum1.p, sc.pairs.p + p0, sc.part12.p, sc.part21.p, nchunk);
      }
      {
        ScopedKernelTimer t("sift_finalize", s);
        hipLaunchKernelGGL(k_sift_finalize_batch, dim3(div_up((uint64_t)mxsum * 16, 256), 1, np), dim3(256), 0, s,
                           sc.pairs.p + p0, sc.sum1.p, sc.part12.p, sc.part21.p, nchunk, max_ratio, max_distance, sc.m12.p, sc.m21.p);
      }
    }
    {
      ScopedKernelTimer t("sift_compact", s);
      hipLaunchKernelGGL(k_sift_keep_compact_batch, dim3(np), dim3(1024), 0, s, sc.pairs.p + p0, sc.m12.p, sc.m21.p,
                         cross_check, d_matches, d_counts + p0);
    }
  }
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}
}  // namespace pcd

using namespace pcd;

extern "C" {

/* tuning hook (tests, fuzzing; not part of the stable ABI): force the number of column chunks of a stripe walk and the
 * bytes of partial results per sub-batch of pcd_sift_match_batch; 0 = the library's choice */
pcd_status pcd_sift_set_tuning(int nchunk, uint64_t batch_partials_bytes) {
  PCD_REQUIRE(nchunk >= 0, "nchunk");
  g_sift_nchunk.store(nchunk, std::memory_order_relaxed);
  g_sift_batch_partials.store(batch_partials_bytes, std::memory_order_relaxed);
  return PCD_OK;
}

pcd_status pcd_sift_match_device(int device, const uint8_t* d_desc1, int n1, const uint8_t* d_desc2, int n2,
                                 float max_ratio, float max_distance, int cross_check, int32_t* d_m12,
                                 int32_t* d_m21, uint32_t* d_matches, int32_t* d_num_matches, void* stream) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(n1 >= 0 && n2 >= 0 && d_num_matches, "sizes / count pointer");
  PCD_REFUSE_CAPTURE(stream);
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
  SiftLock g(g_sift_mu);
  PCD_HIP_TRY(hipSetDevice(device));
  if (!g_sift[device]) g_sift[device] = new SiftScratch();
  PCD_TRY(sift_begin_use(*g_sift[device], s));
  const pcd_status st = sift_device(device, d_desc1, n1, d_desc2, n2, max_ratio, max_distance, cross_check, d_m12, d_m21,
                                    d_matches, d_num_matches, *g_sift[device], s);
  PCD_TRY(sift_end_use(*g_sift[device], s));
  return st;
  });
}

pcd_status pcd_sift_match(int device, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float max_ratio,
                          float max_distance, int cross_check, uint32_t* matches, int32_t* num_matches) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(num_matches && n1 >= 0 && n2 >= 0, "sizes / count pointer");
  *num_matches = 0;
  if (n1 == 0 || n2 == 0) return PCD_OK;
  PCD_REQUIRE(desc1 && desc2 && matches, "null pointer");
  PCD_REQUIRE(device >= 0 && device < 64, "device ordinal");
  PCD_TRY(require_device(device));
  SiftLock g(g_sift_mu);   // the whole host call: it owns the scratch's d1 / d2 / m12 / matches until it has synchronised
  PCD_HIP_TRY(hipSetDevice(device));
  if (!g_sift[device]) g_sift[device] = new SiftScratch();
  SiftScratch* sc = g_sift[device];
  hipStream_t s = nullptr;
  PCD_TRY(sift_begin_use(*sc, s));   // an earlier *_device call on another stream may still be reading the scratch
  PCD_TRY(sc->d1.reserve((size_t)n1 * 128)); PCD_TRY(sc->d2.reserve((size_t)n2 * 128));
  PCD_TRY(sc->m12.reserve(n1)); PCD_TRY(sc->m21.reserve(n2)); PCD_TRY(sc->matches.reserve(2 * (size_t)n1));
  PCD_TRY(sc->count.reserve(1));
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
  });
}


pcd_status pcd_sift_match_batch_device(int device, const uint8_t* d_arena, const uint64_t* first_row, int n_images,
                                       const uint32_t* pairs, int n_pairs, float max_ratio, float max_distance,
                                       int cross_check, uint32_t* d_matches, const uint64_t* match_offset,
                                       int32_t* d_counts, void* stream) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(n_images >= 0 && n_pairs >= 0 && first_row, "sizes / first_row");
  PCD_REFUSE_CAPTURE(stream);
  if (n_pairs == 0) return PCD_OK;
  PCD_REQUIRE(pairs && match_offset && d_counts && d_matches, "null pointer");
  PCD_REQUIRE(first_row[n_images] == 0 || d_arena, "null arena");
  PCD_REQUIRE(device >= 0 && device < 64, "device ordinal");
  PCD_TRY(require_device(device));
  SiftLock g(g_sift_mu);
  PCD_HIP_TRY(hipSetDevice(device));
  if (!g_sift[device]) g_sift[device] = new SiftScratch();
  PCD_TRY(sift_begin_use(*g_sift[device], (hipStream_t)stream));
  const pcd_status st = sift_batch_device(device, d_arena, first_row, n_images, pairs, n_pairs, max_ratio, max_distance,
                                          cross_check, d_matches, match_offset, d_counts, *g_sift[device],
                                          (hipStream_t)stream);
  PCD_TRY(sift_end_use(*g_sift[device], (hipStream_t)stream));
  return st;
  });
}

pcd_status pcd_sift_match_batch(int device, const uint8_t* arena, const uint64_t* first_row, int n_images,
                                const uint32_t* pairs, int n_pairs, float max_ratio, float max_distance, int cross_check,
                                uint32_t* matches, uint64_t matches_capacity, uint64_t* list_offset) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(n_images >= 0 && n_pairs >= 0 && first_row && list_offset, "sizes / first_row / list_offset");
  list_offset[0] = 0;
  if (n_pairs == 0) return PCD_OK;
  PCD_REQUIRE(pairs, "null pair list");
  PCD_REQUIRE(device >= 0 && device < 64, "device ordinal");
  PCD_TRY(require_device(device));
  const uint64_t rows = first_row[n_images];
  PCD_REQUIRE(rows == 0 || arena, "null arena");
  SiftLock g(g_sift_mu);   // the whole host call: arena / matches / counts / dense are the device's shared scratch
  PCD_HIP_TRY(hipSetDevice(device));
  if (!g_sift[device]) g_sift[device] = new SiftScratch();
  SiftScratch* sc = g_sift[device];
  PCD_TRY(sift_begin_use(*sc, nullptr));
  // worst-case list of pair p: one match per descriptor of its first image
  std::vector<uint64_t> off((size_t)n_pairs + 1, 0);
  for (int p = 0; p < n_pairs; ++p) {
    PCD_REQUIRE(pairs[2 * p] < (uint32_t)n_images && pairs[2 * p + 1] < (uint32_t)n_images, "pair names an image outside the arena");
    off[p + 1] = off[p] + (first_row[pairs[2 * p] + 1] - first_row[pairs[2 * p]]);
  }
  PCD_TRY(sc->arena.reserve(rows * 128)); PCD_TRY(sc->matches.reserve(2 * off[n_pairs] + 2));
  PCD_TRY(sc->counts.reserve(n_pairs)); PCD_TRY(sc->dense_off.reserve(2 * ((size_t)n_pairs + 1)));
  hipStream_t s = nullptr;
  if (rows) PCD_HIP_TRY(hipMemcpy(sc->arena.p, arena, rows * 128, hipMemcpyHostToDevice));
  PCD_TRY(pcd_sift_match_batch_device(device, sc->arena.p, first_row, n_images, pairs, n_pairs, max_ratio, max_distance,
                                      cross_check, sc->matches.p, off.data(), sc->counts.p, s));
  std::vector<int> cnt((size_t)n_pairs);
  PCD_HIP_TRY(hipMemcpy(cnt.data(), sc->counts.p, sizeof(int) * (size_t)n_pairs, hipMemcpyDeviceToHost));
  for (int p = 0; p < n_pairs; ++p) list_offset[p + 1] = list_offset[p] + (uint64_t)cnt[p];
  const uint64_t total = list_offset[n_pairs];
  if (total > matches_capacity) {
    set_error("pcd_sift_match_batch: %llu matches, capacity %llu", (unsigned long long)total, (unsigned long long)matches_capacity);
    return PCD_ERR_INVALID;
  }
  if (total == 0) return PCD_OK;
  PCD_REQUIRE(matches, "null match buffer");
  // pack the lists back to back on the device: one download of exactly the matches
  PCD_TRY(sc->dense.reserve(2 * total));
  const size_t np1 = (size_t)n_pairs + 1;
  PCD_HIP_TRY(hipMemcpy(sc->dense_off.p, list_offset, sizeof(uint64_t) * np1, hipMemcpyHostToDevice));
  PCD_HIP_TRY(hipMemcpy(sc->dense_off.p + np1, off.data(), sizeof(uint64_t) * np1, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sift_pack_lists, dim3(n_pairs), dim3(256), 0, s, sc->dense_off.p + np1, sc->counts.p,
                     sc->dense_off.p, sc->matches.p, sc->dense.p);
  PCD_HIP_TRY(hipGetLastError());
  PCD_HIP_TRY(hipMemcpy(matches, sc->dense.p, 2 * total * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return PCD_OK;
  });
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
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(out && max_sift > 0, "null pointer / max_sift");
  PCD_REQUIRE(device >= 0 && device < 64, "device ordinal");
  PCD_TRY(require_device(device));
  pcd_sift_matcher* m = new pcd_sift_matcher();
  m->device = device;
  m->max_sift = max_sift;
  *out = m;
  return PCD_OK;
  });
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
