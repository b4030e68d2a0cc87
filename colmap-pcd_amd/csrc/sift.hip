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
constexpr int kSiftConst = 128 * 128 * 128;
#ifndef PCD_SIFT_WGS
#define PCD_SIFT_WGS 2
#endif
#ifndef PCD_SIFT_ABLATE   // timing-only variants (tools/sift_ablate.sh; results are wrong): 1 no scan, 2 no slow path,
#define PCD_SIFT_ABLATE 0 // 4 no fetch / store of further tiles, 8 no barrier, 16 no MFMA
#endif

// ---- preparation: the re-centred, padded copy the walks read ------------------------------------------------------
// Every image (descriptor set) is copied ONCE per call into scratch with its bytes re-centred (x ^ 0x80) and its rows
// padded with zero descriptors to a multiple of 128, next to one constant per row: cc = 128 sum(row) - 128^3 (a padding
// row: -128^3, i.e. true score 0 against anything).  The walks then stage column tiles with LDS-DMA (no registers, no
// VALU, no bounds tests) and read their row fragments unconditionally.
struct SiftImageDev {
  const uint8_t* src;   // the image's descriptors (n x 128 bytes)
  uint32_t n;           // descriptors
  uint32_t prow;        // first row of the image in the padded copy (a multiple of 128)
};

// one workgroup per 128-row tile: thread (row = tid >> 1, half = tid & 1) moves 64 bytes
__device__ __forceinline__ void sift_prep_tile(const SiftImageDev im, int tile, uint8_t* __restrict__ xa,
                                               int* __restrict__ cc) {
  const uint32_t row = (uint32_t)tile * kSiftTile + (threadIdx.x >> 1), h = threadIdx.x & 1;
  if ((uint32_t)tile * kSiftTile >= im.n) return;   // (the grid is sized for the largest image; an empty one has no tiles)
  uint4 v[4];
  unsigned s = 0;
  if (row < im.n) {
    const uint4* p = reinterpret_cast<const uint4*>(im.src + (size_t)row * 128 + h * 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] = p[c];
      const unsigned w[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) s += (w[k] & 0xFF) + ((w[k] >> 8) & 0xFF) + ((w[k] >> 16) & 0xFF) + (w[k] >> 24);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = make_uint4(0, 0, 0, 0);
  }
  s += __shfl_xor(s, 1);
  uint4* q = reinterpret_cast<uint4*>(xa + (size_t)(im.prow + row) * 128 + h * 64);
#pragma unroll
  for (int c = 0; c < 4; ++c)
    q[c] = make_uint4(v[c].x ^ 0x80808080u, v[c].y ^ 0x80808080u, v[c].z ^ 0x80808080u, v[c].w ^ 0x80808080u);
  if (h == 0) cc[im.prow + row] = 128 * (int)s - kSiftConst;
}

__global__ __launch_bounds__(256) void k_sift_prep_pair(SiftImageDev a, SiftImageDev b, uint8_t* __restrict__ xa,
                                                        int* __restrict__ cc) {
  sift_prep_tile(blockIdx.y == 0 ? a : b, blockIdx.x, xa, cc);
}
__global__ __launch_bounds__(256) void k_sift_prep_images(const SiftImageDev* __restrict__ images,
                                                          uint8_t* __restrict__ xa, int* __restrict__ cc) {
  sift_prep_tile(images[blockIdx.y], blockIdx.x, xa, cc);
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

// ---- persistent row-stripe walk, ONE direction per walk, lane = row (round 4) -------------------------------------
// Workgroup (by, chunk) owns the 512-row stripe `by` of the ROW set and walks `ct` 128-descriptor tiles of the COLUMN
// set.  Tiles are staged with LDS-DMA from the prepared copy (k_sift_prep_*), double-buffered, one barrier per tile; a
// wavefront owns 128 rows and every 32-column block of a tile: 16 MFMAs per block, in two halves of 2 row tiles x 4
// k-steps into TWO accumulator sets -- one half is multiplied while the other is scanned.
//
// The product is taken TRANSPOSED, D = B A^T (the staged column fragment is the MFMA's first operand), so that in the
// result layout a LANE is a ROW of the row set (lane & 31; the two lane halves hold columns 4 apart) and a REGISTER is a
// column of the block.  The row's running (best, second, argbest) is then three registers per lane and row tile that live
// for the whole walk, and a register of scores concerns them only if SOME lane's score beats its row's running second
// best: one v_cmp (eight issued back to back into SGPR pairs) + one scalar test per 64 scores, the fast path the
// fall-through.  With k columns seen the chance that a given row's pair changes is 2 / k, so after the first tiles almost
// every register takes the fast path.  Round 3 kept both directions' top-2 up to date for every score of ONE product
// (6 half-rate VALU per score against 8 MFMAs per 64 x 32 block: 0.17 of the dense i8 peak, bound by the scan); here the
// other direction (best row per column) is the same walk with the two sets exchanged: every tile is multiplied twice,
// as in rounds 1-2, but nothing else is done twice and a walk costs little more than its MFMAs.
//   * 512 rows per stripe: a staged tile (16 KB) feeds 256 MFMAs -- at 128 rows per stripe the walks of one 50-image
//     block pulled 157 GB through L2 (6 TB/s at the speed reached: the bound); LDS reads per block: 4 KB of column
//     fragments + 4 KB of constants for 16 MFMAs;
//   * the row set's fragments of the wavefront's 128 rows stay in registers for the whole walk (64 VGPRs);
//   * the first MFMA of a chain takes the COLUMN constants 128 sum(col) - 128^3 as its C operand (16 registers read from
//     LDS once per block, 4 consecutive columns per 16-byte read: no VALU); the row constant 128 sum(row) is one register
//     per lane and row tile and never added inside the walk -- the running values are kept relative to it;
//   * the staged tile is swizzled (16-byte chunk q of row r at position q ^ (r & 7): the DMA's lane -> source mapping
//     does it) so that the fragment reads of 32 rows at one k-chunk spread over all banks;
//   * the slow path (some lane beats its second best) is six VALU on the whole register, harmless for the lanes that do
//     not: arg <- (v > best ? code : arg), second <- med3(best, second, v), best <- max(best, v); code = 16 x column
//     block sequence number + register, wave-uniform; the column index is decoded once at the end of the walk.  The
//     masks of later registers of a group of eight are taken against the second best as it was before the earlier ones
//     were absorbed: a superset of the lanes that still beat it;
//   * strict "greater" in ascending column order inside a lane = the reference's rule (sift.cc:72-83: the first of equal
//     scores keeps the best place); the lane halves and the chunks are merged with top2_merge (equal best: lower index).
// part [nchunk][n1] int4 {best, second, arg, 0} with true scores.
constexpr int kSiftWaveRows = 64;            // rows of a wavefront (2 MFMA row tiles)
constexpr int kSiftWaves = 8;                         // wavefronts of a workgroup
constexpr int kSiftStripe = kSiftWaves * kSiftWaveRows;   // rows of a stripe (workgroup)

template <int OFF>
__device__ __forceinline__ void sift_dma16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
  // LDS destination = wave-uniform base + OFF + lane * 16; source = gsrc + OFF (the offset counts on both sides)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}
__device__ __forceinline__ void sift_dma4(const int* gsrc, int* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ uint32_t sift_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

struct SiftFrag { v4i fb[4], cq[4]; };   // one column block: 4 k-chunks of the staged descriptors, 16 column constants
template <int BLK>
__device__ __forceinline__ void sift_read_frag(SiftFrag& f, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t cc_a) {
  asm volatile("ds_read_b128 %0, %8 offset:%c12\n\tds_read_b128 %1, %9 offset:%c12\n\t"
               "ds_read_b128 %4, %13 offset:%c14\n\tds_read_b128 %5, %13 offset:%c15\n\t"
               "ds_read_b128 %2, %10 offset:%c12\n\tds_read_b128 %3, %11 offset:%c12\n\t"
               "ds_read_b128 %6, %13 offset:%c16\n\tds_read_b128 %7, %13 offset:%c17"
               : "=&v"(f.fb[0]), "=&v"(f.fb[1]), "=&v"(f.fb[2]), "=&v"(f.fb[3]), "=&v"(f.cq[0]), "=&v"(f.cq[1]), "=&v"(f.cq[2]), "=&v"(f.cq[3])
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(BLK * 4096), "v"(cc_a), "n"(BLK * 128), "n"(BLK * 128 + 32),
                 "n"(BLK * 128 + 64), "n"(BLK * 128 + 96)
               : "memory");
}

// xr / ccr: the row image in the prepared copy (its first row) and its constants; xc / ccc: the column image
__device__ __forceinline__ void sift_rows(const uint8_t* __restrict__ xr, const int* __restrict__ ccr, int n1,
                                          const uint8_t* __restrict__ xc, const int* __restrict__ ccc, int nbx,
                                          int4* __restrict__ part, int ct_per_chunk, const int chunk, const int by) {
  __shared__ __attribute__((aligned(16))) uint8_t sB[2][kSiftTile * 128];
  __shared__ __attribute__((aligned(16))) int sCc[2][kSiftTile];
  const int row0 = by * kSiftStripe;
  const int bx0 = chunk * ct_per_chunk, bx1 = min(bx0 + ct_per_chunk, nbx);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  if (bx0 >= bx1) return;
  const bool active = row0 + wave * kSiftWaveRows < n1;   // wave-uniform: a wavefront without rows only stages

  // staging: wavefront w moves 128 / kSiftWaves rows of a tile, 8 rows per instruction: lane L of instruction j fills
  // position L & 7 of tile row 8 (kDma w + j) + (L >> 3) (r = its place in the 32-row block), which holds chunk
  // (L & 7) ^ swz(r) of the descriptor;
  // wavefronts 0 / 1 move the tile's 128 constants.  swz(r) = (bit 4, bit 3, bit 1) of r: ds_read_b128 serves the lanes
  // {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of a half together (MI355X LDS banking), and over each of these sets
  // (row parity, chunk ^ swz(row)) takes 16 different values = all 64 banks once.
  constexpr int kDma = 16 / kSiftWaves;   // DMA instructions per wavefront and tile (8 rows each)
  uint32_t goff[kDma];
#pragma unroll
  for (int j = 0; j < kDma; ++j) {
    const int row = wave * (8 * kDma) + 8 * j + (lane >> 3), r = row & 31;
    goff[j] = (uint32_t)row * 128u + (uint32_t)(((lane & 7) ^ (((r >> 2) & 6) | ((r >> 1) & 1))) * 16);
  }
  auto issue = [&](int bx, int buf) {
    if (PCD_SIFT_ABLATE & 4) return;
    const uint8_t* g = xc + (size_t)bx * (kSiftTile * 128);   // wave-uniform base + 32-bit lane offsets
    uint8_t* l = sB[buf] + wave * (1024 * kDma);
#pragma unroll
    for (int j = 0; j < kDma; ++j) sift_dma16<0>(g + goff[j], l + 1024 * j);
    if (wave < 2) sift_dma4(ccc + (size_t)bx * kSiftTile + wave * 64 + lane, &sCc[buf][wave * 64]);
  };
  issue(bx0, 0);

  // the wavefront's row fragments and row constants (rows past n1 are padding or another image's rows: never written)
  v4i fa[2][4];
  int rowc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const size_t r = (size_t)row0 + wave * kSiftWaveRows + nt * 32 + lr;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint4 v = *reinterpret_cast<const uint4*>(xr + r * 128 + kk * 32 + lh * 16);
      fa[nt][kk] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
    rowc[nt] = ccr[r] + kSiftConst;
  }
  // running state per row tile, relative to the row constant: true score 0 = -rowc, code -1 = "no column" (sift.cc:66-68)
  int m1[2], m2[2], arg[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) { m1[nt] = -rowc[nt]; m2[nt] = -rowc[nt]; arg[nt] = -1; }

  // fragment addresses inside a tile buffer: row lr of a block, k-chunk 2 kk + lh at position (2 kk + lh) ^ swz(lr)
  uint32_t foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    foff[kk] = (uint32_t)lr * 128u + (uint32_t)(((2 * kk + lh) ^ (((lr >> 2) & 6) | ((lr >> 1) & 1))) * 16);
  const uint32_t sB_base = sift_lds_addr(&sB[0][0]), sCc_base = sift_lds_addr(&sCc[0][0]) + (uint32_t)lh * 16u;

  // thr: what a score has to beat to matter -- the lane's own second best, or the second best of the row's other lane
  // half minus one (refreshed once per tile): a score below the other half's second best is below the row's final one.
  int thr[2] = {m2[0], m2[1]};
  auto slow = [&](const int v, const int j, const int code_s) {
    // volatile: the instructions must stay behind the branch (they are cheap enough to be if-converted)
    int code;   // (a v_cndmask cannot take an SGPR next to VCC: the code goes through a VGPR)
    asm volatile("v_mov_b32 %[c], %[code]\n\t"
                 "v_cmp_gt_i32 vcc, %[v], %[b]\n\t"
                 "v_cndmask_b32 %[a], %[a], %[c], vcc\n\t"
                 "v_med3_i32 %[s], %[b], %[s], %[v]\n\t"
                 "v_max_i32 %[b], %[b], %[v]\n\t"
                 "v_max_i32 %[t], %[t], %[s]"
                 : [b] "+v"(m1[j]), [s] "+v"(m2[j]), [a] "+v"(arg[j]), [t] "+v"(thr[j]), [c] "=&v"(code)
                 : [v] "v"(v), [code] "s"(code_s)
                 : "vcc");
  };
  auto scan_block = [&](const v16i (&acc)[2], const int seq) {
    if (PCD_SIFT_ABLATE & 1) { asm volatile("" ::"v"(acc[0]), "v"(acc[1])); return; }
    const int base = __builtin_amdgcn_readfirstlane(seq * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        unsigned long long mk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (PCD_SIFT_ABLATE & 64) { asm volatile("s_mov_b64 %0, 0" : "=s"(mk[k])); continue; }   // no compares
          mk[k] = __builtin_amdgcn_ballot_w64(acc[j][8 * g + k] > thr[j]);
        }
        if (PCD_SIFT_ABLATE & 32) {   // compares only
#pragma unroll
          for (int k = 0; k < 8; ++k) asm volatile("" ::"s"(mk[k]));
          continue;
        }
        // one scalar test per PAIR of registers (asm goto, which would save the s_cmp behind the s_or, is miscompiled by
        // this hipcc: the asm body is dropped); unlikely: the fast path must be the fall-through
        // (a taken branch costs an instruction refetch)
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          if (!(PCD_SIFT_ABLATE & 2) && __builtin_expect((mk[k] | mk[k + 1]) != 0, 0)) {
            if (mk[k]) slow(acc[j][8 * g + k], j, base + 8 * g + k);
            if (mk[k + 1]) slow(acc[j][8 * g + k + 1], j, base + 8 * g + k + 1);
          }
        }
      }
  };
  // once per tile: the other lane half's second best (v_permlane32_swap: lanes 32 .. 63 of one copy <-> lanes 0 .. 31 of
  // the other)
  auto share_thr = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)m2[j], (unsigned)m2[j], false, false);
      const int other = lh ? (int)sw[0] : (int)sw[1];
      thr[j] = max(thr[j], other - 1);
    }
  };

  v16i acc0[2];
  // fragments of one column block: 4 k-chunks of the staged descriptors + the 16 column constants
  // (inline-asm LDS reads: for an ordinary LDS load hipcc would first drain the DMAs in flight with vmcnt(0))
  auto read_frag = [&](SiftFrag& f, const uint32_t fb_a, const uint32_t cc_a, auto blk_tag) {
    if (PCD_SIFT_ABLATE & 128) {   // fragments read once; opaque "new values" so that the MFMAs stay in the loop
      asm volatile("" : "+v"(f.fb[0]), "+v"(f.fb[1]), "+v"(f.fb[2]), "+v"(f.fb[3]), "+v"(f.cq[0]), "+v"(f.cq[1]), "+v"(f.cq[2]), "+v"(f.cq[3]));
      return;
    }
    sift_read_frag<decltype(blk_tag)::value>(f, fb_a + foff[0], fb_a + foff[1], fb_a + foff[2], fb_a + foff[3], cc_a);
    // the reads have landed (the asm ties the wait to the registers: nothing that uses them moves above it)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.fb[0]), "+v"(f.fb[1]), "+v"(f.fb[2]), "+v"(f.fb[3]), "+v"(f.cq[0]), "+v"(f.cq[1]), "+v"(f.cq[2]), "+v"(f.cq[3])
                 :: "memory");
  };
  auto mfma_block = [&](v16i (&acc)[2], const SiftFrag& f) {
    // register 4 q + i of the result = column 8 q + 4 lh + i of the block
    v16i cc;
#pragma unroll
    for (int q = 0; q < 4; ++q) { cc[4 * q] = f.cq[q][0]; cc[4 * q + 1] = f.cq[q][1]; cc[4 * q + 2] = f.cq[q][2]; cc[4 * q + 3] = f.cq[q][3]; }
    if (PCD_SIFT_ABLATE & 16) return;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.fb[kk], fa[j][kk], kk == 0 ? cc : acc[j], 0, 0, 0);
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  SiftFrag f;
  if (PCD_SIFT_ABLATE & 128) sift_read_frag<0>(f, sB_base + foff[0], sB_base + foff[1], sB_base + foff[2], sB_base + foff[3], sCc_base);
  for (int bx = bx0; bx < bx1; ++bx) {
    const int t = bx - bx0, buf = t & 1;
    if (bx + 1 < bx1) issue(bx + 1, buf ^ 1);   // the buffer's last readers passed the barrier below
    if (active) {
      share_thr();
      const uint32_t fb_a = sB_base + (uint32_t)buf * (kSiftTile * 128);
      const uint32_t cc_a = sCc_base + (uint32_t)buf * (kSiftTile * 4);
      read_frag(f, fb_a, cc_a, std::integral_constant<int, 0>{});
      mfma_block(acc0, f);
      scan_block(acc0, 4 * t);
      read_frag(f, fb_a, cc_a, std::integral_constant<int, 1>{});
      mfma_block(acc0, f);
      scan_block(acc0, 4 * t + 1);
      read_frag(f, fb_a, cc_a, std::integral_constant<int, 2>{});
      mfma_block(acc0, f);
      scan_block(acc0, 4 * t + 2);
      read_frag(f, fb_a, cc_a, std::integral_constant<int, 3>{});
      mfma_block(acc0, f);
      scan_block(acc0, 4 * t + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's part of the next tile has landed
    if (!(PCD_SIFT_ABLATE & 8)) __syncthreads();
  }
  if (!active) return;

  // ---- end of the walk: decode, merge the two lane halves of a row, write the chunk's partial
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    int best = m1[nt] + rowc[nt], second = m2[nt] + rowc[nt], a = -1;
    if (arg[nt] >= 0) {
      const int seq = arg[nt] >> 4, reg = arg[nt] & 15;
      a = (bx0 + (seq >> 2)) * kSiftTile + (seq & 3) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
    }
    const int b2 = __shfl_xor(best, 32), s2 = __shfl_xor(second, 32), a2 = __shfl_xor(a, 32);
    top2_merge(b2, s2, a2, best, second, a);
    const int grow = row0 + wave * kSiftWaveRows + nt * 32 + lr;
    if (lh == 0 && grow < n1) part[(size_t)chunk * n1 + grow] = make_int4(best, second, a, 0);
  }
}

// blockIdx.z = direction: 0 = rows of set 1 over the columns of set 2 (part12), 1 = the sets exchanged (part21).
// ct12 / ct21: column tiles per chunk of the two directions; workgroups past a direction's stripes or chunks leave.
__global__ __launch_bounds__(64 * kSiftWaves, PCD_SIFT_WGS) void k_sift_scores_stripe(const uint8_t* __restrict__ xa,
                                                            const int* __restrict__ cc, uint32_t prow1, int n1,
                                                            uint32_t prow2, int n2, int4* __restrict__ part12,
                                                            int4* __restrict__ part21, int ct12, int ct21) {
  // (one call site: the walk is ~20 KB of code)
  const bool fwd = blockIdx.z == 0;
  const uint32_t pr = fwd ? prow1 : prow2, pc = fwd ? prow2 : prow1;
  const int nr = fwd ? n1 : n2, nc = fwd ? n2 : n1;
  if ((int)blockIdx.y * kSiftStripe >= nr) return;
  sift_rows(xa + (size_t)pr * 128, cc + pr, nr, xa + (size_t)pc * 128, cc + pc, (nc + kSiftTile - 1) / kSiftTile,
            fwd ? part12 : part21, fwd ? ct12 : ct21, blockIdx.x, blockIdx.y);
}

// ---- many image pairs in one launch set (pcd_sift_match_batch_device) ----------------------------------
// All descriptors live in one arena; a pair names two row ranges of it.  blockIdx.z = 2 pair + direction; the grid's
// x / y extents are sized for the largest set of the batch, smaller ones leave early.
struct SiftPairDev {
  uint32_t prow1, n1, prow2, n2; // rows of the two images in the prepared copy, their sizes
  uint64_t part12, part21;       // int4 offsets of the pair's partial results
  uint64_t m12, m21;             // int offsets of the pair's best-match arrays
  uint64_t match;                // offset (in matches) of the pair's output list
};

__global__ __launch_bounds__(64 * kSiftWaves, PCD_SIFT_WGS) void k_sift_scores_batch(const uint8_t* __restrict__ xa,
                                                           const int* __restrict__ cc,
                                                           const SiftPairDev* __restrict__ pairs,
                                                           int4* __restrict__ part12, int4* __restrict__ part21,
                                                           int nchunk) {
  const SiftPairDev pr = pairs[blockIdx.z >> 1];
  const bool fwd = (blockIdx.z & 1) == 0;
  const uint32_t rr = fwd ? pr.prow1 : pr.prow2, rc = fwd ? pr.prow2 : pr.prow1;
  const int nr = (int)(fwd ? pr.n1 : pr.n2), nc = (int)(fwd ? pr.n2 : pr.n1);
  if ((int)blockIdx.y * kSiftStripe >= nr) return;
  const int nb = (nc + kSiftTile - 1) / kSiftTile;   // chunks past the last column tile leave inside sift_rows
  sift_rows(xa + (size_t)rr * 128, cc + rr, nr, xa + (size_t)rc * 128, cc + rc, nb,
            fwd ? part12 + pr.part12 : part21 + pr.part21, (nb + nchunk - 1) / nchunk, blockIdx.x, blockIdx.y);
}

// sift.cc:72-104: merge the per-tile triples in ascending tile order, then the distance / ratio tests.
// One launch for both directions: threads [0, n1) finish set 1 -> 2, threads [n1, n1 + n2) set 2 -> 1.
__device__ __forceinline__ void sift_finalize(const int4* __restrict__ part12, int n1, int nc12,
                                              const int4* __restrict__ part21, int n2, int nc21, float max_ratio,
                                              float max_distance, int* __restrict__ m12, int* __restrict__ m21) {
  // 16 lanes per descriptor: lane p merges chunks p, p + 16, ... in ascending order, then a 4-step butterfly.
  // On equal best scores the lower index wins, which is the earlier chunk (indices ascend with the chunk) --
  // the same winner as the reference's single ascending scan.
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, p = threadIdx.x & 15;
  if (g >= n1 + n2) return;
  const bool first = g < n1;
  const int i = first ? g : g - n1;
  const int4* __restrict__ part = first ? part12 : part21;
  const int n = first ? n1 : n2, nc = first ? nc12 : nc21;
  int best = 0, second = 0, arg = -1;
  for (int b = p; b < nc; b += 16) {
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

// chunks of a walk over nb column tiles cut into nchunk pieces that wrote a partial
__host__ __device__ __forceinline__ int sift_chunks_used(int nb, int nchunk) {
  const int ct = (nb + nchunk - 1) / nchunk;
  return ct > 0 ? (nb + ct - 1) / ct : 0;
}

__global__ __launch_bounds__(256) void k_sift_finalize(const int4* __restrict__ part12, int n1, int nc12,
                                                       const int4* __restrict__ part21, int n2, int nc21,
                                                       float max_ratio, float max_distance, int* __restrict__ m12,
                                                       int* __restrict__ m21) {
  sift_finalize(part12, n1, nc12, part21, n2, nc21, max_ratio, max_distance, m12, m21);
}

__global__ __launch_bounds__(256) void k_sift_finalize_batch(const SiftPairDev* __restrict__ pairs,
                                                             const int4* __restrict__ part12,
                                                             const int4* __restrict__ part21, int nchunk,
                                                             float max_ratio, float max_distance, int* __restrict__ m12,
                                                             int* __restrict__ m21) {
  const SiftPairDev pr = pairs[blockIdx.z];
  const int nb1 = ((int)pr.n1 + kSiftTile - 1) / kSiftTile, nb2 = ((int)pr.n2 + kSiftTile - 1) / kSiftTile;
  sift_finalize(part12 + pr.part12, (int)pr.n1, sift_chunks_used(nb2, nchunk), part21 + pr.part21, (int)pr.n2,
                sift_chunks_used(nb1, nchunk), max_ratio, max_distance, m12 + pr.m12, m21 + pr.m21);
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
  // two rounds of independent loads (all m12 entries, then all m21 entries on clamped indices) instead of `per`
  // dependent pairs in a row: the single workgroup's time is the latency of this chain (17 -> 8 us for 8192 rows)
  int jj[kCompactPer], bb[kCompactPer];
#pragma unroll
  for (int k = 0; k < kCompactPer; ++k) jj[k] = (k < per && i0 + k < n1) ? m12[i0 + k] : -1;
#pragma unroll
  for (int k = 0; k < kCompactPer; ++k) bb[k] = (cross_check && jj[k] != -1) ? m21[jj[k]] : -1;
#pragma unroll
  for (int k = 0; k < kCompactPer; ++k) {
    bool keep = jj[k] != -1;
    if (keep && cross_check) keep = bb[k] != -1 && bb[k] == i0 + k;
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
#pragma unroll
  for (int k = 0; k < kCompactPer; ++k)
    if ((flags >> k) & 1u) {
      matches[2 * pos] = (uint32_t)(i0 + k);
      matches[2 * pos + 1] = (uint32_t)jj[k];
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
  DevBuf<uint8_t> xa;   // the prepared copy (k_sift_prep_*): re-centred bytes, images padded to 128 rows, + one stripe of slack
  DevBuf<int> cc;       // its per-row constants
  DevBuf<SiftImageDev> images;
  PinnedBuf<SiftImageDev> h_images;
  DevBuf<int> m12, m21, count;
  DevBuf<int4> part12, part21;
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
  const int nb1 = (n1 + kSiftTile - 1) / kSiftTile, nb2 = (n2 + kSiftTile - 1) / kSiftTile;
  // stripe walks in both directions: enough (row tile, chunk) workgroups to fill the chip twice over
  // pcd_sift_set_tuning (tests / fuzzing) can force the number of column chunks, e.g. 1 = every stripe walks all tiles
  const int nchunk_set = g_sift_nchunk.load(std::memory_order_relaxed);
  auto chunks = [&](int nrow_tiles, int ncol_tiles) {
    const int stripes = (n1 + kSiftStripe - 1) / kSiftStripe + (n2 + kSiftStripe - 1) / kSiftStripe;
    const int want = nchunk_set > 0 ? std::min(nchunk_set, ncol_tiles) : std::min(ncol_tiles, (512 + stripes - 1) / stripes);
    (void)nrow_tiles;
    return std::max(1, want);
  };
  const int nchunk12 = chunks(nb1, nb2), nchunk21 = chunks(nb2, nb1);
  const int ct12 = (nb2 + nchunk12 - 1) / nchunk12, ct21 = (nb1 + nchunk21 - 1) / nchunk21;
  const int used12 = sift_chunks_used(nb2, nchunk12), used21 = sift_chunks_used(nb1, nchunk21);
  const uint32_t prow1 = 0, prow2 = (uint32_t)nb1 * kSiftTile;
  const size_t prows = (size_t)(nb1 + nb2) * kSiftTile + kSiftStripe;   // a stripe's row fragments are read unconditionally
  PCD_TRY(sc.xa.reserve(prows * 128)); PCD_TRY(sc.cc.reserve(prows));
  PCD_TRY(sc.part12.reserve((size_t)n1 * used12)); PCD_TRY(sc.part21.reserve((size_t)n2 * used21));
  PCD_TRY(sc.keep.reserve(n1)); PCD_TRY(sc.pos.reserve(n1));
  {
    ScopedKernelTimer t("sift_prep", s);
    hipLaunchKernelGGL(k_sift_prep_pair, dim3(std::max(nb1, nb2), 2), dim3(256), 0, s, SiftImageDev{d_d1, (uint32_t)n1, prow1},
                       SiftImageDev{d_d2, (uint32_t)n2, prow2}, sc.xa.p, sc.cc.p);
  }
  {
    ScopedKernelTimer t("sift_scores", s);
    const int ns1 = (n1 + kSiftStripe - 1) / kSiftStripe, ns2 = (n2 + kSiftStripe - 1) / kSiftStripe;
    hipLaunchKernelGGL(k_sift_scores_stripe, dim3(std::max(used12, used21), std::max(ns1, ns2), 2), dim3(64 * kSiftWaves), 0, s, sc.xa.p,
                       sc.cc.p, prow1, n1, prow2, n2, sc.part12.p, sc.part21.p, ct12, ct21);
  }
  {
    ScopedKernelTimer t("sift_finalize", s);
    hipLaunchKernelGGL(k_sift_finalize, dim3(div_up(((uint64_t)n1 + n2) * 16, 256)), dim3(256), 0, s, sc.part12.p, n1,
                       used12, sc.part21.p, n2, used21, max_ratio, max_distance, d_m12, d_m21);
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
  bool wide = false;   // a set too long for the one-workgroup compaction: those batches run pair by pair
  for (int p = 0; p < n_pairs; ++p) {
    PCD_REQUIRE(pair_ids[2 * p] < (uint32_t)n_images && pair_ids[2 * p + 1] < (uint32_t)n_images, "pair names an image outside the arena");
    const uint64_t n1 = first_row[pair_ids[2 * p] + 1] - first_row[pair_ids[2 * p]];
    const uint64_t n2 = first_row[pair_ids[2 * p + 1] + 1] - first_row[pair_ids[2 * p + 1]];
    PCD_REQUIRE(n1 < (1u << 30) && n2 < (1u << 30), "image too large");
    wide = wide || n1 > (uint64_t)1024 * kCompactPer;
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
  // the prepared copy: every image padded to a multiple of 128 rows
  std::vector<SiftImageDev> imgs((size_t)n_images);
  uint64_t prows = 0;
  uint32_t max_tiles = 0;
  for (int i = 0; i < n_images; ++i) {
    const uint64_t n = first_row[i + 1] - first_row[i];
    imgs[i] = SiftImageDev{d_arena + first_row[i] * 128, (uint32_t)n, (uint32_t)prows};
    const uint64_t tiles = (n + kSiftTile - 1) / kSiftTile;
    max_tiles = std::max<uint32_t>(max_tiles, (uint32_t)tiles);
    prows += tiles * kSiftTile;
  }
  PCD_REQUIRE(prows + kSiftStripe < (1ull << 32), "arena larger than 2^32 descriptors");
  PCD_TRY(sc.xa.reserve((prows + kSiftStripe) * 128)); PCD_TRY(sc.cc.reserve(prows + kSiftStripe));
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
      const long stripes0 = (nby0 * kSiftTile + kSiftStripe - 1) / kSiftStripe + (nbx0 * kSiftTile + kSiftStripe - 1) / kSiftStripe;   // stripes of both directions
      const long left = n_pairs - p0;
      const int nchunk_env = g_sift_nchunk.load(std::memory_order_relaxed);
      const long want = nchunk_env > 0 ? std::min<long>(nchunk_env, std::max(nbx0, nby0))
                                       : std::min<long>(std::max(nbx0, nby0), (512 + stripes0 * left - 1) / (stripes0 * left));
      const int nchunk = (int)std::max<long>(1, want);
      size_t o12 = 0, o21 = 0, om12 = 0, om21 = 0;
      int p = p0;
      for (; p < n_pairs; ++p) {
        const uint32_t a = pair_ids[2 * p], b = pair_ids[2 * p + 1];
        uint64_t n1 = first_row[a + 1] - first_row[a], n2 = first_row[b + 1] - first_row[b];
        if (n1 == 0 || n2 == 0) n1 = n2 = 0;   // an empty image: no matches (sift_test.cc:311-318); every kernel skips the pair
        const size_t need12 = (size_t)nchunk * n1, need21 = (size_t)nchunk * n2;
        // (the budget counts 16-byte units: both partial arrays are int4; 2 z-slices per pair in the scores launch)
        if (p > p0 && (o12 + need12 + o21 + need21 > budget || p - p0 >= 32767)) break;
        tab[p] = SiftPairDev{imgs[a].prow, (uint32_t)n1, imgs[b].prow, (uint32_t)n2, o12, o21, om12, om21, match_offset[p]};
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
  PCD_TRY(sc.h_pairs.reserve(n_pairs)); PCD_TRY(sc.h_images.reserve(n_images)); PCD_TRY(sc.images.reserve(n_images));
  std::memcpy(sc.h_pairs.p, tab.data(), sizeof(SiftPairDev) * (size_t)n_pairs);
  std::memcpy(sc.h_images.p, imgs.data(), sizeof(SiftImageDev) * (size_t)n_images);
  PCD_HIP_TRY(hipMemcpyAsync(sc.pairs.p, sc.h_pairs.p, sizeof(SiftPairDev) * (size_t)n_pairs, hipMemcpyHostToDevice, s));
  PCD_HIP_TRY(hipMemcpyAsync(sc.images.p, sc.h_images.p, sizeof(SiftImageDev) * (size_t)n_images, hipMemcpyHostToDevice, s));
  PCD_HIP_TRY(hipEventRecord(sc.ev_tab, s));
  sc.tab_pending = true;
  if (max_tiles) {
    ScopedKernelTimer t("sift_prep", s);
    hipLaunchKernelGGL(k_sift_prep_images, dim3(max_tiles, n_images), dim3(256), 0, s, sc.images.p, sc.xa.p, sc.cc.p);
  }
  for (size_t k = 0; k + 1 < cut.size(); ++k) {
    const int p0 = cut[k], np = cut[k + 1] - cut[k], nchunk = cut_nchunk[k];
    uint32_t mx1 = 0, mx2 = 0, mxsum = 0;
    for (int p = p0; p < p0 + np; ++p) {
      mx1 = std::max(mx1, tab[p].n1); mx2 = std::max(mx2, tab[p].n2); mxsum = std::max(mxsum, tab[p].n1 + tab[p].n2);
    }
    if (mx1 && mx2) {
      {
        ScopedKernelTimer t("sift_scores", s);
        hipLaunchKernelGGL(k_sift_scores_batch, dim3(nchunk, (std::max(mx1, mx2) + kSiftStripe - 1) / kSiftStripe, 2 * np),
                           dim3(64 * kSiftWaves), 0, s, sc.xa.p, sc.cc.p, sc.pairs.p + p0, sc.part12.p, sc.part21.p, nchunk);
      }
      {
        ScopedKernelTimer t("sift_finalize", s);
        hipLaunchKernelGGL(k_sift_finalize_batch, dim3(div_up((uint64_t)mxsum * 16, 256), 1, np), dim3(256), 0, s,
                           sc.pairs.p + p0, sc.part12.p, sc.part21.p, nchunk, max_ratio, max_distance, sc.m12.p, sc.m21.p);
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
