// stencil_kernel.h -- the per-query stencil kernel of the NN search (included by nn.hip).
//
// The brick kernel (brick_kernel.h) compares every query of a brick with every point of the brick's 6x6x6-cell
// region: ~1 300 point-query pairs per query on surface-like clouds, of which a query's own neighbourhood is a small
// part.  This kernel gives every query its OWN region and grows it only for the queries that need it:
//
//   stage <0,K1>    cube of +-K1 cells around the query's home cell
//   stage <K1,K2>   the shell between that cube and the +-K2 cube, only for queries stage 1 could not prove
//   ...             whatever is still unproven goes to the exact pyramid search (k_nn_fallback)
//
// (cubes are grown outward to whole 2x2 (y,z) cell quads, grid.h: a quad row is one contiguous point range along x,
// so the +-1 cube is 4 ranges, the +-2 shell 13, the +-3 shell 25 -- at most two per quad row).
//
// 16 lanes per query, 4 queries per wavefront: lanes = cloud points, as in the other two kernels, but the query is
// per lane group instead of wave-uniform.  Each of a group's first (K+1)^2 lanes looks up the point ranges of one quad
// row; the ranges are cut into STEPS of 16 consecutive records (a step may run past the end of its range into the
// next cells' points: real points, harmless extra candidates; `sorted` ends with kSortedSpare copies of the last
// record) and the steps' first-record indices go into a per-group table in LDS.  The main loop then is table-driven
// and identical for all four groups: one ds_read_b128 gives the next four steps, each step is ONE 16-byte load per
// lane (256 contiguous bytes per group) and FLANN's float distance + one v_min_f64 on the packed key (brick_kernel.h
// compare_point).  Groups with fewer steps than the wavefront's longest re-read record 0 (a real point).  The group
// minimum is four DPP row rotations.  A result is final iff best < (distance to the faces of the scanned cube -
// slack)^2 (nn.hip header); the others are appended, in chunks of 64 slots per wavefront and atomic, to the next
// stage's list (their position in the sorted query array) or, from the last stage, to the fallback list (their
// query id).  A query whose step table would overflow (kStCap steps = 1024 points: duplicated / very dense cells)
// is sent straight to the fallback list, which needs no region.
#pragma once

namespace pcd {

constexpr int kStLanes = 16;             // lanes per query
constexpr int kStQ = 64 / kStLanes;      // queries per wavefront iteration
constexpr int kStCap = 96;               // steps per query and stage (multiple of 4)
#ifndef PCD_ST_WAVES
#define PCD_ST_WAVES 16
#endif
constexpr int kStWaves = PCD_ST_WAVES;   // wavefronts per workgroup
constexpr uint32_t kStChunk = 4;         // consecutive block iterations a workgroup takes at a time
constexpr uint32_t kStNone = 0xFFFFFFFFu;

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// minimum over the 16 lanes of a DPP row of packed keys held as f64 (brick_kernel.h min_key_f64), in every lane
__device__ __forceinline__ double row_min_key(double v) {
  v = min_key_f64(v, dpp_f64<0x128>(v));   // row_ror:8
  v = min_key_f64(v, dpp_f64<0x124>(v));   // row_ror:4
  v = min_key_f64(v, dpp_f64<0x122>(v));   // row_ror:2
  v = min_key_f64(v, dpp_f64<0x121>(v));   // row_ror:1
  return v;
}

__device__ __forceinline__ void stencil_compare(const float4 p, float qx, float qy, float qz, double& best) {
  const float d = l2_simple3(qx, qy, qz, p.x, p.y, p.z);
  best = min_key_f64(best, __builtin_bit_cast(double, make_key(d, __float_as_uint(p.w))));
}

struct StencilOut {
  uint32_t* next_list;      // chunked list of this stage's unproven queries (sorted positions, or query ids when
  uint32_t* next_count;     //   `next_is_fallback`); count = slots handed out (chunks of 64, unused slots = kStNone)
  uint32_t* fb_list;        // fallback list (query ids) for step-table overflows; == next_list on the last stage
  uint32_t* fb_count;
  int next_is_fallback;
};

// BRICK = 1: the stage runs behind the brick kernel (nn.hip run_grid): what has been scanned already is the query's
// brick region (whole quad rows over the region's x-range, brick_kernel.h brick_load_meta), the tentative best
// distance d_t the brick kernel found comes in as the prior key, and every quad row is CLIPPED TO THE BALL of radius
// d_t around the query -- a point farther than d_t cannot be the nearest neighbour, so the rows the ball misses are
// skipped and the others shrink to the cells its x-extent touches (rounding: radius widened by 2e-6 relative and by the
// binning slack; cell bounds widened by the slack).  KPREV is unused then.
template <int KPREV, int K, int BRICK>
__global__ __launch_bounds__(64 * kStWaves, 4) void k_nn_stencil(GridParams g, BrickParams b, const float4* __restrict__ sorted,
                                                        const uint32_t* __restrict__ cell_start,
                                                        const float4* __restrict__ qsorted,
                                                        uint64_t* __restrict__ ksorted,
                                                        const uint32_t* __restrict__ list,   // NULL: positions 0..count-1
                                                        const uint32_t* __restrict__ count_ptr,
                                                        uint64_t* __restrict__ keys, StencilOut out,
                                                        NnCounters* __restrict__ ctr, int collect_stats) {
  static_assert(K >= 1 && K <= 3 && KPREV < K, "one quad row per lane of a 16-lane group: (K + 1)^2 <= 16");
  constexpr int NQR = K + 1;   // quad rows per axis of the +-K cube: ((c + K) >> 1) - ((c - K) >> 1) + 1 for every c
  constexpr bool TWO = KPREV > 0 || BRICK;   // a quad row may have two x-ranges (left and right of the scanned part)
  __shared__ __attribute__((aligned(16))) uint32_t s_tbl[kStWaves][kStQ][kStCap];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = lane >> 4, sub = lane & 15;
  const uint32_t count = *count_ptr;
  const uint32_t niter = (count + kStQ - 1) / kStQ;
  uint32_t* const tb = s_tbl[wave][grp];
  uint32_t ob = 0, oleft = 0;   // this wavefront's chunk of the output list
  unsigned long long st_steps = 0, st_q = 0;

  // Work split.  A BLOCK ITERATION = kStWaves consecutive wavefront iterations = 4 kStWaves queries that are neighbours
  // in the cell-sorted order.  Blocks b, b+8, ... share an XCD: each of the 8 classes walks its own contiguous eighth
  // of the block iterations, in chunks of kStChunk per workgroup.
  const uint32_t nbi = (niter + kStWaves - 1) / kStWaves;
  const uint32_t ncls = (gridDim.x & 7u) == 0 ? 8u : 1u;
  const uint32_t cls = ncls == 8u ? (blockIdx.x & 7u) : 0u, bic = ncls == 8u ? (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t per = (nbi + ncls - 1) / ncls, nblk = gridDim.x / ncls;
  const uint32_t bi_end = min(nbi, (cls + 1) * per);
  for (uint32_t cb = cls * per + bic * kStChunk; cb < bi_end; cb += nblk * kStChunk)
  for (uint32_t bi = cb; bi < min(cb + kStChunk, bi_end); ++bi) {
    const uint32_t it = bi * kStWaves + (uint32_t)wave;
    if (it >= niter) break;
    const uint32_t e = it * kStQ + grp;
    uint32_t j = e < count ? (list ? list[e] : e) : kStNone;
    const bool valid = j != kStNone;
    j = valid ? j : 0u;                       // position 0 exists whenever count > 0
    const float4 qr = qsorted[j];
    const uint64_t prior = ksorted[j];
    const int cx = cell_coord(qr.x, g.origin[0], g.inv_h, g.dims[0]);
    const int cy = cell_coord(qr.y, g.origin[1], g.inv_h, g.dims[1]);
    const int cz = cell_coord(qr.z, g.origin[2], g.inv_h, g.dims[2]);
    // ---- the lane's quad row and its one or two x-ranges ------------------------------------------------------
    const int yqa = (cy - K) >> 1, zqa = (cz - K) >> 1;
    const int ry = sub % NQR, rz = sub / NQR;
    const int yq = yqa + ry, zq = zqa + rz;
    const bool row_ok = valid && sub < NQR * NQR && yq >= 0 && yq < g.qdims[0] && zq >= 0 && zq < g.qdims[1];
    const int nx = g.dims[0];
    int xlo = max(cx - K, 0), xhi = min(cx + K + 1, nx);   // the cube's cells along x
    bool inner = false;
    int ix0 = 0, ix1 = 0;                                  // x-cells of an inner row that are scanned already
    if (BRICK) {
      const int bx = cx / b.Bx, by = (cy + b.S) / b.B, bz = (cz + b.S) / b.B;
      const int ry0 = by * b.B - b.S - b.R, ry1 = ry0 + b.B + 2 * b.R, rz0 = bz * b.B - b.S - b.R, rz1 = rz0 + b.B + 2 * b.R;
      inner = 2 * yq + 2 > ry0 && 2 * yq < ry1 && 2 * zq + 2 > rz0 && 2 * zq < rz1;   // the brick kernel takes whole quads
      ix0 = max(bx * b.Bx - b.R, 0); ix1 = min(bx * b.Bx + b.Bx + b.R, nx);
      // clip the row to the ball of the tentative distance
      const double dt = (double)__uint_as_float((uint32_t)(prior >> 32));
      const double sl = (double)g.slack, h = (double)g.h;
      const double r = sqrt(dt * (1.0 + 4e-6)) + sl, r2 = r * r;   // inf for a query without a tentative result
      const double ya = (double)g.origin[1] + (double)(2 * yq) * h - sl, yb = (double)g.origin[1] + (double)(2 * yq + 2) * h + sl;
      const double za = (double)g.origin[2] + (double)(2 * zq) * h - sl, zb = (double)g.origin[2] + (double)(2 * zq + 2) * h + sl;
      const double dy = fmax(fmax(ya - (double)qr.y, (double)qr.y - yb), 0.0), dz = fmax(fmax(za - (double)qr.z, (double)qr.z - zb), 0.0);
      const double dyz2 = dy * dy + dz * dz;
      if (dyz2 > r2) {
        xhi = xlo;                                         // the ball misses the row
      } else {
        const double ex = sqrt(r2 - dyz2) + 2.0 * sl;      // half-extent along x (inf stays inf)
        const double ca = floor(((double)qr.x - ex - (double)g.origin[0]) / h), cb2 = floor(((double)qr.x + ex - (double)g.origin[0]) / h);
        xlo = max(xlo, (int)fmin(fmax(ca, -1.0), (double)nx));
        xhi = min(xhi, (int)fmin(fmax(cb2, -1.0), (double)nx) + 1);
        xhi = max(xhi, xlo);
      }
    } else if (KPREV > 0) {
      inner = yq >= ((cy - KPREV) >> 1) && yq <= ((cy + KPREV) >> 1) && zq >= ((cz - KPREV) >> 1) && zq <= ((cz + KPREV) >> 1);
      ix0 = max(cx - KPREV, 0); ix1 = min(cx + KPREV + 1, nx);
    }
    const int xa0 = xlo, xb1 = xhi;
    const int xa1 = inner ? min(max(ix0, xlo), xhi) : xhi;
    const int xb0 = inner ? min(max(ix1, xlo), xhi) : xhi;
    const uint32_t rowbase = (uint32_t)quad_row_base(g, row_ok ? yq : 0, row_ok ? zq : 0);
    // every lane issues every load (clamped rows, results masked): no loads under divergent branches
    const uint32_t sA = cell_start[rowbase + 4u * (uint32_t)xa0], eA = cell_start[rowbase + 4u * (uint32_t)xa1];
    uint32_t sB = 0, eB = 0;
    if (TWO) { sB = cell_start[rowbase + 4u * (uint32_t)xb0]; eB = cell_start[rowbase + 4u * (uint32_t)xb1]; }
    const uint32_t nA = row_ok ? (eA - sA + 15u) >> 4 : 0u;
    const uint32_t nB = (TWO && row_ok) ? (eB - sB + 15u) >> 4 : 0u;
    const uint32_t n = nA + nB;
    // ---- steps of the group: prefix over its 16 lanes, total per group, longest group of the wavefront --------
    uint32_t inc = n;
    PCD_DPP_STEP(op_add_u32, inc, 0u, 0x111, 0xf);
    PCD_DPP_STEP(op_add_u32, inc, 0u, 0x112, 0xf);
    PCD_DPP_STEP(op_add_u32, inc, 0u, 0x114, 0xf);
    PCD_DPP_STEP(op_add_u32, inc, 0u, 0x118, 0xf);
    const uint32_t T0 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 15), T1 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 31),
                   T2 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 47), T3 = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    const uint32_t Tg = grp == 0 ? T0 : grp == 1 ? T1 : grp == 2 ? T2 : T3;
    const bool ovf = Tg > (uint32_t)kStCap;   // per group
    const uint32_t c0 = T0 > (uint32_t)kStCap ? 0u : T0, c1 = T1 > (uint32_t)kStCap ? 0u : T1,
                   c2 = T2 > (uint32_t)kStCap ? 0u : T2, c3 = T3 > (uint32_t)kStCap ? 0u : T3;
    const uint32_t Tmax4 = (max(max(c0, c1), max(c2, c3)) + 3u) & ~3u;   // wave-uniform
    // ---- step table: record 0 everywhere (groups with fewer steps re-read it), then every lane's own steps -----
    for (uint32_t t = sub; t < Tmax4; t += 16) tb[t] = 0u;
    if (!ovf) {
      uint32_t pos = inc - n;
      // entries are BYTE offsets of a step's first record (the host keeps this kernel to clouds below 2^28 records)
      for (uint32_t i = 0; i < nA; ++i) tb[pos + i] = (sA + 16u * i) << 4;
      pos += nA;
      for (uint32_t i = 0; i < nB; ++i) tb[pos + i] = (sB + 16u * i) << 4;
    }
    // ---- main loop (LDS operations of one wavefront complete in order: the fence only stops the compiler) -------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    double best = __builtin_bit_cast(double, kKeyInit);
    if (Tmax4) {
      // One asm block for the whole loop: the loads of the next four steps are in flight while the current four are
      // compared (counted s_waitcnt vmcnt(4)), which C++ cannot express -- hipcc waits for every load before its
      // first use and may copy a register that is still a load's destination.  Registers v40..v79 belong to the
      // block: two sets of four 16-byte records (v[40:55], v[56:71]), the four table entries (v[72:75]) and two key
      // pairs (v[76:77], v[78:79]).  Per step: 1 v_add (byte offset of the lane's record) + 1 load + 3 v_sub +
      // 3 v_mul + 2 v_add + 1 v_mov (the index into the low half of the key pair) + 1 v_min_f64 -- the same IEEE
      // operations in the same order as l2_simple3 (grid.h), two points interleaved.
      uint32_t nq = Tmax4 >> 2;                 // quads of steps, wave-uniform
      uint32_t tba = lds_addr(tb);
      const uint32_t lo16 = (uint32_t)sub << 4;
#define PCD_ST_ISSUE(R0, R1, R2, R3)                                                           \
  "ds_read_b128 v[72:75], %[tb]\n\tv_add_u32_e32 %[tb], 16, %[tb]\n\ts_waitcnt lgkmcnt(0)\n\t"   \
  "v_add_u32_e32 v72, v72, %[lo]\n\tv_add_u32_e32 v73, v73, %[lo]\n\t"                         \
  "v_add_u32_e32 v74, v74, %[lo]\n\tv_add_u32_e32 v75, v75, %[lo]\n\t"                         \
  "global_load_dwordx4 " R0 ", v72, %[base]\n\tglobal_load_dwordx4 " R1 ", v73, %[base]\n\t"     \
  "global_load_dwordx4 " R2 ", v74, %[base]\n\tglobal_load_dwordx4 " R3 ", v75, %[base]\n\t"
#define PCD_ST_CMP2(P0, P1, P2, P3, R0, R1, R2, R3)                                            \
  "v_sub_f32 v" #P0 ", %[qx], v" #P0 "\n\tv_sub_f32 v" #R0 ", %[qx], v" #R0 "\n\t"             \
  "v_sub_f32 v" #P1 ", %[qy], v" #P1 "\n\tv_sub_f32 v" #R1 ", %[qy], v" #R1 "\n\t"             \
  "v_sub_f32 v" #P2 ", %[qz], v" #P2 "\n\tv_sub_f32 v" #R2 ", %[qz], v" #R2 "\n\t"             \
  "v_mul_f32 v" #P0 ", v" #P0 ", v" #P0 "\n\tv_mul_f32 v" #R0 ", v" #R0 ", v" #R0 "\n\t"       \
  "v_mul_f32 v" #P1 ", v" #P1 ", v" #P1 "\n\tv_mul_f32 v" #R1 ", v" #R1 ", v" #R1 "\n\t"       \
  "v_mul_f32 v" #P2 ", v" #P2 ", v" #P2 "\n\tv_mul_f32 v" #R2 ", v" #R2 ", v" #R2 "\n\t"       \
  "v_add_f32 v" #P0 ", v" #P0 ", v" #P1 "\n\tv_add_f32 v" #R0 ", v" #R0 ", v" #R1 "\n\t"       \
  "v_mov_b32 v76, v" #P3 "\n\tv_mov_b32 v78, v" #R3 "\n\t"                                   \
  "v_add_f32 v77, v" #P0 ", v" #P2 "\n\tv_add_f32 v79, v" #R0 ", v" #R2 "\n\t"                 \
  "v_min_f64 %[b], %[b], v[76:77]\n\tv_min_f64 %[b], %[b], v[78:79]\n\t"
#define PCD_ST_CMP_A PCD_ST_CMP2(40, 41, 42, 43, 44, 45, 46, 47) PCD_ST_CMP2(48, 49, 50, 51, 52, 53, 54, 55)
#define PCD_ST_CMP_B PCD_ST_CMP2(56, 57, 58, 59, 60, 61, 62, 63) PCD_ST_CMP2(64, 65, 66, 67, 68, 69, 70, 71)
#define PCD_ST_NEXT(target, cond) "s_sub_u32 %[n], %[n], 1\n\ts_cmp_eq_u32 %[n], 0\n\ts_cbranch_" cond " " target "\n\t"
      asm volatile(
          PCD_ST_ISSUE("v[40:43]", "v[44:47]", "v[48:51]", "v[52:55]")
          PCD_ST_NEXT("2f", "scc1")
          "1:\n\t"
          PCD_ST_ISSUE("v[56:59]", "v[60:63]", "v[64:67]", "v[68:71]")
          "s_waitcnt vmcnt(4)\n\t"
          PCD_ST_CMP_A
          PCD_ST_NEXT("3f", "scc1")
          PCD_ST_ISSUE("v[40:43]", "v[44:47]", "v[48:51]", "v[52:55]")
          "s_waitcnt vmcnt(4)\n\t"
          PCD_ST_CMP_B
          PCD_ST_NEXT("1b", "scc0")
          "2:\n\t"   // set A in flight, nothing behind it
          "s_waitcnt vmcnt(0)\n\t"
          PCD_ST_CMP_A
          "s_branch 4f\n"
          "3:\n\t"   // set B in flight
          "s_waitcnt vmcnt(0)\n\t"
          PCD_ST_CMP_B
          "4:\n\t"
          : [b] "+v"(best), [tb] "+v"(tba), [n] "+s"(nq)
          : [lo] "v"(lo16), [base] "s"(sorted), [qx] "v"(qr.x), [qy] "v"(qr.y), [qz] "v"(qr.z)
          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
            "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",
            "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "scc", "memory");
#undef PCD_ST_ISSUE
#undef PCD_ST_CMP2
#undef PCD_ST_CMP_A
#undef PCD_ST_CMP_B
#undef PCD_ST_NEXT
    }
    best = row_min_key(best);
    // ---- epilogue: lane 0 of each group ------------------------------------------------------------------------
    const uint64_t mine = min_u64(__builtin_bit_cast(uint64_t, best), prior);
    int b0[3] = {max(cx - K, 0), max(2 * yqa, 0), max(2 * zqa, 0)};
    int b1[3] = {min(cx + K + 1, g.dims[0]), min(2 * (yqa + NQR), g.dims[1]), min(2 * (zqa + NQR), g.dims[2])};
    const double bound = proven_bound(g, qr.x, qr.y, qr.z, b0, b1);
    const double bd = (double)__uint_as_float((uint32_t)(mine >> 32));
    const bool head = valid && sub == 0;
    const bool unproven = head && !ovf && !(bd < bound);
    const uint32_t qid = __float_as_uint(qr.w);
    if (head) {
      keys[qid] = mine;                       // final, or the starting bound of the next stage / the fallback
      if (unproven || ovf) ksorted[j] = mine;
    }
    if (head && ovf) out.fb_list[atomicAdd(out.fb_count, 1u)] = qid;   // rare: one atomic each
    const unsigned long long um = __ballot(unproven);
    if (um) {
      const uint32_t k = (uint32_t)__popcll(um);
      if (k > oleft) {   // wave-uniform: leave the chunk (rest = sentinel) and reserve the next one
        if (lane < (int)oleft) out.next_list[ob + lane] = kStNone;
        uint32_t nb = 0;
        if (lane == 0) nb = atomicAdd(out.next_count, 64u);
        ob = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
        oleft = 64u;
      }
      if (unproven) out.next_list[ob + __popcll(um & ((1ull << lane) - 1))] = out.next_is_fallback ? qid : j;
      ob += k;
      oleft -= k;
    }
    if (collect_stats) { st_steps += c0 + c1 + c2 + c3; st_q += (uint32_t)__popcll(__ballot(head)); }
  }
  if (lane < (int)oleft) out.next_list[ob + lane] = kStNone;
  if (collect_stats && lane == 0) {
    atomicAdd(&ctr->staged_points, st_steps * 16ull);
    atomicAdd(&ctr->pair_evals, st_steps * 16ull);
    atomicAdd(&ctr->brick_groups, st_q);
  }
}

}  // namespace pcd
