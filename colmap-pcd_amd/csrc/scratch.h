// scratch.h -- grow-only per-cloud device scratch shared by nn.hip and assoc.hip
#pragma once
#include "cloud.h"

namespace pcd {

struct NnCounters {
  unsigned long long brick_groups, staged_points, fallback_queries, fallback_points, pair_evals;
  unsigned int nitems, fb_count;
  unsigned int pad[2];   // pad[0] = length of the squeezed fallback list
  unsigned int n_in_grid;     // brick-sorted positions 0 .. n_in_grid-1 are the finite queries inside the grid
  unsigned int pad2[3];
  unsigned int fb_max_steps, fb_max_leaves;   // statistics passes: longest walk of k_nn_fallback (steps, leaf scans)
  unsigned long long fb_steps, fb_leaves;
};

struct QueryScratch {
  DevBuf<float4> qf4;          // (float)query, w = 1 valid / 0 not finite
  DevBuf<uint64_t> keys;       // host-API result keys
  DevBuf<uint32_t> bk_keys, bk_vals;   // (brick id, query id) pairs, unsorted | sorted halves
  DevBuf<uint32_t> bk_item;            // work items per tile of 1024 sorted positions
  // counting-sort bookkeeping (nn.hip): slot of every brick with a non-empty halo region (kSlotNone otherwise), valid
  // for the brick geometry in bk_slot_key; per batch the slots' query counters, in-tile prefixes and tile totals
  DevBuf<uint32_t> bk_slot, bk_chist;   // bk_chist: coarse histogram | bucket starts | bucket cursors
  int bk_slot_key[5] = {0, 0, 0, 0, 0};
  uint32_t bk_nslots = 0;
  DevBuf<float4> qsorted;      // brick-sorted query records {x,y,z,bits(query id)}
  DevBuf<uint64_t> ksorted;    // their incoming keys, same order
  DevBuf<uint4> items;         // {first query, brick x, brick y, brick z | count << 28}
  DevBuf<uint32_t> fb_list, fb_dense;   // fallback list as the brick kernel fills it (chunked) / squeezed
  DevBuf<NnCounters> counters;
  DevBuf<char> tmp;
  DevBuf<double> d_q;          // staging of host queries
  DevBuf<uint32_t> d_idx; DevBuf<float> d_sq; DevBuf<uint8_t> d_found;
  // host-API association staging
  DevBuf<double> a_mr, a_xyz, a_abcd, a_dist, a_angle, a_d2p;
  DevBuf<uint8_t> a_type;
  DevBuf<uint64_t> a_keys;
  // staged host path (pcd_assoc_staging / pcd_associate_staged): pinned inputs / results, compaction scratch
  PinnedBuf<double> h_q, h_mr;
  PinnedBuf<pcd_assoc_hit> h_hits;
  PinnedBuf<uint32_t> h_count;
  // staged host path: copies in / compute / copies out on their own streams, chunk by chunk (assoc.hip)
  static constexpr int kStageChunks = 4;
  hipStream_t st_in = nullptr, st_comp = nullptr, st_out = nullptr, st_cnt = nullptr;
  hipEvent_t ev_in[kStageChunks] = {}, ev_comp[kStageChunks] = {}, ev_cnt[kStageChunks] = {};
  ~QueryScratch() {
    for (int k = 0; k < kStageChunks; ++k) {
      if (ev_in[k]) (void)hipEventDestroy(ev_in[k]);
      if (ev_comp[k]) (void)hipEventDestroy(ev_comp[k]);
      if (ev_cnt[k]) (void)hipEventDestroy(ev_cnt[k]);
    }
    if (st_in) (void)hipStreamDestroy(st_in);
    if (st_comp) (void)hipStreamDestroy(st_comp);
    if (st_out) (void)hipStreamDestroy(st_out);
    if (st_cnt) (void)hipStreamDestroy(st_cnt);
  }
  DevBuf<pcd_assoc_hit> d_hits;
  DevBuf<uint32_t> hit_pos, hit_count;
};

inline QueryScratch* scratch_of(pcd_cloud* c) {
  if (!c->scratch) c->scratch = new QueryScratch();
  return c->scratch;
}

}  // namespace pcd
