// cloud.h -- device-resident LiDAR cloud index (replaces PointCloudProcess +
// Kdtree of the reference: lidar/ply.cc:9-57, lidar/kdtree.cc:5-8).
//
// HBM layout (all hipMalloc'ed on one device):
//   pts4 [n]   float4 {x,y,z,0}     original (post-NaN-filter) row order  -- epilogue gathers, brute force
//   pn8  [n]   2 x float4 {x,y,z,0, nx,ny,nz,0}  original order           -- epilogue gathers: a winner's point AND
//                                   normal in ONE 32-byte record (one cache line instead of a line of pts4 and one of nrm4)
//   sorted [m] float4 {x,y,z,bits(global_idx)}  finite rows in cell-sorted order (x fastest):
//              one 16-B record per lane per load; a row of cells along x is one contiguous range
//   cell_start [ncells+1] uint32    exclusive prefix of per-cell counts
//   blk_aabb [nodes][8] float       the AABB pyramid of the exact fallback.  Level 0 = LEAVES: every sub-block of
//                                   2x2x2 cells -- an x-range of 2 cells of ONE quad row, i.e. one contiguous point
//                                   range -- as {lo.xyz, hi.x | hi.y, hi.z, bits(first point), bits(count)}; level
//                                   k+1 = 4x4x4 nodes of level k as {lo.xyz, hi.xyz, 0, 0}, up to the first level with <= 64 nodes
//                                   (the walk starts at a virtual top over all of them, PyramidParams below).
//                                   An expansion tests 64 children with the exact float bound; a leaf is scanned.
#pragma once
#include "common.h"

namespace pcd {

constexpr int kSortedSpare = 16;  // records behind the last point of `sorted` (copies of it): range reads may overrun
constexpr int kBlockCells = 4;  // cells per block edge (blocks carry tight AABBs for the exact fallback)

struct GridParams {
  float origin[3];
  float h, inv_h;
  int dims[3];    // cells
  int qdims[2];   // (dims[1]+1)/2, (dims[2]+1)/2: 2x2 (y,z) cell quads, see grid.h cell_index
  int bdims[3];   // blocks of kBlockCells^3 cells
  float slack;    // metres: bound on binning rounding, see nn.hip
};

// 64-ary AABB pyramid: level 0 = leaves (2x2x2-cell sub-blocks), level k+1 = 4x4x4 nodes of level k, up to the first
// level with at most 64 nodes (level nlev-2); level nlev-1 is a VIRTUAL top whose children are all nodes of level
// nlev-2, one per lane.  The real levels live in blk_aabb (8 floats per node) at off[level].
constexpr int kMaxPyrLevels = 10;
struct PyramidParams {
  int nlev;                       // >= 2
  int dims[kMaxPyrLevels][3];
  uint32_t off[kMaxPyrLevels];    // node offset of each level
};

struct QueryScratch;  // nn.hip

// ownership of a global index by a shard: local row or 0xFFFFFFFFFFFFFFFF
struct ShardIndex {
  uint32_t base, stride;
  uint64_t n;
  const uint32_t* g2l;   // NULL: base / stride arithmetic
  uint64_t g2l_n;
};
__device__ __forceinline__ uint64_t shard_local_row(const ShardIndex& si, uint32_t gi) {
  if (si.g2l) {
    const uint32_t l = gi < si.g2l_n ? si.g2l[gi] : 0xFFFFFFFFu;
    return l == 0xFFFFFFFFu ? ~0ull : (uint64_t)l;
  }
  if (gi < si.base || (gi - si.base) % si.stride != 0) return ~0ull;
  const uint64_t li = (uint64_t)(gi - si.base) / si.stride;
  return li < si.n ? li : ~0ull;
}

}  // namespace pcd

struct pcd_cloud {
  int device = 0;
  uint64_t n = 0;  // rows kept
  uint64_t m = 0;  // finite rows indexed in the grid
  uint32_t index_base = 0, index_stride = 1;
  // shards with an arbitrary row -> global index map (pcd_cloud_create_sharded): global index of local row i, and the
  // inverse over the whole cloud (0xFFFFFFFF = not a row of this shard).  Empty: global = index_base + i * index_stride.
  pcd::DevBuf<uint32_t> row_index, g2l;
  uint64_t g2l_n = 0;
  pcd::DevBuf<float4> pts4, pn8, sorted;   // pn8: 2 float4 per row
  pcd::DevBuf<uint32_t> cell_start;
  pcd::DevBuf<float> blk_aabb;
  pcd::GridParams grid{};
  pcd::PyramidParams pyr{};
  uint64_t ncells = 0, nblocks = 0, occupied = 0;
  float bb_lo[3] = {0, 0, 0}, bb_hi[3] = {0, 0, 0};   // tight bounds of the finite rows (valid when m > 0)
  double build_ms = 0;
  pcd::QueryScratch* scratch = nullptr;
  pcd::ShardIndex shard_index() const { return pcd::ShardIndex{index_base, index_stride, n, g2l.p, g2l_n}; }
};

namespace pcd {
// pcd_cloud_create with an explicit global index per row (row_index[n], host; NULL = index_base / index_stride) over
// a cloud of global_n rows in total
pcd_status cloud_create_indexed(const float* xyz, const float* nrm, uint64_t n, const pcd_cloud_options* opts,
                                const uint32_t* row_index, uint64_t global_n, pcd_cloud** out);
}
