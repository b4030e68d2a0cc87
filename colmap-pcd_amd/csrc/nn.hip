// nn.hip -- exact nearest neighbour of 3D feature points in the LiDAR cloud.
//
// Replaces lidar/kdtree.cc:10-21 (Kdtree::GetClosestPoint -> pcl::KdTreeFLANN
// nearestKSearch, k = 1) as called from lidar/ply.cc:90-107.  Arithmetic is
// FLANN's L2_Simple<float> on (x,y,z): ((dx*dx) + dy*dy) + dz*dz in float32
// with separate multiplies and adds (this file is compiled with
// -ffp-contract=off), query = (float)double (ply.cc:92).  Result = the
// minimum of key = float_bits(d) << 32 | index over all points, i.e. ties on
// the float distance go to the lowest index; a point only counts if
// d < FLT_MAX (FLANN's initial worst distance).
//
// Three kernels, all with lanes = cloud points and wave-uniform queries, each
// finishing with a per-wavefront min-reduction of packed keys:
//   k_nn_bruteforce  all pairs; LDS-tiled; reference for the other two.
//   k_nn_brick       one wavefront per group of <= G queries that fall into
//                    the same brick of B^3 grid cells: the cell rows of the
//                    brick grown by R cells are streamed through an LDS tile
//                    (16-B records, coalesced row ranges) and every query of
//                    the group is compared with every staged point.  A query
//                    is final when its best distance is provably smaller than
//                    its distance to the boundary of the staged region.
//   k_nn_fallback    one wavefront per remaining query: depth-first, nearest-first
//                    descent of the 64-ary pyramid of tight AABBs over the grid
//                    (level 0 = leaves of 2x2x2 cells, one contiguous point range each;
//                    a virtual top over the first level with <= 64 nodes), the 64
//                    children of a node one per lane, pruned with the exact float
//                    lower bound.  Exact for any query position, also outside the grid.
// Query bookkeeping in front of k_nn_brick: a two-level counting sort on the brick id
// (k_bk_slots, k_bk_scatter, k_bk_count, k_bk_emit) that yields the brick-sorted query
// records and the work items; queries with no cloud point in their brick's halo go
// straight to the fallback list.
//
// Exactness of the pruning (float distances, not real ones):
//   * AABB bound: lb = l2_simple3(q, clamp(q, lo, hi)) with lo/hi the actual
//     float min/max of the block's points.  Rounding is monotone, so for every
//     point p of the block fl_dist(q,p) >= lb; a block is skipped only if
//     lb > best (strictly: an equal-distance point with a lower index may hide
//     in it).
//   * region bound: margin = min distance from q to the faces of the
//     staged cell range (double).  A point binned outside the range can lie at
//     most `slack` inside it because of float rounding in the binning
//     (grid.slack = 9.6e-7 * max(extent, |coord|) >= 4 roundings of 2^-24),
//     and its float distance is >= true^2 * (1 - 2.4e-7).  A result is final
//     iff best < (margin - slack)^2 * (1 - 1e-6), margin > slack.
#include <cstring>  // rocprim's texture_cache_iterator.hpp needs memset declared first

#include <rocprim/rocprim.hpp>

#include <atomic>
#include <cfloat>

#include "cloud.h"
#include "grid.h"
#include "scratch.h"

namespace pcd {

void free_query_scratch(QueryScratch* s) { delete s; }

constexpr uint64_t kKeyInit = (uint64_t)0x7F7FFFFFu << 32;  // (FLT_MAX, idx 0): nothing with d >= FLT_MAX beats it
constexpr int kMaxRows = 64;                                // rows of a brick region (one per lane)
// Batches up to this size skip the grid path (query sort + brick kernel + fallback = 13 launches) and go
// straight to the per-wavefront hierarchical search, one launch: measured on a 2 M-point cloud
// (tools/nn_latency.py), 20 k queries take 99 us that way against 190 us, 100 k queries 290 against 329.
constexpr uint64_t kSmallBatch = 65536;

// ------------------------------------------------------------ brick math ---
struct BrickParams {
  int B, R;        // brick edge in cells (y, z), halo in cells
  int Bx;          // brick length along x in cells (the rows of the region run along x)
  int nb[3];       // bricks per axis
  uint32_t nbricks;
};

static BrickParams make_bricks(const GridParams& g, int B, int Bx, int R) {
  BrickParams b;
  b.B = B; b.R = R; b.Bx = Bx;
  uint64_t n = 1;
  for (int d = 0; d < 3; ++d) {
    const int e = d == 0 ? Bx : B;
    b.nb[d] = (g.dims[d] + e - 1) / e;
    n *= (uint64_t)b.nb[d];
  }
  b.nbricks = (uint32_t)n;
  return b;
}

// keys still at their initial value (kKeyInit, or a bound with the impossible index) mean "nothing found".  Every
// kernel that writes a FINAL key writes it in this form (the brick kernels for the queries they prove, the pyramid walk
// for all of its queries, the prepare kernels for the queries no search kernel touches): no separate pass over the keys.
__device__ __forceinline__ uint64_t finalized_key(uint64_t k) {
  return (k == kKeyInit || (uint32_t)k == 0xFFFFFFFFu) ? PCD_KEY_NONE : k;
}

// ------------------------------------------------------- query preparation --
// ply.cc:92: feature_point.getVector3fMap() = point_3d.cast<float>()
__global__ void k_prepare_queries(const double* __restrict__ q, uint64_t Q, float4* __restrict__ qf4,
                                  uint64_t* __restrict__ keys) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  float x = (float)q[3 * i], y = (float)q[3 * i + 1], z = (float)q[3 * i + 2];
  bool ok = isfinite(x) && isfinite(y) && isfinite(z);
  qf4[i] = make_float4(x, y, z, ok ? 1.f : 0.f);
  keys[i] = ok ? kKeyInit : PCD_KEY_NONE;   // a query no search kernel touches leaves with its final key
}

// pcd_nn_refine_device: the keys come in with another shard's results.  A query stays active only if this shard
// can still improve it: lower bound of the float distance to the shard's tight bounding box <= incoming distance
// (monotone rounding: fl_dist(q, p) >= fl_dist(q, clamp(q, lo, hi)) for every p in the box; equality is kept --
// a lower index at the same distance may live here).  Inactive queries get w = 0 and no kernel touches their key.
__global__ void k_prepare_refine(const double* __restrict__ q, uint64_t Q, const uint8_t* __restrict__ skip,
                                 float lox, float loy, float loz, float hix, float hiy, float hiz,
                                 float4* __restrict__ qf4, uint64_t* __restrict__ keys) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  float x = (float)q[3 * i], y = (float)q[3 * i + 1], z = (float)q[3 * i + 2];
  bool ok = isfinite(x) && isfinite(y) && isfinite(z) && !(skip && skip[i]);
  uint64_t k = keys[i];
  if (k == PCD_KEY_NONE) k = kKeyInit;
  if (ok) {
    const float px = fminf(fmaxf(x, lox), hix), py = fminf(fmaxf(y, loy), hiy), pz = fminf(fmaxf(z, loz), hiz);
    ok = l2_simple3(x, y, z, px, py, pz) <= __uint_as_float((uint32_t)(k >> 32));
  }
  qf4[i] = make_float4(x, y, z, ok ? 1.f : 0.f);
  keys[i] = ok ? k : finalized_key(k);   // inactive: the key it came with, in its final form
}

// Gate-bounded search (the association entry points): the three call sites reject an association whose point-to-point
// distance exceeds max_search_range (mapper) or 2 m (controller), so a query needs no neighbour farther than that.
// The search starts from key = (bound, index 0xFFFFFFFF) instead of (FLT_MAX, 0): every candidate at or inside the
// bound beats it, everything beyond is pruned by the same exact float bounds as ever, and a key that still carries
// the impossible index at the end means "nothing within the gate" (-> PCD_KEY_NONE, type 0, as the reference's gate
// decides).  bound = float >= (R + e)^2 (1 + 1e-5), e = the query's float rounding (bounded_init_key): the float distance of any point that passes the double-precision gate
// is below it.  NaN range: the reference's `dist > range` is false, nothing is rejected -> unbounded.
// (x, y, z): the query in double.  The gate compares R with the DOUBLE distance |double(p_float) - q_double|
// (lidar_point.cc:27-30), the search minimises the FLOAT distance |p_float - float(q)|: the two differ by up to
// |q_double - float(q)| <= 0.5 ulp per axis -- millimetres for coordinates of kilometres -- plus the rounding of the
// float arithmetic (relative 2e-7, inside the 1e-5 factor).  e = (|x| + |y| + |z|) 2^-23 bounds the first term twice
// over, so every point the gate accepts has a float distance below (R + e)^2 (1 + 1e-5).  (Found by tools/nn_fuzz.py:
// a cloud 8 km from the origin, 11 of 150 000 associations with double distance 0.3899 < R = 0.39 < float distance.)
__device__ __forceinline__ uint64_t bounded_init_key(double R, double x, double y, double z) {
  uint64_t k = kKeyInit;
  if (R == R) {   // not NaN
    const double e = (fabs(x) + fabs(y) + fabs(z)) * (1.0 / 8388608.0);
    const double Rb = (R < 0.0 ? 0.0 : R) + e;
    const double b = Rb * Rb * (1.0 + 1e-5);
    float bf = (float)b;
    if ((double)bf < b) bf = nextafterf(bf, INFINITY);
    if (bf < FLT_MAX) k = ((uint64_t)__float_as_uint(bf) << 32) | 0xFFFFFFFFull;
  }
  return k;
}
__global__ void k_prepare_bounded(const double* __restrict__ q, uint64_t Q, const double* __restrict__ max_range,
                                  uint64_t mr_count, double fixed_range, float4* __restrict__ qf4,
                                  uint64_t* __restrict__ keys) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  float x = (float)q[3 * i], y = (float)q[3 * i + 1], z = (float)q[3 * i + 2];
  bool ok = isfinite(x) && isfinite(y) && isfinite(z);
  qf4[i] = make_float4(x, y, z, ok ? 1.f : 0.f);
  keys[i] = ok ? bounded_init_key(max_range ? max_range[mr_count == 1 ? 0 : i] : fixed_range, q[3 * i], q[3 * i + 1], q[3 * i + 2])
               : PCD_KEY_NONE;
}

__global__ void k_finalize_keys(uint64_t* __restrict__ keys, uint64_t Q) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const uint64_t k = keys[i];
  if (finalized_key(k) != k) keys[i] = PCD_KEY_NONE;
}

__global__ void k_unpack_keys(const uint64_t* __restrict__ keys, uint64_t Q, uint32_t* __restrict__ idx,
                              float* __restrict__ sq, uint8_t* __restrict__ found) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  uint64_t k = keys[i];
  bool f = k != PCD_KEY_NONE;
  idx[i] = f ? (uint32_t)k : 0xFFFFFFFFu;
  sq[i] = f ? __uint_as_float((uint32_t)(k >> 32)) : FLT_MAX;
  found[i] = f ? 1 : 0;
}

// ------------------------------------------------------------ brute force ---
// thread = query, the cloud chunk of blockIdx.y is streamed through a 1024-point LDS tile
// (broadcast reads), chunks are combined with atomicMin on the packed key.
__global__ __launch_bounds__(256) void k_nn_bruteforce(const float4* __restrict__ pts4, uint64_t n,
                                                        uint32_t index_base, uint32_t index_stride,
                                                        const uint32_t* __restrict__ row_index,
                                                        const float4* __restrict__ qf4, uint64_t Q,
                                                        uint64_t chunk, uint64_t* __restrict__ keys) {
  __shared__ float4 tile[1024];
  const uint64_t qi = blockIdx.x * (uint64_t)256 + threadIdx.x;
  float4 q = qi < Q ? qf4[qi] : make_float4(0, 0, 0, 0);
  uint64_t best = kKeyInit;
  const uint64_t beg = (uint64_t)blockIdx.y * chunk;
  const uint64_t end = beg + chunk < n ? beg + chunk : n;
  for (uint64_t t0 = beg; t0 < end; t0 += 1024) {
    const int tn = (int)(end - t0 < 1024 ? end - t0 : 1024);
    __syncthreads();
    for (int j = threadIdx.x; j < tn; j += 256) {
      float4 p = pts4[t0 + j];
      p.w = __uint_as_float(row_index ? row_index[t0 + j] : index_base + (uint32_t)(t0 + j) * index_stride);
      tile[j] = p;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < tn; ++j) {
      float4 p = tile[j];
      float d = l2_simple3(q.x, q.y, q.z, p.x, p.y, p.z);
      uint64_t k = make_key(d, __float_as_uint(p.w));
      best = k < best ? k : best;
    }
  }
  if (qi < Q && q.w != 0.f && best < kKeyInit) atomicMin((unsigned long long*)&keys[qi], (unsigned long long)best);
}

// ------------------------------------------------------ brick bookkeeping ---
// O(Q), independent of the number of bricks: queries are sorted by brick id (rocPRIM radix sort of (brick, query)
// pairs over the bits the brick count needs), runs of equal bricks are cut into work items of <= G queries with
// two scans over the sorted keys, and one kernel writes the item records and the brick-sorted query records.
// Keys: brick id | nbricks = finite query outside the grid (-> exact fallback) | nbricks + 1 = not finite (skipped).
__global__ void k_brick_keys(const float4* __restrict__ qf4, uint64_t Q, GridParams g, BrickParams b,
                             uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= Q) return;
  const float4 q = qf4[i];
  uint32_t bid = b.nbricks + 1u;
  if (q.w != 0.f) {
    const int cx = cell_coord_raw(q.x, g.origin[0], g.inv_h, g.dims[0]);
    const int cy = cell_coord_raw(q.y, g.origin[1], g.inv_h, g.dims[1]);
    const int cz = cell_coord_raw(q.z, g.origin[2], g.inv_h, g.dims[2]);
    const bool in = cx >= 0 && cx < g.dims[0] && cy >= 0 && cy < g.dims[1] && cz >= 0 && cz < g.dims[2];
    bid = in ? (uint32_t)(((uint64_t)(cz / b.B) * b.nb[1] + (cy / b.B)) * b.nb[0] + (cx / b.Bx)) : b.nbricks;
  }
  keys[i] = bid;
  vals[i] = (uint32_t)i;
}

// ---- work items from the sorted keys: two launches (count per tile, then offsets + emit) -------------------------
// A work item starts at sorted position j when its key is a brick id and (j - start of j's run of equal keys) is a
// multiple of G.  A tile = 1024 consecutive positions handled by one workgroup (4 per thread).
constexpr uint32_t kBkTile = 1024;

__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v) { return ~wave_scan_min_u32(~v); }   // inclusive

// per thread: keys of its 4 positions, start of each position's run, item flags; returns the tile's item count
// through *tile_items (valid in every thread).  s_* : LDS scratch of the workgroup.
template <int G>
__device__ __forceinline__ void brick_tile(const uint32_t* __restrict__ keys, uint32_t Q, uint32_t nbricks,
                                           uint32_t tile, uint32_t (&key)[4], uint32_t (&excl_items)[4],
                                           bool (&flag)[4], uint32_t* tile_items, uint32_t* s_wave /*[8]*/,
                                           uint32_t* s_left /*[1]*/) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t j0 = tile * kBkTile + tid * 4;
  const uint32_t tile_start = tile * kBkTile;
  uint32_t prev = j0 > 0 && j0 - 1 < Q ? keys[j0 - 1] : 0xFFFFFFFFu;
  uint32_t hp[4];   // position of the run head at or before each element, 0 = none seen yet in this thread
  uint32_t run = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t j = j0 + k;
    key[k] = j < Q ? keys[j] : 0xFFFFFFFFu;
    const bool head = j < Q && (j == 0 || key[k] != prev);
    run = head ? j : run;
    hp[k] = run;
    prev = key[k];
  }
  // run start carried in from the threads to the left (max-scan of the last head position per thread)
  const uint32_t inc = wave_scan_max_u32(run);
  if (lane == 63) s_wave[wave] = inc;
  // the run that crosses the tile's left edge starts before the tile: first position with that key (sorted array)
  if (tid == 0) {
    uint32_t left = tile_start;
    if (tile_start > 0 && tile_start < Q && keys[tile_start - 1] == keys[tile_start]) {
      const uint32_t k0 = keys[tile_start];
      uint32_t lo = 0, hi = tile_start;   // first index in [0, tile_start] with keys[idx] >= k0 (== k0)
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < k0) lo = mid + 1; else hi = mid; }
      left = lo;
    }
    s_left[0] = left;
  }
  __syncthreads();
  uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)inc, 0);   // placeholder, replaced below
  {
    uint32_t up = __shfl_up(inc, 1);
    carry = lane == 0 ? 0u : up;            // exclusive within the wave
    for (uint32_t w = 0; w < wave; ++w) carry = max(carry, s_wave[w]);
  }
  const uint32_t left = s_left[0];
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t j = j0 + k;
    uint32_t rs = max(hp[k], carry);
    // no head at or before j inside this tile (rs == 0 cannot be a real head unless tile 0, where left == 0 too)
    if (rs == 0 || rs < tile_start) rs = left;
    flag[k] = j < Q && key[k] < nbricks && ((j - rs) % (uint32_t)G) == 0u;
    excl_items[k] = cnt;
    cnt += flag[k] ? 1u : 0u;
  }
  // exclusive sum-scan of the per-thread item counts over the workgroup
  const uint32_t sinc = wave_scan_add_u32(cnt);
  __syncthreads();                        // s_wave is reused
  if (lane == 63) s_wave[wave] = sinc;
  __syncthreads();
  uint32_t base = sinc - cnt, total = 0;
  for (uint32_t w = 0; w < 4; ++w) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) excl_items[k] += base;
  *tile_items = total;
}

template <int G>
__global__ __launch_bounds__(256) void k_brick_tile_count(const uint32_t* __restrict__ keys, uint32_t Q, uint32_t nbricks,
                                                          uint32_t* __restrict__ tile_items) {
  __shared__ uint32_t s_wave[8], s_left[1];
  uint32_t key[4], ex[4], total;
  bool flag[4];
  brick_tile<G>(keys, Q, nbricks, blockIdx.x, key, ex, flag, &total, s_wave, s_left);
  if (threadIdx.x == 0) tile_items[blockIdx.x] = total;
}

template <int G>
__global__ __launch_bounds__(256) void k_brick_emit(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                    const uint32_t* __restrict__ tile_items, const float4* __restrict__ qf4,
                                                    const uint64_t* __restrict__ keys_in, uint32_t Q, uint32_t nbricks,
                                                    uint32_t nb0, uint32_t nb1, uint4* __restrict__ items,
                                                    float4* __restrict__ qsorted, uint64_t* __restrict__ ksorted,
                                                    uint32_t* __restrict__ fb_list, NnCounters* __restrict__ ctr) {
  __shared__ uint32_t s_wave[8], s_left[1], s_off[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // items of the tiles to the left: fixed-order strided sums (few hundred values)
  uint32_t part = 0;
  for (uint32_t t = tid; t < blockIdx.x; t += 256) part += tile_items[t];
  part = wave_scan_add_u32(part);
  if (lane == 63) s_off[wave] = part;
  __syncthreads();
  const uint32_t tile_off = s_off[0] + s_off[1] + s_off[2] + s_off[3];
  __syncthreads();
  uint32_t key[4], ex[4], total;
  bool flag[4];
  brick_tile<G>(keys, Q, nbricks, blockIdx.x, key, ex, flag, &total, s_wave, s_left);
  const uint32_t j0 = blockIdx.x * kBkTile + tid * 4;
  // the four positions of a thread: all loads of a stage are issued before any is used (the dependent chain
  // vals -> qf4 / keys_in would otherwise be walked four times in a row)
  uint32_t v[4];
  float4 q[4];
  uint64_t kin[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = j0 + k < Q ? vals[j0 + k] : 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = j0 + k < Q && key[k] < nbricks;
    q[k] = in ? qf4[v[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
    // the key the query came with (a plain query comes with kKeyInit: no gather)
    kin[k] = (in && keys_in) ? keys_in[v[k]] : kKeyInit;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t j = j0 + k;
    if (j >= Q) break;
    if (key[k] < nbricks) {
      q[k].w = __uint_as_float(v[k]);          // brick-sorted query record {x, y, z, bits(query id)}
      qsorted[j] = q[k];
      ksorted[j] = kin[k];
      if (j + 1 == Q || keys[j + 1] >= nbricks) ctr->n_in_grid = j + 1;   // in-grid queries sort first
      if (flag[k]) {
        uint32_t cnt = 1;
        while (cnt < (uint32_t)G && j + cnt < Q && keys[j + cnt] == key[k]) ++cnt;
        // item record {first query, brick x, brick y, brick z | count << 28}: the brick kernel needs no divisions
        const uint32_t bx = key[k] % nb0, by = (key[k] / nb0) % nb1, bz = key[k] / (nb0 * nb1);
        items[tile_off + ex[k]] = make_uint4(j, bx, by, bz | (cnt << 28));
      }
    } else if (key[k] == nbricks) {
      fb_list[atomicAdd(&ctr->fb_count, 1u)] = v[k];   // outside the grid: straight to the exact fallback
    }
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) ctr->nitems = tile_off + total;
}

// ------------------------------------------- brick bookkeeping, counting sort ---
// Four launches instead of the radix sort's seven (round 2: 0.17 ms at Q = 1 M, three Onesweep digit passes at ~29 us
// each whatever the size).  Per cloud and brick geometry, once: a SLOT for every brick whose halo region holds at
// least one cloud point (k_brick_occupied + an exclusive scan; slots ascend with the brick id); a query in a brick
// without a slot has nothing within the halo and goes straight to the exact fallback -- which is where the brick
// kernel would have sent it after walking an empty region.  Per batch a two-level counting sort on the slot:
//   k_bk_slots    query -> sort key (brick id) or "fallback" (outside the grid / brick with an empty halo region: those
//                 are appended to the chunked fallback list here); histogram of the COARSE keys (id >> shift, at most
//                 4096 buckets): LDS histogram per workgroup, merged with global atomics.
//   k_bk_scatter  every workgroup scans the histogram into bucket starts itself; tiles of 4096 queries: rank inside the
//                 tile by LDS atomics, one global atomic per (tile, bucket) for the tile's place in the bucket,
//                 (key, query) pairs written bucket by bucket.
//   k_bk_count    one workgroup per coarse bucket: queries per FINE key (low bits) in LDS -> the bucket's item count.
//   k_bk_emit     one workgroup per coarse bucket: the same counting sort again, now with the item base known (sum
//                 of the item counts to the left: items stay in brick order, which the XCD-aware walk of k_nn_brick
//                 is worth 0.04 ms for), then the brick-sorted query records, their incoming keys and the work
//                 items (G queries of one brick each).
// (No workgroup waits for another one: a look-back over per-bucket descriptors needs agent-scope release / acquire to
//  cross the XCDs' L2s -- with relaxed polling the descriptors never arrived -- and at that price was slower than this
//  fourth launch.)
// Queries of one brick arrive in any order (LDS / global atomics): WHICH queries share a work item varies from run to
// run, every query's result does not.
constexpr uint32_t kSlotFallback = 0xFFFFFFFEu;   // query: finite, but outside the grid or in a brick without a slot
constexpr uint32_t kSlotSkip = 0xFFFFFFFFu;       // query: not finite
constexpr uint32_t kBkMaxCoarse = 4096, kBkMaxFine = 4096, kBkTileQ = 4096;   // 2^24 bricks: every grid of the 2^26-cell budget

// one thread per brick: does the halo region (whole quad rows, as brick_load_meta walks them) hold any point?
__global__ void k_brick_occupied(GridParams g, BrickParams b, const uint32_t* __restrict__ cell_start,
                                 uint32_t* __restrict__ flag) {
  const uint32_t bid = blockIdx.x * blockDim.x + threadIdx.x;
  if (bid >= b.nbricks) return;
  const int bx = (int)(bid % (uint32_t)b.nb[0]), by = (int)((bid / (uint32_t)b.nb[0]) % (uint32_t)b.nb[1]),
            bz = (int)(bid / ((uint32_t)b.nb[0] * (uint32_t)b.nb[1]));
  const int x0 = max(bx * b.Bx - b.R, 0), x1 = min(bx * b.Bx + b.Bx + b.R, g.dims[0]);
  const int y0 = max(by * b.B - b.R, 0), y1 = min(by * b.B + b.B + b.R, g.dims[1]);
  const int z0 = max(bz * b.B - b.R, 0), z1 = min(bz * b.B + b.B + b.R, g.dims[2]);
  uint32_t any = 0;
  if (x0 < x1 && y0 < y1 && z0 < z1)
    for (int zq = z0 >> 1; zq < (z1 + 1) >> 1; ++zq)
      for (int yq = y0 >> 1; yq < (y1 + 1) >> 1; ++yq) {
        const uint64_t rb = quad_row_base(g, yq, zq);
        any |= cell_start[rb + 4 * (uint64_t)x1] - cell_start[rb + 4 * (uint64_t)x0];
      }
  flag[bid] = any ? 1u : 0u;
}
// 1 bit per brick: the halo region holds a point (32 bricks per thread; 580 KB for workload M's 4.65 M bricks, cache
// resident -- a 4-byte slot per brick made the per-query lookup a 128 MB gather)
__global__ void k_brick_bitmap(const uint32_t* __restrict__ flag, uint32_t nbricks, uint32_t* __restrict__ bits) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w * 32u >= nbricks) return;
  uint32_t m = 0;
  for (uint32_t k = 0; k < 32u && w * 32u + k < nbricks; ++k) m |= (flag[w * 32u + k] ? 1u : 0u) << k;
  bits[w] = m;
}

// block-wide exclusive scan of one value per thread (NW wavefronts); returns the exclusive prefix, *total in every thread
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w /*[NW]*/, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_scan_add_u32(v);
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t w = 0; w < (uint32_t)NW; ++w) { if (w < wave) base += s_w[w]; tot += s_w[w]; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_w, uint32_t* total) {
  return block_excl_scan<4>(v, s_w, total);
}

__global__ __launch_bounds__(1024) void k_bk_slots(const float4* __restrict__ qf4, uint32_t Q, GridParams g, BrickParams b,
                                                   const uint32_t* __restrict__ brick_slot, uint32_t shift, uint32_t ncoarse,
                                                   uint32_t* __restrict__ qslot, uint32_t* __restrict__ chist,
                                                   uint32_t* __restrict__ fb_list, NnCounters* __restrict__ ctr) {
  // few, large workgroups: every workgroup merges its whole LDS histogram into the global one with atomics
  // (1024 workgroups x 2048 buckets = 2 M global atomics made this kernel 145 us)
  __shared__ uint32_t s_h[kBkMaxCoarse], s_w[16], s_base;
  for (uint32_t c = threadIdx.x; c < ncoarse; c += 1024) s_h[c] = 0;
  __syncthreads();
  // rounds of 4 queries per thread: the fallback queries of a round are counted over the WORKGROUP and placed with one
  // global atomic (a chunk per wavefront, as the brick kernel does it, was one same-address returning atomic per
  // wavefront of this short kernel: 4096 of them, ~40 of its 54 us)
  for (uint32_t r0 = blockIdx.x * 4096u; r0 < Q; r0 += gridDim.x * 4096u) {
    uint32_t sv[4], nfb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = r0 + k * 1024u + threadIdx.x;
      uint32_t s = kSlotSkip;
      if (i < Q) {
        const float4 q = qf4[i];
        if (q.w != 0.f) {
          const int cx = cell_coord_raw(q.x, g.origin[0], g.inv_h, g.dims[0]);
          const int cy = cell_coord_raw(q.y, g.origin[1], g.inv_h, g.dims[1]);
          const int cz = cell_coord_raw(q.z, g.origin[2], g.inv_h, g.dims[2]);
          const bool in = cx >= 0 && cx < g.dims[0] && cy >= 0 && cy < g.dims[1] && cz >= 0 && cz < g.dims[2];
          s = kSlotFallback;
          if (in) {
            const uint32_t bid = (uint32_t)(((uint64_t)(cz / b.B) * b.nb[1] + (cy / b.B)) * b.nb[0] + (cx / b.Bx));
            if ((brick_slot[bid >> 5] >> (bid & 31u)) & 1u) { s = bid; atomicAdd(&s_h[bid >> shift], 1u); }
          }
        }
        qslot[i] = s;
      }
      sv[k] = s;
      nfb += s == kSlotFallback ? 1u : 0u;
    }
    uint32_t total;
    uint32_t pos = block_excl_scan<16>(nfb, s_w, &total);
    if (total) {   // workgroup-uniform
      if (threadIdx.x == 0) s_base = atomicAdd(&ctr->fb_count, total);
      __syncthreads();
      pos += s_base;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (sv[k] == kSlotFallback) fb_list[pos++] = r0 + k * 1024u + threadIdx.x;
      __syncthreads();   // s_base is rewritten in the next round
    }
  }
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < ncoarse; c += 1024)
    if (s_h[c]) atomicAdd(&chist[c], s_h[c]);
}

__global__ __launch_bounds__(1024) void k_bk_scatter(const uint32_t* __restrict__ qslot, uint32_t Q, uint32_t shift,
                                                     uint32_t ncoarse, const uint32_t* __restrict__ chist,
                                                     uint32_t* __restrict__ cstart, uint32_t* __restrict__ ccur,
                                                     uint32_t* __restrict__ pslot, uint32_t* __restrict__ pqid,
                                                     NnCounters* __restrict__ ctr) {
  __shared__ uint32_t s_h[kBkMaxCoarse], s_start[kBkMaxCoarse], s_w[16];
  // bucket starts: every workgroup scans the (at most 4096-entry) histogram itself; workgroup 0 keeps the result
  {
    const uint32_t per = (ncoarse + 1023u) / 1024u, c0 = threadIdx.x * per;   // per <= 4
    uint32_t h[4] = {0, 0, 0, 0}, sum = 0;
    for (uint32_t k = 0; k < per && c0 + k < ncoarse; ++k) { h[k] = chist[c0 + k]; sum += h[k]; }
    uint32_t total;
    uint32_t base = block_excl_scan<16>(sum, s_w, &total);
    for (uint32_t k = 0; k < per && c0 + k < ncoarse; ++k) {
      s_start[c0 + k] = base;
      if (blockIdx.x == 0) cstart[c0 + k] = base;
      base += h[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { cstart[ncoarse] = total; ctr->n_in_grid = total; }
  }
  for (uint32_t c = threadIdx.x; c < ncoarse; c += 1024) s_h[c] = 0;
  __syncthreads();
  const uint32_t i0 = blockIdx.x * kBkTileQ + threadIdx.x;
  uint32_t sl[4], lr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = i0 + k * 1024u;
    sl[k] = i < Q ? qslot[i] : kSlotSkip;
    lr[k] = sl[k] < kSlotFallback ? atomicAdd(&s_h[sl[k] >> shift], 1u) : 0u;   // rank inside the tile
  }
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < ncoarse; c += 1024) {
    const uint32_t n = s_h[c];
    s_h[c] = s_start[c] + (n ? atomicAdd(&ccur[c], n) : 0u);     // the tile's place in the bucket (cursors start at 0)
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (sl[k] < kSlotFallback) {
      const uint32_t p = s_h[sl[k] >> shift] + lr[k];
      pslot[p] = sl[k];
      pqid[p] = i0 + k * 1024u;
    }
}

// work items of every coarse bucket: one workgroup per bucket counts the bucket's queries per fine key in LDS
template <int G>
__global__ __launch_bounds__(256) void k_bk_count(uint32_t shift, const uint32_t* __restrict__ cstart,
                                                  const uint32_t* __restrict__ pslot, uint32_t* __restrict__ bitems) {
  __shared__ uint32_t s_cnt[kBkMaxFine], s_w[4];
  const uint32_t nfine = 1u << shift, mask = nfine - 1u;
  const uint32_t beg = cstart[blockIdx.x], end = cstart[blockIdx.x + 1];
  if (beg == end) { if (threadIdx.x == 0) bitems[blockIdx.x] = 0; return; }
  for (uint32_t f = threadIdx.x; f < nfine; f += 256) s_cnt[f] = 0;
  __syncthreads();
  for (uint32_t p = beg + threadIdx.x; p < end; p += 256) atomicAdd(&s_cnt[pslot[p] & mask], 1u);
  __syncthreads();
  uint32_t ai = 0;
  for (uint32_t f = threadIdx.x; f < nfine; f += 256) ai += (s_cnt[f] + G - 1) / G;
  uint32_t ti;
  (void)block_excl_scan_256(ai, s_w, &ti);
  if (threadIdx.x == 0) bitems[blockIdx.x] = ti;
}

template <int G>
__global__ __launch_bounds__(256) void k_bk_emit(const float4* __restrict__ qf4, const uint64_t* __restrict__ keys_in,
                                                 GridParams g, BrickParams b, uint32_t shift,
                                                 const uint32_t* __restrict__ cstart, const uint32_t* __restrict__ pslot,
                                                 const uint32_t* __restrict__ pqid, const uint32_t* __restrict__ bitems,
                                                 uint32_t* __restrict__ chist, uint32_t* __restrict__ ccur,
                                                 uint4* __restrict__ items, float4* __restrict__ qsorted,
                                                 uint64_t* __restrict__ ksorted, NnCounters* __restrict__ ctr) {
  // per fine key: start of its queries / items inside the bucket (kBkMaxFine + 1 entries: the count of key f is
  // s_off[f + 1] - s_off[f]) and the running rank; 48 KB at 4096 fine keys
  __shared__ uint32_t s_off[kBkMaxFine + 1], s_ioff[kBkMaxFine + 1], s_cur[kBkMaxFine], s_w[4];
  const uint32_t nfine = 1u << shift, mask = nfine - 1u;
  const uint32_t beg = cstart[blockIdx.x], end = cstart[blockIdx.x + 1];
  // item base = items of all buckets to the left, in bucket order = brick order (the XCD-aware walk of k_nn_brick
  // depends on it): summed from k_bk_count's per-bucket totals -- no atomics, no waiting on other workgroups
  uint32_t acc = 0;
  for (uint32_t c = threadIdx.x; c < blockIdx.x; c += 256) acc += bitems[c];
  uint32_t ibase;
  (void)block_excl_scan_256(acc, s_w, &ibase);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) ctr->nitems = ibase + bitems[blockIdx.x];
  if (threadIdx.x == 0) { chist[blockIdx.x] = 0; ccur[blockIdx.x] = 0; }   // clean for the next batch
  if (beg == end) return;
  for (uint32_t f = threadIdx.x; f < nfine; f += 256) s_cur[f] = 0;
  __syncthreads();
  for (uint32_t p = beg + threadIdx.x; p < end; p += 256) atomicAdd(&s_cur[pslot[p] & mask], 1u);   // counts
  __syncthreads();
  // exclusive prefixes of the query counts and of the item counts over the bucket's fine keys
  const uint32_t per = (nfine + 255u) / 256u, f0 = threadIdx.x * per;
  {
    uint32_t aq = 0, ai = 0;
    for (uint32_t f = f0; f < min(f0 + per, nfine); ++f) { aq += s_cur[f]; ai += (s_cur[f] + G - 1) / G; }
    uint32_t tq, ti;
    uint32_t bq = block_excl_scan_256(aq, s_w, &tq);
    uint32_t bi = block_excl_scan_256(ai, s_w, &ti);
    for (uint32_t f = f0; f < min(f0 + per, nfine); ++f) {
      const uint32_t c = s_cur[f];
      s_off[f] = bq; s_ioff[f] = bi; bq += c; bi += (c + G - 1) / G;
      s_cur[f] = 0;                     // becomes the running rank
    }
    if (threadIdx.x == 255) { s_off[nfine] = tq; s_ioff[nfine] = ti; }
    __syncthreads();
  }
  for (uint32_t p = beg + threadIdx.x; p < end; p += 256) {
    const uint32_t f = pslot[p] & mask, i = pqid[p];
    const uint32_t r = atomicAdd(&s_cur[f], 1u);
    float4 q = qf4[i];
    const uint32_t pos = beg + s_off[f] + r;
    if (r % (uint32_t)G == 0u) {   // first query of a work item: G consecutive positions of one brick
      const int cx = cell_coord_raw(q.x, g.origin[0], g.inv_h, g.dims[0]);
      const int cy = cell_coord_raw(q.y, g.origin[1], g.inv_h, g.dims[1]);
      const int cz = cell_coord_raw(q.z, g.origin[2], g.inv_h, g.dims[2]);
      const uint32_t left = s_off[f + 1] - s_off[f] - r;
      items[ibase + s_ioff[f] + r / (uint32_t)G] =
          make_uint4(pos, (uint32_t)(cx / b.Bx), (uint32_t)(cy / b.B),
                     (uint32_t)(cz / b.B) | ((left < (uint32_t)G ? left : (uint32_t)G) << 28));
    }
    q.w = __uint_as_float(i);
    qsorted[pos] = q;
    ksorted[pos] = keys_in ? keys_in[i] : kKeyInit;
  }
}

// ------------------------------------------------------------ brick kernel ---
// wave-uniform copy of lane l's value (lands in an SGPR)
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t& total) {
  const uint32_t inc = wave_scan_add_u32(v);
  total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  return inc - v;
}

// squared safe radius (double) around q inside the cell range [c0,c1) per axis; faces on the
// grid boundary do not bound anything (no points beyond them). Returns < 0 when nothing is proven.
__device__ __forceinline__ double proven_bound(const GridParams& g, float qx, float qy, float qz, const int c0[3],
                                               const int c1[3]) {
  const double q[3] = {(double)qx, (double)qy, (double)qz};
  double margin = 1e300;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (c0[d] > 0) margin = fmin(margin, q[d] - ((double)g.origin[d] + (double)c0[d] * (double)g.h));
    if (c1[d] < g.dims[d]) margin = fmin(margin, ((double)g.origin[d] + (double)c1[d] * (double)g.h) - q[d]);
  }
  if (margin >= 1e300) return 1e300;  // the range covers the whole grid
  margin -= (double)g.slack;
  if (!(margin > 0.0)) return -1.0;
  return margin * margin * (1.0 - 1e-6);
}

}  // namespace pcd
#include "brick_kernel.h"
#include "brick_clip_kernel.h"
namespace pcd {


// The brick kernel fills the fallback list in per-wavefront chunks (brick_kernel.h kFbChunk); the unused slots of
// the chunks keep 0xFFFFFFFF.  One pass squeezes them out (wave-aggregated atomics: the order of the dense list is
// not deterministic, the per-query results do not depend on it) so that k_nn_fallback gets an evenly filled list.
constexpr uint32_t kFbcThreads = 1024, kFbcPer = 4;   // 4096 list slots per workgroup
__global__ __launch_bounds__(1024) void k_fb_compact(const uint32_t* __restrict__ list,
                                                     const uint32_t* __restrict__ count_ptr,
                                                     uint32_t* __restrict__ dense, uint32_t* __restrict__ dense_count) {
  // one atomic per WORKGROUP of 4096 slots: same-address device-scope atomics serialise at ~10 ns each on this part
  // (one per wavefront of 64 slots made this pass 45 us for a 390 k-slot list)
  __shared__ uint32_t s_wave[16], s_base;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n = *count_ptr;
  const uint32_t e0 = (blockIdx.x * kFbcThreads + tid) * kFbcPer;
  if (blockIdx.x * kFbcThreads * kFbcPer >= n) return;   // whole workgroup past the end
  uint32_t v[kFbcPer], cnt = 0;
  if (e0 + kFbcPer <= n) {
    const uint4 q = *reinterpret_cast<const uint4*>(list + e0);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
#pragma unroll
    for (uint32_t k = 0; k < kFbcPer; ++k) v[k] = e0 + k < n ? list[e0 + k] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (uint32_t k = 0; k < kFbcPer; ++k) cnt += v[k] != 0xFFFFFFFFu ? 1u : 0u;
  const uint32_t inc = wave_scan_add_u32(cnt);
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (uint32_t w = 0; w < kFbcThreads / 64; ++w) { before += w < wave ? s_wave[w] : 0u; total += s_wave[w]; }
  if (tid == 0) s_base = total ? atomicAdd(dense_count, total) : 0u;
  __syncthreads();
  uint32_t pos = s_base + before + inc - cnt;
#pragma unroll
  for (uint32_t k = 0; k < kFbcPer; ++k)
    if (v[k] != 0xFFFFFFFFu) dense[pos++] = v[k];
}

// --------------------------------------------------------- exact fallback ---
struct FbFused {   // inputs of k_nn_fallback<1 / 2> (the kernel as the only launch of a call)
  const double* q;
  const double* max_range;
  uint64_t mr_count;
  double fixed_range;
};

// One wavefront per query walks the 64-ary AABB pyramid over the grid (level 0 = 2x2x2-cell leaves carrying their
// point range, level k+1 = 4x4x4 nodes of level k, cloud.h): the 64 children of the current node sit one per lane, each
// lane computes the exact float lower bound of its child, and the wave descends into the nearest
// child whose bound does not exceed the best distance so far (depth first, nearest first).
// Every skipped subtree has bound > best, so the result is the exact minimum of the packed keys.
// FUSED: 0 = queries from qf4, keys in and out raw (the grid path's second stage; refining), 1 / 2 = the kernel is the
// only launch of the call (plain / gate-bounded).  A template parameter, not a run-time mode: with the mode tested at
// run time the grid path's instance compiled differently and ran 17 % slower (0.365 -> 0.425 ms on workload M).
template <int FUSED>
__global__ __launch_bounds__(256) void k_nn_fallback(GridParams g, PyramidParams py, const float4* __restrict__ sorted,
                                                      const uint32_t* __restrict__ cell_start,
                                                      const float* __restrict__ aabb,
                                                      const float4* __restrict__ qf4,
                                                      const uint32_t* __restrict__ list,  // NULL: queries 0..count-1
                                                      const uint32_t* __restrict__ count_ptr, uint32_t count_imm,
                                                      uint64_t* __restrict__ keys, NnCounters* __restrict__ ctr,
                                                      int collect_stats, FbFused fu) {
  __shared__ float s_lb[4][kMaxPyrLevels][64];
  __shared__ unsigned long long s_mask[4][kMaxPyrLevels];
  __shared__ int s_node[4][kMaxPyrLevels][3];
  __shared__ int4 s_pyr[kMaxPyrLevels];   // per level {dims x, y, z, node offset}
  if (threadIdx.x < (unsigned)kMaxPyrLevels)
    s_pyr[threadIdx.x] = make_int4(py.dims[threadIdx.x][0], py.dims[threadIdx.x][1], py.dims[threadIdx.x][2], (int)py.off[threadIdx.x]);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform for the compiler: LDS addresses on the scalar unit
  const uint32_t count = count_ptr ? *count_ptr : count_imm;
  const uint32_t nwaves = gridDim.x * 4;
  const int top = py.nlev - 1;  // >= 1
  const int ci = lane & 3, cj = (lane >> 2) & 3, ck = lane >> 4;
  // the lane's child of the virtual top: flat index over the dimensions of level top - 1 (at most 64 nodes)
  const int tdx = py.dims[top - 1][0], tdy = py.dims[top - 1][1];
  const int tcx = lane % tdx, tcy = (lane / tdx) % tdy, tcz = lane / (tdx * tdy);   // tcz >= dims z: no such node
  unsigned long long st_pts = 0, st_q = 0, st_steps = 0, st_leaves = 0;
  uint32_t st_max_steps = 0, st_max_leaves = 0;
  for (uint32_t e = blockIdx.x * 4 + wave; e < count; e += nwaves) {
    const uint32_t qi = list ? list[e] : e;
    float qx, qy, qz;
    uint64_t best;
    if (FUSED == 0) {
      const float4 q = qf4[qi];
      if (q.w == 0.f) continue;  // not finite: stays "not found"
      qx = q.x; qy = q.y; qz = q.z;
      best = keys[qi];  // kKeyInit or the brick kernel's tentative result
    } else {
      // one-launch form (small batches): the query is converted here and the key leaves finalised -- no prepare
      // kernel in front, no finalize kernel behind (lidar/ply.cc:92: feature_point = point_3d.cast<float>())
      qx = (float)fu.q[3 * (size_t)qi]; qy = (float)fu.q[3 * (size_t)qi + 1]; qz = (float)fu.q[3 * (size_t)qi + 2];
      if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) {
        if (lane == 0) keys[qi] = PCD_KEY_NONE;
        continue;
      }
      best = FUSED == 2 ? bounded_init_key(fu.max_range ? fu.max_range[fu.mr_count == 1 ? 0 : qi] : fu.fixed_range,
                                             fu.q[3 * (size_t)qi], fu.q[3 * (size_t)qi + 1], fu.q[3 * (size_t)qi + 2])
                          : kKeyInit;
    }
    // The state of the level being iterated lives in REGISTERS (the children's bounds one per lane, the mask of the
    // children still to visit in an SGPR pair, the node in SGPRs); the LDS arrays are a stack touched only when the walk
    // descends (push) or comes back (pop).  Before, every iteration went through LDS for the mask and the node -- a
    // write -> read round trip in the walk's dependent chain, and most iterations are leaves of ONE level-1 node.
    // The keys are compared as doubles (brick_kernel.h: one v_min_f64 instead of a 64-bit compare + two selects).
    double lane_best = __builtin_bit_cast(double, best);
    float best_d = __uint_as_float((uint32_t)(best >> 32));
    int lev = top;                 // the children of node (nx, ny, nz) of level `lev` are being iterated
    int nx = 0, ny = 0, nz = 0;
    float cur_lb;
    unsigned long long cur_m;
    uint32_t rs = 0, rn = 0;       // lev == 1: the lane's leaf's point range
    uint32_t q_steps = 0, q_leaves = 0;
    // children of node (nx,ny,nz) of level `lev` live on level lev-1.  ONE memory round trip per expansion: the level's
    // dimensions and offset come from LDS (as kernel arguments indexed by `lev` they were four dependent scalar loads),
    // the child's two 16-byte loads are unconditional on a clamped index and the bound is selected afterwards (under
    // `if (non-empty)` the compiler split them into a first pair of dwords, the test, and a second, dependent pair of
    // loads): the walk is a chain of such steps and its latency is the kernel's time.
#define PCD_FB_EXPAND()                                                                                               \
    {                                                                                                                 \
      const int4 dm = s_pyr[lev - 1]; /* {dims x, y, z, node offset} */                                               \
      /* (the virtual top's children are ALL nodes of the level below, lane = flat node index) */                     \
      const int cx = lev == top ? tcx : 4 * nx + ci, cy = lev == top ? tcy : 4 * ny + cj,                             \
                cz = lev == top ? tcz : 4 * nz + ck;                                                                  \
      const bool in = cx < dm.x && cy < dm.y && cz < dm.z;                                                            \
      const uint32_t id = (uint32_t)dm.w + (in ? (uint32_t)((cz * dm.y + cy) * dm.x + cx) : 0u); /* < 2^32 nodes */   \
      const float4 lo = *reinterpret_cast<const float4*>(aabb + 8 * (size_t)id);     /* lo.x lo.y lo.z hi.x */        \
      const float4 hi = *reinterpret_cast<const float4*>(aabb + 8 * (size_t)id + 4); /* hi.y hi.z start count */      \
      /* clamp(q, lo, hi) = the median of the three for lo <= hi: ONE v_med3_f32 (fminf(fmaxf()) costs two ops + */     \
      /* three canonicalising v_max x, x, x); an empty node's inverted box is masked below */                         \
      const float px = __builtin_amdgcn_fmed3f(qx, lo.x, lo.w), pyc = __builtin_amdgcn_fmed3f(qy, lo.y, hi.x),        \
                  pz = __builtin_amdgcn_fmed3f(qz, lo.z, hi.y);                                                       \
      const float lbv = l2_simple3(qx, qy, qz, px, pyc, pz);                                                          \
      cur_lb = (in && lo.x <= lo.w) ? lbv : INFINITY; /* empty nodes have an inverted box */                          \
      rs = __float_as_uint(hi.z); rn = __float_as_uint(hi.w); /* leaves carry their range */                          \
      cur_m = __ballot(cur_lb <= best_d);                                                                             \
    }
    PCD_FB_EXPAND();
    while (true) {
      ++q_steps;
      unsigned long long m = cur_m & __ballot(cur_lb <= best_d);
      if (m == 0) {
        if (lev == top) break;
        ++lev;   // pop
        cur_lb = s_lb[wave][lev][lane];
        const unsigned long long pm = s_mask[wave][lev];
        cur_m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pm >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)pm);
        nx = __builtin_amdgcn_readfirstlane(s_node[wave][lev][0]);
        ny = __builtin_amdgcn_readfirstlane(s_node[wave][lev][1]);
        nz = __builtin_amdgcn_readfirstlane(s_node[wave][lev][2]);
        continue;
      }
      const int sl = wave_argmin_u32(__float_as_uint(cur_lb), m);  // nearest remaining child (lb >= 0: bit order)
      m &= ~(1ull << sl);
      cur_m = m;
      if (lev == 1) {
        // the child is a LEAF (cloud.h: a sub-block of 2x2x2 cells, one contiguous point range with a tight box): scan
        // it, 128 points per trip with the loads issued back to back (clamped indices, no branches around the loads: a
        // re-read is harmless), then tighten the bound -- the other leaves of the node are tested against it
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)rs, sl), n0 = (uint32_t)__builtin_amdgcn_readlane((int)rn, sl);
        for (uint32_t base = 0; base < n0; base += 128) {
          float4 p[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const uint32_t gi = base + k * 64 + lane;
            p[k] = sorted[s0 + (gi < n0 ? gi : n0 - 1)];
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float d = l2_simple3(qx, qy, qz, p[k].x, p[k].y, p[k].z);
            lane_best = min_key_f64(lane_best, __builtin_bit_cast(double, make_key(d, __float_as_uint(p[k].w))));
          }
        }
        st_pts += n0;
        ++q_leaves;
        best = wave_min_u64(__builtin_bit_cast(uint64_t, lane_best));
        best_d = __uint_as_float((uint32_t)(best >> 32));
        // The node's OTHER leaves that still pass the tightened bound: all of them in one batch -- four leaves' loads in
        // flight at a time, no reduction in between.  They would each be scanned anyway (a leaf on a tilted surface has a
        // fat box: a far query at distance d finds ~d / 0.1 m leaves whose box is nearer than d although none of their
        // points is -- 129 leaf scans in the longest walk of workload M, one walk step each before); the rare leaf that
        // a batch neighbour would have ruled out costs one load.
        unsigned long long mb = m & __ballot(cur_lb <= best_d);
        if (mb) {
          cur_m = m & ~mb;
          while (mb) {
            // four leaves per trip, one 64-point load each (a leaf holds ~50 points; the rare longer one loops)
            uint32_t a[4], c[4];
            uint32_t cmax = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int r = mb ? __ffsll((long long)mb) - 1 : -1;
              mb &= mb - 1;   // (0 stays 0)
              a[k] = r >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)rs, r) : a[0];
              c[k] = r >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)rn, r) : 0u;
              cmax = c[k] > cmax ? c[k] : cmax;
              st_pts += c[k];
              q_leaves += r >= 0 ? 1 : 0;
            }
            for (uint32_t base = 0; base < cmax; base += 64) {
              float4 p[4];
              const uint32_t gi = base + lane;
#pragma unroll
              for (int k = 0; k < 4; ++k) p[k] = sorted[a[k] + (gi < c[k] ? gi : (c[k] ? c[k] - 1 : 0u))];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float d = l2_simple3(qx, qy, qz, p[k].x, p[k].y, p[k].z);
                lane_best = min_key_f64(lane_best, __builtin_bit_cast(double, make_key(d, __float_as_uint(p[k].w))));
              }
            }
          }
          best = wave_min_u64(__builtin_bit_cast(uint64_t, lane_best));
          best_d = __uint_as_float((uint32_t)(best >> 32));
        }
      } else {
        // push the level, descend into child `sl`
        s_lb[wave][lev][lane] = cur_lb;
        if (lane == 0) { s_mask[wave][lev] = cur_m; s_node[wave][lev][0] = nx; s_node[wave][lev][1] = ny; s_node[wave][lev][2] = nz; }
        if (lev == top) {
          nx = __builtin_amdgcn_readlane(tcx, sl); ny = __builtin_amdgcn_readlane(tcy, sl); nz = __builtin_amdgcn_readlane(tcz, sl);
        } else {
          nx = 4 * nx + (sl & 3); ny = 4 * ny + ((sl >> 2) & 3); nz = 4 * nz + (sl >> 4);
        }
        --lev;
        PCD_FB_EXPAND();
      }
    }
#undef PCD_FB_EXPAND
    if (lane == 0) keys[qi] = finalized_key(best);   // the walk is exact: whatever it ends with is final
    st_q += 1;
    st_steps += q_steps; st_leaves += q_leaves;
    st_max_steps = q_steps > st_max_steps ? q_steps : st_max_steps;
    st_max_leaves = q_leaves > st_max_leaves ? q_leaves : st_max_leaves;
  }
  if (collect_stats) {   // (uniform) one set of global atomics per WORKGROUP: same-address atomics serialise at ~10 ns
    __shared__ unsigned long long s_st[4];   // each, and this grid has 65 k wavefronts (per wavefront: a 3 ms launch)
    __shared__ unsigned int s_mx[2];
    if (threadIdx.x < 4) s_st[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_mx[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0) {
      atomicAdd(&s_st[0], st_pts); atomicAdd(&s_st[1], st_q); atomicAdd(&s_st[2], st_steps); atomicAdd(&s_st[3], st_leaves);
      atomicMax(&s_mx[0], st_max_steps); atomicMax(&s_mx[1], st_max_leaves);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_st[1]) {
      atomicAdd(&ctr->fallback_points, s_st[0]);
      atomicAdd(&ctr->pair_evals, s_st[0]);
      atomicAdd(&ctr->fallback_queries, s_st[1]);
      atomicAdd(&ctr->fb_steps, s_st[2]); atomicAdd(&ctr->fb_leaves, s_st[3]);
      atomicMax(&ctr->fb_max_steps, s_mx[0]); atomicMax(&ctr->fb_max_leaves, s_mx[1]);
    }
  }
}

// ------------------------------------------------------------- host driver ---
// k_nn_fallback's grid: up to 4 x 16384 wavefronts, i.e. one or two queries per wavefront for lists up to 128 k and
// the hardware's workgroup dispatch as the load balancer (query costs spread 1:4).  2048 / 4096 / 16384 / 65536
// workgroups: 0.384 / 0.387 / 0.368 / 0.369 ms at workload M, 0.136 / 0.125 / 0.126 / 0.125 ms on an eighth of it.
constexpr unsigned g_fb_max_blocks = 16384;
// process-global tuning state (pcd_nn_set_*: experiments and tests, not part of the stable ABI): atomics, and every call
// works on ONE snapshot taken at its top, so a setter racing a search on another thread cannot pair the slot bitmap of
// one brick geometry with the kernel of another
static std::atomic<int> g_brick_B{2}, g_brick_R{2}, g_collect_stats{0};
#ifdef PCD_ABLATE   // occupancy experiments (tools/nn_ablate.py): workgroups of the brick kernels per CU
static const int g_brick_blocks_per_cu = std::getenv("PCD_BRICK_BLOCKS") ? std::atoi(std::getenv("PCD_BRICK_BLOCKS")) : PCD_BRICK_MINWAVES;
#else
constexpr int g_brick_blocks_per_cu = PCD_BRICK_MINWAVES;
#endif
// first stage of the grid path: 0 = the clipped brick kernel (brick_clip_kernel.h; needs the default brick geometry
// B = R = 2), 1 = the same kernel with the clip switched off (A/B timing: it then stages the whole region like
// round 3's kernel), 2 = round 3's brick kernel (brick_kernel.h; also what other brick geometries run on).
static std::atomic<int> g_nn_kernel{0};

static std::atomic<int> g_bk_sort{0};   // 1: force the radix-sort bookkeeping (A/B timing, tests)

// slot table of the cloud for this brick geometry (built on first use, rebuilt when the geometry changes)
static pcd_status brick_slots(pcd_cloud* c, QueryScratch* sc, const BrickParams& b, hipStream_t s) {
  const int key[5] = {b.B, b.R, 0, b.Bx, (int)b.nbricks};
  if (sc->bk_slot.p && std::memcmp(key, sc->bk_slot_key, sizeof key) == 0) return PCD_OK;
  PCD_TRY(sc->bk_slot.reserve((size_t)b.nbricks / 32 + 1));
  DevBuf<uint32_t> flag;
  PCD_TRY(flag.reserve((size_t)b.nbricks + 1));
  hipLaunchKernelGGL(k_brick_occupied, dim3(div_up(b.nbricks, 256)), dim3(256), 0, s, c->grid, b, c->cell_start.p, flag.p);
  hipLaunchKernelGGL(k_brick_bitmap, dim3(div_up(div_up(b.nbricks, 32), 256)), dim3(256), 0, s, flag.p, b.nbricks, sc->bk_slot.p);
  PCD_HIP_TRY(hipStreamSynchronize(s));   // one-off per cloud and geometry; `flag` goes out of scope
  sc->bk_nslots = b.nbricks;
  std::memcpy(sc->bk_slot_key, key, sizeof key);
  return PCD_OK;
}

template <int G>
static pcd_status run_grid(pcd_cloud* c, QueryScratch* sc, uint64_t Q, uint64_t* d_keys, hipStream_t s, bool refine) {
  const GridParams& g = c->grid;
  // one snapshot of the tuning state for the whole call
  const int nn_kernel = g_nn_kernel.load(), bk_sort = g_bk_sort.load(), collect_stats = g_collect_stats.load();
  int B = g_brick_B.load(), R = g_brick_R.load();
  if ((B + 2 * R) * (B + 2 * R) > kMaxRows) { B = 2; R = 2; }
  // x-long bricks (Bx = 2B, 3B, 4B: fewer, fuller groups but a longer region per query) were measured on workload M:
  // 0.638 / 0.698 / 0.768 ms against 0.633 ms for cubes (profiles/r02_nn_config_sweeps.txt)
  // x-long bricks (3, 4 cells): with the in-kernel clip 0.529 / 0.551 ms against 0.509 (profiles/r04_nn_experiments.txt)
  const bool clip = nn_kernel != 2 && B == 2 && R == 2;   // other geometries run on round 3's kernel
  const BrickParams b = make_bricks(g, B, B, R);
  PCD_TRY(sc->qsorted.reserve(Q));
  PCD_TRY(sc->ksorted.reserve(Q));
  // fallback list: one slot per query + the chunk slack of every wavefront of the brick kernel (brick_kernel.h)
  // A wavefront leaves a chunk when the next item's unproven queries (<= 8) do not fit: at most 7 of 64 slots stay
  // unused per chunk, so the reserved slots are <= used * 64 / 57 + one chunk per wavefront, used <= Q.
  const size_t fb_cap = Q + Q / 8 + 8 + (size_t)256 * g_brick_blocks_per_cu * 4 * kFbChunk +
                        (size_t)2048 * 4 * 64;   // + the last chunk of every wavefront of k_bk_emit
  PCD_TRY(sc->fb_list.reserve(fb_cap));   // no memset: every reserved slot is written (a query id or the sentinel)
  PCD_TRY(sc->bk_keys.reserve(2 * Q));
  PCD_TRY(sc->bk_vals.reserve(2 * Q));
  PCD_TRY(sc->bk_item.reserve(div_up(Q, kBkTile) + 1));
  PCD_TRY(sc->items.reserve(Q + 1));
  PCD_TRY(sc->counters.reserve(1));
  PCD_HIP_TRY(hipMemsetAsync(sc->counters.p, 0, sizeof(NnCounters), s));
  PCD_TRY(brick_slots(c, sc, b, s));
  // sort key = brick id: coarse key = id >> shift (at most 4096 buckets), fine key = the low bits (at most 4096)
  uint32_t shift = 8;
  while ((((uint64_t)b.nbricks + (1u << shift) - 1) >> shift) > 2304 && shift < 16) ++shift;
  const uint32_t ncoarse = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)b.nbricks + (1u << shift) - 1) >> shift));
  if ((1u << shift) <= kBkMaxFine && ncoarse <= kBkMaxCoarse && bk_sort == 0) {
    ScopedKernelTimer t("nn_brick_bookkeeping", s);
    if (!sc->bk_chist.p) {                 // histogram + cursors: zero once, k_bk_emit leaves them zero
      PCD_TRY(sc->bk_chist.reserve(4 * (size_t)kBkMaxCoarse + 4));
      PCD_HIP_TRY(hipMemsetAsync(sc->bk_chist.p, 0, (4 * (size_t)kBkMaxCoarse + 4) * sizeof(uint32_t), s));
    }
    uint32_t *chist = sc->bk_chist.p, *cstart = chist + kBkMaxCoarse + 1, *ccur = cstart + kBkMaxCoarse + 1,
             *bitems = ccur + kBkMaxCoarse + 1;
    uint32_t *qslot = sc->bk_keys.p, *pslot = sc->bk_keys.p + Q, *pqid = sc->bk_vals.p;
    const unsigned sblocks = (unsigned)std::min<uint64_t>(div_up(Q, 4096), 256);
    hipLaunchKernelGGL(k_bk_slots, dim3(sblocks), dim3(1024), 0, s, sc->qf4.p, (uint32_t)Q, g, b, sc->bk_slot.p, shift,
                       ncoarse, qslot, chist, sc->fb_list.p, sc->counters.p);
    hipLaunchKernelGGL(k_bk_scatter, dim3(div_up(Q, kBkTileQ)), dim3(1024), 0, s, qslot, (uint32_t)Q, shift, ncoarse, chist,
                       cstart, ccur, pslot, pqid, sc->counters.p);
    hipLaunchKernelGGL(k_bk_count<G>, dim3(ncoarse), dim3(256), 0, s, shift, cstart, pslot, bitems);
    hipLaunchKernelGGL(k_bk_emit<G>, dim3(ncoarse), dim3(256), 0, s, sc->qf4.p, refine ? d_keys : (const uint64_t*)nullptr,
                       g, b, shift, cstart, pslot, pqid, bitems, chist, ccur, sc->items.p, sc->qsorted.p, sc->ksorted.p,
                       sc->counters.p);
  } else {
    // (grids with more than 8 M occupied bricks: the radix-sort bookkeeping of round 2)
    ScopedKernelTimer t("nn_brick_bookkeeping", s);
    uint32_t *k0 = sc->bk_keys.p, *k1 = sc->bk_keys.p + Q, *v0 = sc->bk_vals.p, *v1 = sc->bk_vals.p + Q;
    hipLaunchKernelGGL(k_brick_keys, dim3(div_up(Q, 256)), dim3(256), 0, s, sc->qf4.p, Q, g, b, k0, v0);
    unsigned end_bit = 1;
    while (end_bit < 32 && ((uint64_t)1 << end_bit) <= (uint64_t)b.nbricks + 1) ++end_bit;
    size_t tb = 0;
#ifndef PCD_SORT_MERGE_LIMIT
#define PCD_SORT_MERGE_LIMIT 262144
#endif
    // rocPRIM's default takes its merge sort up to 2^20 items: Onesweep above 256 k
    using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                               rocprim::default_config, PCD_SORT_MERGE_LIMIT>;
    PCD_HIP_TRY(rocprim::radix_sort_pairs<SortCfg>(nullptr, tb, k0, k1, v0, v1, (unsigned)Q, 0u, end_bit, s));
    PCD_TRY(sc->tmp.reserve(tb));
    PCD_HIP_TRY(rocprim::radix_sort_pairs<SortCfg>(sc->tmp.p, tb, k0, k1, v0, v1, (unsigned)Q, 0u, end_bit, s));
    const unsigned ntiles = div_up(Q, kBkTile);
    hipLaunchKernelGGL(k_brick_tile_count<G>, dim3(ntiles), dim3(256), 0, s, k1, (uint32_t)Q, b.nbricks, sc->bk_item.p);
    hipLaunchKernelGGL(k_brick_emit<G>, dim3(ntiles), dim3(256), 0, s, k1, v1, sc->bk_item.p, sc->qf4.p,
                       refine ? d_keys : (const uint64_t*)nullptr, (uint32_t)Q, b.nbricks, (uint32_t)b.nb[0],
                       (uint32_t)b.nb[1], sc->items.p, sc->qsorted.p, sc->ksorted.p, sc->fb_list.p, sc->counters.p);
  }
  {
    ScopedKernelTimer t("nn_brick", s);
    const unsigned blocks = (unsigned)std::min<uint64_t>(div_up(div_up(Q, G), 4) + 1, 256 * (uint64_t)g_brick_blocks_per_cu);
    if (clip) {
      const int fl = (collect_stats & ~2) | (nn_kernel == 1 ? 2 : 0);
      hipLaunchKernelGGL(k_nn_brick_clip, dim3(blocks), dim3(256), 0, s, g, c->sorted.p, c->cell_start.p, sc->qsorted.p,
                         sc->ksorted.p, sc->items.p, sc->counters.p, d_keys, sc->fb_list.p, &sc->counters.p->fb_count, fl);
    } else
      hipLaunchKernelGGL(k_nn_brick<G>, dim3(blocks), dim3(256), 0, s, g, b, c->sorted.p, c->cell_start.p,
                         sc->qsorted.p, sc->ksorted.p, sc->items.p, sc->counters.p, d_keys, sc->fb_list.p,
                         &sc->counters.p->fb_count, collect_stats);
  }
  {
    ScopedKernelTimer t("nn_fallback", s);
    PCD_TRY(sc->fb_dense.reserve(Q));
    hipLaunchKernelGGL(k_fb_compact, dim3(div_up(fb_cap, kFbcThreads * kFbcPer)), dim3(kFbcThreads), 0, s, sc->fb_list.p,
                       &sc->counters.p->fb_count, sc->fb_dense.p, &sc->counters.p->pad[0]);
    const unsigned blocks = (unsigned)std::min<uint64_t>(div_up(Q, 4), g_fb_max_blocks);
    hipLaunchKernelGGL(k_nn_fallback<0>, dim3(blocks), dim3(256), 0, s, g, c->pyr, c->sorted.p, c->cell_start.p,
                       c->blk_aabb.p, sc->qf4.p, sc->fb_dense.p, &sc->counters.p->pad[0], 0u, d_keys,
                       sc->counters.p, collect_stats, FbFused{nullptr, nullptr, 0, 0.0});
  }
  return PCD_OK;
}

struct NnBound {        // gate-bounded search: per-query / scalar ranges, or one fixed range; count == 0: unbounded
  const double* d_max_range = nullptr;
  uint64_t count = 0;
  double fixed = 0.0;
};

static pcd_status nn_device(pcd_cloud* c, const double* d_q, uint64_t Q, int algo, uint64_t* d_keys,
                            hipStream_t s, bool refine = false, const uint8_t* d_skip = nullptr,
                            const NnBound* bound = nullptr) {
  QueryScratch* sc = scratch_of(c);
  if (Q == 0) return PCD_OK;
  PCD_REQUIRE(Q < 0xFFFFFFF0ull, "more than 2^32 queries in one call");
  const int collect_stats = g_collect_stats.load();
  // Small batches (and PCD_NN_FALLBACK_ONLY): ONE launch -- k_nn_fallback converts the queries itself and writes
  // finalised keys, so the per-call latency is one kernel instead of prepare + memset + search + finalize.
  const bool one_launch = c->m > 0 && !refine &&
                          (algo == PCD_NN_FALLBACK_ONLY || (algo == PCD_NN_AUTO && Q <= kSmallBatch));
  if (one_launch) {
    PCD_TRY(sc->counters.reserve(1));
    if (collect_stats) PCD_HIP_TRY(hipMemsetAsync(sc->counters.p, 0, sizeof(NnCounters), s));
    ScopedKernelTimer t("nn_fallback", s);
    const unsigned blocks = (unsigned)std::min<uint64_t>(div_up(Q, 4), g_fb_max_blocks);
    const FbFused fu{d_q, bound ? bound->d_max_range : nullptr, bound ? bound->count : 0, bound ? bound->fixed : 0.0};
    if (bound)
      hipLaunchKernelGGL(k_nn_fallback<2>, dim3(blocks), dim3(256), 0, s, c->grid, c->pyr, c->sorted.p, c->cell_start.p,
                         c->blk_aabb.p, (const float4*)nullptr, (const uint32_t*)nullptr,
                         (const uint32_t*)nullptr, (uint32_t)Q, d_keys, sc->counters.p, collect_stats, fu);
    else
      hipLaunchKernelGGL(k_nn_fallback<1>, dim3(blocks), dim3(256), 0, s, c->grid, c->pyr, c->sorted.p, c->cell_start.p,
                         c->blk_aabb.p, (const float4*)nullptr, (const uint32_t*)nullptr,
                         (const uint32_t*)nullptr, (uint32_t)Q, d_keys, sc->counters.p, collect_stats, fu);
    PCD_HIP_TRY(hipGetLastError());
    return PCD_OK;
  }
  PCD_TRY(sc->qf4.reserve(Q));
  {
    ScopedKernelTimer t("nn_prepare", s);
    if (bound)
      hipLaunchKernelGGL(k_prepare_bounded, dim3(div_up(Q, 256)), dim3(256), 0, s, d_q, Q, bound->d_max_range,
                         bound->count, bound->fixed, sc->qf4.p, d_keys);
    else if (refine)
      hipLaunchKernelGGL(k_prepare_refine, dim3(div_up(Q, 256)), dim3(256), 0, s, d_q, Q, d_skip, c->bb_lo[0],
                         c->bb_lo[1], c->bb_lo[2], c->bb_hi[0], c->bb_hi[1], c->bb_hi[2], sc->qf4.p, d_keys);
    else
      hipLaunchKernelGGL(k_prepare_queries, dim3(div_up(Q, 256)), dim3(256), 0, s, d_q, Q, sc->qf4.p, d_keys);
  }
  if (c->m > 0) {
    if (algo == PCD_NN_BRUTEFORCE) {
      ScopedKernelTimer t("nn_bruteforce", s);
      // enough chunks to fill the chip even for few queries
      const unsigned qblocks = div_up(Q, 256);
      unsigned chunks = std::max(1u, std::min<unsigned>(div_up(c->n, 4096), 2048u / std::max(1u, qblocks)));
      chunks = std::min(chunks, 65535u);
      const uint64_t chunk = ((c->n + chunks - 1) / chunks + 1023) / 1024 * 1024;
      chunks = div_up(c->n, chunk);
      hipLaunchKernelGGL(k_nn_bruteforce, dim3(qblocks, chunks), dim3(256), 0, s, c->pts4.p, c->n, c->index_base,
                         c->index_stride, c->row_index.p, sc->qf4.p, Q, chunk, d_keys);
    } else if (algo == PCD_NN_FALLBACK_ONLY || (algo == PCD_NN_AUTO && Q <= kSmallBatch)) {
      // (refining another shard's keys: the keys come in, so the prepare / finalize kernels stay)
      PCD_TRY(sc->counters.reserve(1));
      PCD_HIP_TRY(hipMemsetAsync(sc->counters.p, 0, sizeof(NnCounters), s));
      ScopedKernelTimer t("nn_fallback", s);
      const unsigned blocks = (unsigned)std::min<uint64_t>(div_up(Q, 4), g_fb_max_blocks);
      hipLaunchKernelGGL(k_nn_fallback<0>, dim3(blocks), dim3(256), 0, s, c->grid, c->pyr, c->sorted.p, c->cell_start.p,
                         c->blk_aabb.p, sc->qf4.p, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t)Q,
                         d_keys, sc->counters.p, collect_stats, FbFused{nullptr, nullptr, 0, 0.0});
    } else if (algo == PCD_NN_AUTO || algo == PCD_NN_GRID) {
      PCD_TRY(run_grid<8>(c, sc, Q, d_keys, s, refine || bound != nullptr));   // incoming keys matter: carry them
    } else {
      set_error("unknown nn algo %d", algo);
      return PCD_ERR_INVALID;
    }
  }
  if (c->m == 0 || algo == PCD_NN_BRUTEFORCE) {   // raw keys left behind: the brute-force kernel's atomicMin, an empty cloud
    ScopedKernelTimer t("nn_finalize", s);
    hipLaunchKernelGGL(k_finalize_keys, dim3(div_up(Q, 256)), dim3(256), 0, s, d_keys, Q);
  }
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

pcd_status nn_query_device_internal(pcd_cloud* c, const double* d_q, uint64_t Q, int algo, uint64_t* d_keys,
                                    hipStream_t s) {
  return nn_device(c, d_q, Q, algo, d_keys, s);
}
// bounded by the association gate: d_max_range (1 or Q entries) or, when NULL, fixed_range
pcd_status nn_query_bounded_internal(pcd_cloud* c, const double* d_q, uint64_t Q, const double* d_max_range,
                                     uint64_t mr_count, double fixed_range, uint64_t* d_keys, hipStream_t s) {
  NnBound b;
  b.d_max_range = d_max_range; b.count = mr_count; b.fixed = fixed_range;
  return nn_device(c, d_q, Q, PCD_NN_AUTO, d_keys, s, false, nullptr, &b);
}

}  // namespace pcd

using namespace pcd;

extern "C" {

pcd_status pcd_nn_query_device(pcd_cloud* c, const double* d_q_xyz, uint64_t Q, int algo, uint64_t* d_keys,
                               void* stream) {
  PCD_REQUIRE(c, "null cloud");
  PCD_REQUIRE(Q == 0 || (d_q_xyz && d_keys), "null pointer");
  PCD_REFUSE_CAPTURE(stream);
  PCD_HIP_TRY(hipSetDevice(c->device));
  return nn_device(c, d_q_xyz, Q, algo, d_keys, (hipStream_t)stream);
}

pcd_status pcd_nn_refine_device(pcd_cloud* c, const double* d_q_xyz, uint64_t Q, const uint8_t* d_skip,
                                uint64_t* d_keys, void* stream) {
  PCD_REQUIRE(c, "null cloud");
  PCD_REQUIRE(Q == 0 || (d_q_xyz && d_keys), "null pointer");
  PCD_REFUSE_CAPTURE(stream);
  PCD_HIP_TRY(hipSetDevice(c->device));
  return nn_device(c, d_q_xyz, Q, PCD_NN_AUTO, d_keys, (hipStream_t)stream, /*refine=*/true, d_skip);
}

pcd_status pcd_nn_query_algo(pcd_cloud* c, const double* q_xyz, uint64_t Q, int algo, uint32_t* idx, float* sqdist,
                             uint8_t* found) {
  PCD_REQUIRE(c, "null cloud");
  PCD_REQUIRE(Q == 0 || (q_xyz && idx && sqdist && found), "null pointer");
  if (Q == 0) return PCD_OK;
  PCD_HIP_TRY(hipSetDevice(c->device));
  QueryScratch* sc = scratch_of(c);
  hipStream_t s = nullptr;
  PCD_TRY(sc->d_q.reserve(3 * Q));
  PCD_TRY(sc->keys.reserve(Q));
  PCD_TRY(sc->d_idx.reserve(Q));
  PCD_TRY(sc->d_sq.reserve(Q));
  PCD_TRY(sc->d_found.reserve(Q));
  PCD_HIP_TRY(hipMemcpyAsync(sc->d_q.p, q_xyz, 3 * Q * sizeof(double), hipMemcpyHostToDevice, s));
  PCD_TRY(nn_device(c, sc->d_q.p, Q, algo, sc->keys.p, s));
  hipLaunchKernelGGL(k_unpack_keys, dim3(div_up(Q, 256)), dim3(256), 0, s, sc->keys.p, Q, sc->d_idx.p, sc->d_sq.p,
                     sc->d_found.p);
  PCD_HIP_TRY(hipMemcpyAsync(idx, sc->d_idx.p, Q * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  PCD_HIP_TRY(hipMemcpyAsync(sqdist, sc->d_sq.p, Q * sizeof(float), hipMemcpyDeviceToHost, s));
  PCD_HIP_TRY(hipMemcpyAsync(found, sc->d_found.p, Q * sizeof(uint8_t), hipMemcpyDeviceToHost, s));
  PCD_HIP_TRY(hipStreamSynchronize(s));
  return PCD_OK;
}

pcd_status pcd_nn_query(pcd_cloud* c, const double* q_xyz, uint64_t Q, uint32_t* idx, float* sqdist, uint8_t* found) {
  return pcd_nn_query_algo(c, q_xyz, Q, PCD_NN_AUTO, idx, sqdist, found);
}

pcd_status pcd_nn_last_stats(pcd_cloud* c, pcd_nn_stats* st) {
  PCD_REQUIRE(c && st, "null pointer");
  std::memset(st, 0, sizeof *st);
  if (!c->scratch || !c->scratch->counters.p) return PCD_OK;
  PCD_HIP_TRY(hipSetDevice(c->device));
  PCD_HIP_TRY(hipDeviceSynchronize());
  NnCounters h;
  PCD_HIP_TRY(hipMemcpy(&h, c->scratch->counters.p, sizeof h, hipMemcpyDeviceToHost));
  st->brick_groups = h.brick_groups;
  st->staged_points = h.staged_points;
  st->fallback_queries = h.fallback_queries;
  st->fallback_points = h.fallback_points;
  st->pair_evals = h.pair_evals;
  static const bool print_walks = std::getenv("PCD_FB_STATS") != nullptr;   // read once per process
  if (print_walks)
    std::fprintf(stderr, "[pcd] fallback walk: %llu queries, %.1f steps / %.1f leaf scans per query, max %u / %u\n",
                 (unsigned long long)h.fallback_queries, h.fallback_queries ? (double)h.fb_steps / h.fallback_queries : 0.0,
                 h.fallback_queries ? (double)h.fb_leaves / h.fallback_queries : 0.0, h.fb_max_steps, h.fb_max_leaves);
  return PCD_OK;
}

/* tuning hook: which kernel serves the grid path's first stage (g_nn_kernel above).  Negative: leave it as it is. */
pcd_status pcd_nn_set_search(int kernel) {
  if (kernel < 0) return PCD_OK;
  PCD_REQUIRE(kernel <= 2, "kernel must be 0 (clipped brick kernel), 1 (the same, clip off) or 2 (round 3's brick kernel)");
  g_nn_kernel = kernel;
  return PCD_OK;
}

pcd_status pcd_nn_set_bookkeeping(int radix_sort) {
  g_bk_sort = radix_sort ? 1 : 0;
  return PCD_OK;
}

/* tuning hooks (not part of the stable ABI; used by bench.py and tests) */
pcd_status pcd_nn_set_tuning(int brick_cells, int halo_cells, int collect_stats) {
  if (brick_cells > 0) g_brick_B = brick_cells;
  if (halo_cells >= 0) g_brick_R = halo_cells;
#ifdef PCD_ABLATE
  g_collect_stats = collect_stats;
#else
  g_collect_stats = collect_stats & 1;   // the ablation bits exist only in -DPCD_ABLATE builds
#endif
  return PCD_OK;
}

}  // extern "C"
