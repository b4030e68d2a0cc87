/*
 * oracle/nn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float32, no FMA: compile with -ffp-contract=off)
 * of the nearest-neighbour association path of colmap-pcd.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY STATUS: "parity unpinned" by the reference -- src/lidar has no
 * tests and ships no fixtures, and PCL/FLANN are not
 * installed, so the reference cannot be built here.  The restatement is
 * pinned instead by (1) the published FLANN L2_Simple<float> arithmetic,
 * (2) agreement brute-force <-> exact KD-tree below, (3) scipy cKDTree
 * on float32-representable inputs (tests/test_oracle_cpu.py).
 *
 * Reference lines followed:
 *   src/lidar/ply.cc:33-57    PointCloudDirectionTrans (axis swap, NaN rows dropped)
 *   src/lidar/ply.cc:90-107   SearchNearestNeiborByKdtree (double->float query,
 *                             float->double result, NaN / |n|<1e-6 reject)
 *   src/lidar/kdtree.cc:5-21  BuildMap / GetClosestPoint (k = 1, exact)
 *   [3P] pcl::KdTreeFLANN<PointT, flann::L2_Simple<float>> (PCL 1.10-1.12),
 *        flann::KDTreeSingleIndex (FLANN 1.9.1), leaf_max_size 15, 3 dims:
 *          diff = a[i] - b[i]; result += diff*diff;   (float, i = x,y,z)
 *        a candidate replaces the current best only if dist < worst, the
 *        initial worst being FLT_MAX (so an all-Inf cloud finds nothing).
 *   Tie rule of THIS build (FLANN's depends on unreproducible tree order):
 *        equal float distance -> lowest post-filter cloud index.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- ply.cc:33-57 ------------------------------------------------------ */
/* in : raw LiDAR-frame rows, xyz_in/nrm_in [n][3]
 * out: visual-frame rows with NaN rows dropped, order preserved.
 * returns number of rows kept. */
uint64_t oracle_direction_trans(const float* xyz_in, const float* nrm_in, uint64_t n,
                                float* xyz_out, float* nrm_out) {
  uint64_t m = 0;
  for (uint64_t i = 0; i < n; ++i) {
    float x = -xyz_in[3 * i + 1];
    float y = -xyz_in[3 * i + 2];
    float z = xyz_in[3 * i + 0];
    float nx = -nrm_in[3 * i + 1];
    float ny = -nrm_in[3 * i + 2];
    float nz = nrm_in[3 * i + 0];
    if (isnan(x) || isnan(y) || isnan(z) || isnan(nx) || isnan(ny) || isnan(nz)) continue;
    xyz_out[3 * m + 0] = x; xyz_out[3 * m + 1] = y; xyz_out[3 * m + 2] = z;
    nrm_out[3 * m + 0] = nx; nrm_out[3 * m + 1] = ny; nrm_out[3 * m + 2] = nz;
    ++m;
  }
  return m;
}

/* ---- FLANN L2_Simple<float>, 3 dims ------------------------------------ */
static inline float l2_simple3(const float* a, const float* b) {
  float result = 0.0f;
  float diff;
  diff = a[0] - b[0]; result += diff * diff;
  diff = a[1] - b[1]; result += diff * diff;
  diff = a[2] - b[2]; result += diff * diff;
  return result;
}

/* ---- exact brute force: the ground truth ------------------------------- */
/* q_xyz are doubles (Eigen::Vector3d at ply.cc:90), cast to float first
 * (ply.cc:92).  found[i] = 0 when nothing beats FLT_MAX or the query is
 * not finite. */
void oracle_nn_bruteforce(const float* xyz, uint64_t n, const double* q_xyz, uint64_t nq,
                          uint32_t* idx, float* sqdist, uint8_t* found) {
  for (uint64_t j = 0; j < nq; ++j) {
    float q[3] = {(float)q_xyz[3 * j], (float)q_xyz[3 * j + 1], (float)q_xyz[3 * j + 2]};
    float best = FLT_MAX;
    uint32_t bi = 0xFFFFFFFFu;
    if (isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2])) {
      for (uint64_t i = 0; i < n; ++i) {
        float d = l2_simple3(q, xyz + 3 * i);
        if (d < best) { best = d; bi = (uint32_t)i; } /* strict: first (lowest) index wins ties */
      }
    }
    idx[j] = bi;
    sqdist[j] = best;
    found[j] = (bi != 0xFFFFFFFFu);
  }
}

/* ---- exact single KD-tree (FLANN KDTreeSingleIndex shape) --------------- */
/* Used (a) as a second, independent exact implementation to cross-check
 * the brute force and (b) as the timed CPU baseline in bench.py: it is the
 * data structure the reference queries in its serial loops
 * (controllers/bundle_adjustment.cc:130-185). */
typedef struct {
  int32_t left, right;   /* child node ids, -1 for leaf            */
  int32_t begin, end;    /* leaf: range in the reordered point set */
  int32_t cut_dim;
  float cut_lo, cut_hi;  /* max of left side / min of right side along cut_dim */
} kd_node;

typedef struct {
  uint64_t n;
  float* pts;        /* reordered xyz [n][3] */
  uint32_t* orig;    /* reordered -> original index */
  kd_node* nodes;
  int32_t num_nodes, cap_nodes;
  float bb_lo[3], bb_hi[3];
  int leaf_max;
} kd_tree;

static int32_t kd_new_node(kd_tree* t) {
  if (t->num_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
  }
  return t->num_nodes++;
}

static int32_t kd_build_rec(kd_tree* t, const float* xyz, uint32_t* ind, int32_t begin, int32_t end,
                            float* lo, float* hi) {
  int32_t id = kd_new_node(t);
  if (end - begin <= t->leaf_max) {
    t->nodes[id].left = t->nodes[id].right = -1;
    t->nodes[id].begin = begin;
    t->nodes[id].end = end;
    t->nodes[id].cut_dim = 0;
    t->nodes[id].cut_lo = t->nodes[id].cut_hi = 0.f;
    /* tighten bbox to the leaf's points (FLANN does the same) */
    for (int d = 0; d < 3; ++d) { lo[d] = FLT_MAX; hi[d] = -FLT_MAX; }
    for (int32_t i = begin; i < end; ++i)
      for (int d = 0; d < 3; ++d) {
        float v = xyz[3 * (size_t)ind[i] + d];
        if (v < lo[d]) lo[d] = v;
        if (v > hi[d]) hi[d] = v;
      }
    return id;
  }
  /* middle split on the widest bbox dimension */
  int dim = 0;
  float span = hi[0] - lo[0];
  for (int d = 1; d < 3; ++d)
    if (hi[d] - lo[d] > span) { span = hi[d] - lo[d]; dim = d; }
  float cut = 0.5f * (lo[dim] + hi[dim]);
  /* clamp the cut into the actual data range along dim */
  float mn = FLT_MAX, mx = -FLT_MAX;
  for (int32_t i = begin; i < end; ++i) {
    float v = xyz[3 * (size_t)ind[i] + dim];
    if (v < mn) mn = v;
    if (v > mx) mx = v;
  }
  if (cut < mn) cut = mn;
  if (cut > mx) cut = mx;
  /* three-way partition: < cut | == cut | > cut, then balance the equals */
  int32_t l = begin, r = end - 1;
  for (;;) {
    while (l <= r && xyz[3 * (size_t)ind[l] + dim] < cut) ++l;
    while (l <= r && xyz[3 * (size_t)ind[r] + dim] >= cut) --r;
    if (l > r) break;
    uint32_t tmp = ind[l]; ind[l] = ind[r]; ind[r] = tmp; ++l; --r;
  }
  int32_t lim1 = l;
  r = end - 1;
  for (;;) {
    while (l <= r && xyz[3 * (size_t)ind[l] + dim] <= cut) ++l;
    while (l <= r && xyz[3 * (size_t)ind[r] + dim] > cut) --r;
    if (l > r) break;
    uint32_t tmp = ind[l]; ind[l] = ind[r]; ind[r] = tmp; ++l; --r;
  }
  int32_t lim2 = l;
  int32_t half = (end - begin) / 2, split;
  if (lim1 - begin > half) split = lim1;
  else if (lim2 - begin < half) split = lim2;
  else split = begin + half;
  if (split == begin || split == end) split = begin + half; /* all equal along dim */

  float llo[3], lhi[3], rlo[3], rhi[3];
  memcpy(llo, lo, sizeof llo); memcpy(lhi, hi, sizeof lhi);
  memcpy(rlo, lo, sizeof rlo); memcpy(rhi, hi, sizeof rhi);
  lhi[dim] = cut;
  rlo[dim] = cut;
  int32_t lc = kd_build_rec(t, xyz, ind, begin, split, llo, lhi);
  int32_t rc = kd_build_rec(t, xyz, ind, split, end, rlo, rhi);
  t->nodes[id].left = lc;
  t->nodes[id].right = rc;
  t->nodes[id].begin = begin;
  t->nodes[id].end = end;
  t->nodes[id].cut_dim = dim;
  t->nodes[id].cut_lo = lhi[dim];
  t->nodes[id].cut_hi = rlo[dim];
  for (int d = 0; d < 3; ++d) {
    lo[d] = llo[d] < rlo[d] ? llo[d] : rlo[d];
    hi[d] = lhi[d] > rhi[d] ? lhi[d] : rhi[d];
  }
  return id;
}

/* points with a non-finite coordinate never enter the tree (they can never
 * beat FLT_MAX, see header) but keep their index slot. */
kd_tree* oracle_kdtree_build(const float* xyz, uint64_t n, int leaf_max) {
  kd_tree* t = (kd_tree*)calloc(1, sizeof(kd_tree));
  t->leaf_max = leaf_max > 0 ? leaf_max : 15;
  uint32_t* ind = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
  uint64_t m = 0;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint64_t i = 0; i < n; ++i) {
    const float* p = xyz + 3 * i;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    ind[m++] = (uint32_t)i;
    for (int d = 0; d < 3; ++d) {
      if (p[d] < lo[d]) lo[d] = p[d];
      if (p[d] > hi[d]) hi[d] = p[d];
    }
  }
  t->n = m;
  memcpy(t->bb_lo, lo, sizeof lo);
  memcpy(t->bb_hi, hi, sizeof hi);
  if (m) kd_build_rec(t, xyz, ind, 0, (int32_t)m, lo, hi);
  t->pts = (float*)malloc(sizeof(float) * 3 * (size_t)(m ? m : 1));
  for (uint64_t i = 0; i < m; ++i) memcpy(t->pts + 3 * i, xyz + 3 * (size_t)ind[i], 3 * sizeof(float));
  t->orig = ind;
  return t;
}

void oracle_kdtree_free(kd_tree* t) {
  if (!t) return;
  free(t->pts); free(t->orig); free(t->nodes); free(t);
}

typedef struct { float best; uint32_t bi; } kd_result;

static void kd_search_rec(const kd_tree* t, int32_t id, const float* q, float mindist, float* dists,
                          kd_result* r) {
  const kd_node* nd = &t->nodes[id];
  if (nd->left < 0) {
    for (int32_t i = nd->begin; i < nd->end; ++i) {
      float d = l2_simple3(q, t->pts + 3 * (size_t)i);
      uint32_t oi = t->orig[i];
      if (d < r->best || (d == r->best && oi < r->bi)) { r->best = d; r->bi = oi; }
    }
    return;
  }
  int dim = nd->cut_dim;
  float val = q[dim];
  float diff1 = val - nd->cut_lo, diff2 = val - nd->cut_hi;
  int32_t best_child, other_child;
  float cut_dist;
  if (diff1 + diff2 < 0) { best_child = nd->left; other_child = nd->right; cut_dist = diff2 * diff2; }
  else { best_child = nd->right; other_child = nd->left; cut_dist = diff1 * diff1; }
  kd_search_rec(t, best_child, q, mindist, dists, r);
  float dst = dists[dim];
  mindist = mindist + cut_dist - dst;
  dists[dim] = cut_dist;
  /* <= (not <): an equal-distance point with a lower index may live there.
   * mindist is accumulated in float in a different order than the leaf
   * distance, so it is deflated by a few ulp before the test (extra visits
   * only; exactness is checked against the brute force in tests/). */
  if (mindist * (1.0f - 8.0f * FLT_EPSILON) <= r->best) kd_search_rec(t, other_child, q, mindist, dists, r);
  dists[dim] = dst;
}

void oracle_kdtree_query(const kd_tree* t, const double* q_xyz, uint64_t nq, uint32_t* idx,
                         float* sqdist, uint8_t* found) {
  for (uint64_t j = 0; j < nq; ++j) {
    float q[3] = {(float)q_xyz[3 * j], (float)q_xyz[3 * j + 1], (float)q_xyz[3 * j + 2]};
    kd_result r = {FLT_MAX, 0xFFFFFFFFu};
    if (t->n && isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2])) {
      float dists[3] = {0, 0, 0};
      /* NOTE: the float bound below can exceed the float leaf distance by
       * rounding when summed in a different order, so it is deflated by a
       * relative 4 ulp before use; this only costs extra visits. */
      float mind = 0.f;
      for (int d = 0; d < 3; ++d) {
        float df = 0.f;
        if (q[d] < t->bb_lo[d]) df = q[d] - t->bb_lo[d];
        if (q[d] > t->bb_hi[d]) df = q[d] - t->bb_hi[d];
        dists[d] = df * df * (1.0f - 4.0f * FLT_EPSILON);
        mind += dists[d];
      }
      kd_search_rec(t, 0, q, mind * (1.0f - 4.0f * FLT_EPSILON), dists, &r);
    }
    idx[j] = r.bi;
    sqdist[j] = r.best;
    found[j] = (r.bi != 0xFFFFFFFFu);
  }
}

/* ---- ply.cc:90-107: query wrapper result ------------------------------- */
/* out6[j] = (double)xyz, (double)normal of the winner; ok[j] = 0 when the
 * search fails, out6 has a NaN, or ||n|| < 1e-6 (double). */
void oracle_search_nearest_neibor(const float* xyz, const float* nrm, const uint32_t* idx,
                                  const uint8_t* found, uint64_t nq, double* out6, uint8_t* ok) {
  for (uint64_t j = 0; j < nq; ++j) {
    double* o = out6 + 6 * j;
    ok[j] = 0;
    for (int k = 0; k < 6; ++k) o[k] = 0.0;
    if (!found[j]) continue;
    size_t i = idx[j];
    for (int k = 0; k < 3; ++k) { o[k] = (double)xyz[3 * i + k]; o[3 + k] = (double)nrm[3 * i + k]; }
    int bad = 0;
    for (int k = 0; k < 6; ++k) bad |= isnan(o[k]);
    if (bad) continue;
    double nn = sqrt(o[3] * o[3] + o[4] * o[4] + o[5] * o[5]);
    if (nn < 1e-6) continue;
    ok[j] = 1;
  }
}
