/*
 * oracle/assoc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, double, compile with -ffp-contract=off) of the
 * LiDAR plane-association step that follows the KD-tree lookup.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it.
 *
 * PARITY STATUS: "parity unpinned" by the reference (src/lidar and the
 * association loops have no tests); pinned by closed-form checks in
 * tests/test_oracle_cpu.py (hand-computed cases) and by numpy float64
 * re-derivation.
 *
 * Reference lines followed:
 *   src/lidar/lidar_point.cc:5-16   ctor -> Normalize
 *   src/lidar/lidar_point.cc:21-25  ComputeDist            |pt.n + d|
 *   src/lidar/lidar_point.cc:27-30  ComputePointToPointDist ||xyz - pt||
 *   src/lidar/lidar_point.cc:32-37  ComputeAngle           |n.(pt-xyz)| / ||pt-xyz||
 *   src/lidar/lidar_point.cc:39-50  Normalize
 *   src/optim/bundle_adjustment.cc:358-410          MatchClosestLidarPoint   (gate mode 0)
 *   src/sfm/incremental_mapper.cc:1413-1469         AdjustGlobalBundleByLidar (gate mode 1)
 *   src/controllers/bundle_adjustment.cc:130-185    BundleAdjustmentController::Run (gate mode 2)
 */
#include <math.h>
#include <stdint.h>

enum { GATE_MAPPER_LOCAL = 0, GATE_MAPPER_GLOBAL = 1, GATE_CONTROLLER = 2 };
enum { TYPE_NONE = 0, TYPE_ICP = 1, TYPE_ICP_GROUND = 2 };

/* One query.  X = 3D feature point (double), l6 = winner (xyz, normal) as
 * doubles-of-floats, ok = result of SearchNearestNeiborByKdtree.
 * Outputs: abcd (normalised plane), type, dist (point-to-point), angle,
 * dist2plane.  Returns the type (0 = no association recorded). */
static int assoc_one(const double* X, const double* l6, int ok, double max_range, int gate_mode,
                     double* abcd, double* dist, double* angle, double* dist2plane) {
  abcd[0] = abcd[1] = abcd[2] = abcd[3] = 0.0;
  *dist = 0.0; *angle = 0.0; *dist2plane = 0.0;
  if (!ok) return TYPE_NONE;
  const double lx = l6[0], ly = l6[1], lz = l6[2];
  const double nx = l6[3], ny = l6[4], nz = l6[5];
  /* plane << norm, 0 - l_pt.dot(norm) is overwritten by Normalize(); only abc survive. */
  /* lidar_point.cc:39-50 */
  double a = nx, b = ny, c = nz;
  double norm = sqrt(a * a + b * b + c * c);
  a = a / norm;
  b = b / norm;
  c = c / norm;
  double d = 0 - a * lx - b * ly - c * lz;
  abcd[0] = a; abcd[1] = b; abcd[2] = c; abcd[3] = d;

  /* lidar_point.cc:21-37 */
  const double vx = X[0] - lx, vy = X[1] - ly, vz = X[2] - lz; /* pt - xyz_ */
  /* (xyz_ - pt).norm(): squares are sign-independent */
  const double p2p = sqrt(vx * vx + vy * vy + vz * vz);
  const double d2p = fabs((X[0] * a + X[1] * b + X[2] * c) + d);
  const double ang = fabs((a * vx + b * vy + c * vz) / p2p);
  *dist2plane = d2p;

  /* classification on the RAW normal (IEEE division: x/0 = inf passes, 0/0 = NaN fails) */
  const int ground = (fabs(ny / nx) > 10) && (fabs(ny / nz) > 10);
  const int type = ground ? TYPE_ICP_GROUND : TYPE_ICP;

  if (gate_mode == GATE_CONTROLLER) {
    /* controllers/bundle_adjustment.cc:156-160 */
    if (d2p > 1 || p2p > 2) return TYPE_NONE;
  } else {
    /* bundle_adjustment.cc:398-400 / incremental_mapper.cc:1462-1463 */
    if (p2p > max_range) return TYPE_NONE;
  }
  *dist = p2p;   /* bundle_adjustment.cc:403; the other two call sites leave it unset */
  *angle = ang;  /* bundle_adjustment.cc:404 */
  return type;
}

/* Batched.  max_range has nq entries (per-point schedule,
 * incremental_mapper.cc:1159-1163) or is NULL for the controller mode. */
void oracle_associate(const double* X, const double* l6, const uint8_t* ok, const double* max_range,
                      uint64_t nq, int gate_mode, double* out_abcd, uint8_t* out_type,
                      double* out_dist, double* out_angle, double* out_dist2plane) {
  for (uint64_t j = 0; j < nq; ++j) {
    double mr = max_range ? max_range[j] : 0.0;
    out_type[j] = (uint8_t)assoc_one(X + 3 * j, l6 + 6 * j, ok[j], mr, gate_mode, out_abcd + 4 * j,
                                     out_dist + j, out_angle + j, out_dist2plane + j);
  }
}

/* incremental_mapper.cc:1159-1163 / 1423-1427: per-point search-range schedule */
void oracle_search_range_schedule(const int32_t* opt_num, uint64_t n, double kd_max, double kd_min,
                                  double drop, double* out) {
  for (uint64_t i = 0; i < n; ++i) {
    double r = kd_max - opt_num[i] * drop;
    if (r <= kd_min) r = kd_min;
    out[i] = r;
  }
}

/* base/reconstruction.cc:771-805 FilterLidarOutlier: an association is dropped when the point-to-point
 * distance between the (possibly re-optimised) 3D point and its LiDAR point exceeds the bound of its
 * type (Proj: max_proj_dist_error, Icp / IcpGround: max_icp_dist_error).  type: 0 none (kept untouched),
 * 1 Icp, 2 IcpGround, 3 Proj.  out[i] = 1 when the association is erased. */
void oracle_filter_lidar_outlier(const double* X, const double* lidar_xyz, const uint8_t* type, uint64_t n,
                                 double max_proj_dist_error, double max_icp_dist_error, uint8_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    out[i] = 0;
    if (type[i] == 0) continue;
    const double vx = lidar_xyz[3 * i] - X[3 * i], vy = lidar_xyz[3 * i + 1] - X[3 * i + 1],
                 vz = lidar_xyz[3 * i + 2] - X[3 * i + 2];
    const double dist = sqrt(vx * vx + vy * vy + vz * vz);
    const double bound = type[i] == 3 ? max_proj_dist_error : max_icp_dist_error;
    if (dist > bound) out[i] = 1;
  }
}
