/*
 * oracle/proj_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, compile with -ffp-contract=off) of the
 * depth-projection association `PcdProj`: 1 m voxel
 * "submaps", frustum culling on voxel centres, splat projection of every
 * LiDAR point of the surviving voxels into a down-scaled image with a
 * nearest-wins buffer on the feature pixels.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline may load it.
 *
 * PARITY STATUS: "parity unpinned".  src/lidar has no tests or fixtures and the
 * file needs Eigen, PCL and OpenCV headers this image lacks, so the reference
 * itself cannot be compiled here.  The float evaluation order below follows
 * Eigen's fixed-size code paths as read from its sources' documented behaviour:
 *   3x3 * 3x1 lazy product coefficient and Vector3f::squaredNorm both reduce a
 *   3-term sum as  t0 + (t1 + t2)  (redux_novec_unroller splits at Length/2);
 *   Quaterniond::toRotationMatrix does not normalise.
 * The reference loop is racy under OpenMP (unlocked update of the per-pixel
 * winner, pcd_projection.cc:433-461); this restatement is its single-thread
 * order: voxels in lexicographic key order, points in cloud order, a later
 * point replaces the winner only when strictly nearer.
 *
 * Reference lines followed:
 *   src/lidar/pcd_projection.h:71-78     GetKeyType (round(coord / submap size))
 *   src/lidar/pcd_projection.h:131-152   QuadPyramid / GetPlane
 *   src/lidar/pcd_projection.cc:223-255  BuildSubMap
 *   src/lidar/pcd_projection.cc:258-297  SearchSubMap (frustum corners, float)
 *   src/lidar/pcd_projection.cc:499-559  SearchImageMap (5 plane tests on voxel centres)
 *   src/lidar/pcd_projection.cc:305-468  ImageMapProj (splat, nearest wins)
 *   src/lidar/pcd_projection.cc:561-594  DistortOpenCV
 *   src/lidar/pcd_projection.cc:13-89    SetNewImage #1 (feature set, 6-vector out)
 *   src/lidar/pcd_projection.cc:102-220  SetNewImage #2 (plane/ray intersection out)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef struct {
  double depth_image_scale;
  int32_t max_proj_scale, min_proj_scale;
  double min_proj_dist;
  float submap_length, submap_width, submap_height, choose_meter;
  double min_lidar_proj_dist;
} proj_options;

typedef struct {
  double qvec[4], tvec[3];
  double params[8];           /* fx fy cx cy k1 k2 p1 p2 */
  uint64_t width, height;
  uint64_t feat_begin, feat_end;
} proj_image;

typedef struct { int32_t k[3]; uint32_t idx; } keyed;

static int cmp_keyed(const void* a, const void* b) {
  const keyed* p = (const keyed*)a; const keyed* q = (const keyed*)b;
  for (int i = 0; i < 3; ++i) if (p->k[i] != q->k[i]) return p->k[i] < q->k[i] ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx);
}

static float dot3_eigen(float a0, float b0, float a1, float b1, float a2, float b2) {
  const float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  const float s = p1 + p2;
  return p0 + s;
}

/* pcd_projection.h:139-146 */
static void get_plane(const float* a, const float* b, const float* c, float* pl) {
  const float ab[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
  const float ac[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]};
  const float n0 = ab[1] * ac[2] - ab[2] * ac[1];
  const float n1 = ab[2] * ac[0] - ab[0] * ac[2];
  const float n2 = ab[0] * ac[1] - ab[1] * ac[0];
  float d = n0 * a[0] + n1 * a[1];
  d = d + n2 * a[2];
  pl[0] = n0; pl[1] = n1; pl[2] = n2; pl[3] = -d;
}

static int plane_inside(const float* pl, float x, float y, float z) {
  float v = pl[0] * x + pl[1] * y;
  v = v + pl[2] * z;
  v = v + pl[3];
  return v <= 0.0f;
}

/* Eigen::Quaterniond(w,x,y,z).toRotationMatrix(), no normalisation. */
static void quat_to_rot(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* pcd_projection.cc:561-594 */
static void distort_opencv(const double* prm, double u, double v, double* ou, double* ov) {
  const double fx = prm[0], fy = prm[1], cx = prm[2], cy = prm[3];
  const double k1 = prm[4], k2 = prm[5], p1 = prm[6], p2 = prm[7];
  const double x = (u - cx) / fx;
  const double y = (v - cy) / fy;
  const double r2 = x * x + y * y;
  const double dRa = 1. + k1 * r2 + k2 * r2 * r2;
  const double dRb = 1;
  const double dTx = 2. * p1 * x * y + p2 * (r2 + 2. * x * x);
  const double dTy = p1 * (r2 + 2. * y * y) + 2. * p2 * x * y;
  double du = x * dRa * dRb + dTx;
  double dv = y * dRa * dRb + dTy;
  *ou = du * fx + cx;
  *ov = dv * fy + cy;
}

/* The four function-local statics of pcd_projection.cc:391-397, as computed on
 * the first camera the process sees (b_y really does use the unscaled
 * min_proj_scale). */
void oracle_proj_scale_coeffs(const proj_options* o, double fx, double fy, double* c4) {
  const double s = o->depth_image_scale;
  const double max_x = (double)o->max_proj_scale * (fx / 3039.0) * (s / 0.2);
  const double max_y = (double)o->max_proj_scale * (fy / 3039.0) * (s / 0.2);
  const double min_x = (double)o->min_proj_scale * (fx / 3039.0) * (s / 0.2);
  const double min_y = (double)o->min_proj_scale * (fy / 3039.0) * (s / 0.2);
  const double a_x = (max_x - min_x) / (o->min_proj_dist - (double)o->choose_meter);
  const double b_x = min_x - a_x * (double)o->choose_meter;
  const double a_y = (max_y - min_y) / (o->min_proj_dist - (double)o->choose_meter);
  const double b_y = (double)o->min_proj_scale - a_y * (double)o->choose_meter;
  c4[0] = a_x; c4[1] = b_x; c4[2] = a_y; c4[3] = b_y;
}

/*
 * xyz: n x 3 floats (visual frame, as PcdProj receives them after
 * ply.cc:38-54).  coeffs: the four latched statics.  Outputs per feature:
 * found (0/1), index (cloud row of the winner), dist (its float norm).
 * Returns the number of (image, voxel) pairs that survived culling, or -1.
 */
int64_t oracle_proj_images(const float* xyz, uint64_t n, const proj_options* o, const double* coeffs,
                           uint64_t n_images, const proj_image* imgs, const double* feat_xy,
                           uint8_t* found, uint32_t* index, float* dist_out) {
  /* BuildSubMap: std::map ordered by (kx, ky, kz); nodes keep cloud order */
  keyed* ks = (keyed*)malloc(sizeof(keyed) * (n ? n : 1));
  if (!ks) return -1;
  uint64_t m = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!isfinite(x) || !isfinite(y) || !isfinite(z)) continue;   /* round(inf)->int is UB in the reference */
    ks[m].k[0] = (int32_t)roundf(x / o->submap_length);
    ks[m].k[1] = (int32_t)roundf(y / o->submap_height);
    ks[m].k[2] = (int32_t)roundf(z / o->submap_width);
    ks[m].idx = (uint32_t)i;
    ++m;
  }
  qsort(ks, m, sizeof(keyed), cmp_keyed);

  int64_t pairs = 0;
  const double scale = o->depth_image_scale;
  for (uint64_t ii = 0; ii < n_images; ++ii) {
    const proj_image* im = &imgs[ii];
    const double* prm = im->params;
    const int img_h = (int)((double)im->height * scale);
    const int img_w = (int)((double)im->width * scale);
    const uint64_t nf = im->feat_end - im->feat_begin;
    for (uint64_t f = im->feat_begin; f < im->feat_end; ++f) { found[f] = 0; index[f] = 0xFFFFFFFFu; dist_out[f] = 0.f; }
    if (img_h <= 0 || img_w <= 0) continue;
    const size_t npx = (size_t)img_h * (size_t)img_w;
    uint8_t* is_feat = (uint8_t*)calloc(npx, 1);
    float* zdist = (float*)malloc(npx * sizeof(float));
    uint32_t* zidx = (uint32_t*)malloc(npx * sizeof(uint32_t));
    if (!is_feat || !zdist || !zidx) { free(is_feat); free(zdist); free(zidx); free(ks); return -1; }
    memset(zidx, 0xFF, npx * sizeof(uint32_t));
    for (uint64_t f = 0; f < nf; ++f) {
      const double* xy = &feat_xy[2 * (im->feat_begin + f)];
      const int u = (int)(xy[0] * scale), v = (int)(xy[1] * scale);
      if (u < 0 || u >= img_w || v < 0 || v >= img_h) continue;
      is_feat[(size_t)v * img_w + u] = 1;
    }

    /* SetNewImage: scaled intrinsics, pose */
    double Rd[9];
    quat_to_rot(im->qvec, Rd);
    float R[9], t[3];
    for (int k = 0; k < 9; ++k) R[k] = (float)Rd[k];
    for (int k = 0; k < 3; ++k) t[k] = (float)im->tvec[k];
    const double ifx = prm[0] * scale, ify = prm[1] * scale, icx = prm[2] * scale, icy = prm[3] * scale;

    /* SearchSubMap */
    float Rt[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[3 * r + c] = R[3 * c + r];
    float twc[3];
    for (int r = 0; r < 3; ++r) twc[r] = dot3_eigen(-Rt[3 * r], t[0], -Rt[3 * r + 1], t[1], -Rt[3 * r + 2], t[2]);
    const float xb_min = (float)(-icx / ifx), xb_max = (float)(((double)img_w - icx) / ifx);
    const float yb_min = (float)(-icy / ify), yb_max = (float)(((double)img_h - icy) / ify);
    const float cdir[4][2] = {{xb_max, yb_max}, {xb_max, yb_min}, {xb_min, yb_min}, {xb_min, yb_max}};
    float corner[4][3];
    for (int c = 0; c < 4; ++c) {
      const float d0 = 0.0f + cdir[c][0], d1 = 0.0f + cdir[c][1], d2 = 1.0f + 0.0f;   /* center_v + corner */
      for (int r = 0; r < 3; ++r) {
        float v = dot3_eigen(Rt[3 * r], d0, Rt[3 * r + 1], d1, Rt[3 * r + 2], d2);
        v = v * o->choose_meter;
        corner[c][r] = twc[r] + v;
      }
    }
    float pl[5][4];
    get_plane(corner[0], corner[3], corner[2], pl[0]);
    get_plane(twc, corner[0], corner[1], pl[1]);
    get_plane(twc, corner[1], corner[2], pl[2]);
    get_plane(twc, corner[2], corner[3], pl[3]);
    get_plane(twc, corner[3], corner[0], pl[4]);

    const double fx = prm[0], fy = prm[1], cx = prm[2], cy = prm[3];
    const double max_sx = (double)o->max_proj_scale * (fx / 3039.0) * (scale / 0.2);
    const double max_sy = (double)o->max_proj_scale * (fy / 3039.0) * (scale / 0.2);

    /* SearchImageMap + ImageMapProj in single-thread order */
    uint64_t s = 0;
    while (s < m) {
      uint64_t e = s + 1;
      while (e < m && ks[e].k[0] == ks[s].k[0] && ks[e].k[1] == ks[s].k[1] && ks[e].k[2] == ks[s].k[2]) ++e;
      const float vx = (float)ks[s].k[0] * o->submap_length;
      const float vy = (float)ks[s].k[1] * o->submap_height;
      const float vz = (float)ks[s].k[2] * o->submap_width;
      int in = 1;
      for (int p = 0; p < 5; ++p) in &= plane_inside(pl[p], vx, vy, vz);
      if (in) {
        ++pairs;
        for (uint64_t j = s; j < e; ++j) {
          const uint32_t pi = ks[j].idx;
          const float* pw = &xyz[3 * (size_t)pi];
          float pc[3];
          for (int r = 0; r < 3; ++r) pc[r] = dot3_eigen(R[3 * r], pw[0], R[3 * r + 1], pw[1], R[3 * r + 2], pw[2]) + t[r];
          if (pc[2] < 0) continue;
          const double u_ori = fx * (double)(pc[0] / pc[2]) + cx;
          const double v_ori = fy * (double)(pc[1] / pc[2]) + cy;
          double ud, vd;
          distort_opencv(prm, u_ori, v_ori, &ud, &vd);
          const float depth = pc[2];
          int sx, sy;
          if ((double)depth < o->min_lidar_proj_dist) continue;
          else if (o->min_lidar_proj_dist <= (double)depth && (double)depth <= o->min_proj_dist) { sx = (int)max_sx; sy = (int)max_sy; }
          else if ((double)depth > o->min_proj_dist) { sx = (int)(coeffs[0] * depth + coeffs[1]); sy = (int)(coeffs[2] * depth + coeffs[3]); }
          else continue;
          const double ur = round(ud * scale), vr = round(vd * scale);
          if (!(fabs(ur) < 1e9) || !(fabs(vr) < 1e9)) continue;   /* int(round(inf/nan)) is UB in the reference */
          const int u0 = (int)ur, v0 = (int)vr;
          const float nrm = sqrtf(pc[0] * pc[0] + (pc[1] * pc[1] + pc[2] * pc[2]));
          for (int u = u0 - sx; u <= u0 + sx; ++u)
            for (int v = v0 - sy; v <= v0 + sy; ++v) {
              if (u < 0 || u >= img_w || v < 0 || v >= img_h) continue;
              const size_t px = (size_t)v * img_w + u;
              if (!is_feat[px]) continue;
              if (zidx[px] == 0xFFFFFFFFu || zdist[px] > nrm) { zidx[px] = pi; zdist[px] = nrm; }
            }
        }
      }
      s = e;
    }

    for (uint64_t f = 0; f < nf; ++f) {
      const uint64_t g = im->feat_begin + f;
      const double* xy = &feat_xy[2 * g];
      const int u = (int)(xy[0] * scale), v = (int)(xy[1] * scale);
      if (u < 0 || u >= img_w || v < 0 || v >= img_h) continue;
      const size_t px = (size_t)v * img_w + u;
      if (zidx[px] != 0xFFFFFFFFu) { found[g] = 1; index[g] = zidx[px]; dist_out[g] = zdist[px]; }
    }
    free(is_feat); free(zdist); free(zidx);
  }
  free(ks);
  return pairs;
}

/* SetNewImage #1 (pcd_projection.cc:66-76): winner as 6 doubles-of-floats. */
void oracle_proj_lidar6(const float* xyz, const float* nrm, uint64_t nf, const uint8_t* found,
                        const uint32_t* index, double* l6) {
  for (uint64_t f = 0; f < nf; ++f) {
    double* o = &l6[6 * f];
    if (!found[f]) { for (int k = 0; k < 6; ++k) o[k] = 0.0; continue; }
    const size_t i = index[f];
    for (int k = 0; k < 3; ++k) { o[k] = (double)xyz[3 * i + k]; o[3 + k] = (double)nrm[3 * i + k]; }
  }
}

/* SetNewImage #2 (pcd_projection.cc:183-205): intersect the pixel ray with the
 * winner's plane (plane built from the winner's coordinates as stored, i.e. no
 * world->camera transform, exactly as the reference does). */
void oracle_proj_ray_plane(const double* params, uint64_t nf, const double* feat_xy, const uint8_t* found,
                           const double* l6, double* cam_xyz) {
  const double fx = params[0], fy = params[1], cx = params[2], cy = params[3];
  for (uint64_t f = 0; f < nf; ++f) {
    double* o = &cam_xyz[3 * f];
    o[0] = o[1] = o[2] = 0.0;
    if (!found[f]) continue;
    const double* p = &l6[6 * f];
    const double a = p[3], b = p[4], c = p[5];
    const double d = 0 - a * p[0] - b * p[1] - c * p[2];
    const double u = feat_xy[2 * f], v = feat_xy[2 * f + 1];
    const double z = -d / (a * (u - cx) / fx + b * (v - cy) / fy + c);
    o[0] = z * (u - cx) / fx;
    o[1] = z * (v - cy) / fy;
    o[2] = z;
  }
}
