// proj.hip -- depth-projection association (the reference's PcdProj).
//
// Reference behaviour rebuilt here (lidar/pcd_projection.{h,cc}):
//   BuildSubMap      .cc:223-255  voxel "submaps" keyed by round(coord / submap size)
//   SearchSubMap     .cc:258-297  float frustum (apex + 4 corners at choose_meter)
//   SearchImageMap   .cc:499-559  a submap survives when its key centre is inside the 5 planes
//   ImageMapProj     .cc:305-468  splat every point of the surviving submaps into the scaled image; per feature
//                                 pixel the point with the smallest camera-frame norm wins
//   DistortOpenCV    .cc:561-594
//   SetNewImage      .cc:13-89, 102-220  the two read-outs (6-vector; ray/plane intersection)
//
// MI355X design: the cloud is re-sorted once by (submap key, cloud row), which IS the order the reference walks
// (std::map of keys, nodes in push order), so "position in the sorted cloud" is the reference's single-thread
// visiting rank.  A batch of images is processed together: one thread per (image, submap) culls, one wavefront
// per surviving pair splats its points, winners are u64 atomicMin of (bits(norm) << 32 | sorted position) on a
// per-image dense buffer touched only at feature pixels (a bitmap filters the rest).  The minimum of that key
// is the reference's strict-"nearer replaces" rule in single-thread order, and it is deterministic where the
// reference's OpenMP loop is racy.
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "cloud.h"
#include "grid.h"

namespace pcd {

constexpr uint64_t kProjNone = ~0ull;

struct ProjImageDev {
  float R[9], t[3];
  float pl[5][4];
  double prm[8];
  int sx_near, sy_near;      // static_cast<int>(max_proj_scale_{x,y}) of this camera
  int w, h, row_words, pad;
  uint64_t zoff;             // u64 elements into the winner buffer
  uint64_t boff;             // u32 words into the feature bitmap
  uint64_t feat_begin, feat_end;
};

struct ProjConst {
  double scale, min_lidar_proj_dist, min_proj_dist;
  double a_x, b_x, a_y, b_y;
};

struct ProjKeyRange { int lo[3]; int hi[3]; unsigned invalid; };

// Eigen's fixed-size 3-term reductions evaluate t0 + (t1 + t2)
__device__ __host__ __forceinline__ float dot3_e(float a0, float b0, float a1, float b1, float a2, float b2) {
  const float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  const float s = p1 + p2;
  return p0 + s;
}

__device__ __forceinline__ bool finite3(float4 p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

// pcd_projection.h:71-78
__device__ __forceinline__ void submap_key(float4 p, float len, float hei, float wid, int* k) {
  k[0] = (int)roundf((p.x / len));
  k[1] = (int)roundf((p.y / hei));
  k[2] = (int)roundf((p.z / wid));
}

__global__ void k_proj_key_range(const float4* __restrict__ pts4, uint64_t n, float len, float hei, float wid,
                                 ProjKeyRange* __restrict__ r) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  bool bad = false;
  if (i < n) {
    const float4 p = pts4[i];
    if (finite3(p)) {
      int k[3];
      submap_key(p, len, hei, wid, k);
      for (int a = 0; a < 3; ++a) lo[a] = hi[a] = k[a];
    } else {
      bad = true;
    }
  }
  for (int a = 0; a < 3; ++a) {
    int l = lo[a], h = hi[a];
    for (int o = 32; o; o >>= 1) {
      l = min(l, __shfl_xor(l, o));
      h = max(h, __shfl_xor(h, o));
    }
    if ((threadIdx.x & 63) == 0) {
      if (l != INT_MAX) atomicMin(&r->lo[a], l);
      if (h != INT_MIN) atomicMax(&r->hi[a], h);
    }
  }
  const unsigned long long b = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&r->invalid, (unsigned)__popcll(b));
}

__global__ void k_proj_keys(const float4* __restrict__ pts4, uint64_t n, float len, float hei, float wid,
                            int lox, int loy, int loz, int bx, int by, int bz, uint64_t* __restrict__ keys,
                            uint32_t* __restrict__ vals) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts4[i];
  uint64_t key = 1ull << (bx + by + bz);   // non-finite rows sort to the end
  if (finite3(p)) {
    int k[3];
    submap_key(p, len, hei, wid, k);
    key = ((uint64_t)(uint32_t)(k[0] - lox) << (by + bz)) | ((uint64_t)(uint32_t)(k[1] - loy) << bz) |
          (uint64_t)(uint32_t)(k[2] - loz);
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

__global__ void k_proj_heads(const uint64_t* __restrict__ keys, uint64_t m, uint32_t* __restrict__ head) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= m) return;
  head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void k_proj_fill(const float4* __restrict__ pts4, const uint64_t* __restrict__ keys,
                            const uint32_t* __restrict__ order, const uint32_t* __restrict__ head,
                            const uint32_t* __restrict__ sub_of, uint64_t m, int lox, int loy, int loz, int by,
                            int bz, float4* __restrict__ sorted, uint32_t* __restrict__ sub_start,
                            int4* __restrict__ sub_key) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t src = order[i];
  float4 p = pts4[src];
  p.w = __uint_as_float(src);
  sorted[i] = p;
  if (head[i]) {
    const uint32_t sidx = sub_of[i];
    const uint64_t key = keys[i];
    sub_start[sidx] = (uint32_t)i;
    sub_key[sidx] = make_int4((int)(key >> (by + bz)) + lox, (int)((key >> bz) & ((1ull << by) - 1)) + loy,
                              (int)(key & ((1ull << bz) - 1)) + loz, 0);
  }
}

// pcd_projection.cc:25-31: uv = (xy * scale).cast<int>(), bounds check.
__device__ __forceinline__ bool feature_pixel(const double* __restrict__ xy, double scale, int w, int h, int* u,
                                              int* v) {
  const double fu = xy[0] * scale, fv = xy[1] * scale;
  if (!(fabs(fu) < 2e9) || !(fabs(fv) < 2e9)) return false;
  *u = (int)fu;
  *v = (int)fv;
  return *u >= 0 && *u < w && *v >= 0 && *v < h;
}

__global__ void k_proj_feat_init(const ProjImageDev* __restrict__ imgs, uint32_t n_img,
                                 const double* __restrict__ feat_xy, double scale, uint32_t* __restrict__ bitmap,
                                 uint64_t* __restrict__ zbuf) {
  const uint32_t ii = blockIdx.y;
  const ProjImageDev& im = imgs[ii];
  const uint64_t nf = im.feat_end - im.feat_begin;
  for (uint64_t f = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; f < nf; f += (uint64_t)gridDim.x * blockDim.x) {
    int u, v;
    if (!feature_pixel(&feat_xy[2 * (im.feat_begin + f)], scale, im.w, im.h, &u, &v)) continue;
    atomicOr(&bitmap[im.boff + (uint64_t)v * im.row_words + (u >> 5)], 1u << (u & 31));
    zbuf[im.zoff + (uint64_t)v * im.w + u] = kProjNone;
  }
}

// pcd_projection.cc:523-553: ((a*x + b*y) + c*z) + d <= 0 on the key centre, all five planes
__global__ void k_proj_cull(const ProjImageDev* __restrict__ imgs, const int4* __restrict__ sub_key, uint32_t n_sub,
                            float len, float hei, float wid, uint2* __restrict__ items,
                            unsigned* __restrict__ n_items) {
  const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ii = blockIdx.y;
  bool in = false;
  if (sidx < n_sub && imgs[ii].w > 0 && imgs[ii].h > 0) {
    const int4 k = sub_key[sidx];
    const float x = (float)k.x * len, y = (float)k.y * hei, z = (float)k.z * wid;
    in = true;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const float* pl = imgs[ii].pl[p];
      float v = pl[0] * x + pl[1] * y;
      v = v + pl[2] * z;
      v = v + pl[3];
      in = in && (v <= 0.0f);
    }
  }
  const unsigned long long m = __ballot(in);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(n_items, (unsigned)__popcll(m));
  base = __shfl(base, 0);
  if (in) items[base + __popcll(m & ((1ull << lane) - 1))] = make_uint2(ii, sidx);
}

// One wavefront per surviving (image, submap); lanes stride over the submap's points.
#ifndef PCD_SPLAT_ROWS
#define PCD_SPLAT_ROWS 4
#endif
constexpr int kSplatRows = PCD_SPLAT_ROWS;   // bitmap rows fetched together by a lane of k_proj_splat
                                             // (1 / 4 / 8 / 16 rows: 1.46 / 1.06 / 1.12 / 1.20 ms on tools/proj_probe.py)
__global__ __launch_bounds__(256) void k_proj_splat(const ProjImageDev* __restrict__ imgs,
                                                    const uint2* __restrict__ items,
                                                    const unsigned* __restrict__ n_items,
                                                    const uint32_t* __restrict__ sub_start,
                                                    const float4* __restrict__ sorted, ProjConst c,
                                                    const uint32_t* __restrict__ bitmap,
                                                    uint64_t* __restrict__ zbuf) {
  const unsigned total = *n_items;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (unsigned it = wave; it < total; it += nwaves) {
    const uint2 item = items[__builtin_amdgcn_readfirstlane(it)];
    const unsigned ii = __builtin_amdgcn_readfirstlane(item.x);
    const unsigned sidx = __builtin_amdgcn_readfirstlane(item.y);
    const ProjImageDev& im = imgs[ii];
    const uint32_t beg = sub_start[sidx], end = sub_start[sidx + 1];
    const double fx = im.prm[0], fy = im.prm[1], cx = im.prm[2], cy = im.prm[3];
    const double k1 = im.prm[4], k2 = im.prm[5], p1 = im.prm[6], p2 = im.prm[7];
    const int W = im.w, H = im.h, RW = im.row_words;
    const uint32_t* __restrict__ bm = bitmap + im.boff;
    uint64_t* __restrict__ zb = zbuf + im.zoff;
    for (uint32_t pos = beg + lane; pos < end; pos += 64) {
      const float4 pw = sorted[pos];
      // pt_c = rot_cw * pt_w + t_cw  (float)
      const float xc = dot3_e(im.R[0], pw.x, im.R[1], pw.y, im.R[2], pw.z) + im.t[0];
      const float yc = dot3_e(im.R[3], pw.x, im.R[4], pw.y, im.R[5], pw.z) + im.t[1];
      const float zc = dot3_e(im.R[6], pw.x, im.R[7], pw.y, im.R[8], pw.z) + im.t[2];
      if (zc < 0) continue;
      const double depth = (double)zc;
      int sx, sy;
      if (depth < c.min_lidar_proj_dist) continue;
      else if (c.min_lidar_proj_dist <= depth && depth <= c.min_proj_dist) { sx = im.sx_near; sy = im.sy_near; }
      else if (depth > c.min_proj_dist) { sx = (int)(c.a_x * depth + c.b_x); sy = (int)(c.a_y * depth + c.b_y); }
      else continue;
      if (sx < 0 || sy < 0) continue;
      const double u_ori = fx * (double)(xc / zc) + cx;
      const double v_ori = fy * (double)(yc / zc) + cy;
      // DistortOpenCV
      const double x = (u_ori - cx) / fx;
      const double y = (v_ori - cy) / fy;
      const double r2 = x * x + y * y;
      const double dRa = 1. + k1 * r2 + k2 * r2 * r2;
      const double dTx = 2. * p1 * x * y + p2 * (r2 + 2. * x * x);
      const double dTy = p1 * (r2 + 2. * y * y) + 2. * p2 * x * y;
      const double ud = (x * dRa * 1.0 + dTx) * fx + cx;
      const double vd = (y * dRa * 1.0 + dTy) * fy + cy;
      const double ur = round(ud * c.scale), vr = round(vd * c.scale);
      if (!(fabs(ur) < 1e9) || !(fabs(vr) < 1e9)) continue;
      const int u0 = (int)ur, v0 = (int)vr;
      const int ulo = max(u0 - sx, 0), uhi = min(u0 + sx, W - 1);
      const int vlo = max(v0 - sy, 0), vhi = min(v0 + sy, H - 1);
      if (ulo > uhi || vlo > vhi) continue;
      const float nrm = sqrtf(xc * xc + (yc * yc + zc * zc));
      const uint64_t key = ((uint64_t)__float_as_uint(nrm) << 32) | pos;
      const int wlo = ulo >> 5, whi = uhi >> 5;
      auto hit_word = [&](uint32_t bits, int w, int v) {
        if (!bits) return;
        if (w == wlo) bits &= 0xFFFFFFFFu << (ulo & 31);
        if (w == whi) bits &= 0xFFFFFFFFu >> (31 - (uhi & 31));
        while (bits) {
          const int b = __builtin_ctz(bits);
          bits &= bits - 1;
          uint64_t* slot = zb + (uint64_t)v * W + (w * 32 + b);
          if (*(volatile uint64_t*)slot > key) atomicMin((unsigned long long*)slot, (unsigned long long)key);
        }
      };
      if (whi - wlo <= 1) {
        // the usual case, a splat one or two bitmap words wide: the words of four rows are fetched together (clamped
        // row index, no branch around the loads) before any of them is looked at -- one round trip per four rows
        // instead of one per row
        const bool two = whi != wlo;
        for (int v = vlo; v <= vhi; v += kSplatRows) {
          uint32_t b0[kSplatRows], b1[kSplatRows];
#pragma unroll
          for (int k = 0; k < kSplatRows; ++k) {
            const uint32_t* row = bm + (uint64_t)min(v + k, vhi) * RW;
            b0[k] = row[wlo];
            b1[k] = row[whi];
          }
#pragma unroll
          for (int k = 0; k < kSplatRows; ++k) {
            if (v + k > vhi) break;
            hit_word(b0[k], wlo, v + k);
            if (two) hit_word(b1[k], whi, v + k);
          }
        }
      } else {
        for (int v = vlo; v <= vhi; ++v) {
          const uint32_t* row = bm + (uint64_t)v * RW;
          for (int w = wlo; w <= whi; ++w) hit_word(row[w], w, v);
        }
      }
    }
  }
}

__global__ void k_proj_readout(const ProjImageDev* __restrict__ imgs, const double* __restrict__ feat_xy,
                               double scale, const uint64_t* __restrict__ zbuf, const float4* __restrict__ sorted,
                               const float4* __restrict__ pn8,
                               uint8_t* __restrict__ found, uint32_t* __restrict__ index, float* __restrict__ dist,
                               double* __restrict__ l6, double* __restrict__ cam_xyz) {
  const uint32_t ii = blockIdx.y;
  const ProjImageDev& im = imgs[ii];
  const uint64_t nf = im.feat_end - im.feat_begin;
  for (uint64_t f = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; f < nf; f += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t g = im.feat_begin + f;
    int u, v;
    uint64_t key = kProjNone;
    if (im.w > 0 && im.h > 0 && feature_pixel(&feat_xy[2 * g], scale, im.w, im.h, &u, &v))
      key = zbuf[im.zoff + (uint64_t)v * im.w + u];
    const bool ok = key != kProjNone;
    uint32_t idx = 0xFFFFFFFFu;
    double o[6] = {0, 0, 0, 0, 0, 0}, cxyz[3] = {0, 0, 0};
    float d = 0.f;
    if (ok) {
      idx = __float_as_uint(sorted[(uint32_t)key].w);
      d = __uint_as_float((uint32_t)(key >> 32));
      const float4 p = pn8[2 * (size_t)idx], nn = pn8[2 * (size_t)idx + 1];
      o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = nn.x; o[4] = nn.y; o[5] = nn.z;
      // pcd_projection.cc:183-205
      const double fx = im.prm[0], fy = im.prm[1], cx = im.prm[2], cy = im.prm[3];
      const double a = o[3], b = o[4], cc = o[5];
      const double dd = 0 - a * o[0] - b * o[1] - cc * o[2];
      const double uu = feat_xy[2 * g], vv = feat_xy[2 * g + 1];
      const double z = -dd / (a * (uu - cx) / fx + b * (vv - cy) / fy + cc);
      cxyz[0] = z * (uu - cx) / fx;
      cxyz[1] = z * (vv - cy) / fy;
      cxyz[2] = z;
    }
    if (found) found[g] = ok ? 1 : 0;
    if (index) index[g] = idx;
    if (dist) dist[g] = d;
    if (l6) for (int k = 0; k < 6; ++k) l6[6 * g + k] = o[k];
    if (cam_xyz) for (int k = 0; k < 3; ++k) cam_xyz[3 * g + k] = cxyz[k];
  }
}

}  // namespace pcd

using namespace pcd;

struct pcd_proj {
  pcd_cloud* cloud = nullptr;
  pcd_proj_options opt{};
  uint64_t m = 0;         // finite rows
  uint32_t n_sub = 0;
  DevBuf<float4> sorted;
  DevBuf<uint32_t> sub_start;
  DevBuf<int4> sub_key;
  bool latched = false;
  double coeffs[4] = {0, 0, 0, 0};
  // per-call scratch (grow-only)
  DevBuf<ProjImageDev> d_imgs;
  DevBuf<double> d_feat;
  DevBuf<uint32_t> bitmap;
  DevBuf<uint64_t> zbuf;
  DevBuf<uint2> items;
  DevBuf<unsigned> n_items;
  DevBuf<uint8_t> o_found;
  DevBuf<uint32_t> o_index;
  DevBuf<float> o_dist;
  DevBuf<double> o_l6, o_cam;
  uint64_t last_items = 0;
};

namespace {

int bits_for(int64_t range) {
  int b = 1;
  while ((1ll << b) <= range) ++b;
  return b;
}

// Eigen::Quaterniond(w,x,y,z).toRotationMatrix(): no normalisation
void quat_to_rot(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// pcd_projection.h:139-146
void get_plane(const float* a, const float* b, const float* c, float* pl) {
  const float ab[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
  const float ac[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]};
  const float n0 = ab[1] * ac[2] - ab[2] * ac[1];
  const float n1 = ab[2] * ac[0] - ab[0] * ac[2];
  const float n2 = ab[0] * ac[1] - ab[1] * ac[0];
  float d = n0 * a[0] + n1 * a[1];
  d = d + n2 * a[2];
  pl[0] = n0; pl[1] = n1; pl[2] = n2; pl[3] = -d;
}

void scale_coeffs(const pcd_proj_options& o, double fx, double fy, double* c4) {
  const double s = o.depth_image_scale;
  const double max_x = (double)o.max_proj_scale * (fx / 3039.0) * (s / 0.2);
  const double max_y = (double)o.max_proj_scale * (fy / 3039.0) * (s / 0.2);
  const double min_x = (double)o.min_proj_scale * (fx / 3039.0) * (s / 0.2);
  const double min_y = (double)o.min_proj_scale * (fy / 3039.0) * (s / 0.2);
  c4[0] = (max_x - min_x) / (o.min_proj_dist - (double)o.choose_meter);
  c4[1] = min_x - c4[0] * (double)o.choose_meter;
  c4[2] = (max_y - min_y) / (o.min_proj_dist - (double)o.choose_meter);
  c4[3] = (double)o.min_proj_scale - c4[2] * (double)o.choose_meter;   // sic: pcd_projection.cc:397
}

// SetNewImage head + SearchSubMap, all in the reference's float/double mix
void prepare_image(const pcd_proj_options& o, const pcd_proj_image& in, ProjImageDev* d) {
  const double scale = o.depth_image_scale;
  d->h = (int)((double)in.height * scale);
  d->w = (int)((double)in.width * scale);
  d->row_words = d->w > 0 ? (d->w + 31) / 32 : 0;
  d->pad = 0;
  double Rd[9];
  quat_to_rot(in.qvec, Rd);
  for (int k = 0; k < 9; ++k) d->R[k] = (float)Rd[k];
  for (int k = 0; k < 3; ++k) d->t[k] = (float)in.tvec[k];
  for (int k = 0; k < 8; ++k) d->prm[k] = in.params[k];
  const double ifx = in.params[0] * scale, ify = in.params[1] * scale;
  const double icx = in.params[2] * scale, icy = in.params[3] * scale;
  float Rt[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[3 * r + c] = d->R[3 * c + r];
  float twc[3];
  for (int r = 0; r < 3; ++r) twc[r] = dot3_e(-Rt[3 * r], d->t[0], -Rt[3 * r + 1], d->t[1], -Rt[3 * r + 2], d->t[2]);
  const float xb_min = (float)(-icx / ifx), xb_max = (float)(((double)d->w - icx) / ifx);
  const float yb_min = (float)(-icy / ify), yb_max = (float)(((double)d->h - icy) / ify);
  const float dir[4][2] = {{xb_max, yb_max}, {xb_max, yb_min}, {xb_min, yb_min}, {xb_min, yb_max}};
  float corner[4][3];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 3; ++r) {
      float v = dot3_e(Rt[3 * r], dir[c][0], Rt[3 * r + 1], dir[c][1], Rt[3 * r + 2], 1.0f);
      v = v * o.choose_meter;
      corner[c][r] = twc[r] + v;
    }
  get_plane(corner[0], corner[3], corner[2], d->pl[0]);
  get_plane(twc, corner[0], corner[1], d->pl[1]);
  get_plane(twc, corner[1], corner[2], d->pl[2]);
  get_plane(twc, corner[2], corner[3], d->pl[3]);
  get_plane(twc, corner[3], corner[0], d->pl[4]);
  d->sx_near = (int)((double)o.max_proj_scale * (in.params[0] / 3039.0) * (scale / 0.2));
  d->sy_near = (int)((double)o.max_proj_scale * (in.params[1] / 3039.0) * (scale / 0.2));
  d->feat_begin = in.feat_begin;
  d->feat_end = in.feat_end;
}

pcd_status build_submaps(pcd_proj* p) {
  pcd_cloud* c = p->cloud;
  const uint64_t n = c->n;
  hipStream_t s = nullptr;
  const pcd_proj_options& o = p->opt;
  p->m = 0;
  p->n_sub = 0;
  PCD_TRY(p->sub_start.reserve(1));
  if (n == 0) return PCD_OK;
  DevBuf<ProjKeyRange> d_range;
  PCD_TRY(d_range.reserve(1));
  ProjKeyRange init{{INT_MAX, INT_MAX, INT_MAX}, {INT_MIN, INT_MIN, INT_MIN}, 0};
  PCD_HIP_TRY(hipMemcpyAsync(d_range.p, &init, sizeof(init), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_proj_key_range, dim3(div_up(n, 256)), dim3(256), 0, s, c->pts4.p, n, o.submap_length,
                     o.submap_height, o.submap_width, d_range.p);
  ProjKeyRange r;
  PCD_HIP_TRY(hipMemcpy(&r, d_range.p, sizeof(r), hipMemcpyDeviceToHost));
  p->m = n - r.invalid;
  if (p->m == 0) return PCD_OK;
  const int bx = bits_for((int64_t)r.hi[0] - r.lo[0]), by = bits_for((int64_t)r.hi[1] - r.lo[1]),
            bz = bits_for((int64_t)r.hi[2] - r.lo[2]);
  if (bx + by + bz > 62) {
    set_error("pcd_proj_create: submap key range needs %d bits (cloud extent / submap size too large)", bx + by + bz);
    return PCD_ERR_UNSUPPORTED;
  }
  DevBuf<uint64_t> k0, k1;
  DevBuf<uint32_t> v0, v1, head, sub_of;
  PCD_TRY(k0.reserve(n)); PCD_TRY(k1.reserve(n)); PCD_TRY(v0.reserve(n)); PCD_TRY(v1.reserve(n));
  hipLaunchKernelGGL(k_proj_keys, dim3(div_up(n, 256)), dim3(256), 0, s, c->pts4.p, n, o.submap_length,
                     o.submap_height, o.submap_width, r.lo[0], r.lo[1], r.lo[2], bx, by, bz, k0.p, v0.p);
  {
    size_t tb = 0;
    PCD_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, k0.p, k1.p, v0.p, v1.p, n, 0, bx + by + bz + 1, s));
    DevBuf<char> tmp;
    PCD_TRY(tmp.reserve(tb));
    PCD_HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, k0.p, k1.p, v0.p, v1.p, n, 0, bx + by + bz + 1, s));
  }
  const uint64_t m = p->m;
  PCD_TRY(head.reserve(m)); PCD_TRY(sub_of.reserve(m));
  hipLaunchKernelGGL(k_proj_heads, dim3(div_up(m, 256)), dim3(256), 0, s, k1.p, m, head.p);
  {
    size_t tb = 0;
    PCD_HIP_TRY(rocprim::exclusive_scan(nullptr, tb, head.p, sub_of.p, 0u, m, rocprim::plus<uint32_t>(), s));
    DevBuf<char> tmp;
    PCD_TRY(tmp.reserve(tb));
    PCD_HIP_TRY(rocprim::exclusive_scan(tmp.p, tb, head.p, sub_of.p, 0u, m, rocprim::plus<uint32_t>(), s));
  }
  uint32_t last_sub = 0, last_head = 0;
  PCD_HIP_TRY(hipMemcpy(&last_sub, sub_of.p + (m - 1), 4, hipMemcpyDeviceToHost));
  PCD_HIP_TRY(hipMemcpy(&last_head, head.p + (m - 1), 4, hipMemcpyDeviceToHost));
  p->n_sub = last_sub + last_head;
  PCD_TRY(p->sorted.reserve(m));
  PCD_TRY(p->sub_start.reserve((size_t)p->n_sub + 1));
  PCD_TRY(p->sub_key.reserve(p->n_sub));
  hipLaunchKernelGGL(k_proj_fill, dim3(div_up(m, 256)), dim3(256), 0, s, c->pts4.p, k1.p, v1.p, head.p, sub_of.p, m,
                     r.lo[0], r.lo[1], r.lo[2], by, bz, p->sorted.p, p->sub_start.p, p->sub_key.p);
  const uint32_t m32 = (uint32_t)m;
  PCD_HIP_TRY(hipMemcpyAsync(p->sub_start.p + p->n_sub, &m32, 4, hipMemcpyHostToDevice, s));
  PCD_HIP_TRY(hipStreamSynchronize(s));
  PCD_HIP_TRY(hipGetLastError());
  return PCD_OK;
}

}  // namespace

extern "C" {

void pcd_proj_default_options(pcd_proj_options* o) {
  if (!o) return;
  o->depth_image_scale = 0.2;
  o->max_proj_scale = 10;
  o->min_proj_scale = 2;
  o->min_proj_dist = 2;
  o->submap_length = o->submap_width = o->submap_height = 1.0f;
  o->choose_meter = 40.0f;
  o->min_lidar_proj_dist = 0.0;
}

pcd_status pcd_proj_create(pcd_cloud* cloud, const pcd_proj_options* options, pcd_proj** out) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(cloud && options && out, "null pointer");
  PCD_REQUIRE(options->submap_length > 0 && options->submap_width > 0 && options->submap_height > 0, "submap size");
  PCD_REQUIRE(options->depth_image_scale > 0, "depth_image_scale");
  PCD_REQUIRE(cloud->index_stride == 1 && cloud->index_base == 0, "projection needs the whole cloud (not a shard)");
  PCD_REQUIRE(cloud->n < 0xFFFFFFFFull, "cloud too large");
  PCD_HIP_TRY(hipSetDevice(cloud->device));
  pcd_proj* p = new pcd_proj();
  p->cloud = cloud;
  p->opt = *options;
  pcd_status st = build_submaps(p);
  if (st != PCD_OK) {
    delete p;
    return st;
  }
  *out = p;
  return PCD_OK;
  });
}

void pcd_proj_destroy(pcd_proj* p) {
  if (!p) return;
  (void)hipSetDevice(p->cloud->device);
  delete p;
}

uint64_t pcd_proj_num_submaps(const pcd_proj* p) { return p ? p->n_sub : 0; }
uint64_t pcd_proj_last_pairs(const pcd_proj* p) { return p ? p->last_items : 0; }

pcd_status pcd_proj_scale_coeffs(pcd_proj* p, int set, double* coeffs4, int* latched) {
  PCD_REQUIRE(p, "null pointer");
  if (set) {
    PCD_REQUIRE(coeffs4, "null coeffs");
    std::memcpy(p->coeffs, coeffs4, sizeof(p->coeffs));
    p->latched = true;
  } else if (coeffs4) {
    std::memcpy(coeffs4, p->coeffs, sizeof(p->coeffs));
  }
  if (latched) *latched = p->latched ? 1 : 0;
  return PCD_OK;
}

pcd_status pcd_proj_set_new_images(pcd_proj* p, uint64_t n_images, const pcd_proj_image* images, uint64_t n_feat,
                                   const double* feat_xy, uint8_t* found, uint32_t* lidar_index, float* dist,
                                   double* lidar6, double* cam_xyz) {
  return pcd::guard([&]() -> pcd_status {
  PCD_REQUIRE(p, "null pointer");
  if (n_images == 0) return PCD_OK;
  PCD_REQUIRE(images, "null images");
  PCD_REQUIRE(n_feat == 0 || feat_xy, "null features");
  for (uint64_t i = 0; i < n_images; ++i) {
    PCD_REQUIRE(images[i].feat_begin <= images[i].feat_end && images[i].feat_end <= n_feat, "feature range");
    PCD_REQUIRE(images[i].params[0] != 0 && images[i].params[1] != 0, "focal length");
  }
  PCD_HIP_TRY(hipSetDevice(p->cloud->device));
  hipStream_t s = nullptr;
  if (!p->latched) {   // the reference's function-local statics latch on the first camera it projects with
    scale_coeffs(p->opt, images[0].params[0], images[0].params[1], p->coeffs);
    p->latched = true;
  }
  const ProjConst pc{p->opt.depth_image_scale, p->opt.min_lidar_proj_dist, p->opt.min_proj_dist,
                     p->coeffs[0], p->coeffs[1], p->coeffs[2], p->coeffs[3]};
  const uint64_t nfa = std::max<uint64_t>(n_feat, 1);
  PCD_TRY(p->d_feat.reserve(2 * nfa));
  PCD_TRY(p->o_found.reserve(nfa)); PCD_TRY(p->o_index.reserve(nfa)); PCD_TRY(p->o_dist.reserve(nfa));
  PCD_TRY(p->o_l6.reserve(6 * nfa)); PCD_TRY(p->o_cam.reserve(3 * nfa));
  PCD_TRY(p->n_items.reserve(1));
  if (n_feat) PCD_HIP_TRY(hipMemcpyAsync(p->d_feat.p, feat_xy, 2 * n_feat * sizeof(double), hipMemcpyHostToDevice, s));
  // features outside every image range keep found = 0
  PCD_HIP_TRY(hipMemsetAsync(p->o_found.p, 0, nfa, s));
  PCD_HIP_TRY(hipMemsetAsync(p->o_index.p, 0xFF, nfa * 4, s));
  PCD_HIP_TRY(hipMemsetAsync(p->o_dist.p, 0, nfa * 4, s));
  PCD_HIP_TRY(hipMemsetAsync(p->o_l6.p, 0, 6 * nfa * 8, s));
  PCD_HIP_TRY(hipMemsetAsync(p->o_cam.p, 0, 3 * nfa * 8, s));

  // chunk the batch so that the (image, submap) pair list and the winner buffers stay bounded
  const uint64_t kMaxItems = 32ull << 20, kMaxZ = 1ull << 30;   // 256 MB of pairs, 8 GB of winner slots
  const uint64_t per_chunk = std::max<uint64_t>(1, kMaxItems / std::max<uint32_t>(p->n_sub, 1));
  std::vector<ProjImageDev> h;
  p->last_items = 0;
  uint64_t i0 = 0;
  while (i0 < n_images) {
    h.clear();
    uint64_t zoff = 0, boff = 0, maxf = 0;
    uint64_t i1 = i0;
    while (i1 < n_images && (i1 - i0) < per_chunk) {
      ProjImageDev d;
      prepare_image(p->opt, images[i1], &d);
      const uint64_t px = d.w > 0 && d.h > 0 ? (uint64_t)d.w * d.h : 0;
      if (i1 > i0 && zoff + px > kMaxZ) break;
      d.zoff = zoff;
      d.boff = boff;
      zoff += px;
      boff += d.h > 0 ? (uint64_t)d.row_words * d.h : 0;
      maxf = std::max(maxf, d.feat_end - d.feat_begin);
      h.push_back(d);
      ++i1;
    }
    const uint32_t ni = (uint32_t)h.size();
    PCD_TRY(p->d_imgs.reserve(ni));
    PCD_TRY(p->zbuf.reserve(std::max<uint64_t>(zoff, 1)));
    PCD_TRY(p->bitmap.reserve(std::max<uint64_t>(boff, 1)));
    PCD_TRY(p->items.reserve(std::max<uint64_t>((uint64_t)ni * p->n_sub, 1)));
    PCD_HIP_TRY(hipMemcpyAsync(p->d_imgs.p, h.data(), ni * sizeof(ProjImageDev), hipMemcpyHostToDevice, s));
    PCD_HIP_TRY(hipMemsetAsync(p->bitmap.p, 0, std::max<uint64_t>(boff, 1) * 4, s));
    PCD_HIP_TRY(hipMemsetAsync(p->n_items.p, 0, sizeof(unsigned), s));
    const unsigned fblocks = std::max(1u, std::min(div_up(maxf, 256), 1024u));
    if (maxf) {
      ScopedKernelTimer t("proj_feat_init", s);
      hipLaunchKernelGGL(k_proj_feat_init, dim3(fblocks, ni), dim3(256), 0, s, p->d_imgs.p, ni, p->d_feat.p,
                         pc.scale, p->bitmap.p, p->zbuf.p);
    }
    if (maxf && p->n_sub) {
      {
        ScopedKernelTimer t("proj_cull", s);
        hipLaunchKernelGGL(k_proj_cull, dim3(div_up(p->n_sub, 256), ni), dim3(256), 0, s, p->d_imgs.p, p->sub_key.p,
                           p->n_sub, p->opt.submap_length, p->opt.submap_height, p->opt.submap_width, p->items.p,
                           p->n_items.p);
      }
      {
        ScopedKernelTimer t("proj_splat", s);
        hipLaunchKernelGGL(k_proj_splat, dim3(256 * 8), dim3(256), 0, s, p->d_imgs.p, p->items.p, p->n_items.p,
                           p->sub_start.p, p->sorted.p, pc, p->bitmap.p, p->zbuf.p);
      }
    }
    if (maxf) {
      ScopedKernelTimer t("proj_readout", s);
      hipLaunchKernelGGL(k_proj_readout, dim3(fblocks, ni), dim3(256), 0, s, p->d_imgs.p, p->d_feat.p, pc.scale,
                         p->zbuf.p, p->sorted.p, p->cloud->pn8.p, p->o_found.p, p->o_index.p,
                         p->o_dist.p, p->o_l6.p, p->o_cam.p);
    }
    PCD_HIP_TRY(hipGetLastError());
    unsigned cnt = 0;
    PCD_HIP_TRY(hipMemcpyAsync(&cnt, p->n_items.p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    PCD_HIP_TRY(hipStreamSynchronize(s));   // h (pinned by the async upload) is reused by the next chunk
    p->last_items += cnt;
    i0 = i1;
  }
  auto back = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    return (dst && bytes) ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s) : hipSuccess;
  };
  PCD_HIP_TRY(back(found, p->o_found.p, n_feat));
  PCD_HIP_TRY(back(lidar_index, p->o_index.p, n_feat * 4));
  PCD_HIP_TRY(back(dist, p->o_dist.p, n_feat * 4));
  PCD_HIP_TRY(back(lidar6, p->o_l6.p, 6 * n_feat * 8));
  PCD_HIP_TRY(back(cam_xyz, p->o_cam.p, 3 * n_feat * 8));
  PCD_HIP_TRY(hipStreamSynchronize(s));
  return PCD_OK;
  });
}

}  // extern "C"
