// ba_math.h -- device math of the bundle-adjustment residual blocks (fp64).
//
// Restates (independently of oracle/ba_oracle.cc, which differentiates with Jets):
//   base/cost_functions.h:100-135, :319-355  reprojection functor:  P = R(q) X + t,
//        (u,v) = P.xy / P.z, (x,y) = Model::WorldToImage(params,u,v), r = (x,y) - obs
//   base/cost_functions.h:204-232            LiDAR plane functor:   r = w * sqrt((0-s)(0-s)), s = X.n + d
//   base/camera_models.h:614-1482            WorldToImage of the 11 camera models
//   [3P Ceres] UnitQuaternionRotatePoint is the polynomial X + 2w(v x X) + 2 v x (v x X); autodiff
//        differentiates that polynomial w.r.t. all four quaternion components, so the analytic
//        Jacobian below is the derivative of the same polynomial (not of a normalised rotation).
//        QuaternionManifold::PlusJacobian, SoftLOne / Cauchy losses, Corrector (rho'' <= 0 branch).
// Jacobian of the rigid part is hand-derived; the camera model's 2x2 Jacobian d(x,y)/d(u,v) is
// obtained with 2-wide forward-mode duals so that all 11 models share one code path.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pcdhip.h"

namespace pcd {

// value + d/du + d/dv
struct D2 {
  double a, p, q;
};
__device__ __forceinline__ D2 mk(double a, double p = 0.0, double q = 0.0) { return D2{a, p, q}; }
__device__ __forceinline__ D2 operator+(D2 f, D2 g) { return D2{f.a + g.a, f.p + g.p, f.q + g.q}; }
__device__ __forceinline__ D2 operator-(D2 f, D2 g) { return D2{f.a - g.a, f.p - g.p, f.q - g.q}; }
__device__ __forceinline__ D2 operator*(D2 f, D2 g) { return D2{f.a * g.a, f.a * g.p + f.p * g.a, f.a * g.q + f.q * g.a}; }
__device__ __forceinline__ D2 operator/(D2 f, D2 g) {
  const double gi = 1.0 / g.a, fg = f.a * gi;
  return D2{fg, (f.p - fg * g.p) * gi, (f.q - fg * g.q) * gi};
}
__device__ __forceinline__ D2 operator+(D2 f, double s) { return D2{f.a + s, f.p, f.q}; }
__device__ __forceinline__ D2 operator+(double s, D2 f) { return D2{f.a + s, f.p, f.q}; }
__device__ __forceinline__ D2 operator-(D2 f, double s) { return D2{f.a - s, f.p, f.q}; }
__device__ __forceinline__ D2 operator*(D2 f, double s) { return D2{f.a * s, f.p * s, f.q * s}; }
__device__ __forceinline__ D2 operator*(double s, D2 f) { return D2{f.a * s, f.p * s, f.q * s}; }
__device__ __forceinline__ D2 operator/(D2 f, double s) { const double i = 1.0 / s; return D2{f.a * i, f.p * i, f.q * i}; }
__device__ __forceinline__ D2 dsqrt(D2 f) { const double r = sqrt(f.a), t = 1.0 / (2.0 * r); return D2{r, f.p * t, f.q * t}; }
__device__ __forceinline__ D2 datan(D2 f) { const double t = 1.0 / (1.0 + f.a * f.a); return D2{atan(f.a), f.p * t, f.q * t}; }
__device__ __forceinline__ double dtan_c(double x) { return tan(x); }

// theta_d / r style fisheye distortion: camera_models.h:963-990, :1272-1290, :1348-1370
__device__ __forceinline__ void fisheye(D2 u, D2 v, double k1, double k2, double k3, double k4, D2& xu, D2& xv) {
  const D2 r = dsqrt(u * u + v * v);
  if (r.a > 2.220446049250313e-16) {
    const D2 th = datan(r), th2 = th * th, th4 = th2 * th2;
    const D2 ser = (1.0 + k1 * th2) + k2 * th4 + k3 * (th4 * th2) + k4 * (th4 * th4);
    const D2 thd = th * ser;
    xu = u + (u * thd / r - u);
    xv = v + (v * thd / r - v);
  } else {
    xu = u;
    xv = v;
  }
}

// (x,y) and their derivatives w.r.t. (u,v); params are constants here (intrinsics held fixed)
__device__ __forceinline__ void world_to_image_d2(int model, const double* __restrict__ p, double u0, double v0, D2& x,
                                                  D2& y) {
  const D2 u = mk(u0, 1.0, 0.0), v = mk(v0, 0.0, 1.0);
  switch (model) {
    case 0:  // SIMPLE_PINHOLE f cx cy
      x = p[0] * u + p[1]; y = p[0] * v + p[2]; break;
    case 1:  // PINHOLE fx fy cx cy
      x = p[0] * u + p[2]; y = p[1] * v + p[3]; break;
    case 2: {  // SIMPLE_RADIAL f cx cy k
      const D2 r2 = u * u + v * v, rad = p[3] * r2;
      x = p[0] * (u + u * rad) + p[1]; y = p[0] * (v + v * rad) + p[2]; break; }
    case 3: {  // RADIAL f cx cy k1 k2
      const D2 r2 = u * u + v * v, rad = p[3] * r2 + p[4] * r2 * r2;
      x = p[0] * (u + u * rad) + p[1]; y = p[0] * (v + v * rad) + p[2]; break; }
    case 4: {  // OPENCV fx fy cx cy k1 k2 p1 p2
      const D2 u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
      const D2 rad = p[4] * r2 + p[5] * r2 * r2;
      const D2 du = u * rad + (2.0 * p[6]) * uv + p[7] * (r2 + 2.0 * u2);
      const D2 dv = v * rad + (2.0 * p[7]) * uv + p[6] * (r2 + 2.0 * v2);
      x = p[0] * (u + du) + p[2]; y = p[1] * (v + dv) + p[3]; break; }
    case 5: {  // OPENCV_FISHEYE fx fy cx cy k1 k2 k3 k4
      D2 xu, xv; fisheye(u, v, p[4], p[5], p[6], p[7], xu, xv);
      x = p[0] * xu + p[2]; y = p[1] * xv + p[3]; break; }
    case 6: {  // FULL_OPENCV fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6
      const D2 u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2;
      const D2 rad = (1.0 + p[4] * r2 + p[5] * r4 + p[8] * r6) / (1.0 + p[9] * r2 + p[10] * r4 + p[11] * r6);
      const D2 du = u * rad + (2.0 * p[6]) * uv + p[7] * (r2 + 2.0 * u2) - u;
      const D2 dv = v * rad + (2.0 * p[7]) * uv + p[6] * (r2 + 2.0 * v2) - v;
      x = p[0] * (u + du) + p[2]; y = p[1] * (v + dv) + p[3]; break; }
    case 7: {  // FOV fx fy cx cy omega
      const double om = p[4], om2 = om * om;
      const D2 rad2 = u * u + v * v;
      D2 fac;
      if (om2 < 1e-4) {
        fac = (om2 * rad2) / 3.0 - om2 / 12.0 + 1.0;
      } else if (rad2.a < 1e-4) {
        const double t = dtan_c(om / 2.0);
        fac = ((-2.0 * t) * ((4.0 * t * t) * rad2 - 3.0)) / (3.0 * om);
      } else {
        const D2 rad = dsqrt(rad2);
        const D2 num = datan(rad * (2.0 * dtan_c(om / 2.0)));
        fac = num / (rad * om);
      }
      x = p[0] * (u * fac) + p[2]; y = p[1] * (v * fac) + p[3]; break; }
    case 8: {  // SIMPLE_RADIAL_FISHEYE f cx cy k
      D2 xu, xv; fisheye(u, v, p[3], 0.0, 0.0, 0.0, xu, xv);
      x = p[0] * xu + p[1]; y = p[0] * xv + p[2]; break; }
    case 9: {  // RADIAL_FISHEYE f cx cy k1 k2
      D2 xu, xv; fisheye(u, v, p[3], p[4], 0.0, 0.0, xu, xv);
      x = p[0] * xu + p[1]; y = p[0] * xv + p[2]; break; }
    default: {  // 10 THIN_PRISM_FISHEYE fx fy cx cy k1 k2 p1 p2 k3 k4 sx1 sy1
      const D2 r = dsqrt(u * u + v * v);
      D2 uu = u, vv = v;
      if (r.a > 2.220446049250313e-16) { const D2 th = datan(r); uu = th * u / r; vv = th * v / r; }
      const D2 u2 = uu * uu, uv = uu * vv, v2 = vv * vv, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2;
      const D2 rad = p[4] * r2 + p[5] * r4 + p[8] * r6 + p[9] * r8;
      const D2 du = uu * rad + (2.0 * p[6]) * uv + p[7] * (r2 + 2.0 * u2) + p[10] * r2;
      const D2 dv = vv * rad + (2.0 * p[7]) * uv + p[6] * (r2 + 2.0 * v2) + p[11] * r2;
      x = p[0] * (uu + du) + p[2]; y = p[1] * (vv + dv) + p[3]; break; }
  }
}

struct ReprojBlock {
  double r[2];
  double M[6];   // d r / d P  (2x3) = A * d(u,v)/dP ;  J_t = M
  double D[9];   // dP/dX (3x3, the rotation polynomial)
  double dPdq[12];  // dP/dq (3x4): columns w,x,y,z
};

// q = (w,x,y,z), t, X world point.  Fills r and the factors of all Jacobians.
__device__ __forceinline__ void reproj_eval(int model, const double* __restrict__ cam, const double q[4],
                                            const double t[3], const double X[3], double ox, double oy,
                                            ReprojBlock& b) {
  const double w = q[0], a = q[1], bq = q[2], c = q[3];
  // uv = 2 v x X ; P = X + w uv + v x uv + t      (same polynomial as Ceres' UnitQuaternionRotatePoint)
  const double cx = bq * X[2] - c * X[1], cy = c * X[0] - a * X[2], cz = a * X[1] - bq * X[0];
  const double ux = 2.0 * cx, uy = 2.0 * cy, uz = 2.0 * cz;
  const double Px = X[0] + w * ux + (bq * uz - c * uy) + t[0];
  const double Py = X[1] + w * uy + (c * ux - a * uz) + t[1];
  const double Pz = X[2] + w * uz + (a * uy - bq * ux) + t[2];
  const double iz = 1.0 / Pz;
  const double u = Px * iz, v = Py * iz;
  D2 x, y;
  world_to_image_d2(model, cam, u, v, x, y);
  b.r[0] = x.a - ox;
  b.r[1] = y.a - oy;
  // d(u,v)/dP = [[iz, 0, -u iz], [0, iz, -v iz]]
  b.M[0] = x.p * iz; b.M[1] = x.q * iz; b.M[2] = -(x.p * u + x.q * v) * iz;
  b.M[3] = y.p * iz; b.M[4] = y.q * iz; b.M[5] = -(y.p * u + y.q * v) * iz;
  // D = I + 2w[v]x + 2(v v^T - |v|^2 I)
  b.D[0] = 1.0 - 2.0 * (bq * bq + c * c); b.D[1] = 2.0 * (a * bq - w * c);       b.D[2] = 2.0 * (a * c + w * bq);
  b.D[3] = 2.0 * (a * bq + w * c);        b.D[4] = 1.0 - 2.0 * (a * a + c * c);  b.D[5] = 2.0 * (bq * c - w * a);
  b.D[6] = 2.0 * (a * c - w * bq);        b.D[7] = 2.0 * (bq * c + w * a);       b.D[8] = 1.0 - 2.0 * (a * a + bq * bq);
  // dP/dw = 2 v x X ; dP/dv_k = 2w (e_k x X) + 2 e_k (v.X) + 2 v X_k - 4 X v_k
  const double vX = a * X[0] + bq * X[1] + c * X[2];
  const double vq[3] = {a, bq, c};
  b.dPdq[0] = ux; b.dPdq[4] = uy; b.dPdq[8] = uz;
  // e_0 x X = (0, -X2, X1); e_1 x X = (X2, 0, -X0); e_2 x X = (-X1, X0, 0)
  const double eX[3][3] = {{0.0, -X[2], X[1]}, {X[2], 0.0, -X[0]}, {-X[1], X[0], 0.0}};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      b.dPdq[4 * i + 1 + k] = 2.0 * w * eX[k][i] + (i == k ? 2.0 * vX : 0.0) + 2.0 * vq[i] * X[k] - 4.0 * X[i] * vq[k];
}

// ambient blocks as Ceres sees them (row-major 2x4, 2x3, 2x3)
__device__ __forceinline__ void reproj_jacobians(const ReprojBlock& b, double Jq[8], double Jt[6], double JX[6]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      Jq[4 * r + k] = b.M[3 * r] * b.dPdq[k] + b.M[3 * r + 1] * b.dPdq[4 + k] + b.M[3 * r + 2] * b.dPdq[8 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Jt[3 * r + k] = b.M[3 * r + k];
      JX[3 * r + k] = b.M[3 * r] * b.D[k] + b.M[3 * r + 1] * b.D[3 + k] + b.M[3 * r + 2] * b.D[6 + k];
    }
  }
}

// [3P Ceres] rho(s), rho'(s) for TrivialLoss / SoftLOneLoss(a) / CauchyLoss(a); rho'' <= 0 for all three,
// so the Corrector scales residual and Jacobian by sqrt(rho') and nothing else.
__device__ __forceinline__ void loss_eval(int type, double scale, double s, double& rho0, double& rho1) {
  if (type == PCD_LOSS_TRIVIAL) { rho0 = s; rho1 = 1.0; return; }
  const double bb = scale * scale, cc = 1.0 / bb, sum = 1.0 + s * cc;
  if (type == PCD_LOSS_SOFT_L1) {
    const double tmp = sqrt(sum);
    rho0 = 2.0 * bb * (tmp - 1.0);
    rho1 = fmax(2.2250738585072014e-308, 1.0 / tmp);
  } else {
    const double inv = 1.0 / sum;
    rho0 = bb * log(sum);
    rho1 = fmax(2.2250738585072014e-308, inv);
  }
}

// QuaternionManifold::PlusJacobian (4x3): Jq_tangent = Jq(2x4) * plus
__device__ __forceinline__ void quat_tangent(const double q[4], const double Jq[8], double Jqt[6]) {
  const double pl[12] = {-q[1], -q[2], -q[3], q[0], q[3], -q[2], -q[3], q[0], q[1], q[2], -q[1], q[0]};
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Jqt[3 * r + c] = Jq[4 * r] * pl[c] + Jq[4 * r + 1] * pl[3 + c] + Jq[4 * r + 2] * pl[6 + c] + Jq[4 * r + 3] * pl[9 + c];
}

// LiDAR plane block: r = w*|s|, J = w*sign(s)*n with sign(0) = 0 (Ceres' Jet of sqrt gives NaN at s == 0;
// strict != 0 reproduces that)
__device__ __forceinline__ void lidar_eval(const double X[3], const double abcd[4], double w, int strict, double& r,
                                           double J[3]) {
  const double s = X[0] * abcd[0] + X[1] * abcd[1] + X[2] * abcd[2] + abcd[3];
  const double e = 0.0 - s;
  r = w * sqrt(e * e);
  double sg = s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0);
  if (strict && s == 0.0) sg = __longlong_as_double(0x7FF8000000000000ll);
  J[0] = w * sg * abcd[0]; J[1] = w * sg * abcd[1]; J[2] = w * sg * abcd[2];
}

}  // namespace pcd
