// ba_cam_jac.h -- d(projection)/d(camera parameters) for the 11 camera models.
//
// The reference gets these columns from Ceres' autodiff of the templated functors
// (base/cost_functions.h:49-141 over base/camera_models.h WorldToImage) whenever
// BundleAdjustmentOptions::refine_focal_length / refine_principal_point / refine_extra_params
// (optim/bundle_adjustment.h:76-81) leave the camera block variable.  Here: K-wide forward duals over the
// parameters (u, v are constants), one instantiation per model, fp64.  Output block layout is Ceres'
// row-major 2 x K, written with a fixed row stride of PCD_CAM_JAC_STRIDE (= 12, the widest model) columns.
#pragma once
#include <hip/hip_runtime.h>

namespace pcd {

template <int K>
struct DK {
  double a;
  double d[K];
};

template <int K>
__device__ __forceinline__ DK<K> dk_var(double a, int i) {
  DK<K> r;
  r.a = a;
#pragma unroll
  for (int k = 0; k < K; ++k) r.d[k] = k == i ? 1.0 : 0.0;
  return r;
}
template <int K>
__device__ __forceinline__ DK<K> operator+(DK<K> f, DK<K> g) {
  DK<K> r; r.a = f.a + g.a;
#pragma unroll
  for (int k = 0; k < K; ++k) r.d[k] = f.d[k] + g.d[k];
  return r;
}
template <int K>
__device__ __forceinline__ DK<K> operator-(DK<K> f, DK<K> g) {
  DK<K> r; r.a = f.a - g.a;
#pragma unroll
  for (int k = 0; k < K; ++k) r.d[k] = f.d[k] - g.d[k];
  return r;
}
template <int K>
__device__ __forceinline__ DK<K> operator*(DK<K> f, DK<K> g) {
  DK<K> r; r.a = f.a * g.a;
#pragma unroll
  for (int k = 0; k < K; ++k) r.d[k] = f.a * g.d[k] + f.d[k] * g.a;
  return r;
}
template <int K>
__device__ __forceinline__ DK<K> operator/(DK<K> f, DK<K> g) {
  const double gi = 1.0 / g.a, fg = f.a * gi;
  DK<K> r; r.a = fg;
#pragma unroll
  for (int k = 0; k < K; ++k) r.d[k] = (f.d[k] - fg * g.d[k]) * gi;
  return r;
}
template <int K>
__device__ __forceinline__ DK<K> operator+(DK<K> f, double s) { f.a += s; return f; }
template <int K>
__device__ __forceinline__ DK<K> operator+(double s, DK<K> f) { f.a += s; return f; }
template <int K>
__device__ __forceinline__ DK<K> operator-(DK<K> f, double s) { f.a -= s; return f; }
template <int K>
__device__ __forceinline__ DK<K> operator*(DK<K> f, double s) {
  f.a *= s;
#pragma unroll
  for (int k = 0; k < K; ++k) f.d[k] *= s;
  return f;
}
template <int K>
__device__ __forceinline__ DK<K> operator*(double s, DK<K> f) { return f * s; }
template <int K>
__device__ __forceinline__ DK<K> operator/(DK<K> f, double s) { return f * (1.0 / s); }
template <int K>
__device__ __forceinline__ DK<K> dk_tan(DK<K> f) {
  const double t = tan(f.a), s = 1.0 + t * t;
  DK<K> r; r.a = t;
#pragma unroll
  for (int k = 0; k < K; ++k) r.d[k] = f.d[k] * s;
  return r;
}
template <int K>
__device__ __forceinline__ DK<K> dk_atan(DK<K> f) {
  const double s = 1.0 / (1.0 + f.a * f.a);
  DK<K> r; r.a = atan(f.a);
#pragma unroll
  for (int k = 0; k < K; ++k) r.d[k] = f.d[k] * s;
  return r;
}

__host__ __device__ constexpr int cam_num_params(int model) {
  return model == 0 ? 3 : model == 1 ? 4 : model == 2 ? 4 : model == 3 ? 5 : model == 4 ? 8 : model == 5 ? 8
       : model == 6 ? 12 : model == 7 ? 5 : model == 8 ? 4 : model == 9 ? 5 : 12;
}

// theta_d / r style fisheye distortion with variable coefficients (camera_models.h:963-990, :1272-1290, :1348-1370)
template <int K>
__device__ __forceinline__ void fisheye_dk(double u, double v, DK<K> k1, DK<K> k2, DK<K> k3, DK<K> k4, bool has2,
                                           bool has34, DK<K>& xu, DK<K>& xv) {
  const double r = sqrt(u * u + v * v);
  DK<K> zu; zu.a = u;
  DK<K> zv; zv.a = v;
#pragma unroll
  for (int k = 0; k < K; ++k) zu.d[k] = zv.d[k] = 0.0;
  if (r > 2.220446049250313e-16) {
    const double th = atan(r), th2 = th * th, th4 = th2 * th2;
    DK<K> ser = 1.0 + k1 * th2;
    if (has2) ser = ser + k2 * th4;
    if (has34) ser = ser + k3 * (th4 * th2) + k4 * (th4 * th4);
    const DK<K> thd = th * ser;
    xu = zu + (thd * (u / r) - u);
    xv = zv + (thd * (v / r) - v);
  } else {
    xu = zu;
    xv = zv;
  }
}

// J: [2][stride], columns >= K are left untouched (the caller zero-fills)
template <int MODEL>
__device__ __forceinline__ void cam_param_jacobian(const double* __restrict__ p, double u, double v,
                                                   double* __restrict__ J, int stride) {
  constexpr int K = cam_num_params(MODEL);
  typedef DK<K> T;
  T q[K];
#pragma unroll
  for (int i = 0; i < K; ++i) q[i] = dk_var<K>(p[i], i);
  T x, y;
  if constexpr (MODEL == 0) {          // SIMPLE_PINHOLE f cx cy
    x = q[0] * u + q[1]; y = q[0] * v + q[2];
  } else if constexpr (MODEL == 1) {   // PINHOLE fx fy cx cy
    x = q[0] * u + q[2]; y = q[1] * v + q[3];
  } else if constexpr (MODEL == 2) {   // SIMPLE_RADIAL f cx cy k
    const double r2 = u * u + v * v;
    const T rad = q[3] * r2;
    x = q[0] * (u + rad * u) + q[1]; y = q[0] * (v + rad * v) + q[2];
  } else if constexpr (MODEL == 3) {   // RADIAL f cx cy k1 k2
    const double r2 = u * u + v * v;
    const T rad = q[3] * r2 + q[4] * (r2 * r2);
    x = q[0] * (u + rad * u) + q[1]; y = q[0] * (v + rad * v) + q[2];
  } else if constexpr (MODEL == 4) {   // OPENCV fx fy cx cy k1 k2 p1 p2
    const double u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
    const T rad = q[4] * r2 + q[5] * (r2 * r2);
    const T du = rad * u + q[6] * (2.0 * uv) + q[7] * (r2 + 2.0 * u2);
    const T dv = rad * v + q[7] * (2.0 * uv) + q[6] * (r2 + 2.0 * v2);
    x = q[0] * (u + du) + q[2]; y = q[1] * (v + dv) + q[3];
  } else if constexpr (MODEL == 5) {   // OPENCV_FISHEYE fx fy cx cy k1 k2 k3 k4
    T xu, xv;
    fisheye_dk<K>(u, v, q[4], q[5], q[6], q[7], true, true, xu, xv);
    x = q[0] * xu + q[2]; y = q[1] * xv + q[3];
  } else if constexpr (MODEL == 6) {   // FULL_OPENCV fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6
    const double u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2;
    const T rad = (1.0 + q[4] * r2 + q[5] * r4 + q[8] * r6) / (1.0 + q[9] * r2 + q[10] * r4 + q[11] * r6);
    const T du = rad * u + q[6] * (2.0 * uv) + q[7] * (r2 + 2.0 * u2) - u;
    const T dv = rad * v + q[7] * (2.0 * uv) + q[6] * (r2 + 2.0 * v2) - v;
    x = q[0] * (u + du) + q[2]; y = q[1] * (v + dv) + q[3];
  } else if constexpr (MODEL == 7) {   // FOV fx fy cx cy omega
    const T om = q[4], om2 = om * om;
    const double rad2 = u * u + v * v;
    T fac;
    if (om2.a < 1e-4) {
      fac = (om2 * rad2) / 3.0 - om2 / 12.0 + 1.0;
    } else if (rad2 < 1e-4) {
      const T t = dk_tan(om / 2.0);
      fac = ((-2.0 * t) * ((4.0 * (t * t)) * rad2 - 3.0)) / (3.0 * om);
    } else {
      const double rad = sqrt(rad2);
      const T num = dk_atan((2.0 * dk_tan(om / 2.0)) * rad);
      fac = num / (rad * om);
    }
    x = q[0] * (fac * u) + q[2]; y = q[1] * (fac * v) + q[3];
  } else if constexpr (MODEL == 8) {   // SIMPLE_RADIAL_FISHEYE f cx cy k
    T xu, xv;
    fisheye_dk<K>(u, v, q[3], q[3], q[3], q[3], false, false, xu, xv);
    x = q[0] * xu + q[1]; y = q[0] * xv + q[2];
  } else if constexpr (MODEL == 9) {   // RADIAL_FISHEYE f cx cy k1 k2
    T xu, xv;
    fisheye_dk<K>(u, v, q[3], q[4], q[4], q[4], true, false, xu, xv);
    x = q[0] * xu + q[1]; y = q[0] * xv + q[2];
  } else {                             // 10 THIN_PRISM_FISHEYE fx fy cx cy k1 k2 p1 p2 k3 k4 sx1 sy1
    const double r = sqrt(u * u + v * v);
    double uu = u, vv = v;
    if (r > 2.220446049250313e-16) { const double th = atan(r); uu = th * u / r; vv = th * v / r; }
    const double u2 = uu * uu, uv = uu * vv, v2 = vv * vv, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2;
    const T rad = q[4] * r2 + q[5] * r4 + q[8] * r6 + q[9] * r8;
    const T du = rad * uu + q[6] * (2.0 * uv) + q[7] * (r2 + 2.0 * u2) + q[10] * r2;
    const T dv = rad * vv + q[7] * (2.0 * uv) + q[6] * (r2 + 2.0 * v2) + q[11] * r2;
    x = q[0] * (uu + du) + q[2]; y = q[1] * (vv + dv) + q[3];
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    J[i] = x.d[i];
    J[stride + i] = y.d[i];
  }
}

__device__ __forceinline__ void cam_param_jacobian_any(int model, const double* __restrict__ p, double u, double v,
                                                       double* __restrict__ J, int stride) {
  switch (model) {
    case 0: cam_param_jacobian<0>(p, u, v, J, stride); break;
    case 1: cam_param_jacobian<1>(p, u, v, J, stride); break;
    case 2: cam_param_jacobian<2>(p, u, v, J, stride); break;
    case 3: cam_param_jacobian<3>(p, u, v, J, stride); break;
    case 4: cam_param_jacobian<4>(p, u, v, J, stride); break;
    case 5: cam_param_jacobian<5>(p, u, v, J, stride); break;
    case 6: cam_param_jacobian<6>(p, u, v, J, stride); break;
    case 7: cam_param_jacobian<7>(p, u, v, J, stride); break;
    case 8: cam_param_jacobian<8>(p, u, v, J, stride); break;
    case 9: cam_param_jacobian<9>(p, u, v, J, stride); break;
    default: cam_param_jacobian<10>(p, u, v, J, stride); break;
  }
}

}  // namespace pcd
