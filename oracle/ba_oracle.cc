// oracle/ba_oracle.cc -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the residual / Jacobian evaluation that colmap-pcd's
// BundleAdjuster hands to Ceres.  Like the reference it differentiates the
// templated functors with forward-mode dual numbers ("Jets"), so the GPU
// kernels' hand-derived analytic Jacobians are checked against an
// independent derivation.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library.
//
// PARITY STATUS
//   * reprojection residuals: pinned by the reference's own known-answer
//     tests src/base/cost_functions_test.cc:41-104 (tests/golden/cost_function_kats.json).
//   * LiDAR plane residual (cost_functions.h:150-241), all Jacobians, loss
//     correction and the quaternion manifold: "parity unpinned" by the
//     reference (no test covers them; Ceres is not installed).  Pinned by
//     closed forms, finite differences and torch float64 autograd in tests/.
//
// Reference lines followed
//   src/base/cost_functions.h:100-135   BundleAdjustmentCostFunction::operator()
//   src/base/cost_functions.h:204-232   BundleAdjustmentLidarCostFunction::operator()
//   src/base/cost_functions.h:319-355   BundleAdjustmentConstantPoseCostFunction::operator()
//   src/base/cost_functions.h:610-627   SetQuaternionManifold / SetSubsetManifold
//   src/base/camera_models.h:614-1482   WorldToImage / Distortion of the 11 models
//   src/optim/bundle_adjustment.cc:53-68    CreateLossFunction
//   src/optim/bundle_adjustment.cc:993-1040 AddLidarToProblem (NaN guard, weights)
//   [3P, Ceres 2.1.0, restated from its published sources]
//     ceres::UnitQuaternionRotatePoint (rotation.h), Jet arithmetic (jet.h),
//     QuaternionManifold::PlusJacobian (manifold.cc), SoftLOneLoss /
//     CauchyLoss (loss_function.cc), Corrector (corrector.cc, rho'' <= 0 branch).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

// ---------------------------------------------------------------- Jet ----
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
  explicit Jet(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0; }
  Jet(double x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0; v[k] = 1; }
};
template <int N> Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> Jet<N> operator-(const Jet<N>& f) { Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi; return h; }
template <int N> Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> Jet<N>& operator-=(Jet<N>& f, const Jet<N>& g) { f = f - g; return f; }
template <int N> Jet<N>& operator/=(Jet<N>& f, const Jet<N>& g) { f = f / g; return f; }
template <int N> Jet<N> operator*(const Jet<N>& f, double s) { Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> Jet<N> operator+(double s, const Jet<N>& f) { return f + s; }
template <int N> Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> Jet<N> jsqrt(const Jet<N>& f) { Jet<N> h; h.a = std::sqrt(f.a); const double t = 1.0 / (2.0 * h.a); for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * t; return h; }
template <int N> Jet<N> jatan(const Jet<N>& f) { Jet<N> h; h.a = std::atan(f.a); const double t = 1.0 / (1.0 + f.a * f.a); for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * t; return h; }
template <int N> Jet<N> jtan(const Jet<N>& f) { Jet<N> h; h.a = std::tan(f.a); const double t = 1.0 + h.a * h.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * t; return h; }
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jatan(double x) { return std::atan(x); }
inline double jtan(double x) { return std::tan(x); }

template <typename T> struct Lit { static T of(double x) { return T(x); } };
template <> struct Lit<double> { static double of(double x) { return x; } };

// ------------------------------------------------------- camera models ----
// ids as in camera_models.h:187-347 CAMERA_MODEL_DEFINITIONS
enum { SIMPLE_PINHOLE = 0, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV, OPENCV_FISHEYE, FULL_OPENCV, FOV,
       SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE, THIN_PRISM_FISHEYE, NUM_MODELS };
const int kNumParams[NUM_MODELS] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12};

#define L(x) Lit<T>::of(x)

template <typename T> void fisheye_dist(const T& u, const T& v, const T& poly_k1, const T& poly_k2,
                                        const T& poly_k3, const T& poly_k4, int order, T* du, T* dv) {
  // camera_models.h:963-990 (OPENCV_FISHEYE), :1272-1290, :1348-1370
  const T r = jsqrt(u * u + v * v);
  if (r > L(std::numeric_limits<double>::epsilon())) {
    const T theta = jatan(r);
    const T theta2 = theta * theta;
    T series = L(1) + poly_k1 * theta2;
    if (order >= 2) { const T theta4 = theta2 * theta2; series = series + poly_k2 * theta4;
      if (order >= 4) { const T theta6 = theta4 * theta2; const T theta8 = theta4 * theta4;
        series = series + poly_k3 * theta6 + poly_k4 * theta8; } }
    const T thetad = theta * series;
    *du = u * thetad / r - u;
    *dv = v * thetad / r - v;
  } else {
    *du = L(0);
    *dv = L(0);
  }
}

template <typename T>
void world_to_image(int model, const T* p, const T u, const T v, T* x, T* y) {
  switch (model) {
    case SIMPLE_PINHOLE:  // :614-626
      *x = p[0] * u + p[1]; *y = p[0] * v + p[2]; return;
    case PINHOLE:         // :663-676
      *x = p[0] * u + p[2]; *y = p[1] * v + p[3]; return;
    case SIMPLE_RADIAL: { // :714-757
      const T u2 = u * u, v2 = v * v, r2 = u2 + v2; const T radial = p[3] * r2;
      const T du = u * radial, dv = v * radial;
      *x = u + du; *y = v + dv; *x = p[0] * *x + p[1]; *y = p[0] * *y + p[2]; return; }
    case RADIAL: {        // :783-827
      const T u2 = u * u, v2 = v * v, r2 = u2 + v2; const T radial = p[3] * r2 + p[4] * r2 * r2;
      const T du = u * radial, dv = v * radial;
      *x = u + du; *y = v + dv; *x = p[0] * *x + p[1]; *y = p[0] * *y + p[2]; return; }
    case OPENCV: {        // :853-902
      const T k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7];
      const T u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
      const T radial = k1 * r2 + k2 * r2 * r2;
      const T du = u * radial + L(2) * p1 * uv + p2 * (r2 + L(2) * u2);
      const T dv = v * radial + L(2) * p2 * uv + p1 * (r2 + L(2) * v2);
      *x = u + du; *y = v + dv; *x = p[0] * *x + p[2]; *y = p[1] * *y + p[3]; return; }
    case OPENCV_FISHEYE: { // :929-990
      T du, dv; fisheye_dist(u, v, p[4], p[5], p[6], p[7], 4, &du, &dv);
      *x = u + du; *y = v + dv; *x = p[0] * *x + p[2]; *y = p[1] * *y + p[3]; return; }
    case FULL_OPENCV: {   // :1024-1085
      const T k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7], k3 = p[8], k4 = p[9], k5 = p[10], k6 = p[11];
      const T u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2;
      const T radial = (L(1) + k1 * r2 + k2 * r4 + k3 * r6) / (L(1) + k4 * r2 + k5 * r4 + k6 * r6);
      const T du = u * radial + L(2) * p1 * uv + p2 * (r2 + L(2) * u2) - u;
      const T dv = v * radial + L(2) * p2 * uv + p1 * (r2 + L(2) * v2) - v;
      *x = u + du; *y = v + dv; *x = p[0] * *x + p[2]; *y = p[1] * *y + p[3]; return; }
    case FOV: {           // :1105-1160
      const T omega = p[4]; const T kEps = L(1e-4);
      const T radius2 = u * u + v * v; const T omega2 = omega * omega; T factor;
      if (omega2 < kEps) {
        factor = (omega2 * radius2) / L(3) - omega2 / L(12) + L(1);
      } else if (radius2 < kEps) {
        const T t = jtan(omega / L(2));
        factor = (L(-2) * t * (L(4) * radius2 * t * t - L(3))) / (L(3) * omega);
      } else {
        const T radius = jsqrt(radius2);
        const T numerator = jatan(radius * L(2) * jtan(omega / L(2)));
        factor = numerator / (radius * omega);
      }
      *x = u * factor; *y = v * factor; *x = p[0] * *x + p[2]; *y = p[1] * *y + p[3]; return; }
    case SIMPLE_RADIAL_FISHEYE: { // :1240-1290
      T du, dv; fisheye_dist(u, v, p[3], p[3], p[3], p[3], 1, &du, &dv);
      *x = u + du; *y = v + dv; *x = p[0] * *x + p[1]; *y = p[0] * *y + p[2]; return; }
    case RADIAL_FISHEYE: { // :1316-1370
      T du, dv; fisheye_dist(u, v, p[3], p[4], p[4], p[4], 2, &du, &dv);
      *x = u + du; *y = v + dv; *x = p[0] * *x + p[1]; *y = p[0] * *y + p[2]; return; }
    case THIN_PRISM_FISHEYE: { // :1405-1482
      const T r = jsqrt(u * u + v * v); T uu, vv;
      if (r > L(std::numeric_limits<double>::epsilon())) { const T theta = jatan(r); uu = theta * u / r; vv = theta * v / r; }
      else { uu = u; vv = v; }
      const T k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7], k3 = p[8], k4 = p[9], sx1 = p[10], sy1 = p[11];
      const T u2 = uu * uu, uv = uu * vv, v2 = vv * vv, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2, r8 = r6 * r2;
      const T radial = k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8;
      const T du = uu * radial + L(2) * p1 * uv + p2 * (r2 + L(2) * u2) + sx1 * r2;
      const T dv = vv * radial + L(2) * p2 * uv + p1 * (r2 + L(2) * v2) + sy1 * r2;
      *x = uu + du; *y = vv + dv; *x = p[0] * *x + p[2]; *y = p[1] * *y + p[3]; return; }
  }
}
#undef L

// [3P] ceres::UnitQuaternionRotatePoint, Ceres 2.1.0 rotation.h
template <typename T> void unit_quaternion_rotate_point(const T q[4], const T pt[3], T result[3]) {
  T uv0 = q[2] * pt[2] - q[3] * pt[1];
  T uv1 = q[3] * pt[0] - q[1] * pt[2];
  T uv2 = q[1] * pt[1] - q[2] * pt[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  result[0] = pt[0] + q[0] * uv0;
  result[1] = pt[1] + q[0] * uv1;
  result[2] = pt[2] + q[0] * uv2;
  result[0] += q[2] * uv2 - q[3] * uv1;
  result[1] += q[3] * uv0 - q[1] * uv2;
  result[2] += q[1] * uv1 - q[2] * uv0;
}

// cost_functions.h:100-135 (also :319-355 with q,t taken as constants)
template <typename T>
void reprojection_functor(int model, const T* qvec, const T* tvec, const T* point3D, const T* cam,
                          double ox, double oy, T* residuals) {
  T projection[3];
  unit_quaternion_rotate_point(qvec, point3D, projection);
  projection[0] += tvec[0];
  projection[1] += tvec[1];
  projection[2] += tvec[2];
  projection[0] /= projection[2];
  projection[1] /= projection[2];
  world_to_image(model, cam, projection[0], projection[1], &residuals[0], &residuals[1]);
  residuals[0] -= Lit<T>::of(ox);
  residuals[1] -= Lit<T>::of(oy);
}

// cost_functions.h:204-232
template <typename T>
void lidar_functor(const T* X, double a, double b, double c, double d, double w, T* residuals) {
  const T s = X[0] * a + X[1] * b + X[2] * c + d;
  residuals[0] = Lit<T>::of(w) * jsqrt((Lit<T>::of(0.) - s) * (Lit<T>::of(0.) - s));
}

// Evaluate one variable-pose reprojection block with full ambient Jacobians:
// partial index layout [q0..q3 | t0..t2 | X0..X2 | cam0..camK-1]
template <int K>
void eval_reproj_block(int model, const double* q, const double* t, const double* X, const double* cam,
                       double ox, double oy, double* r, double* Jq, double* Jt, double* JX, double* Jc) {
  constexpr int N = 10 + K;
  typedef Jet<N> J;
  J jq[4], jt[3], jX[3], jc[12], res[2];  // 12 = max K; the model switch never reads past K
  for (int i = 0; i < 4; ++i) jq[i] = J(q[i], i);
  for (int i = 0; i < 3; ++i) jt[i] = J(t[i], 4 + i);
  for (int i = 0; i < 3; ++i) jX[i] = J(X[i], 7 + i);
  for (int i = 0; i < K; ++i) jc[i] = J(cam[i], 10 + i);
  reprojection_functor<J>(model, jq, jt, jX, jc, ox, oy, res);
  for (int row = 0; row < 2; ++row) {
    r[row] = res[row].a;
    if (Jq) for (int i = 0; i < 4; ++i) Jq[row * 4 + i] = res[row].v[i];
    if (Jt) for (int i = 0; i < 3; ++i) Jt[row * 3 + i] = res[row].v[4 + i];
    if (JX) for (int i = 0; i < 3; ++i) JX[row * 3 + i] = res[row].v[7 + i];
    if (Jc) for (int i = 0; i < K; ++i) Jc[row * K + i] = res[row].v[10 + i];
  }
}

void eval_reproj_dispatch(int model, const double* q, const double* t, const double* X, const double* cam,
                          double ox, double oy, double* r, double* Jq, double* Jt, double* JX, double* Jc) {
  switch (kNumParams[model]) {
    case 3: eval_reproj_block<3>(model, q, t, X, cam, ox, oy, r, Jq, Jt, JX, Jc); break;
    case 4: eval_reproj_block<4>(model, q, t, X, cam, ox, oy, r, Jq, Jt, JX, Jc); break;
    case 5: eval_reproj_block<5>(model, q, t, X, cam, ox, oy, r, Jq, Jt, JX, Jc); break;
    case 8: eval_reproj_block<8>(model, q, t, X, cam, ox, oy, r, Jq, Jt, JX, Jc); break;
    case 12: eval_reproj_block<12>(model, q, t, X, cam, ox, oy, r, Jq, Jt, JX, Jc); break;
  }
}

// [3P] Ceres loss functions: rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s)
void loss_evaluate(int type, double scale, double s, double rho[3]) {
  if (type == 0) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return; }          // TrivialLoss
  const double b = scale * scale, c = 1.0 / b;
  if (type == 1) {                                                            // SoftLOneLoss(a)
    const double sum = 1.0 + s * c, tmp = std::sqrt(sum);
    rho[0] = 2.0 * b * (tmp - 1.0);
    rho[1] = std::max(std::numeric_limits<double>::min(), 1.0 / tmp);
    rho[2] = -(c * rho[1]) / (2.0 * sum);
    return;
  }
  const double sum = 1.0 + s * c, inv = 1.0 / sum;                            // CauchyLoss(a)
  rho[0] = b * std::log(sum);
  rho[1] = std::max(std::numeric_limits<double>::min(), inv);
  rho[2] = -c * (inv * inv);
}

// [3P] QuaternionManifold::PlusJacobian (4x3, row-major) at x = (w,x,y,z)
void quat_plus_jacobian(const double* x, double* j) {
  j[0] = -x[1]; j[1] = -x[2]; j[2] = -x[3];
  j[3] = x[0];  j[4] = x[3];  j[5] = -x[2];
  j[6] = -x[3]; j[7] = x[0];  j[8] = x[1];
  j[9] = x[2];  j[10] = -x[1]; j[11] = x[0];
}

}  // namespace

// =========================================================== C interface ===
extern "C" {

struct oracle_ba_problem {
  int32_t num_cameras;
  const int32_t* cam_model;      // [C]
  const int32_t* cam_param_off;  // [C] offset into cam_params
  const double* cam_params;
  int32_t num_images;
  const double* poses;           // [I][7] qw qx qy qz tx ty tz
  const int32_t* image_camera;   // [I]
  const uint8_t* image_const_pose;   // [I] 1: constant (functor :256-370)
  const uint8_t* image_const_tvec;   // [I] bit k set: tvec[k] constant (SubsetManifold), may be NULL
  int32_t num_points;
  const double* points;          // [P][3]
  const uint8_t* point_const;    // [P] may be NULL
  int64_t num_obs;
  const int32_t* obs_image;      // [O]
  const int32_t* obs_point;      // [O]
  const double* obs_xy;          // [O][2]
  int64_t num_lidar;
  const int32_t* lidar_point;    // [L]
  const double* lidar_abcd;      // [L][4]
  const double* lidar_weight;    // [L]
  int32_t loss_type;             // 0 TRIVIAL 1 SOFT_L1 2 CAUCHY
  double loss_scale;
  int32_t cam_jac_stride;        // columns reserved per row in Jc (>= max K)
};

int oracle_camera_num_params(int model) { return (model >= 0 && model < NUM_MODELS) ? kNumParams[model] : -1; }

// plain-double evaluation of one block (what Evaluate(params, residuals, nullptr) returns)
void oracle_reproj_residual(int model, const double* q, const double* t, const double* X, const double* cam,
                            const double* obs, double* r) {
  reprojection_functor<double>(model, q, t, X, cam, obs[0], obs[1], r);
}

void oracle_reproj_block(int model, const double* q, const double* t, const double* X, const double* cam,
                         const double* obs, double* r, double* Jq, double* Jt, double* JX, double* Jc) {
  eval_reproj_dispatch(model, q, t, X, cam, obs[0], obs[1], r, Jq, Jt, JX, Jc);
}

// strict != 0 reproduces the Jet of sqrt at s = 0 (0/0 -> NaN); otherwise sign(0) = 0.
void oracle_lidar_block(const double* X, const double* abcd, double w, int strict, double* r, double* JX) {
  typedef Jet<3> J;
  J jX[3] = {J(X[0], 0), J(X[1], 1), J(X[2], 2)}, res[1];
  lidar_functor<J>(jX, abcd[0], abcd[1], abcd[2], abcd[3], w, res);
  r[0] = res[0].a;
  if (JX) for (int i = 0; i < 3; ++i) {
    JX[i] = res[0].v[i];
    if (!strict && res[0].a == 0.0) JX[i] = 0.0;
  }
}

void oracle_loss(int type, double scale, double s, double* rho3) { loss_evaluate(type, scale, s, rho3); }

// Raw (un-corrected, ambient) residuals and Jacobians of every block: what
// each CostFunction::Evaluate returns to Ceres.
//   residuals [2*O + L]; Jq [O][2][4]; Jt [O][2][3]; JX [O][2][3];
//   Jc [O][2][stride]; JL [L][3].  Any Jacobian pointer may be NULL.
// Blocks of constant-pose images leave Jq/Jt rows zero (the functor has no
// such parameter blocks, cost_functions.h:256-370).
void oracle_ba_evaluate_raw(const oracle_ba_problem* p, double* residuals, double* Jq, double* Jt,
                            double* JX, double* Jc, double* JL) {
  const int S = p->cam_jac_stride;
  for (int64_t o = 0; o < p->num_obs; ++o) {
    const int im = p->obs_image[o], pt = p->obs_point[o], cm = p->image_camera[im];
    const int model = p->cam_model[cm];
    const int K = kNumParams[model];
    const double* pose = p->poses + 7 * (size_t)im;
    double r[2], jq[8], jt[6], jx[6], jc[24];
    eval_reproj_dispatch(model, pose, pose + 4, p->points + 3 * (size_t)pt,
                         p->cam_params + p->cam_param_off[cm], p->obs_xy[2 * o], p->obs_xy[2 * o + 1],
                         r, jq, jt, jx, jc);
    residuals[2 * o] = r[0];
    residuals[2 * o + 1] = r[1];
    const bool cpose = p->image_const_pose && p->image_const_pose[im];
    if (Jq) for (int i = 0; i < 8; ++i) Jq[8 * o + i] = cpose ? 0.0 : jq[i];
    if (Jt) for (int i = 0; i < 6; ++i) Jt[6 * o + i] = cpose ? 0.0 : jt[i];
    if (JX) for (int i = 0; i < 6; ++i) JX[6 * o + i] = jx[i];
    if (Jc) for (int row = 0; row < 2; ++row)
      for (int i = 0; i < S; ++i) Jc[(2 * o + row) * S + i] = i < K ? jc[row * K + i] : 0.0;
  }
  for (int64_t l = 0; l < p->num_lidar; ++l) {
    double r[1], jx[3];
    oracle_lidar_block(p->points + 3 * (size_t)p->lidar_point[l], p->lidar_abcd + 4 * l, p->lidar_weight[l], 0, r, jx);
    residuals[2 * p->num_obs + l] = r[0];
    if (JL) for (int i = 0; i < 3; ++i) JL[3 * l + i] = jx[i];
  }
}

// Camera blocks of the normal equations with refined intrinsics (ParameterizeCameras,
// optim/bundle_adjustment.cc:1047-1100): refine[cam_param_off[c] + k] = 1 when parameter k of camera c is
// optimised; the columns of constant parameters vanish (whole block constant, or SubsetManifold).
//   Hcam [C][S][S] = sum Jc^T Jc, gcam [C][S] = sum Jc^T r   over the observations of the camera's images
//   Ecam [I][S][6] = sum Jc^T Jp   (pose tangent as in oracle_ba_normal_equations)
//   Wcam [O][S][3] = Jc^T JX       per observation              S = cam_jac_stride; any output may be NULL
void oracle_ba_camera_blocks(const oracle_ba_problem* p, const uint8_t* refine, double* Hcam, double* gcam,
                             double* Ecam, double* Wcam) {
  const int S = p->cam_jac_stride;
  if (Hcam) std::memset(Hcam, 0, sizeof(double) * S * S * (size_t)p->num_cameras);
  if (gcam) std::memset(gcam, 0, sizeof(double) * S * (size_t)p->num_cameras);
  if (Ecam) std::memset(Ecam, 0, sizeof(double) * S * 6 * (size_t)p->num_images);
  if (Wcam) std::memset(Wcam, 0, sizeof(double) * S * 3 * (size_t)p->num_obs);
  for (int64_t o = 0; o < p->num_obs; ++o) {
    const int im = p->obs_image[o], pt = p->obs_point[o], cm = p->image_camera[im];
    const int model = p->cam_model[cm], K = kNumParams[model];
    const double* pose = p->poses + 7 * (size_t)im;
    double r[2], jq[8], jt[6], jx[6], jc[24];
    eval_reproj_dispatch(model, pose, pose + 4, p->points + 3 * (size_t)pt, p->cam_params + p->cam_param_off[cm],
                         p->obs_xy[2 * o], p->obs_xy[2 * o + 1], r, jq, jt, jx, jc);
    double rho[3];
    loss_evaluate(p->loss_type, p->loss_scale, r[0] * r[0] + r[1] * r[1], rho);
    const double sr = std::sqrt(rho[1]);
    const bool cpose = p->image_const_pose && p->image_const_pose[im];
    const bool cpt = p->point_const && p->point_const[pt];
    const unsigned tmask = (p->image_const_tvec ? p->image_const_tvec[im] : 0);
    double plus[12], Jp[12], Jx[6], Jc[24];
    quat_plus_jacobian(pose, plus);
    for (int row = 0; row < 2; ++row) {
      for (int c = 0; c < 3; ++c) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += jq[row * 4 + k] * plus[k * 3 + c];
        Jp[row * 6 + c] = cpose ? 0.0 : sr * acc;
        Jp[row * 6 + 3 + c] = (cpose || ((tmask >> c) & 1)) ? 0.0 : sr * jt[row * 3 + c];
      }
      for (int c = 0; c < 3; ++c) Jx[row * 3 + c] = cpt ? 0.0 : sr * jx[row * 3 + c];
      for (int k = 0; k < K; ++k)
        Jc[row * 12 + k] = (refine && refine[p->cam_param_off[cm] + k]) ? sr * jc[row * K + k] : 0.0;
    }
    const double rc[2] = {sr * r[0], sr * r[1]};
    for (int a = 0; a < K; ++a) {
      if (Hcam) for (int b = 0; b < K; ++b)
        Hcam[((size_t)cm * S + a) * S + b] += Jc[a] * Jc[b] + Jc[12 + a] * Jc[12 + b];
      if (gcam) gcam[(size_t)cm * S + a] += Jc[a] * rc[0] + Jc[12 + a] * rc[1];
      if (Ecam) for (int b = 0; b < 6; ++b)
        Ecam[((size_t)im * S + a) * 6 + b] += Jc[a] * Jp[b] + Jc[12 + a] * Jp[6 + b];
      if (Wcam) for (int b = 0; b < 3; ++b)
        Wcam[((size_t)o * S + a) * 3 + b] = Jc[a] * Jx[b] + Jc[12 + a] * Jx[3 + b];
    }
  }
}

// Residual-only evaluation: what each CostFunction::Evaluate(parameters, residuals, nullptr) returns when
// Ceres evaluates a trial step -- the functors instantiated with plain doubles, no Jets.
void oracle_ba_residuals(const oracle_ba_problem* p, double* residuals) {
  for (int64_t o = 0; o < p->num_obs; ++o) {
    const int im = p->obs_image[o], pt = p->obs_point[o], cm = p->image_camera[im];
    const double* pose = p->poses + 7 * (size_t)im;
    reprojection_functor<double>(p->cam_model[cm], pose, pose + 4, p->points + 3 * (size_t)pt,
                                 p->cam_params + p->cam_param_off[cm], p->obs_xy[2 * o], p->obs_xy[2 * o + 1],
                                 residuals + 2 * o);
  }
  for (int64_t l = 0; l < p->num_lidar; ++l) {
    const double* X = p->points + 3 * (size_t)p->lidar_point[l];
    const double* pl = p->lidar_abcd + 4 * l;
    lidar_functor<double>(X, pl[0], pl[1], pl[2], pl[3], p->lidar_weight[l], residuals + 2 * p->num_obs + l);
  }
}

// Post-BA filters' inputs.  Per observation:
//   sq_err = CalculateSquaredReprojectionError (base/projection.cc:104-117): P = R(q/|q|) X + t
//            (base/pose.cc QuaternionRotatePoint normalises first; |q| == 0 -> identity),
//            DBL_MAX if P.z < eps, else ||WorldToImage(P.xy / P.z) - obs||^2
//   depth  = P.z, what HasPointPositiveDepth (base/projection.cc) tests against eps
//            (base/reconstruction.cc:837-855 FilterObservationsWithNegativeDepth)
void oracle_ba_observation_errors(const oracle_ba_problem* p, double* sq_err, double* depth) {
  for (int64_t o = 0; o < p->num_obs; ++o) {
    const int im = p->obs_image[o], pt = p->obs_point[o], cm = p->image_camera[im];
    const double* pose = p->poses + 7 * (size_t)im;
    const double* X = p->points + 3 * (size_t)pt;
    double q[4] = {pose[0], pose[1], pose[2], pose[3]};
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n == 0) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else for (double& v : q) v /= n;
    double P[3];
    unit_quaternion_rotate_point(q, X, P);
    for (int k = 0; k < 3; ++k) P[k] += pose[4 + k];
    if (depth) depth[o] = P[2];
    if (!sq_err) continue;
    if (P[2] < std::numeric_limits<double>::epsilon()) { sq_err[o] = std::numeric_limits<double>::max(); continue; }
    double x, y;
    world_to_image<double>(p->cam_model[cm], p->cam_params + p->cam_param_off[cm], P[0] / P[2], P[1] / P[2], &x, &y);
    const double dx = x - p->obs_xy[2 * o], dy = y - p->obs_xy[2 * o + 1];
    sq_err[o] = dx * dx + dy * dy;
  }
}

// Cost 1/2 sum rho(||r_block||^2) plus loss-corrected, manifold-projected
// normal-equation blocks:
//   Himg [I][6][6], gimg [I][6]   pose tangent = (3 quaternion-tangent, 3 tvec); zero for constant poses
//   Hpt  [P][3][3], gpt  [P][3]   zero for constant points
//   W    [O][6][3]                Jp^T JX per observation (may be NULL)
// Any output may be NULL.  Returns the cost.
double oracle_ba_normal_equations(const oracle_ba_problem* p, double* Himg, double* gimg, double* Hpt,
                                  double* gpt, double* W) {
  if (Himg) std::memset(Himg, 0, sizeof(double) * 36 * (size_t)p->num_images);
  if (gimg) std::memset(gimg, 0, sizeof(double) * 6 * (size_t)p->num_images);
  if (Hpt) std::memset(Hpt, 0, sizeof(double) * 9 * (size_t)p->num_points);
  if (gpt) std::memset(gpt, 0, sizeof(double) * 3 * (size_t)p->num_points);
  double cost = 0.0;
  for (int64_t o = 0; o < p->num_obs; ++o) {
    const int im = p->obs_image[o], pt = p->obs_point[o], cm = p->image_camera[im];
    const int model = p->cam_model[cm];
    const double* pose = p->poses + 7 * (size_t)im;
    double r[2], jq[8], jt[6], jx[6], jc[24];
    eval_reproj_dispatch(model, pose, pose + 4, p->points + 3 * (size_t)pt,
                         p->cam_params + p->cam_param_off[cm], p->obs_xy[2 * o], p->obs_xy[2 * o + 1],
                         r, jq, jt, jx, jc);
    double rho[3];
    loss_evaluate(p->loss_type, p->loss_scale, r[0] * r[0] + r[1] * r[1], rho);
    cost += 0.5 * rho[0];
    const double sr = std::sqrt(rho[1]);  // Corrector, rho'' <= 0 branch
    const bool cpose = p->image_const_pose && p->image_const_pose[im];
    const bool cpt = p->point_const && p->point_const[pt];
    double Jp[12];  // 2x6
    double plus[12];
    quat_plus_jacobian(pose, plus);
    const unsigned tmask = (p->image_const_tvec ? p->image_const_tvec[im] : 0);
    for (int row = 0; row < 2; ++row) {
      for (int c = 0; c < 3; ++c) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += jq[row * 4 + k] * plus[k * 3 + c];
        Jp[row * 6 + c] = cpose ? 0.0 : sr * acc;
        Jp[row * 6 + 3 + c] = (cpose || ((tmask >> c) & 1)) ? 0.0 : sr * jt[row * 3 + c];
      }
    }
    double Jx[6];
    for (int i = 0; i < 6; ++i) Jx[i] = cpt ? 0.0 : sr * jx[i];
    const double rc[2] = {sr * r[0], sr * r[1]};
    if (Himg) for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b)
      Himg[36 * (size_t)im + 6 * a + b] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
    if (gimg) for (int a = 0; a < 6; ++a) gimg[6 * (size_t)im + a] += Jp[a] * rc[0] + Jp[6 + a] * rc[1];
    if (Hpt) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
      Hpt[9 * (size_t)pt + 3 * a + b] += Jx[a] * Jx[b] + Jx[3 + a] * Jx[3 + b];
    if (gpt) for (int a = 0; a < 3; ++a) gpt[3 * (size_t)pt + a] += Jx[a] * rc[0] + Jx[3 + a] * rc[1];
    if (W) for (int a = 0; a < 6; ++a) for (int b = 0; b < 3; ++b)
      W[18 * o + 3 * a + b] = Jp[a] * Jx[b] + Jp[6 + a] * Jx[3 + b];
  }
  for (int64_t l = 0; l < p->num_lidar; ++l) {
    const int pt = p->lidar_point[l];
    double r[1], jx[3];
    oracle_lidar_block(p->points + 3 * (size_t)pt, p->lidar_abcd + 4 * l, p->lidar_weight[l], 0, r, jx);
    double rho[3];
    loss_evaluate(p->loss_type, p->loss_scale, r[0] * r[0], rho);
    cost += 0.5 * rho[0];
    const double sr = std::sqrt(rho[1]);
    const bool cpt = p->point_const && p->point_const[pt];
    if (cpt) continue;
    const double rc = sr * r[0];
    double Jx[3] = {sr * jx[0], sr * jx[1], sr * jx[2]};
    if (Hpt) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Hpt[9 * (size_t)pt + 3 * a + b] += Jx[a] * Jx[b];
    if (gpt) for (int a = 0; a < 3; ++a) gpt[3 * (size_t)pt + a] += Jx[a] * rc;
  }
  return cost;
}

}  // extern "C"
