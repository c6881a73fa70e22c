// sdvl_math.h — Rigid / camera / small dense algebra shared by the host layer and the gfx950 kernels.
// Mirrors the Eigen + extra/se3 operations the reference path uses:
//   Rigid (quaternion + translation): extra/se3.h:32-78, extra/se3.cc:28-177
//   Camera::Project / Unproject / IsInsideImage: camera.cc:69-79, camera.h:93-98
//   Jacobian3DToPlane, AbsMax: extra/utils.cc:28-42,99-118
//   Matrix<double,6,6>::ldlt().solve: image_align.cc:102, feature_align.cc:402
// Built with -ffp-contract=off on both host and device so that every expression rounds exactly as written.
#ifndef SDVL_MATH_H_
#define SDVL_MATH_H_

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SDVL_HD __host__ __device__ inline
#else
#define SDVL_HD inline
#endif

namespace sdvl {

struct V2 { double x, y; };
struct V3 { double x, y, z; };
struct M3 { double m[9]; };  // row-major

SDVL_HD V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
SDVL_HD V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
SDVL_HD V3 vscale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
SDVL_HD V3 vscale_l(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
SDVL_HD double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SDVL_HD double vnorm(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// sin and cos of a double in [0, 2*pi + eps] (the ORB orientation angle, extra/orb_detector.cc:361-362), fdlibm style: quadrant by Cody-Waite reduction with a
// 33-bit head of pi/2 (n <= 4, so n * head is exact), then the k_sin / k_cos kernels with the reduction's tail.  Within
// 1 ulp (double) of the true value like libm's and ocml's, so the float the caller rounds to is the same (DESIGN.md "frozen
// interpretations"); unlike ocml's sin()/cos() there is no large-argument path, which cost ~15 VGPRs in every kernel that
// describes corners.  Plain IEEE double arithmetic: identical on host and device (tests/test_abi_cpu.py checks the host
// build, sdvlh_sincos_2pi, against libm; every float in [0, 6.3] was compared once: no float differs).
SDVL_HD void sincos_2pi(double x, double *sn, double *cs) {
  const double n = __builtin_rint(x * 6.36619772367581382433e-01);
  const double r0 = x - n * 1.57079632673412561417e+00;  // exact product, (nearly always) exact difference
  const double w = n * 6.07710050650619224932e-11;
  const double y = r0 - w;
  const double t = (r0 - y) - w;  // tail of the reduced argument
  const double z = y * y;
  // k_sin(y, t)
  const double v = z * y;
  const double rs = 8.33333333332248946124e-03 +
                    z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double ks = y - ((z * (0.5 * t - v * rs) - t) - v * -1.66666666666666324348e-01);
  // k_cos(y, t), |y| <= pi/4 + eps
  const double rc = z * (4.16666666666666019037e-02 +
                         z * (-1.38888888888741095749e-03 +
                              z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double ay = y < 0 ? -y : y;
  double kc;
  if (ay < 0.3) {
    kc = 1.0 - (0.5 * z - (z * rc - y * t));
  } else {
    const double qx = ay > 0.78125 ? 0.28125 : 0.25 * ay;  // fdlibm truncates |y|/4 to its high word; any qx near |y|/4 keeps the sum exact
    const double hz = 0.5 * z - qx;
    kc = (1.0 - qx) - (hz - (z * rc - y * t));
  }
  const int q = static_cast<int>(n) & 3;
  const double s_ = (q & 1) ? kc : ks, c_ = (q & 1) ? ks : kc;
  *sn = (q & 2) ? -s_ : s_;
  *cs = (q == 1 || q == 2) ? -c_ : c_;
}

SDVL_HD V3 mvec(const M3 &R, V3 v) {
  return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
          R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}

SDVL_HD M3 mmul(const M3 &A, const M3 &B) {
  M3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}

// Quaterniond::toRotationMatrix()
SDVL_HD M3 quat_to_mat(double w, double x, double y, double z) {
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  M3 R;
  R.m[0] = 1.0 - (tyy + tzz); R.m[1] = txy - twz;         R.m[2] = txz + twy;
  R.m[3] = txy + twz;         R.m[4] = 1.0 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy;         R.m[7] = tyz + twx;         R.m[8] = 1.0 - (txx + tyy);
  return R;
}

struct Rigid {
  double q0, q1, q2, q3;
  V3 t;
};

SDVL_HD Rigid se3_identity() { return {1.0, 0.0, 0.0, 0.0, {0.0, 0.0, 0.0}}; }
SDVL_HD Rigid se3_from7(const double *p) { return {p[0], p[1], p[2], p[3], {p[4], p[5], p[6]}}; }
SDVL_HD void se3_to7(const Rigid &s, double *p) {
  p[0] = s.q0; p[1] = s.q1; p[2] = s.q2; p[3] = s.q3; p[4] = s.t.x; p[5] = s.t.y; p[6] = s.t.z;
}
SDVL_HD M3 se3_rot(const Rigid &s) { return quat_to_mat(s.q0, s.q1, s.q2, s.q3); }

// Rigid::Inverse, extra/se3.cc:59-70
SDVL_HD Rigid se3_inverse(const Rigid &s) {
  Rigid r;
  const double n2 = s.q0 * s.q0 + s.q1 * s.q1 + s.q2 * s.q2 + s.q3 * s.q3;
  if (n2 > 0.0) {
    r.q0 = s.q0 / n2; r.q1 = -s.q1 / n2; r.q2 = -s.q2 / n2; r.q3 = -s.q3 / n2;
  } else {
    r.q0 = 0.0; r.q1 = 0.0; r.q2 = 0.0; r.q3 = 0.0;
  }
  const V3 rt = mvec(quat_to_mat(r.q0, r.q1, r.q2, r.q3), s.t);
  r.t = {-rt.x, -rt.y, -rt.z};
  return r;
}

// Rigid * Vector3d, extra/se3.h:68
SDVL_HD V3 se3_apply(const Rigid &s, V3 p) { return vadd(mvec(se3_rot(s), p), s.t); }

// Rigid::operator*, extra/se3.cc:166-177
SDVL_HD Rigid se3_mul(const Rigid &a, const Rigid &b) {
  Rigid r;
  const double w = a.q0 * b.q0 - a.q1 * b.q1 - a.q2 * b.q2 - a.q3 * b.q3;
  const double x = a.q0 * b.q1 + a.q1 * b.q0 + a.q2 * b.q3 - a.q3 * b.q2;
  const double y = a.q0 * b.q2 + a.q2 * b.q0 + a.q3 * b.q1 - a.q1 * b.q3;
  const double z = a.q0 * b.q3 + a.q3 * b.q0 + a.q1 * b.q2 - a.q2 * b.q1;
  const double n = sqrt(w * w + x * x + y * y + z * z);
  r.q0 = w / n; r.q1 = x / n; r.q2 = y / n; r.q3 = z / n;
  r.t = vadd(a.t, mvec(se3_rot(a), b.t));
  return r;
}

// Rigid::Exp, extra/se3.cc:72-94,114-138
SDVL_HD Rigid se3_exp(const double *u) {
  const double kEps = 1e-10;
  const V3 ups = {u[0], u[1], u[2]};
  const V3 om = {u[3], u[4], u[5]};
  const double theta = vnorm(om);
  const double half_theta = 0.5 * theta;
  double imag;
  double real, sin_half;
  sincos(half_theta, &sin_half, &real);  // same values as sin() and cos(), one argument reduction
  if (theta < kEps) {
    const double t2 = theta * theta;
    const double t4 = t2 * t2;
    imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4;
  } else {
    imag = sin_half / theta;
  }
  Rigid r;
  r.q0 = real; r.q1 = imag * om.x; r.q2 = imag * om.y; r.q3 = imag * om.z;
  M3 Om;
  Om.m[0] = 0;     Om.m[1] = -om.z; Om.m[2] = om.y;
  Om.m[3] = om.z;  Om.m[4] = 0;     Om.m[5] = -om.x;
  Om.m[6] = -om.y; Om.m[7] = om.x;  Om.m[8] = 0;
  const M3 Om2 = mmul(Om, Om);
  M3 V;
  if (theta < kEps) {
    V = quat_to_mat(r.q0, r.q1, r.q2, r.q3);
  } else {
    const double t2 = theta * theta;
    double sin_theta, cos_theta;
    sincos(theta, &sin_theta, &cos_theta);
    const double ca = (1 - cos_theta) / (t2);
    const double cb = (theta - sin_theta) / (t2 * theta);
    for (int i = 0; i < 9; i++) V.m[i] = (((i % 4) == 0 ? 1.0 : 0.0) + ca * Om.m[i]) + cb * Om2.m[i];
  }
  r.t = mvec(V, ups);
  return r;
}

// Rigid::Log, extra/se3.cc:96-112,140-164
SDVL_HD void se3_log(const Rigid &s, double *out) {
  const double kEps = 1e-10;
  const double n = sqrt(s.q1 * s.q1 + s.q2 * s.q2 + s.q3 * s.q3);
  const double w = s.q0;
  double k;
  if (n < kEps) k = 2. / w - 2. * (n * n) / (w * (w * w));
  else k = 2 * atan(n / w) / n;
  const double theta = k * n;
  const V3 om = {k * s.q1, k * s.q2, k * s.q3};
  M3 Om;
  Om.m[0] = 0;     Om.m[1] = -om.z; Om.m[2] = om.y;
  Om.m[3] = om.z;  Om.m[4] = 0;     Om.m[5] = -om.x;
  Om.m[6] = -om.y; Om.m[7] = om.x;  Om.m[8] = 0;
  const M3 Om2 = mmul(Om, Om);
  double c2;
  if (theta < kEps) c2 = (1. / 12.);
  else c2 = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
  M3 Vinv;
  for (int i = 0; i < 9; i++) Vinv.m[i] = (((i % 4) == 0 ? 1.0 : 0.0) - 0.5 * Om.m[i]) + c2 * Om2.m[i];
  const V3 up = mvec(Vinv, s.t);
  out[0] = up.x; out[1] = up.y; out[2] = up.z; out[3] = om.x; out[4] = om.y; out[5] = om.z;
}

struct Cam {
  double width, height, fx, fy, u0, v0;
};

SDVL_HD V2 cam_project(const Cam &c, V3 p) { return {c.u0 + c.fx * p.x / p.z, c.v0 + c.fy * p.y / p.z}; }
SDVL_HD V3 cam_unproject(const Cam &c, V2 p) {
  V3 v = {(p.x - c.u0) / c.fx, (p.y - c.v0) / c.fy, 1.0};
  const double n = vnorm(v);
  return {v.x / n, v.y / n, v.z / n};
}
SDVL_HD bool cam_inside(const Cam &c, int px, int py, int m) {
  return px >= m && px < c.width - m && py >= m && py < c.height - m;
}
SDVL_HD bool cam_inside_level(const Cam &c, int px, int py, int m, int l) {
  return px >= m && px < c.width / (1 << l) - m && py >= m && py < c.height / (1 << l) - m;
}

// Jacobian3DToPlane (2x6), extra/utils.cc:99-118; J[0..5] row 0, J[6..11] row 1
SDVL_HD void jacobian_3d_to_plane(V3 p, double *J) {
  const double x = p.x, y = p.y;
  const double z_inv = 1. / p.z;
  const double z_inv_2 = z_inv * z_inv;
  J[0] = -z_inv;
  J[1] = 0.0;
  J[2] = x * z_inv_2;
  J[3] = y * J[2];
  J[4] = -(1.0 + x * J[2]);
  J[5] = y * z_inv;
  J[6] = 0.0;
  J[7] = -z_inv;
  J[8] = y * z_inv_2;
  J[9] = 1.0 + y * J[8];
  J[10] = -J[3];
  J[11] = -x * z_inv;
}

SDVL_HD double abs_max6(const double *v) {
  double mx = -1;
  for (int i = 0; i < 6; i++) {
    const double a = fabs(v[i]);
    if (a > mx) mx = a;
  }
  return mx;
}

// Eigen LDLT (lower, diagonal pivoting, unblocked) + solve with pseudo-inverse of D.  A is 6x6 row-major,
// only its lower triangle is read.
SDVL_HD void ldlt_solve6(const double *Ain, const double *bin, double *x) {
  double a[36];
  int tr[6];
  for (int i = 0; i < 36; i++) a[i] = Ain[i];
  for (int k = 0; k < 6; k++) {
    int idx = k;
    double big = fabs(a[7 * k]);
    for (int i = k + 1; i < 6; i++) {
      const double v = fabs(a[7 * i]);
      if (v > big) { big = v; idx = i; }
    }
    tr[k] = idx;
    if (k != idx) {
      for (int j = 0; j < k; j++) { const double tmp = a[6 * k + j]; a[6 * k + j] = a[6 * idx + j]; a[6 * idx + j] = tmp; }
      for (int i = idx + 1; i < 6; i++) { const double tmp = a[6 * i + k]; a[6 * i + k] = a[6 * i + idx]; a[6 * i + idx] = tmp; }
      { const double tmp = a[7 * k]; a[7 * k] = a[7 * idx]; a[7 * idx] = tmp; }
      for (int i = k + 1; i < idx; i++) { const double tmp = a[6 * i + k]; a[6 * i + k] = a[6 * idx + i]; a[6 * idx + i] = tmp; }
    }
    if (k > 0) {
      double temp[6];
      for (int j = 0; j < k; j++) temp[j] = a[7 * j] * a[6 * k + j];
      double acc = 0.0;
      for (int j = 0; j < k; j++) acc += a[6 * k + j] * temp[j];
      a[7 * k] -= acc;
      for (int i = k + 1; i < 6; i++) {
        double a2 = 0.0;
        for (int j = 0; j < k; j++) a2 += a[6 * i + j] * temp[j];
        a[6 * i + k] -= a2;
      }
    }
    const double akk = a[7 * k];
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < 6; j++) tr[j] = j;
      break;
    }
    if (valid)
      for (int i = k + 1; i < 6; i++) a[6 * i + k] /= akk;
  }
  double d[6];
  for (int i = 0; i < 6; i++) d[i] = bin[i];
  for (int k = 0; k < 6; k++)
    if (tr[k] != k) { const double tmp = d[k]; d[k] = d[tr[k]]; d[tr[k]] = tmp; }
  for (int i = 0; i < 6; i++) {
    double acc = d[i];
    for (int j = 0; j < i; j++) acc -= a[6 * i + j] * d[j];
    d[i] = acc;
  }
  const double tol = 1.0 / 1.7976931348623157e308;
  for (int i = 0; i < 6; i++) {
    if (fabs(a[7 * i]) > tol) d[i] /= a[7 * i];
    else d[i] = 0.0;
  }
  for (int i = 5; i >= 0; i--) {
    double acc = d[i];
    for (int j = i + 1; j < 6; j++) acc -= a[6 * j + i] * d[j];
    d[i] = acc;
  }
  for (int k = 5; k >= 0; k--)
    if (tr[k] != k) { const double tmp = d[k]; d[k] = d[tr[k]]; d[tr[k]] = tmp; }
  for (int i = 0; i < 6; i++) x[i] = d[i];
}

#if defined(__HIPCC__)
// Same factorisation and solve with every array index a compile-time constant (run-time pivot handled by predicated
// swaps): stays in registers on the device where ldlt_solve6's dynamic indexing would go to scratch.
// ldlt_solve6 (sdvl_math.h) with compile-time indices only: swaps become predicated moves, so everything stays in registers
// kUniform: every active lane of the wave solves the SAME system (or only one lane is active): the pivot index is then
// made a scalar, the swap blocks become scalar branches and only the one that applies is executed.
// Two halves, so that a caller that solves several right-hand sides with ONE matrix factorises once:
// ldlt_factor6_reg leaves the factor (unit lower triangle + D on the diagonal) in a[36] and the transpositions in tr[6],
// ldlt_apply6_reg solves with them.  ldlt_solve6_reg = factor + apply, operation for operation as before.
template <bool kUniform = false>
SDVL_HD void ldlt_factor6_reg(const double *Ain, double *a, int *tr) {
#pragma unroll
  for (int i = 0; i < 36; i++) a[i] = Ain[i];
  bool alive = true;  // false once the k == 0 pivot is exactly zero (all-zero diagonal): nothing more to do
#pragma unroll
  for (int k = 0; k < 6; k++) {
    int idx = k;
    double big = fabs(a[7 * k]);
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double v = fabs(a[7 * i]);
      if (v > big) { big = v; idx = i; }
    }
    if (!alive) idx = k;
#if defined(__HIP_DEVICE_COMPILE__)
    if (kUniform) idx = __builtin_amdgcn_readfirstlane(idx);
#endif
    tr[k] = idx;
    if (alive) {
      // swap rows/cols k <-> idx of the lower triangle (Eigen ldlt unblocked), predicated on the run-time idx
#pragma unroll
      for (int c = k + 1; c < 6; c++) {
        if (c == idx) {
#pragma unroll
          for (int j = 0; j < k; j++) { const double t = a[6 * k + j]; a[6 * k + j] = a[6 * c + j]; a[6 * c + j] = t; }
#pragma unroll
          for (int i = c + 1; i < 6; i++) { const double t = a[6 * i + k]; a[6 * i + k] = a[6 * i + c]; a[6 * i + c] = t; }
          { const double t = a[7 * k]; a[7 * k] = a[7 * c]; a[7 * c] = t; }
#pragma unroll
          for (int i = k + 1; i < c; i++) { const double t = a[6 * i + k]; a[6 * i + k] = a[6 * c + i]; a[6 * c + i] = t; }
        }
      }
      if (k > 0) {
        double temp[6];
#pragma unroll
        for (int j = 0; j < k; j++) temp[j] = a[7 * j] * a[6 * k + j];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < k; j++) acc += a[6 * k + j] * temp[j];
        a[7 * k] -= acc;
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
          double a2 = 0.0;
#pragma unroll
          for (int j = 0; j < k; j++) a2 += a[6 * i + j] * temp[j];
          a[6 * i + k] -= a2;
        }
      }
      const double akk = a[7 * k];
      const bool valid = fabs(akk) > 0.0;
      if (k == 0 && !valid) {
        alive = false;
      } else if (valid) {
#pragma unroll
        for (int i = k + 1; i < 6; i++) a[6 * i + k] /= akk;
      }
    }
  }
}

SDVL_HD void ldlt_apply6_reg(const double *a, const int *tr, const double *bin, double *x) {
  double d[6];
#pragma unroll
  for (int i = 0; i < 6; i++) d[i] = bin[i];
#pragma unroll
  for (int k = 0; k < 6; k++) {
#pragma unroll
    for (int c = k + 1; c < 6; c++)
      if (tr[k] == c) { const double t = d[k]; d[k] = d[c]; d[c] = t; }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double acc = d[i];
#pragma unroll
    for (int j = 0; j < i; j++) acc -= a[6 * i + j] * d[j];
    d[i] = acc;
  }
  const double tol = 1.0 / 1.7976931348623157e308;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    if (fabs(a[7 * i]) > tol) d[i] /= a[7 * i];
    else d[i] = 0.0;
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double acc = d[i];
#pragma unroll
    for (int j = i + 1; j < 6; j++) acc -= a[6 * j + i] * d[j];
    d[i] = acc;
  }
#pragma unroll
  for (int k = 5; k >= 0; k--) {
#pragma unroll
    for (int c = k + 1; c < 6; c++)
      if (tr[k] == c) { const double t = d[k]; d[k] = d[c]; d[c] = t; }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] = d[i];
}

template <bool kUniform = false>
SDVL_HD void ldlt_solve6_reg(const double *Ain, const double *bin, double *x) {
  double a[36];
  int tr[6];
  ldlt_factor6_reg<kUniform>(Ain, a, tr);
  ldlt_apply6_reg(a, tr, bin, x);
}
#endif  // __HIPCC__

}  // namespace sdvl

#endif  // SDVL_MATH_H_
