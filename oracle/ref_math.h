// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the SDVL front-end hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// PARITY UNPINNED: the reference has no tests / golden vectors and cannot be compiled here (needs
// OpenCV + Eigen + Pangolin, none present); OpenCV/Eigen pieces are restated from their published
// algorithms.  Independent numpy/scipy cross-checks live in tests/.
//
// ref_math.h — the Eigen / SE3 / Camera operations the path relies on, restated in plain C++.
//   Quaternion <-> matrix, Hamilton product, inverse:  Eigen 3.x Quaternion
//   SE3:        /root/reference/extra/se3.cc:28-177, extra/se3.h:32-78
//   Camera:     /root/reference/camera.cc:69-79, camera.h:93-116
//   Jacobian3DToPlane, AbsMax: /root/reference/extra/utils.cc:28-42, 99-118
//   LDLT 6x6 (pivoted, Eigen 3.3 unblocked algorithm): image_align.cc:102, feature_align.cc:402
// Compile with -ffp-contract=off: every float/double expression below is evaluated exactly as written.
#ifndef SDVL_ORACLE_REF_MATH_H_
#define SDVL_ORACLE_REF_MATH_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace sdvlref {

struct Vec2 { double x, y; };
struct Vec3 { double x, y, z; };
struct Mat3 { double m[3][3]; };
struct Mat2 { double a, b, c, d; };  // [a b; c d]

inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(const Vec3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(double s, const Vec3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline double Dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double Norm(const Vec3 &a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

inline Vec3 MatVec(const Mat3 &R, const Vec3 &v) {
  return {R.m[0][0] * v.x + R.m[0][1] * v.y + R.m[0][2] * v.z,
          R.m[1][0] * v.x + R.m[1][1] * v.y + R.m[1][2] * v.z,
          R.m[2][0] * v.x + R.m[2][1] * v.y + R.m[2][2] * v.z};
}

inline Mat3 MatMul(const Mat3 &A, const Mat3 &B) {
  Mat3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
  return C;
}

// Eigen::Quaterniond(w,x,y,z).toRotationMatrix()  
inline Mat3 QuatToMat(double w, double x, double y, double z) {
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Mat3 R;
  R.m[0][0] = 1.0 - (tyy + tzz); R.m[0][1] = txy - twz;         R.m[0][2] = txz + twy;
  R.m[1][0] = txy + twz;         R.m[1][1] = 1.0 - (txx + tzz); R.m[1][2] = tyz - twx;
  R.m[2][0] = txz - twy;         R.m[2][1] = tyz + twx;         R.m[2][2] = 1.0 - (txx + tyy);
  return R;
}

// SE3: rigid transform stored as quaternion (w,x,y,z) + translation, extra/se3.h:32-78
struct SE3 {
  double q0 = 1.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  Vec3 t{0.0, 0.0, 0.0};

  Mat3 Rotation() const { return QuatToMat(q0, q1, q2, q3); }

  // extra/se3.cc:59-70 ; Eigen q.inverse() = conjugate / squaredNorm
  SE3 Inverse() const {
    SE3 r;
    const double n2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
    if (n2 > 0.0) {
      r.q0 = q0 / n2; r.q1 = -q1 / n2; r.q2 = -q2 / n2; r.q3 = -q3 / n2;
    } else {
      r.q0 = 0.0; r.q1 = 0.0; r.q2 = 0.0; r.q3 = 0.0;
    }
    const Vec3 rt = MatVec(QuatToMat(r.q0, r.q1, r.q2, r.q3), t);
    r.t = {-rt.x, -rt.y, -rt.z};
    return r;
  }

  // extra/se3.h:68
  Vec3 operator*(const Vec3 &p) const { return MatVec(Rotation(), p) + t; }

  // extra/se3.cc:166-177 ; Hamilton product then normalize
  SE3 operator*(const SE3 &o) const {
    SE3 r;
    double w = q0 * o.q0 - q1 * o.q1 - q2 * o.q2 - q3 * o.q3;
    double x = q0 * o.q1 + q1 * o.q0 + q2 * o.q3 - q3 * o.q2;
    double y = q0 * o.q2 + q2 * o.q0 + q3 * o.q1 - q1 * o.q3;
    double z = q0 * o.q3 + q3 * o.q0 + q1 * o.q2 - q2 * o.q1;
    const double n = std::sqrt(w * w + x * x + y * y + z * z);
    r.q0 = w / n; r.q1 = x / n; r.q2 = y / n; r.q3 = z / n;
    r.t = t + MatVec(Rotation(), o.t);
    return r;
  }
};

const double kSmallEps = 1e-10;  // extra/se3.h:30

// extra/se3.cc:72-94 (+ RotationExp :114-130, RotationHat :132-138)
inline SE3 SE3Exp(const double u[6]) {
  const Vec3 upsilon{u[0], u[1], u[2]};
  const Vec3 omega{u[3], u[4], u[5]};
  const double theta = Norm(omega);
  const double half_theta = 0.5 * theta;
  double imag_factor;
  const double real_factor = std::cos(half_theta);
  if (theta < kSmallEps) {
    const double theta_sq = theta * theta;
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    imag_factor = std::sin(half_theta) / theta;
  }
  SE3 r;
  r.q0 = real_factor; r.q1 = imag_factor * omega.x; r.q2 = imag_factor * omega.y; r.q3 = imag_factor * omega.z;

  Mat3 Om;
  Om.m[0][0] = 0;        Om.m[0][1] = -omega.z; Om.m[0][2] = omega.y;
  Om.m[1][0] = omega.z;  Om.m[1][1] = 0;        Om.m[1][2] = -omega.x;
  Om.m[2][0] = -omega.y; Om.m[2][1] = omega.x;  Om.m[2][2] = 0;
  const Mat3 Om2 = MatMul(Om, Om);
  Mat3 V;
  if (theta < kSmallEps) {
    V = QuatToMat(r.q0, r.q1, r.q2, r.q3);
  } else {
    const double theta_sq = theta * theta;
    const double ca = (1 - std::cos(theta)) / (theta_sq);
    const double cb = (theta - std::sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        V.m[i][j] = ((i == j ? 1.0 : 0.0) + ca * Om.m[i][j]) + cb * Om2.m[i][j];
  }
  r.t = MatVec(V, upsilon);
  return r;
}

// extra/se3.cc:96-112 (+ RotationLog :140-164, including the fall-through quirk at :152-159)
inline void SE3Log(const SE3 &s, double out[6]) {
  const double n = std::sqrt(s.q1 * s.q1 + s.q2 * s.q2 + s.q3 * s.q3);
  const double w = s.q0;
  const double squared_w = w * w;
  double two_atan_nbyw_by_n;
  if (n < kSmallEps) {
    two_atan_nbyw_by_n = 2. / w - 2. * (n * n) / (w * squared_w);
  } else {
    two_atan_nbyw_by_n = 2 * std::atan(n / w) / n;  // the |w|<eps branch value is overwritten (se3.cc:152-159)
  }
  const double theta = two_atan_nbyw_by_n * n;
  const Vec3 om{two_atan_nbyw_by_n * s.q1, two_atan_nbyw_by_n * s.q2, two_atan_nbyw_by_n * s.q3};
  Mat3 Om;
  Om.m[0][0] = 0;     Om.m[0][1] = -om.z; Om.m[0][2] = om.y;
  Om.m[1][0] = om.z;  Om.m[1][1] = 0;     Om.m[1][2] = -om.x;
  Om.m[2][0] = -om.y; Om.m[2][1] = om.x;  Om.m[2][2] = 0;
  const Mat3 Om2 = MatMul(Om, Om);
  Mat3 Vinv;
  double c2;
  if (theta < kSmallEps) c2 = (1. / 12.);
  else c2 = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Vinv.m[i][j] = ((i == j ? 1.0 : 0.0) - 0.5 * Om.m[i][j]) + c2 * Om2.m[i][j];
  const Vec3 up = MatVec(Vinv, s.t);
  out[0] = up.x; out[1] = up.y; out[2] = up.z; out[3] = om.x; out[4] = om.y; out[5] = om.z;
}

// Pinhole camera, camera.cc:69-79, camera.h:93-98
struct Camera {
  double width = 640, height = 480, fx = 300, fy = 300, u0 = 320, v0 = 240;

  Vec2 Project(const Vec3 &p) const { return {u0 + fx * p.x / p.z, v0 + fy * p.y / p.z}; }
  Vec3 Unproject(const Vec2 &p) const {
    Vec3 v{(p.x - u0) / fx, (p.y - v0) / fy, 1.0};
    const double n = Norm(v);
    return {v.x / n, v.y / n, v.z / n};
  }
  bool IsInsideImage(int px, int py, int m = 0) const {
    return px >= m && px < width - m && py >= m && py < height - m;
  }
  bool IsInsideImage(int px, int py, int m, int l) const {
    return px >= m && px < width / (1 << l) - m && py >= m && py < height / (1 << l) - m;
  }
};

// extra/utils.cc:99-118
inline void Jacobian3DToPlane(const Vec3 &p, double J[2][6]) {
  const double x = p.x, y = p.y;
  const double z_inv = 1. / p.z;
  const double z_inv_2 = z_inv * z_inv;
  J[0][0] = -z_inv;
  J[0][1] = 0.0;
  J[0][2] = x * z_inv_2;
  J[0][3] = y * J[0][2];
  J[0][4] = -(1.0 + x * J[0][2]);
  J[0][5] = y * z_inv;
  J[1][0] = 0.0;
  J[1][1] = -z_inv;
  J[1][2] = y * z_inv_2;
  J[1][3] = 1.0 + y * J[1][2];
  J[1][4] = -J[0][3];
  J[1][5] = -x * z_inv;
}

// extra/utils.cc:28-42
inline double AbsMax6(const double v[6]) {
  double max = -1;
  for (int i = 0; i < 6; i++) {
    const double a = std::fabs(v[i]);
    if (a > max) max = a;
  }
  return max;
}

// Eigen::LDLT<Matrix<double,6,6>>(A).solve(b): robust Cholesky with diagonal pivoting, lower triangle,
// unblocked in-place algorithm of Eigen 3.3 (ldlt_inplace<Lower>::unblocked) followed by
// P^T L^-T D^-1 L^-1 P b, with D entries at or below 1/highest() treated as zero (Appendix C).
inline void LdltSolve6(const double Ain[6][6], const double bin[6], double x[6]) {
  const int n = 6;
  double mat[6][6];
  int tr[6];
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) mat[i][j] = Ain[i][j];
  bool zero_diag = false;
  for (int k = 0; k < n; k++) {
    int idx = k;
    double big = std::fabs(mat[k][k]);
    for (int i = k + 1; i < n; i++) {
      const double v = std::fabs(mat[i][i]);
      if (v > big) { big = v; idx = i; }
    }
    tr[k] = idx;
    if (k != idx) {
      const int s = n - idx - 1;
      for (int j = 0; j < k; j++) { double tmp = mat[k][j]; mat[k][j] = mat[idx][j]; mat[idx][j] = tmp; }
      for (int i = 0; i < s; i++) {
        double tmp = mat[idx + 1 + i][k]; mat[idx + 1 + i][k] = mat[idx + 1 + i][idx]; mat[idx + 1 + i][idx] = tmp;
      }
      { double tmp = mat[k][k]; mat[k][k] = mat[idx][idx]; mat[idx][idx] = tmp; }
      for (int i = k + 1; i < idx; i++) { double tmp = mat[i][k]; mat[i][k] = mat[idx][i]; mat[idx][i] = tmp; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      double temp[6];
      for (int j = 0; j < k; j++) temp[j] = mat[j][j] * mat[k][j];
      double acc = 0.0;
      for (int j = 0; j < k; j++) acc += mat[k][j] * temp[j];
      mat[k][k] -= acc;
      for (int i = 0; i < rs; i++) {
        double a2 = 0.0;
        for (int j = 0; j < k; j++) a2 += mat[k + 1 + i][j] * temp[j];
        mat[k + 1 + i][k] -= a2;
      }
    }
    const double akk = mat[k][k];
    const bool pivot_valid = std::fabs(akk) > 0.0;
    if (k == 0 && !pivot_valid) {
      for (int j = 0; j < n; j++) tr[j] = j;
      zero_diag = true;
      break;
    }
    if (rs > 0 && pivot_valid)
      for (int i = 0; i < rs; i++) mat[k + 1 + i][k] /= akk;
  }
  (void)zero_diag;
  double d[6];
  for (int i = 0; i < n; i++) d[i] = bin[i];
  for (int k = 0; k < n; k++) if (tr[k] != k) { double tmp = d[k]; d[k] = d[tr[k]]; d[tr[k]] = tmp; }
  for (int i = 0; i < n; i++) {            // L^-1 (unit lower)
    double acc = d[i];
    for (int j = 0; j < i; j++) acc -= mat[i][j] * d[j];
    d[i] = acc;
  }
  const double tol = 1.0 / std::numeric_limits<double>::max();
  for (int i = 0; i < n; i++) {            // D^-1 (pseudo-inverse)
    if (std::fabs(mat[i][i]) > tol) d[i] /= mat[i][i];
    else d[i] = 0.0;
  }
  for (int i = n - 1; i >= 0; i--) {       // L^-T (unit upper)
    double acc = d[i];
    for (int j = i + 1; j < n; j++) acc -= mat[j][i] * d[j];
    d[i] = acc;
  }
  for (int k = n - 1; k >= 0; k--) if (tr[k] != k) { double tmp = d[k]; d[k] = d[tr[k]]; d[tr[k]] = tmp; }
  for (int i = 0; i < n; i++) x[i] = d[i];
}

}  // namespace sdvlref

#endif  // SDVL_ORACLE_REF_MATH_H_
