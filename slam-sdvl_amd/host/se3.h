// se3.h — sdvl::SE3 with the reference's interface (extra/se3.h:32-78) on top of csrc/sdvl_math.h.
#ifndef SDVL_HOST_SE3_H_
#define SDVL_HOST_SE3_H_

#include "../csrc/sdvl_math.h"
#include "types.h"

namespace sdvl {

class SE3 {  // named SE3 internally; `SE3` in sdvl_math.h is the POD the kernels share
 public:
  SE3() : s_(se3_identity()) {}
  explicit SE3(const Rigid &s) : s_(s) {}
  static SE3 FromArray(const double *p) { return SE3(se3_from7(p)); }
  void ToArray(double *p) const { se3_to7(s_, p); }
  const Rigid &pod() const { return s_; }

  Vector3d GetTranslation() const { return Vector3d(s_.t.x, s_.t.y, s_.t.z); }
  M3 GetRotation() const { return se3_rot(s_); }
  void Restart() { s_ = se3_identity(); }
  SE3 Inverse() const { return SE3(se3_inverse(s_)); }
  static SE3 Exp(const Vector6d &u) { return SE3(se3_exp(u.v)); }
  static Vector6d Log(const SE3 &s) { Vector6d r; se3_log(s.s_, r.v); return r; }
  Vector3d operator*(const Vector3d &p) const {
    const V3 r = se3_apply(s_, {p(0), p(1), p(2)});
    return Vector3d(r.x, r.y, r.z);
  }
  SE3 operator*(const SE3 &o) const { return SE3(se3_mul(s_, o.s_)); }

 private:
  Rigid s_;
};

}  // namespace sdvl

#endif  // SDVL_HOST_SE3_H_
