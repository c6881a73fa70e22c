// minimal_deps.cc — a SECOND, independent implementation of what frontend.cc leaves undefined (frontend_deps.h): the six members of
// the reference's Camera and Point that the hot path's classes call out to, plus the constructors a program needs to make the
// objects.  It shares no code with standalone.cc; `make frontend_link_check` links frontend.o against THIS file only — no
// standalone.cc, no mapper.cc, no capi.cc — which proves that the front end has no hidden dependency on the rest of the host layer
// (INTEGRATION.md route A: in the reference's tree camera.cc and point.cc take this file's place).
//   Camera::Project / Unproject     camera.cc:69-79
//   Point::GetPosition              point.cc:128-142
//   Point::GetStd                   point.h:60
//   Point::Promote / Unpromote      point.cc:102-118
#include <cmath>

#include "frontend.h"

namespace sdvl {

Camera::Camera(int width, int height, double fx, double fy, double u0, double v0) {
  width_ = width;
  height_ = height;
  fx_ = fx;
  fy_ = fy;
  u0_ = u0;
  v0_ = v0;
}

void Camera::Project(const Vector3d &p, Vector2d *out) const {
  const double inv_z = p(2);
  (*out)(0) = u0_ + fx_ * p(0) / inv_z;
  (*out)(1) = v0_ + fy_ * p(1) / inv_z;
}

void Camera::Unproject(const Vector2d &px, Vector3d *ray) const {
  double v[3] = {(px(0) - u0_) / fx_, (px(1) - v0_) / fy_, 1.0};
  const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  for (int i = 0; i < 3; i++) (*ray)(i) = v[i] / len;
}

Point::Point() {
  static int next_id = 0;
  id_ = next_id++;
  status_ = P_NOT_FOUND;
  delete_ = false;
  fixed_ = false;
  last_frame_ = -1;
  n_successful_ = 0;
  n_failed_ = 0;
  rho_ = 1.0;
  sigma2_ = 1.0;
  a_ = b_ = 10;
  z_range_ = 6.0;
}

void Point::InitFixed(const std::shared_ptr<Feature> &f, double depth, double sigma2, const Vector3d &p3d) {
  feature_ = f;
  rho_ = 1.0 / depth;
  sigma2_ = sigma2;
  p3d_ = p3d;
  fixed_ = true;
}

double Point::GetStd() { return std::sqrt(sigma2_); }

Vector3d Point::GetPosition() const {
  if (fixed_) return p3d_;
  // a candidate lies on the ray of its first observation at depth 1 / rho (point.cc:133-141)
  const Vector3d &ray = feature_->GetVector();
  const double depth = 1.0 / rho_;
  return feature_->GetFrameRaw()->GetWorldPose() * Vector3d(depth * ray(0), depth * ray(1), depth * ray(2));
}

bool Point::Promote() {
  n_failed_ = 0;
  n_successful_ += 1;
  return true;
}

bool Point::Unpromote() {
  n_failed_ += 1;
  b_ += 1;
  return n_failed_ > Config::MaxFailed();
}

// The check program constructs a Map (feature_align.h:46 takes one): its out-of-line members, map.cc's part.  frontend.cc itself
// only calls Map::DeletePoint, which is inline.
void Map::AddKeyframe(const std::shared_ptr<Frame> &frame, bool) {
  keyframes_.push_back(frame);
  last_kf_ = frame;
}
void Map::EmptyTrash() { points_trash_.clear(); }
bool Map::NeedKeyframe(const std::shared_ptr<Frame> &, int) { return false; }

}  // namespace sdvl
