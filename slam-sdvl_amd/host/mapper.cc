// mapper.cc — the reference's mapper (map.cc, point.cc depth filter) on top of the batched K7 search kernel: SURVEY §8f
// row 3.  Sequential mode only (main.cc:148-149); bundle adjustment (extra/bundle.cc) is out of scope and not run.
// Every function cites the reference lines it restates; arithmetic keeps the reference's expression order (the file is
// built with -ffp-contract=off like the rest of the host layer).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "config.h"
#include "sdvl_host.h"

#include <chrono>
#include <thread>

namespace sdvl {

using std::shared_ptr;
using std::vector;

namespace {

// extra/utils.cc:193-205: A = [R v_ref | v_cur], depth2 = -(A^T A)^-1 A^T t, |depth2[0]|
bool GetDepthFromTriangulation(const SE3 &pose, const Vector3d &v_ref, const Vector3d &v_cur, double *depth) {
  const M3 R = pose.GetRotation();
  const V3 a0 = mvec(R, {v_ref(0), v_ref(1), v_ref(2)});
  const V3 a1 = {v_cur(0), v_cur(1), v_cur(2)};
  const double m00 = vdot(a0, a0), m01 = vdot(a0, a1), m11 = vdot(a1, a1);
  const double det = m00 * m11 - m01 * m01;
  if (det < 0.000001) return false;
  const double invdet = 1.0 / det;
  const double i00 = m11 * invdet, i01 = -m01 * invdet;
  const double n00 = -i00, n01 = -i01;
  const double r0x = n00 * a0.x + n01 * a1.x, r0y = n00 * a0.y + n01 * a1.y, r0z = n00 * a0.z + n01 * a1.z;
  const Vector3d t = pose.GetTranslation();
  const double d0 = r0x * t(0) + r0y * t(1) + r0z * t(2);
  *depth = std::fabs(d0);
  return true;
}

// extra/utils.cc:207-213
double GetParallax(const Vector3d &src1, const Vector3d &src2, const Vector3d &p3d) {
  V3 v1 = {src1(0) - p3d(0), src1(1) - p3d(1), src1(2) - p3d(2)}, v2 = {src2(0) - p3d(0), src2(1) - p3d(1), src2(2) - p3d(2)};
  const double n1 = vnorm(v1), n2 = vnorm(v2);
  v1 = {v1.x / n1, v1.y / n1, v1.z / n1};
  v2 = {v2.x / n2, v2.y / n2, v2.z / n2};
  return vdot(v1, v2);
}

void FillRequest(sdvl_search_req *rq, Frame *cur, Frame *ref, const Vector2d &px, const Vector3d &bearing, int level, const uchar *desc,
                 double idepth, double idepth_std, bool fixed, const Vector2d &px0) {
  rq->cur = cur->device();
  rq->ref = ref->device();
  cur->GetPose().ToArray(rq->cur_pose);
  ref->GetPose().ToArray(rq->ref_pose);
  rq->px[0] = px(0); rq->px[1] = px(1);
  rq->bearing[0] = bearing(0); rq->bearing[1] = bearing(1); rq->bearing[2] = bearing(2);
  rq->idepth = idepth;
  rq->idepth_std = idepth_std;
  rq->px0[0] = px0(0); rq->px0[1] = px0(1);
  rq->level = level;
  rq->fixed = fixed ? 1 : 0;
  if (desc) std::memcpy(rq->desc, desc, 32);
  else std::memset(rq->desc, 0, 32);
}

void FillRequestFromFeature(sdvl_search_req *rq, Frame *cur, Feature *feature, Frame *ref, double idepth, double idepth_std, bool fixed,
                            const Vector2d &px0) {
  FillRequest(rq, cur, ref, feature->GetPosition(), feature->GetVector(), feature->GetLevel(),
              feature->HasDescriptor() ? feature->DescriptorData().data() : nullptr, idepth, idepth_std, fixed, px0);
}

}  // namespace

// ------------------------------------------------------------------------------------------------- Point (depth filter)
// point.cc:189-201
double Point::ComputeTau(const SE3 &pose, const Vector3d &v, double depth, double px_error_angle) {
  const double PI = 3.14159265;
  const Vector3d t = pose.GetTranslation();
  const V3 tt = {t(0), t(1), t(2)};
  const V3 a = {v(0) * depth - t(0), v(1) * depth - t(1), v(2) * depth - t(2)};
  const double t_norm = vnorm(tt), a_norm = vnorm(a);
  const double alpha = std::acos((v(0) * t(0) + v(1) * t(1) + v(2) * t(2)) / t_norm);
  const double beta = std::acos((a.x * -t(0) + a.y * -t(1) + a.z * -t(2)) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = PI - alpha - beta_plus;
  const double depth_plus = t_norm * std::sin(beta_plus) / std::sin(gamma_plus);
  return depth_plus - depth;
}

// point.cc:203-217
double Point::PDFNormal(double mean, double sd, double x) {
  const double PI = 3.14159265;
  double result = 0.0;
  if (sd <= 0) return result;
  double exponent = x - mean;
  exponent *= -exponent;
  exponent /= 2 * sd * sd;
  result = std::exp(exponent);
  result /= sd * std::sqrt(2.0 * PI);
  return result;
}

// point.cc:64-100
void Point::Update(const shared_ptr<Frame> &frame, double depth, double px_error_angle) {
  Frame *f0 = feature_->GetFrameRaw();
  const SE3 pose = f0->GetPose() * frame->GetPose().Inverse();
  const double tau = ComputeTau(pose, feature_->GetVector(), depth, px_error_angle);
  const double tau_inverse = 0.5 * (1.0 / std::max(0.0000001, depth - tau) - 1.0 / (depth + tau));
  const double tau2 = tau_inverse * tau_inverse;
  const double x = 1. / depth;
  const double norm_scale = std::sqrt(sigma2_ + tau2);
  if (std::isnan(norm_scale)) return;
  const double s2 = 1. / (1. / sigma2_ + 1. / tau2);
  const double m = s2 * (rho_ / sigma2_ + x / tau2);
  double C1 = a_ / (a_ + b_) * PDFNormal(rho_, norm_scale, x);
  double C2 = b_ / (a_ + b_) * 1. / z_range_;
  const double normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  const double f = C1 * (a_ + 1.) / (a_ + b_ + 1.) + C2 * a_ / (a_ + b_ + 1.);
  const double e = C1 * (a_ + 1.) * (a_ + 2.) / ((a_ + b_ + 1.) * (a_ + b_ + 2.)) + C2 * a_ * (a_ + 1.0) / ((a_ + b_ + 1.0) * (a_ + b_ + 2.0));
  const double rho_new = C1 * m + C2 * rho_;
  sigma2_ = C1 * (s2 + m * m) + C2 * (sigma2_ + rho_ * rho_) - rho_new * rho_new;
  rho_ = rho_new;
  a_ = (e - f) / (f - e / f);
  b_ = a_ * (1.0 - f) / f;
  const Vector3d pos = GetPosition();
  cos_alpha_ = GetParallax(f0->GetWorldPosition(), frame->GetWorldPosition(), pos);
  last_distance_ = frame->DistanceTo(pos);
  n_failed_ = 0;
}

// point.cc:164-178
bool Point::HasConverged() {
  if (fixed_) return true;
  const double std_d = std::sqrt(sigma2_) / (rho_ * rho_);
  const double l = 4 * std_d * cos_alpha_ / last_distance_;
  if (l < 0.1) {
    p3d_ = GetPosition();
    fixed_ = true;
    return true;
  }
  return false;
}

// point.cc:180-187
bool Point::SeenFrom(const shared_ptr<Frame> &frame) const {
  for (auto it = features_.begin(); it != features_.end(); it++) {
    Frame *f = (*it)->GetFrameRaw();
    if (f && frame->GetID() == f->GetID()) return true;
  }
  return false;
}

// ----------------------------------------------------------------------------------------------------- Frame (map side)
// frame.cc:70-92
double Frame::GetSceneDepth() {
  if (scene_depth_hint_valid_) return scene_depth_hint_;
  vector<double> depth_vec;
  if (flat_) {  // features still flat records of the tracking tables: same points in the same order, no objects needed
    for (const sdvl_track_feature_out &f : flat_->feats)
      if (f.point >= 0) depth_vec.push_back(GetRelativePos((*flat_->points)[f.point]->GetPosition())(2));
  }
  for (auto it = features_.begin(); it != features_.end(); it++) {
    if (!(*it)) continue;
    Point *point = (*it)->GetPointRaw();
    if (!point) continue;
    depth_vec.push_back(GetRelativePos(point->GetPosition())(2));
  }
  if (depth_vec.empty()) return 0.0;
  auto mid = depth_vec.begin() + static_cast<long>(std::floor(depth_vec.size() / 2));  // GetMedianVector, extra/utils.cc:215-220
  std::nth_element(depth_vec.begin(), mid, depth_vec.end());
  return *mid;
}

// frame.cc:105-113
bool Frame::IsPointVisible(const Vector3d &p) {
  const Vector3d rel_p = GetRelativePos(p);
  if (rel_p(2) < 0.0) return false;
  Vector2d image_p;
  camera_->Project(rel_p, &image_p);
  return camera_->IsInsideImage(Vector2i(static_cast<int>(image_p(0)), static_cast<int>(image_p(1))));
}

// frame.h:129-136
double Frame::DistanceTo(const Frame &frame) const {
  const Vector3d a = GetWorldPosition(), b = frame.GetWorldPosition();
  return vnorm({a(0) - b(0), a(1) - b(1), a(2) - b(2)});
}
double Frame::DistanceTo(const Vector3d &p) const {
  const Vector3d a = GetWorldPosition();
  return vnorm({a(0) - p(0), a(1) - p(1), a(2) - p(2)});
}

// frame.cc:185-207
void Frame::GetBestConnections(vector<shared_ptr<Frame>> *connections, int n) {
  int count = 0;
  connections->clear();
  if (n == 0 || n >= static_cast<int>(connections_.size())) {
    for (auto it = connections_.begin(); it != connections_.end(); it++)
      if (!it->first->ToDelete()) connections->push_back(it->first);
  } else {
    std::sort(connections_.begin(), connections_.end(),
              [](const std::pair<shared_ptr<Frame>, int> &l, const std::pair<shared_ptr<Frame>, int> &r) { return l.second > r.second; });
    for (auto it = connections_.begin(); it != connections_.end() && count < n; it++)
      if (!it->first->ToDelete()) {
        connections->push_back(it->first);
        count++;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------- MapperMap
// map.cc:143-158
void MapperMap::AddKeyframe(const shared_ptr<Frame> &frame, bool search) {
  if (search) keyframe_queue_.push_back(frame);
  else initial_kf_id_ = std::max(initial_kf_id_, frame->GetID());
  num_kfs_++;
  version_++;
  frame->SetKeyframeID(num_kfs_);
  keyframes_.push_back(frame);
  retired_.push_back(frame);
  last_kf_ = frame;
}

// map.cc:190-205,692-706
void MapperMap::LimitKeyframes(const shared_ptr<Frame> &frame) {
  if (static_cast<int>(keyframes_.size()) < Config::MaxKeyframes()) return;
  const Vector3d pos = frame->GetWorldPosition();
  shared_ptr<Frame> kf;
  double maxdist = 0.0;
  for (auto it = keyframes_.begin(); it != keyframes_.end(); it++) {
    const Vector3d q = (*it)->GetWorldPosition();
    const double dist = vnorm({q(0) - pos(0), q(1) - pos(1), q(2) - pos(2)});
    if (dist > maxdist) {
      maxdist = dist;
      kf = *it;
    }
  }
  if (!kf) return;
  kf->SetDelete();
  frame_trash_.push_back(kf);
}

// map.cc:207-259
void MapperMap::EmptyTrash() {
  if (!frame_trash_.empty()) version_++;
  for (auto fit = frame_trash_.begin(); fit != frame_trash_.end(); fit++) {
    if ((*fit)->IsKeyframe())
      for (auto it = keyframes_.begin(); it != keyframes_.end(); it++)
        if (*it == *fit) {
          keyframes_.erase(it);
          break;
        }
    (*fit)->RemoveFeatures();
    (*fit)->SetDelete();
  }
  frame_trash_.clear();
  Map::EmptyTrash();
}

MapperMap::Stats MapperMap::GetStats() const {
  Stats s = stats_;
  s.candidates = static_cast<int>(candidates_.size());
  s.keyframes = static_cast<int>(keyframes_.size());
  return s;
}

// head of Map::UpdateMap, map.cc:75-108
bool MapperMap::BeginUpdate() {
  cur_.reset();
  if (relocalizing_) return false;
  if (frame_queue_.empty() && keyframe_queue_.empty()) return false;
  if (!keyframe_queue_.empty()) {
    while (!frame_queue_.empty()) {
      frame_trash_.push_back(frame_queue_.front());
      frame_queue_.pop_front();
    }
    cur_ = keyframe_queue_.front();
    keyframe_queue_.pop_front();
  } else {
    cur_ = frame_queue_.front();
    frame_queue_.pop_front();
  }
  updates_++;
  version_++;  // the depth filter moves candidates, connections add features to keyframes
  depth_mean_ = cur_->GetSceneDepth();
  pass_ = 0;
  occurrence_.assign(candidates_.size(), 0);
  // the reference pushes every new candidate twice (map.cc:381,389): entry k is the occurrence_[k]-th of its point.  The two
  // entries are pushed back to back and erasures keep the order, so the entries of a point stay neighbours.
  for (size_t k = 1; k < candidates_.size(); k++)
    if (candidates_[k] == candidates_[k - 1]) occurrence_[k] = occurrence_[k - 1] + 1;
  return true;
}

// Map::UpdateCandidates, map.cc:402-498, one pass = the `pass_`-th occurrence of every point in the list.  Points do not
// interact, so processing the list occurrence by occurrence leaves every point in the state the sequential loop does.
static int g_device_filter = -1;  // -1: not decided yet (environment)
void MapperMap::SetDeviceFilter(bool on) { g_device_filter = on ? 1 : 0; }
bool MapperMap::DeviceFilter() {
  if (g_device_filter < 0) {
    const char *e = std::getenv("SDVL_HOST_DEPTH_FILTER");
    g_device_filter = (e && e[0] == '1') ? 0 : 1;
  }
  return g_device_filter != 0;
}

sdvl_depth_params MapperMap::FilterParams() const {
  sdvl_depth_params fp;
  fp.px_error_angle = std::atan(1.0 / (2.0 * camera_->GetFx())) * 2.0;  // Camera::GetPixelErrorAngle, camera.h:104-107
  fp.min_depth = Config::MapScale() * Config::ScaleMinDist();
  fp.scale_min_dist = Config::ScaleMinDist();
  fp.max_failed = Config::MaxFailed();
  fp.pad_ = 0;
  return fp;
}

bool MapperMap::EmitCandidates(vector<sdvl_search_req> *reqs, vector<sdvl_depth_state> *states) {
  cand_work_.clear();
  if (!cur_) return false;
  req_base_ = static_cast<int>(reqs->size());
  bool any = false;
  const size_t n_cand = candidates_.size();
  const Rigid &cp = cur_->GetPose().pod();
  const M3 cur_R = se3_rot(cp);
  const V3 cur_t = cp.t;
  const V3 cur_w = cur_->GetWorldPose().pod().t;
  const Frame *ref_cached = nullptr;
  M3 ref_R = cur_R;
  V3 ref_t = cur_t;
  for (size_t k = 0; k < n_cand; k++) {
    // the loop walks Point -> first Feature -> its keyframe, three dependent loads per candidate scattered over the heap:
    // ask for them a few candidates ahead
    if (k + 8 < n_cand) __builtin_prefetch(candidates_[k + 8].get());
    if (k + 4 < n_cand) {
      const Feature *f4 = candidates_[k + 4]->GetInitFeatureRaw();
      __builtin_prefetch(f4);
      __builtin_prefetch(reinterpret_cast<const char *>(f4) + 64);
    }
    if (occurrence_[k] != pass_) continue;
    any = true;
    Point *point = candidates_[k].get();
    CandWork w{k, -1, kDeleted};
    // the checks that precede SearchPoint (map.cc:421-452): decided here, acted upon in ApplyCandidates — nothing the
    // decision depends on can change in between (a point appears once per pass)
    if (!point->ToDelete()) {
      Feature *feature = point->GetInitFeatureRaw();
      Frame *f0 = feature->GetFrameRaw();
      // Point::GetPosition (point.cc:128-142) with the rotation matrix of the first observation's keyframe kept from the
      // previous candidate (neighbouring candidates share their keyframe), Frame::IsPointVisible (frame.cc:105-113) with the
      // current frame's: the same statements (se3_apply = R p + t), the quaternion -> matrix conversion once instead of twice
      // per candidate
      if (f0 != ref_cached) {
        const Rigid &wp = f0->GetWorldPose().pod();
        ref_R = se3_rot(wp);
        ref_t = wp.t;
        ref_cached = f0;
      }
      V3 pos;
      if (point->IsFixed()) {
        const Vector3d p = point->GetPosition();
        pos = {p(0), p(1), p(2)};
      } else {
        const Vector3d &v = feature->GetVector();
        const double sc = 1.0 / point->GetInverseDepth();
        pos = vadd(mvec(ref_R, {sc * v(0), sc * v(1), sc * v(2)}), ref_t);
      }
      bool visible = false;
      {
        const V3 rel = vadd(mvec(cur_R, pos), cur_t);
        if (!(rel.z < 0.0)) {
          Vector2d image_p;
          camera_->Project(Vector3d(rel.x, rel.y, rel.z), &image_p);
          visible = camera_->IsInsideImage(Vector2i(static_cast<int>(image_p(0)), static_cast<int>(image_p(1))));
        }
      }
      if (!visible) {
        w.state = kInvisible;
      } else {
        const V3 dw = {cur_w.x - ref_t.x, cur_w.y - ref_t.y, cur_w.z - ref_t.z};  // Frame::DistanceTo(frame), frame.h:129-131
        const double distance = vnorm(dw);
        if (distance / depth_mean_ < 0.01) {
          w.state = kTooClose;
        } else {
          w.state = kSearch;
          reqs->emplace_back();
          FillRequestFromFeature(&reqs->back(), cur_.get(), feature, f0, point->GetInverseDepth(), point->GetStd(), false, Vector2d(0, 0));
          w.req = static_cast<int>(reqs->size()) - 1 - req_base_;
          if (states) {
            states->emplace_back();
            point->GetFilterState(&states->back());
            states->back().depth_mean = depth_mean_;
          }
        }
      }
    }
    cand_work_.push_back(w);
  }
  return any;
}

void MapperMap::ApplyCandidates(const sdvl_search_res *res_all, const sdvl_depth_out *fout_all) {
  if (!cur_) return;
  const sdvl_search_res *res = res_all + req_base_;
  const sdvl_depth_out *fout = fout_all ? fout_all + req_base_ : nullptr;
  const double px_error_angle = std::atan(1.0 / (2.0 * camera_->GetFx())) * 2.0;  // Camera::GetPixelErrorAngle, camera.h:104-107
  const int min_kf_id = last_kf_->GetKeyframeID() - 2 * Config::MaxSearchKeyframes();
  vector<char> erase(candidates_.size(), 0);
  for (size_t wi = 0; wi < cand_work_.size(); wi++) {
    const CandWork &w = cand_work_[wi];
    if (wi + 6 < cand_work_.size()) __builtin_prefetch(candidates_[cand_work_[wi + 6].index].get());
    const shared_ptr<Point> &point = candidates_[w.index];
    if (w.state == kDeleted) {
      DeletePoint(point);
      erase[w.index] = 1;
      continue;
    }
    if (w.state == kInvisible) {
      Frame *lf = point->GetLastFeature()->GetFrameRaw();
      if (lf->GetKeyframeID() < min_kf_id) {
        DeletePoint(point);
        erase[w.index] = 1;
      }
      continue;
    }
    if (w.state == kTooClose) continue;
    Feature *feature = point->GetInitFeatureRaw();
    Frame *f0 = feature->GetFrameRaw();
    const sdvl_search_res &r = res[w.req];
    if (fout) {  // the loop body ran on the device (depth_filter_kernel): book what it decided
      const sdvl_depth_out &o = fout[w.req];
      point->ApplyFilterOut(o);
      if (o.outcome & SDVL_DEPTH_DELETED) {
        if (point->TrackRow() >= 0) point->SetDeviceTrashed();  // its table row carries the deletion already
        DeletePoint(point);
      } else if ((o.outcome & 0xFF) == SDVL_DEPTH_CONVERGED || (o.outcome & 0xFF) == SDVL_DEPTH_FIXED_STALE) {
        erase[w.index] = 1;
        stats_.converged++;
      }
      continue;
    }
    if (!r.found) {
      if (point->Unpromote()) DeletePoint(point);
      continue;
    }
    const Vector2d imgpos(r.px[0], r.px[1]);
    const SE3 pose = cur_->GetPose() * f0->GetPose().Inverse();
    const Vector3d v3d = camera_->Unproject(imgpos);
    double depth = 0.0;
    if (!GetDepthFromTriangulation(pose, feature->GetVector(), v3d, &depth)) continue;
    const Vector3d &fv = feature->GetVector();
    const Vector3d p3d = f0->GetWorldPose() * Vector3d(depth * fv(0), depth * fv(1), depth * fv(2));
    const double cos_alpha = GetParallax(f0->GetWorldPosition(), cur_->GetWorldPosition(), p3d);
    if (cos_alpha >= 0.999999) continue;
    if (depth < Config::MapScale() * Config::ScaleMinDist() || depth < depth_mean_ * Config::ScaleMinDist()) continue;
    point->Update(cur_, depth, px_error_angle);
    if (point->HasConverged()) {
      erase[w.index] = 1;
      stats_.converged++;
    }
  }
  // erasures keep the order of the survivors; occurrence numbers are unchanged for them
  size_t o = 0;
  for (size_t k = 0; k < candidates_.size(); k++)
    if (!erase[k]) {
      candidates_[o] = candidates_[k];
      occurrence_[o] = occurrence_[k];
      o++;
    }
  candidates_.resize(o);
  occurrence_.resize(o);
  pass_++;
}

// Map::CheckConnections, map.cc:500-558.  The reference iterates a std::map keyed by frame POINTER; frozen to frame id.
void MapperMap::CheckConnections() {
  std::map<int, std::pair<shared_ptr<Frame>, int>> kfs;
  vector<shared_ptr<Feature>> &features = cur_->GetFeatures();
  for (auto it = features.begin(); it != features.end(); it++) {
    Point *point = (*it)->GetPointRaw();
    if (!point || point->ToDelete()) continue;
    std::list<shared_ptr<Feature>> &ffeatures = point->GetFeatures();
    for (auto fit = ffeatures.begin(); fit != ffeatures.end(); fit++) {
      Frame *ff = (*fit)->GetFrameRaw();
      if (ff->ToDelete()) continue;
      if (ff->GetID() == cur_->GetID()) continue;
      auto &e = kfs[ff->GetID()];
      if (!e.first) e.first = (*fit)->GetFrame();
      e.second++;
    }
  }
  if (kfs.empty()) return;
  int best_n = 0;
  bool saved = false;
  shared_ptr<Frame> best_kf;
  const int min_connections = Config::MinMatches() / 2;
  for (auto it = kfs.begin(); it != kfs.end(); it++) {
    const shared_ptr<Frame> &kf = it->second.first;
    const int n = it->second.second;
    if (n > best_n) {
      best_n = n;
      best_kf = kf;
    }
    if (n >= min_connections) {
      cur_->AddConnection(std::make_pair(kf, n));
      kf->AddConnection(std::make_pair(cur_, n));
      saved = true;
      stats_.connected++;
    }
  }
  if (!saved && best_n > 0) {
    cur_->AddConnection(std::make_pair(best_kf, best_n));
    best_kf->AddConnection(std::make_pair(cur_, best_n));
    stats_.connected++;
  }
}

// Map::AddConnectionsPoints, map.cc:560-617.  std::set keyed by point POINTER there; frozen to point id.
void MapperMap::EmitConnectionsPoints(vector<sdvl_search_req> *reqs) {
  acp_work_.clear();
  req_base_ = static_cast<int>(reqs->size());
  vector<shared_ptr<Frame>> best_kfs;
  cur_->GetBestConnections(&best_kfs, Config::MaxSearchKeyframes());
  if (best_kfs.empty()) return;
  // the reference collects into a std::set keyed by pointer; here: by point id, unique (see the class comment)
  vector<std::pair<int, shared_ptr<Point>>> ids;
  // Point::SeenFrom(cur_) (point.cc:180-187) walks the point's feature list; the points cur_ sees are exactly those behind its
  // own features, so they are stamped once and the test below is one load
  const int cur_id = cur_->GetID();
  {
    vector<shared_ptr<Feature>> &own = cur_->GetFeatures();
    for (auto it = own.begin(); it != own.end(); it++)
      if (*it)
        if (Point *p = (*it)->GetPointRaw()) p->SetSeenStamp(cur_id);
  }
  for (auto it_kf = best_kfs.begin(); it_kf != best_kfs.end(); it_kf++) {
    vector<shared_ptr<Feature>> &features = (*it_kf)->GetFeatures();
    for (auto it = features.begin(); it != features.end(); it++) {
      if (!*it) continue;
      Point *point = (*it)->GetPointRaw();
      if (!point || point->ToDelete()) continue;
      if (point->SeenStamp() == cur_id) continue;
      ids.push_back({point->GetID(), (*it)->GetPoint()});
    }
  }
  typedef std::pair<int, shared_ptr<Point>> IdPoint;
  std::sort(ids.begin(), ids.end(), [](const IdPoint &a, const IdPoint &b) { return a.first < b.first; });
  ids.erase(std::unique(ids.begin(), ids.end(), [](const IdPoint &a, const IdPoint &b) { return a.first == b.first; }), ids.end());
  for (auto it = ids.begin(); it != ids.end(); it++) {
    const shared_ptr<Point> &pt = it->second;
    shared_ptr<Feature> feature = pt->GetInitFeature();
    if (!feature) continue;
    Vector2d pos;
    if (!cur_->Project(pt->GetPosition(), &pos)) continue;
    if (!camera_->IsInsideImage(Vector2i(static_cast<int>(pos(0)), static_cast<int>(pos(1))), Config::PatchSize())) continue;
    reqs->emplace_back();
    FillRequestFromFeature(&reqs->back(), cur_.get(), feature.get(), feature->GetFrameRaw(), pt->GetInverseDepth(), pt->GetStd(), pt->IsFixed(), pos);
    acp_work_.push_back({pt, static_cast<int>(reqs->size()) - 1 - req_base_});
  }
}

void MapperMap::ApplyConnectionsPoints(const sdvl_search_res *res_all) {
  const sdvl_search_res *res = res_all + req_base_;
  for (auto &w : acp_work_) {
    const sdvl_search_res &r = res[w.second];
    if (!r.found) continue;
    shared_ptr<Feature> feature = std::make_shared<Feature>(cur_, Vector2d(r.px[0], r.px[1]), r.level);
    feature->SetPoint(w.first);
    cur_->AddFeature(feature);
    w.first->AddFeature(feature);
    stats_.linked++;
  }
  acp_work_.clear();
}

// Map::InitCandidates up to FilterCorners, map.cc:262-283
bool MapperMap::PrepareInitCandidates() {
  for (auto it = keyframes_.begin(); it != keyframes_.end(); it++) (*it)->SetSelected(false);  // ResetSelected
  cur_->SetSelected(true);
  best_kfs_.clear();
  cur_->GetBestConnections(&best_kfs_, Config::MaxSearchKeyframes());
  return !best_kfs_.empty();
}

// every (connected keyframe, filtered corner) pair the loops at map.cc:296-388 could visit
void MapperMap::EmitInitCandidates(vector<sdvl_search_req> *reqs) {
  ic_req_.clear();
  req_base_ = static_cast<int>(reqs->size());
  if (best_kfs_.empty()) return;
  depth_mean_ = cur_->GetSceneDepth();
  vector<int> &fcorners = cur_->GetFilteredCorners();
  const int nc = static_cast<int>(fcorners.size());
  ic_req_.assign(best_kfs_.size() * static_cast<size_t>(nc), -1);
  // what a request says about the corner is the same for every connected keyframe: position, bearing, level, descriptor
  // and the current frame's side are filled once per corner, the keyframe's side per pair
  ic_proto_.resize(nc);
  for (int c = 0; c < nc; c++) {
    const Vector3i corner = cur_->FilteredCorner(c);
    const int scale = (1 << corner(2));
    const Vector2d px(corner(0) * scale, corner(1) * scale);
    const Vector3d bearing = camera_->Unproject(px);
    FillRequest(&ic_proto_[c], cur_.get(), cur_.get(), px, bearing, corner(2), Config::UseORB() ? cur_->FilteredDescriptor(c) : nullptr,
                1.0 / depth_mean_, 1.0, false, Vector2d(0, 0));
  }
  for (size_t k = 0; k < best_kfs_.size(); k++) {
    Frame *cframe = best_kfs_[k].get();
    const double distance = cur_->DistanceTo(*cframe);
    if (distance / depth_mean_ < 0.01) continue;
    double cpose[7];
    cframe->GetPose().ToArray(cpose);
    const sdvl_frame *cdev = cframe->device();
    const size_t first = reqs->size();
    reqs->insert(reqs->end(), ic_proto_.begin(), ic_proto_.end());
    for (int c = 0; c < nc; c++) {
      sdvl_search_req &rq = (*reqs)[first + c];
      rq.cur = cdev;  // the search runs IN the connected keyframe, from the new keyframe's corner (map.cc:352)
      std::memcpy(rq.cur_pose, cpose, sizeof(cpose));
      ic_req_[k * nc + c] = static_cast<int>(first) + c - req_base_;
    }
  }
}

// the loops of Map::InitCandidates (map.cc:290-392) replayed over the search results
void MapperMap::ApplyInitCandidates(const sdvl_search_res *res_all) {
  if (best_kfs_.empty()) return;
  const sdvl_search_res *res = res_all + req_base_;
  vector<int> &fcorners = cur_->GetFilteredCorners();
  const int nc = static_cast<int>(fcorners.size());
  vector<bool> imatches(fcorners.size(), false);
  for (size_t k = 0; k < best_kfs_.size(); k++) {
    shared_ptr<Frame> cframe = best_kfs_[k];
    cframe->SetSelected(true);
    kf_scan_.clear();
    const double distance = cur_->DistanceTo(*cframe);
    if (distance / depth_mean_ < 0.01) continue;
    for (int count = 0; count < nc; count++) {
      if (imatches[count]) continue;
      const Vector3i corner = cur_->FilteredCorner(count);
      const int scale = (1 << corner(2));
      // map.cc:345-351 constructs the candidate point and its feature before the search; a miss throws them away again. Here
      // a miss only consumes the point id the constructor would have taken (ids stay those of the reference's order).
      const sdvl_search_res &r = res[ic_req_[k * nc + count]];
      if (!r.found) {
        Point::ConsumeId();
        continue;
      }
      shared_ptr<Point> candidate = std::make_shared<Point>();
      shared_ptr<Feature> feature = std::make_shared<Feature>(cur_, Vector2d(corner(0) * scale, corner(1) * scale), corner(2));
      if (Config::UseORB()) feature->SetDescriptor(cur_->FilteredDescriptor(count));
      const Vector2d imgpos(r.px[0], r.px[1]);
      const int level = r.level;
      bool mfound = false;
      vector<shared_ptr<Feature>> &features = cframe->GetFeatures();
      // positions and liveness of the keyframe's features in one contiguous array (built once per keyframe, extended when a
      // feature is added below): the scan runs for every found seed, and walking Feature -> Point each time is a cache miss apiece
      if (kf_scan_.empty() && !features.empty()) {
        kf_scan_.reserve(features.size() + 64);
        for (size_t q = 0; q < features.size(); q++) {
          const Feature *fq = features[q].get();
          const Point *praw = fq ? fq->GetPointRaw() : nullptr;
          KfScan e;
          e.x = fq ? fq->GetPosition()(0) : 0.0;
          e.y = fq ? fq->GetPosition()(1) : 0.0;
          e.live = praw && !praw->ToDelete();
          kf_scan_.push_back(e);
        }
      }
      for (size_t q = 0; q < features.size() && !mfound; q++) {  // index loop: AddFeature below may grow the vector
        if (!kf_scan_[q].live) continue;
        const double d1 = imgpos(0) - kf_scan_[q].x, d2 = imgpos(1) - kf_scan_[q].y;
        if (std::sqrt(d1 * d1 + d2 * d2) < 1.0) {  // Distance2D, extra/utils.cc:222-226
          shared_ptr<Point> point = features[q]->GetPoint();
          feature->SetPoint(point);
          cur_->AddFeature(feature);
          point->AddFeature(feature);
          mfound = true;
          stats_.linked++;
        }
      }
      if (mfound) continue;
      const SE3 pose = cframe->GetPose() * cur_->GetPose().Inverse();
      shared_ptr<Feature> feature2 = std::make_shared<Feature>(cframe, imgpos, level);
      double depth = 0.0;
      if (!GetDepthFromTriangulation(pose, feature->GetVector(), feature2->GetVector(), &depth)) continue;
      const Vector3d &fv = feature->GetVector();
      const Vector3d p3d = cur_->GetWorldPose() * Vector3d(depth * fv(0), depth * fv(1), depth * fv(2));
      const double cos_alpha = GetParallax(cur_->GetWorldPosition(), cframe->GetWorldPosition(), p3d);
      if (cos_alpha >= 0.999999) continue;
      if (depth < Config::MapScale() * Config::ScaleMinDist() || depth < depth_mean_ * Config::ScaleMinDist()) continue;
      candidate->InitCandidate(feature, depth);
      cur_->AddFeature(feature);
      candidate->AddFeature(feature);
      feature->SetPoint(candidate);
      cframe->AddFeature(feature2);
      kf_scan_.push_back(KfScan{feature2->GetPosition()(0), feature2->GetPosition()(1), true});
      candidate->AddFeature(feature2);
      feature2->SetPoint(candidate);
      imatches[count] = true;
      candidates_.push_back(candidate);
      candidates_.push_back(candidate);  // twice, as map.cc:381 and :389 do (`fixed` is false there)
      stats_.initialized++;
    }
  }
  best_kfs_.clear();
}

// Map::CheckRedundantKeyframes, map.cc:619-690
void MapperMap::CheckRedundantKeyframes() {
  const int min_features = 3;
  if (last_kf_checked_ == last_kf_->GetID()) return;
  last_kf_checked_ = last_kf_->GetID();
  vector<shared_ptr<Frame>> fov_kfs;
  last_kf_->GetBestConnections(&fov_kfs, 0);
  for (auto it = fov_kfs.begin(); it != fov_kfs.end(); it++) {
    shared_ptr<Frame> kf = *it;
    if (kf->ToDelete() || kf->GetID() <= initial_kf_id_) continue;
    int nredundant = 0, npoints = 0;
    vector<shared_ptr<Feature>> &features = kf->GetFeatures();
    for (auto it_fts = features.begin(); it_fts != features.end(); it_fts++) {
      if (!*it_fts) continue;
      Point *point = (*it_fts)->GetPointRaw();
      if (!point || point->ToDelete()) continue;
      npoints++;
      const int level1 = (*it_fts)->GetLevel();
      std::list<shared_ptr<Feature>> &ffeatures = point->GetFeatures();
      const int size = static_cast<int>(ffeatures.size());
      if (size > min_features) {
        int nmatches = 0;
        for (auto fit = ffeatures.begin(); fit != ffeatures.end(); fit++) {
          Frame *fkf = (*fit)->GetFrameRaw();
          if (fkf->ToDelete() || fkf->GetID() == kf->GetID()) continue;
          if ((*fit)->GetLevel() <= level1 + 1) {
            nmatches++;
            if (nmatches >= min_features) break;
          }
        }
        if (nmatches >= min_features) nredundant++;
      }
    }
    if (nredundant > 0.8 * npoints) {
      kf->SetDelete();
      frame_trash_.push_back(kf);
    }
  }
}

// tail of Map::UpdateMap for an ordinary frame, map.cc:117-125
void MapperMap::FinishUpdate() {
  if (!cur_) return;
  if (!cur_->IsKeyframe()) {
    CheckRedundantKeyframes();
    frame_trash_.push_back(cur_);
  }
  cur_.reset();
}

// Map::UpdateMap for ONE tracker (one K7 launch per phase); SDVLBatch runs the same phases for many trackers at once
// map.cc:49-71
MapperMap::~MapperMap() { Stop(); }

void MapperMap::Start() {
  if (running_) return;
  tracker_device_ = Device::CurrentOrNull();
  running_ = true;
  thread_ = std::thread(&MapperMap::Run, this);
}

void MapperMap::Stop() {
  running_ = false;
  if (thread_.joinable()) thread_.join();
}

void MapperMap::Run() {
  // one context (= HIP stream, staging, result buffers) per host thread; the tracker's frames are shared read-only, and
  // the points this thread creates continue the tracker's id sequence
  Device dev(tracker_device_ ? tracker_device_->gpu() : 0);
  if (tracker_device_) dev.point_ids = tracker_device_->point_ids;
  Device::SetCurrent(&dev);
  while (running_) {
    {
      std::unique_lock<std::mutex> lock(mutex_map_);
      UpdateMap();
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(2));  // map.cc:68
  }
  {
    std::unique_lock<std::mutex> lock(mutex_map_);
    cur_.reset();
  }
  Device::SetCurrent(nullptr);
}

void MapperMap::UpdateMap() {
  if (!BeginUpdate()) return;
  Device *dev = Device::Current();
  vector<sdvl_search_req> reqs;
  vector<sdvl_search_res> res;
  vector<sdvl_depth_state> states;
  vector<sdvl_depth_out> fout;
  const bool on_device = DeviceFilter();
  const sdvl_depth_params fp = FilterParams();
  for (;;) {
    reqs.clear();
    states.clear();
    if (!EmitCandidates(&reqs, on_device ? &states : nullptr)) break;
    res.clear();
    if (on_device) {
      // no table patching from here: in threaded mode this runs on the mapper's own context, concurrently with the tracker's
      for (sdvl_depth_state &st : states) st.track_row = -1;
      Matcher::SearchPointsFilter(dev, reqs, states, *camera_, fp, nullptr, &res, &fout);
      if (fout.empty()) fout.resize(1);
      ApplyCandidates(res.data(), fout.data());
    } else {
      Matcher::SearchPoints(dev, reqs, *camera_, &res);
      ApplyCandidates(res.data());
    }
  }
  if (IsKeyframeUpdate()) {
    CheckConnections();
    reqs.clear();
    EmitConnectionsPoints(&reqs);
    res.clear();
    Matcher::SearchPoints(dev, reqs, *camera_, &res);
    ApplyConnectionsPoints(res.data());
    if (PrepareInitCandidates()) {
      cur_->FilterCorners();
      reqs.clear();
      EmitInitCandidates(&reqs);
      res.clear();
      Matcher::SearchPoints(dev, reqs, *camera_, &res);
      ApplyInitCandidates(res.data());
    }
  }
  FinishUpdate();
}

}  // namespace sdvl
