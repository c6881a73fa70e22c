// standalone.cc — the part of the host layer the reference's own tree already has: Camera (camera.cc), Point (point.cc), Map and
// the plane-map stub (map.cc), SDVL (sdvl.cc) and, on top, the batch driver SDVLBatch (B trackers per submission, device-resident
// tracking tables).  The hot path's classes are in frontend.cc.
// Reference line numbers are cited at each function; device work goes through the C-ABI (include/sdvl_hip.h) only.
#include "sdvl_host.h"
#include "host_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <new>
#include <stdexcept>
#include <chrono>
#include <thread>

#include <sys/mman.h>

namespace sdvl {

using std::shared_ptr;
using std::vector;

// -------------------------------------------------------------------------------------------------------- Camera
Camera::Camera() {
  const CameraParameters &p = Config::GetCameraParameters();
  width_ = p.width; height_ = p.height; fx_ = p.fx; fy_ = p.fy; u0_ = p.u0; v0_ = p.v0;
  SetDistortions(p.d1, p.d2, p.d3, p.d4, p.d5);
}
Camera::Camera(int width, int height, double fx, double fy, double u0, double v0)
    : width_(width), height_(height), fx_(fx), fy_(fy), u0_(u0), v0_(v0) {}

// camera.cc:39-67
void Camera::SetDistortions(double d0, double d1, double d2, double d3, double d4) {
  d_[0] = d0; d_[1] = d1; d_[2] = d2; d_[3] = d3; d_[4] = d4;
  has_distortion_ = !(d_[0] == 0.0);
}

// camera.cc:100-105
void Camera::UndistortImage(const Image &in, Image *out) const {
  Device *dev = Device::Current();
  if (!dev) throw std::runtime_error("Camera::UndistortImage: no sdvl::Device bound to this thread");
  const int w = in.cols, h = in.rows;
  void *buf = nullptr;
  dev->Check(sdvl_device_malloc(dev->ctx(), static_cast<int64_t>(w) * h, &buf), "sdvl_device_malloc");
  sdvl_ctx *ctx = dev->ctx();
  std::shared_ptr<void> owner(buf, [ctx](void *p) { sdvl_device_free(ctx, p); });
  const void *src = in.dev_src ? in.dev_src : static_cast<const void *>(in.data);
  const sdvl_camera cam = abi();
  const sdvl_distortion dist = distortion();  // d0 == 0 -> plain copy, like in.clone()
  dev->Check(sdvl_undistort(ctx, 1, &src, in.step, in.dev_src ? 1 : 0, w, h, &cam, &dist, &buf, w), "sdvl_undistort");
  *out = Image::WrapDevice(buf, w, h, w, false);
  out->dev_owner = owner;
}

// camera.cc:69-79
void Camera::Project(const Vector3d &p3D, Vector2d *p2D) const {
  (*p2D)(0) = u0_ + fx_ * p3D(0) / p3D(2);
  (*p2D)(1) = v0_ + fy_ * p3D(1) / p3D(2);
}
void Camera::Unproject(const Vector2d &p2D, Vector3d *p3D) const {
  const double x = (p2D(0) - u0_) / fx_, y = (p2D(1) - v0_) / fy_, z = 1.0;
  const double n = std::sqrt(x * x + y * y + z * z);
  (*p3D)(0) = x / n; (*p3D)(1) = y / n; (*p3D)(2) = z / n;
}

static std::atomic<int> g_point_counter{0};

// point.cc:32-43
Point::Point() {
  Device *cur = Device::CurrentOrNull();
  id_ = cur ? (*cur->point_ids)++ : g_point_counter++;
  status_ = P_NOT_FOUND;
  last_frame_ = -1;
  n_failed_ = 0;
  n_successful_ = 0;
  delete_ = false;
  fixed_ = false;
  a_ = b_ = 10;
  rho_ = 1.0;
  sigma2_ = 1.0;
  z_range_ = 6.0;
}
void Point::ConsumeId() {
  if (Device *cur = Device::CurrentOrNull()) (*cur->point_ids)++;
  else g_point_counter++;
}
double Point::GetStd() { return std::sqrt(sigma2_); }

// point.cc:48-62
void Point::InitCandidate(const shared_ptr<Feature> &p, double depth) {
  feature_ = p;
  a_ = 10; b_ = 10;
  rho_ = 1.0 / depth;
  sigma2_ = 1.0;
  z_range_ = std::sqrt(sigma2_ * 36);
}
void Point::InitFixed(const shared_ptr<Feature> &f, double depth, double sigma2, const Vector3d &p3d) {
  InitCandidate(f, depth);
  sigma2_ = sigma2;
  fixed_ = true;
  p3d_ = p3d;
}

void Point::GetFilterState(sdvl_depth_state *st) const {
  st->rho = rho_; st->sigma2 = sigma2_; st->a = a_; st->b = b_; st->z_range = z_range_;
  st->cos_alpha = cos_alpha_; st->last_distance = last_distance_;
  st->position[0] = p3d_(0); st->position[1] = p3d_(1); st->position[2] = p3d_(2);
  st->fixed = fixed_ ? 1 : 0;
  st->n_failed = n_failed_;
  st->track_row = track_row_;
  st->pad_ = 0;
}

// what Map::UpdateCandidates' loop body (map.cc:454-497) left of the point, computed by depth_filter_kernel
void Point::ApplyFilterOut(const sdvl_depth_out &o) {
  const int what = o.outcome & 0xFF;
  if (what == SDVL_DEPTH_NOT_FOUND) {  // Unpromote
    n_failed_ = o.n_failed;
    b_ = o.b;
  } else if (what == SDVL_DEPTH_UPDATED || what == SDVL_DEPTH_CONVERGED) {  // Update (+ HasConverged)
    rho_ = o.rho; sigma2_ = o.sigma2; a_ = o.a; b_ = o.b;
    cos_alpha_ = o.cos_alpha; last_distance_ = o.last_distance;
    n_failed_ = 0;
    if (what == SDVL_DEPTH_CONVERGED && !fixed_) {
      p3d_ = Vector3d(o.position[0], o.position[1], o.position[2]);
      fixed_ = true;
    }
  } else if (what == SDVL_DEPTH_FIXED_STALE && !fixed_) {  // Update returned early, HasConverged fixed the old estimate
    p3d_ = Vector3d(o.position[0], o.position[1], o.position[2]);
    fixed_ = true;
  }
}

// point.cc:128-142
Vector3d Point::GetPosition() const {
  if (fixed_) return p3d_;
  Frame *fr = feature_->GetFrameRaw();
  const SE3 se3 = fr->GetWorldPose();
  const Vector3d &v = feature_->GetVector();
  const double s = 1.0 / rho_;
  return se3 * Vector3d(s * v(0), s * v(1), s * v(2));
}

// point.cc:144-162
void Point::SetPosition(const Vector3d &pos) {
  if (fixed_) {
    p3d_ = pos;
    return;
  }
  shared_ptr<Frame> frame = feature_->GetFrame();
  Vector2d p2d;
  frame->Project(pos, &p2d);
  Vector3d v3d = frame->GetCamera()->Unproject(p2d);
  feature_->SetVector(v3d);
  rho_ = 1.0 / frame->DistanceTo(pos);
}

// point.cc:105-118
bool Point::Promote() { n_successful_++; n_failed_ = 0; return true; }
bool Point::Unpromote() {
  n_failed_++;
  b_++;
  return n_failed_ > Config::MaxFailed();
}

// ------------------------------------------------------------------------------------------------------------ Map
// map.cc:170-188
bool Map::NeedKeyframe(const shared_ptr<Frame> &frame, int) {
  const int npoints = frame->GetNumPoints();
  const bool enough_its = (frame->GetID() - last_kf_->GetID()) >= Config::MinKeyframeIts();
  const bool lost_many = npoints < last_matches_ * Config::LostRatio();
  const bool lost_some = npoints < last_matches_ * 0.9;
  last_matches_ = std::max(last_matches_, npoints);
  if ((enough_its && lost_some) || lost_many) {
    last_matches_ = npoints;
    return true;
  }
  return false;
}

// map.cc:145-159
void Map::AddKeyframe(const shared_ptr<Frame> &frame, bool) {
  version_++;
  keyframes_.push_back(frame);
  last_kf_ = frame;
}

// map.cc:207-259 (points)
void Map::EmptyTrash() {
  if (!points_trash_.empty()) version_++;
  for (auto &p : points_trash_) {
    if (!p->ToDelete() && p->TrackRow() >= 0 && !p->DeviceTrashed()) tables_dirty_ = true;
    std::list<shared_ptr<Feature>> &features = p->GetFeatures();
    for (auto it = features.begin(); it != features.end(); it++) (*it)->SetPoint(nullptr);
    features.clear();
    p->SetDelete();
  }
  points_trash_.clear();
}

// map.cc:190-205,692-706 for the plane-map stub (see sdvl_host.h)
void PlaneMap::LimitKeyframes(const shared_ptr<Frame> &frame) {
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
  if (!kf || kf == frame) return;
  kf->SetDelete();
  culled_.push_back(kf);
}

void PlaneMap::EmptyTrash() {
  if (!culled_.empty()) version_++;
  for (const shared_ptr<Frame> &kf : culled_) {
    // the points this keyframe seeded (their first observation is one of its features) go with it
    vector<shared_ptr<Feature>> &features = kf->GetFeatures();
    for (auto it = features.begin(); it != features.end(); it++) {
      if (!*it) continue;
      shared_ptr<Point> p = (*it)->GetPoint();
      if (p && !p->ToDelete() && p->GetInitFeature() == *it) DeletePoint(p);
    }
    for (auto it = keyframes_.begin(); it != keyframes_.end(); it++)
      if (*it == kf) {
        keyframes_.erase(it);
        break;
      }
  }
  Map::EmptyTrash();  // (clears the doomed points' feature lists, the culled keyframe's features among them)
  for (const shared_ptr<Frame> &kf : culled_) kf->RemoveFeatures();
  culled_.clear();
}

void PlaneMap::InitCandidates(const shared_ptr<Frame> &kf) {
  kf->FilterCorners();
  SeedFromFiltered(kf);
}

void PlaneMap::SeedFromFiltered(const shared_ptr<Frame> &kf) {
  const SE3 world = kf->GetWorldPose();
  const M3 Rw = world.GetRotation();
  const Vector3d tw = world.GetTranslation();
  const int n_filtered = kf->NumFiltered();
  for (int k = 0; k < n_filtered; k++) {
    const Vector3i corner = kf->FilteredCorner(k);
    const int scale = (1 << corner(2));
    shared_ptr<Feature> feature = kf->NewFeature(Vector2d(corner(0) * scale, corner(1) * scale), corner(2));
    if (Config::UseORB()) feature->SetDescriptor(kf->FilteredDescriptor(k));  // filled by FilterCorners (frame.cc:145-161)
    const Vector3d &v = feature->GetVector();
    const V3 ray = mvec(Rw, {v(0), v(1), v(2)});
    const double denom = n_(0) * ray.x + n_(1) * ray.y + n_(2) * ray.z;
    if (!(std::fabs(denom) > 1e-9)) continue;
    const double s = (d_ - (n_(0) * tw(0) + n_(1) * tw(1) + n_(2) * tw(2))) / denom;
    if (!(s > 0.05)) continue;
    shared_ptr<Point> pt = kf->NewPoint();
    const double rho = 1.0 / s;
    pt->InitFixed(feature, s, (0.05 * rho) * (0.05 * rho), world * Vector3d(s * v(0), s * v(1), s * v(2)));
    version_++;
    feature->SetPoint(pt);
    kf->AddFeature(feature);
    pt->AddFeature(feature);
  }
}

// --------------------------------------------------------------------------------------------------- FeatureAlign
// feature_align.cc:33-54

// ----------------------------------------------------------------------------------------------------------- SDVL
SDVL::SDVL(Camera *camera, Map *map, const SE3 &first_pose)
    : camera_(camera), map_(map), rng_(1), feature_align_(map, camera, Config::MaxMatches(), &rng_), first_pose_(first_pose) {
  state_ = STATE_FIRST_FRAME;
  tracking_quality_ = TRACKING_GOOD;
  lost_frames_ = 0;
  matches_ = attempts_ = 0;
  frame_counter_ = 0;
}

static Device *OwnDeviceIfNone() {
  if (Device::CurrentOrNull()) return nullptr;
  const char *g = std::getenv("SDVL_GPU");
  return new Device(g ? std::atoi(g) : 0);  // binds itself to the thread
}

// sdvl.cc:36-50
SDVL::SDVL(Camera *camera)
    : own_device_(OwnDeviceIfNone()),
      own_map_(new MapperMap(Vector3d(0.0, 0.0, 1.0), Config::MapScale(), camera)),
      camera_(camera),
      map_(own_map_.get()),
      rng_(1),
      feature_align_(map_, camera, Config::MaxMatches(), &rng_),
      first_pose_(SE3()) {
  state_ = STATE_FIRST_FRAME;
  tracking_quality_ = TRACKING_GOOD;
  lost_frames_ = 0;
  matches_ = attempts_ = 0;
  frame_counter_ = 0;
}

SDVL::~SDVL() {
  if (own_map_) own_map_->Stop();
  self_batch_.reset();  // its tracking tables go before the frames and the device do
  current_frame_.reset();
  last_frame_.reset();
  last_kf_.reset();
  pending_kf_.reset();
  own_map_.reset();     // frames go back to the device's pool before the device does
}

void SDVL::SetBootstrapPlane(const Vector3d &n, double d) {
  if (PlaneMap *pm = dynamic_cast<PlaneMap *>(map_)) pm->SetPlane(n, d);
}

// the Point counters and feature lists HandleFrame's batch keeps on the device reach the host objects the queries below read
void SDVL::SyncSelfBatch() {
  if (self_batch_) self_batch_->SyncHostState();
}

const StageTimes *SDVL::HandleFrameStageTimes() const { return self_batch_ ? &self_batch_->stage_times : nullptr; }

// sdvl.cc:283-291
void SDVL::GetCameraTrail(vector<std::pair<SE3, bool>> *positions) {
  std::unique_lock<std::mutex> lock(map_->GetMutex());
  positions->clear();
  for (auto it = map_->GetKeyframes().begin(); it != map_->GetKeyframes().end(); it++)
    positions->push_back(std::make_pair((*it)->GetWorldPose(), (*it)->IsSelected()));
}

// sdvl.cc:293-324: converged points twice, the others as the two ends of their depth interval
void SDVL::GetPoints(vector<Vector3d> *positions) {
  std::unique_lock<std::mutex> lock(map_->GetMutex());
  SyncSelfBatch();
  positions->clear();
  for (auto it = map_->GetKeyframes().begin(); it != map_->GetKeyframes().end(); it++) {
    for (auto feature = (*it)->GetFeatures().begin(); feature != (*it)->GetFeatures().end(); feature++) {
      shared_ptr<Point> p = (*feature)->GetPoint();
      if (!p || p->ToDelete()) continue;
      if (p->HasConverged()) {
        positions->push_back(p->GetPosition());
        positions->push_back(p->GetPosition());
      } else {
        const double zmin = 1.0 / (p->GetInverseDepth() + 2.0 * p->GetStd());
        const double zmax = 1.0 / (std::max(p->GetInverseDepth() - 2.0 * p->GetStd(), 0.00000001));
        const Vector3d &v = p->GetInitFeature()->GetVector();
        const SE3 pose = p->GetInitFeature()->GetFrame()->GetWorldPose();
        positions->push_back(pose * Vector3d(v(0) * zmin, v(1) * zmin, v(2) * zmin));
        positions->push_back(pose * Vector3d(v(0) * zmax, v(1) * zmax, v(2) * zmax));
      }
    }
  }
}

// sdvl.cc:326-349: (x, y, Point status) of the last frame's features, then its outliers as status P_OUTLIER
void SDVL::GetLastFeatures(vector<Vector3i> *positions) {
  std::unique_lock<std::mutex> lock(map_->GetMutex());
  SyncSelfBatch();
  positions->clear();
  if (!last_frame_) return;
  vector<shared_ptr<Feature>> &features = last_frame_->GetFeatures();
  for (auto feature = features.begin(); feature != features.end(); feature++) {
    shared_ptr<Point> point = (*feature)->GetPoint();
    if (!point || point->ToDelete()) continue;
    const Vector2d &pos = (*feature)->GetPosition();
    positions->push_back(Vector3i(static_cast<int>(pos(0)), static_cast<int>(pos(1)), static_cast<int>(point->GetStatus())));
  }
  vector<Vector2d> &outliers = last_frame_->GetOutliers();
  for (auto it = outliers.begin(); it != outliers.end(); it++)
    positions->push_back(Vector3i(static_cast<int>((*it)(0)), static_cast<int>((*it)(1)), static_cast<int>(Point::P_OUTLIER)));
}

SE3 SDVL::GetPose() const {
  if (last_frame_) return last_frame_->GetWorldPose();
  return SE3();
}

// sdvl.cc:240-264
void SDVL::CalcTrackingQuality(int matches, int attempts) {
  const double ratio = (attempts == 0) ? 0.0 : static_cast<double>(matches) / static_cast<double>(attempts);
  if (ratio > 0.2) {
    tracking_quality_ = TRACKING_GOOD;
    lost_frames_ = 0;
    return;
  }
  if (matches < Config::MinMatches()) {
    tracking_quality_ = TRACKING_BAD;
    lost_frames_++;
    return;
  }
  lost_frames_ = 0;
  tracking_quality_ = TRACKING_INSUFFICIENT;
}

void SDVL::SetNextImage(const Image &next) {
  next_image_.assign(1, next);
}

bool SDVL::HandleFrame(const Image &img) {
  std::unique_lock<std::mutex> lock(map_->GetMutex());  // the mapper thread of threaded mode stays out meanwhile
  static const bool one_shot = std::getenv("SDVL_HANDLEFRAME_ONE_SHOT") != nullptr;
  FrameStats st;
  if (one_shot) {  // rounds 1-4: a one-call batch, no device-resident tables, every stage driven from the host
    next_image_.clear();
    SDVLBatch one(Device::Current(), {this}, 1);
    one.persistent_ = false;
    one.HandleFrames({img}, &st);
    return true;
  }
  Device *dev = Device::Current();
  if (!self_batch_ || self_batch_->dev_ != dev) {  // first call, or the caller moved to another Device (= stream)
    if (self_batch_) self_batch_->SyncHostState();
    self_batch_.reset(new SDVLBatch(dev, {this}, 1));
    track_.valid = false;
  }
  if (!next_image_.empty()) {
    self_batch_->SetNextImages(next_image_);
    next_image_.clear();
  }
  try {
    self_batch_->HandleFrames({img}, &st);
  } catch (...) {
    // a step that failed half way leaves the set's tables and its own bookkeeping in an unknown state: the next call starts with a
    // fresh batch and rebuilds the table from the host objects (what the one-shot form did after every call)
    self_batch_.reset();
    track_.valid = false;
    throw;
  }
  return true;
}

void SDVL::Mapping() {
  std::unique_lock<std::mutex> lock(map_->GetMutex());
  if (pending_kf_) {
    PlaneMap *pm = dynamic_cast<PlaneMap *>(map_);
    if (pm) {
      pending_kf_->FilterCorners();
      pm->SeedFromFiltered(pending_kf_);
    } else {
      map_->InitCandidates(pending_kf_);
    }
    pending_kf_ = nullptr;
  }
  if (MapperMap *m = dynamic_cast<MapperMap *>(map_)) m->UpdateMap();  // main.cc:148-149
}

// ------------------------------------------------------------------------------------------------------ SDVLBatch
namespace {
// persistent worker pool for the per-sequence host stages
class Pool {
 public:
  explicit Pool(int n) : stop_(false), gen_(0), pending_(0) {
    for (int i = 0; i < n; i++) workers_.emplace_back([this] { Run(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
  }
  void For(int n, const std::function<void(int)> &fn) {
    if (n <= 0) return;
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      n_ = n;
      dev_ = Device::CurrentOrNull();
      next_.store(0);
      pending_ = static_cast<int>(workers_.size());
      gen_++;
    }
    cv_.notify_all();
    Drain();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
    if (err_) {
      std::exception_ptr e = err_;
      err_ = nullptr;
      std::rethrow_exception(e);
    }
  }

 private:
  void Drain() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n_) break;
      try {
        (*fn_)(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(m_);
        if (!err_) err_ = std::current_exception();
      }
    }
  }
  void Run() {
    unsigned seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        Device::SetCurrent(dev_);  // the helpers work for the caller's device (Point ids, scratch) and on its GPU
      }
      Drain();
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  bool stop_;
  unsigned gen_;
  int pending_;
  const std::function<void(int)> *fn_ = nullptr;
  Device *dev_ = nullptr;
  int n_ = 0;
  std::atomic<int> next_{0};
  std::exception_ptr err_;
};
}  // namespace

static Pool *g_pool_of(void *&slot, int threads) {
  if (!slot && threads > 1) slot = new Pool(threads - 1);
  return static_cast<Pool *>(slot);
}

SDVLBatch::SDVLBatch(Device *dev, const vector<SDVL *> &trackers, int host_threads) : dev_(dev), trk_(trackers), threads_(host_threads) {
  for (size_t i = 0; i < trk_.size(); i++) trk_[i]->track_.slot = static_cast<int>(i);
  // Round 5: a lone camera (a batch of up to four) waits for chains of 0.2 ms: its context polls without sleeping for the first 500 us of a
  // wait (a sleeping poll wakes ~15 us late: 3.51 -> 3.83 k frames/s).  Larger batches keep the sleeping polls: 16 cameras in one batch gain
  // nothing, configuration C's groups of 16 lose (46.8 -> 43.3 k: their host threads need the CPU), a farm waits ~9 ms with the CPUs short.
  const int spin_us = trk_.size() <= 4 ? 500 : 0;
  if (dev_ && dev_->ctx()) (void)sdvl_ctx_set_wait_spin(dev_->ctx(), spin_us);
}
SDVLBatch::~SDVLBatch() {
  if (track_) {
    Device::SetCurrent(dev_);
    sdvl_track_destroy(dev_->ctx(), track_);
  }
  if (reloc_store_) {
    Device::SetCurrent(dev_);
    (void)sdvl_align_store_destroy(dev_->ctx(), reloc_store_);
  }
}

static thread_local void *g_pool_slot = nullptr;
static thread_local int g_pool_threads = 0;

void SDVLBatch::ParallelFor(int n, const std::function<void(int)> &fn) {
  if (threads_ <= 1 || n <= 1) {
    for (int i = 0; i < n; i++) fn(i);
    return;
  }
  if (g_pool_slot && g_pool_threads != threads_) {
    delete static_cast<Pool *>(g_pool_slot);
    g_pool_slot = nullptr;
  }
  g_pool_threads = threads_;
  g_pool_of(g_pool_slot, threads_)->For(n, fn);
}

static int g_device_pose = -1;  // -1: not decided yet (environment)
void SDVLBatch::SetNextImages(const vector<Image> &next) { next_imgs_ = next; }

void SDVLBatch::SetDevicePose(bool on) { g_device_pose = on ? 1 : 0; }
bool SDVLBatch::DevicePose() {
  if (g_device_pose < 0) {
    const char *e = std::getenv("SDVL_POSE_HOST");
    g_device_pose = (e && e[0] == '1') ? 0 : 1;
  }
  return g_device_pose != 0;
}

static int g_track_tables = -1;  // -1: not decided yet (environment)
void SDVLBatch::SetTrackTables(bool on) { g_track_tables = on ? 1 : 0; }
bool SDVLBatch::TrackTables() {
  if (g_track_tables < 0) g_track_tables = std::getenv("SDVL_NO_TRACK_TABLES") ? 0 : 1;
  return g_track_tables != 0;
}

// The counters the device advanced (Promote / Unpromote / SetLastFrame / status) reach the Point objects.
void SDVLBatch::SyncStats(SDVL &t) {
  SDVL::TrackState &ts = t.track_;
  if (!ts.stats_dirty || !ts.points) return;
  const size_t n = std::min(ts.stats.size(), ts.points->size());
  for (size_t p = 0; p < n; p++) {
    const sdvl_track_point_stat &s = ts.stats[p];
    (*ts.points)[p]->SetTrackCounters(s.score, s.n_failed, s.last_frame, static_cast<Point::PointStatus>(s.status & 0xFF));
  }
  ts.stats_dirty = false;
}

void SDVLBatch::SyncHostState() {
  for (SDVL *t : trk_) {
    SyncStats(*t);
    if (t->last_frame_ && t->last_frame_->HasFlatFeatures()) t->last_frame_->GetFeatures();
  }
}

// last_frame's features and the points behind them as table rows (include/sdvl_hip.h: sdvl_track_point / _feature).
// false: something the tables cannot express (a point without a first observation or whose keyframe is gone) — the step
// then runs on the host path.
bool SDVLBatch::BuildTable(SDVL &t) {
  SyncStats(t);
  vector<sdvl_track_point> *points = &t.track_.up_points;
  vector<sdvl_track_feature> *feats = &t.track_.up_feats;
  points->clear();
  feats->clear();
  t.track_.up_register.clear();
  vector<shared_ptr<Feature>> &features = t.last_frame_->GetFeatures();
  // the alignment stage takes at most SDVL_MAX_ALIGN_FEATURES features per job: a frame with more (a keyframe of configuration C
  // that has gathered matches + connections + candidates) goes through the host-driven step instead of failing the batch
  if (static_cast<int>(features.size()) > std::min(track_cap_, static_cast<int>(SDVL_MAX_ALIGN_FEATURES))) return false;
  if (t.track_.points)
    for (const shared_ptr<Point> &old : *t.track_.points) old->SetTrackRow(-1);
  const int row_base = t.track_.slot * track_cap_;
  auto table = std::make_shared<Frame::PointTable>();
  table->reserve(features.size());
  // a point may sit behind several features of a keyframe: ProjectPoints takes the first and skips the rest
  // (feature_align.cc:310: the point already carries this frame's id)
  struct Slot { Point *p; int idx; };
  static thread_local vector<Slot> hash;
  size_t hcap = 64;
  while (hcap < features.size() * 2) hcap *= 2;
  hash.assign(hcap, Slot{nullptr, -1});
  const size_t n_features = features.size();
  for (size_t fi = 0; fi < n_features; fi++) {
    // Feature -> Point -> first Feature -> keyframe: four dependent loads per row, asked for a few features ahead
    if (fi + 8 < n_features) __builtin_prefetch(features[fi + 8].get());
    if (fi + 4 < n_features && features[fi + 4]) {
      const Point *p4 = features[fi + 4]->GetPointRaw();
      __builtin_prefetch(p4);
      if (p4) __builtin_prefetch(reinterpret_cast<const char *>(p4) + 64);
    }
    if (fi + 2 < n_features && features[fi + 2]) {
      const Point *p2 = features[fi + 2]->GetPointRaw();
      if (p2) __builtin_prefetch(p2->GetInitFeatureRaw());
    }
    const shared_ptr<Feature> &ftp = features[fi];
    Feature *ft = ftp.get();
    sdvl_track_feature f;
    if (!ft) return false;
    f.px[0] = ft->GetPosition()(0); f.px[1] = ft->GetPosition()(1);
    f.bearing[0] = ft->GetVector()(0); f.bearing[1] = ft->GetVector()(1); f.bearing[2] = ft->GetVector()(2);
    f.level = ft->GetLevel();
    f.point = -1;
    Point *pt = ft->GetPointRaw();
    if (pt && !pt->ToDelete()) {
      size_t h = (reinterpret_cast<uintptr_t>(pt) >> 4) & (hcap - 1);
      while (hash[h].p && hash[h].p != pt) h = (h + 1) & (hcap - 1);
      if (hash[h].p) {
        f.point = hash[h].idx | SDVL_TRACK_DUPLICATE;
      } else {
        Feature *init = pt->GetInitFeatureRaw();
        Frame *ref = init ? init->GetFrameRaw() : nullptr;
        if (!init || !ref || ref->owner() != dev_) return false;
        const int idx = static_cast<int>(table->size());
        hash[h] = Slot{pt, idx};
        f.point = idx;
        table->push_back(ft->GetPoint());
        pt->SetTrackRow(row_base + idx);
        points->emplace_back();
        sdvl_track_point &tp = points->back();
        const Vector3d P = pt->GetPosition();
        tp.position[0] = P(0); tp.position[1] = P(1); tp.position[2] = P(2);
        tp.px[0] = init->GetPosition()(0); tp.px[1] = init->GetPosition()(1);
        tp.bearing[0] = init->GetVector()(0); tp.bearing[1] = init->GetVector()(1); tp.bearing[2] = init->GetVector()(2);
        tp.idepth = pt->GetInverseDepth();
        tp.idepth_std = pt->GetStd();
        tp.ref = ref->device();
        tp.level = init->GetLevel();
        tp.fixed = pt->IsFixed() ? 1 : 0;
        tp.score = pt->Score();
        tp.n_failed = pt->GetFailed();
        tp.last_frame = pt->GetLastFrame();
        tp.status = static_cast<int32_t>(pt->GetStatus());
        if (init->HasDescriptor()) std::memcpy(tp.desc, init->DescriptorData().data(), 32);
        else std::memset(tp.desc, 0, 32);
        // a keyframe that never was the current frame of a tracked step (bootstrap, host-path steps): the caller registers
        // it (device calls of one context come from one thread)
        if (!ref->IsRegistered()) t.track_.up_register.push_back(ref);
      }
    }
    feats->push_back(f);
  }
  if (static_cast<int>(points->size()) > track_cap_) return false;
  t.track_.points = table;
  t.track_.stats.clear();
  t.track_.stats_dirty = false;
  return true;
}

// The points PlaneMap::SeedFromFiltered has just put on keyframe `kf` (its object features from track_.first_seed on) as table rows
// behind the rows the tables hold: point index = position in the tracker's point table, which grows by the same points.
bool SDVLBatch::AppendSeeds(SDVL &t, const shared_ptr<Frame> &kf) {
  SDVL::TrackState &ts = t.track_;
  if (!ts.points) return false;
  ts.up_points.clear();
  ts.up_feats.clear();
  const vector<shared_ptr<Feature>> &objs = kf->ObjectFeatures();
  const int row_base = ts.slot * track_cap_;
  if (static_cast<int>(ts.points->size() + (objs.size() - ts.first_seed)) > track_cap_) return false;
  if (kf->NumFlatFeatures() + static_cast<int>(objs.size()) > std::min(track_cap_, static_cast<int>(SDVL_MAX_ALIGN_FEATURES))) return false;
  if (kf->owner() != dev_ || !kf->IsRegistered()) return false;
  ts.up_points.reserve(objs.size() - ts.first_seed);
  ts.up_feats.reserve(objs.size() - ts.first_seed);
  for (size_t fi = ts.first_seed; fi < objs.size(); fi++) {
    Feature *ft = objs[fi].get();
    Point *pt = ft ? ft->GetPointRaw() : nullptr;
    if (!ft || !pt || pt->GetInitFeatureRaw() != ft) return false;  // a seed is the first observation of its own point
    sdvl_track_feature f;
    f.px[0] = ft->GetPosition()(0); f.px[1] = ft->GetPosition()(1);
    f.bearing[0] = ft->GetVector()(0); f.bearing[1] = ft->GetVector()(1); f.bearing[2] = ft->GetVector()(2);
    f.level = ft->GetLevel();
    const int idx = static_cast<int>(ts.points->size());
    f.point = idx;
    ts.points->push_back(ft->GetPoint());
    pt->SetTrackRow(row_base + idx);
    ts.up_points.emplace_back();
    sdvl_track_point &tp = ts.up_points.back();
    const Vector3d P = pt->GetPosition();
    tp.position[0] = P(0); tp.position[1] = P(1); tp.position[2] = P(2);
    tp.px[0] = f.px[0]; tp.px[1] = f.px[1];
    tp.bearing[0] = f.bearing[0]; tp.bearing[1] = f.bearing[1]; tp.bearing[2] = f.bearing[2];
    tp.idepth = pt->GetInverseDepth();
    tp.idepth_std = pt->GetStd();
    tp.ref = kf->device();
    tp.level = f.level;
    tp.fixed = pt->IsFixed() ? 1 : 0;
    tp.score = pt->Score();
    tp.n_failed = pt->GetFailed();
    tp.last_frame = pt->GetLastFrame();
    tp.status = static_cast<int32_t>(pt->GetStatus());
    if (ft->HasDescriptor()) std::memcpy(tp.desc, ft->DescriptorData().data(), 32);
    else std::memset(tp.desc, 0, 32);
    ts.up_feats.push_back(f);
  }
  return true;
}

// Tables of the trackers `need` rebuilt from their last_frame's host objects and sent to the device in ONE submission.
// built == nullptr: all or nothing (false, nothing uploaded, if one of them cannot be expressed as a table);
// otherwise the expressible ones are uploaded and (*built)[q] says which.
bool SDVLBatch::UploadTables(const vector<int> &need, vector<char> *built) {
  vector<char> ok(need.size(), 0);
  ParallelFor(static_cast<int>(need.size()), [&](int q) { ok[q] = BuildTable(*trk_[need[q]]) ? 1 : 0; });
  if (!built)
    for (char o : ok)
      if (!o) return false;
  vector<int32_t> up_trk, up_buf, up_np, up_nf;
  tr_up_points_.clear();
  tr_up_feats_.clear();
  vector<const sdvl_frame *> reg_frames;   // every reference frame the rebuilt tables name for the first time: ONE submission
  vector<double> reg_poses;
  for (size_t q = 0; q < need.size(); q++) {
    if (!ok[q]) continue;
    const int i = need[q];
    SDVL::TrackState &ts = trk_[i]->track_;
    for (Frame *ref : ts.up_register)
      if (!ref->IsRegistered()) {
        double pose[7];
        ref->GetPose().ToArray(pose);
        reg_frames.push_back(ref->device());
        reg_poses.insert(reg_poses.end(), pose, pose + 7);
        ref->SetRegistered();
      }
    up_trk.push_back(i);
    up_buf.push_back(0);
    up_np.push_back(static_cast<int32_t>(ts.up_points.size()));
    up_nf.push_back(static_cast<int32_t>(ts.up_feats.size()));
    tr_up_points_.insert(tr_up_points_.end(), ts.up_points.begin(), ts.up_points.end());
    tr_up_feats_.insert(tr_up_feats_.end(), ts.up_feats.begin(), ts.up_feats.end());
  }
  if (!reg_frames.empty())
    dev_->Check(sdvl_frames_register(dev_->ctx(), static_cast<int>(reg_frames.size()), reg_frames.data(), reg_poses.data()), "sdvl_frames_register");
  if (!up_trk.empty())
    dev_->Check(sdvl_track_upload(dev_->ctx(), track_, static_cast<int>(up_trk.size()), up_trk.data(), up_buf.data(), up_np.data(), tr_up_points_.data(),
                                  up_nf.data(), tr_up_feats_.data()), "sdvl_track_upload");
  for (size_t q = 0; q < need.size(); q++)
    if (ok[q]) {
      trk_[need[q]]->track_.feat_buf = 0;
      trk_[need[q]]->track_.valid = true;
    }
  if (built) *built = ok;
  return true;
}

static std::atomic<unsigned long long> g_reloc_epoch{1};  // process-wide: a cache never mistakes another batch's store for its own

// The alignments of RelocalizeLost: jobs whose feature ranges name records of the batch's store, in launches of bounded size
// (a farm whose trackers are all lost asks for B x |keyframes| alignments per step).
void SDVLBatch::RelocAlign(const vector<sdvl_align_job> &jobs, const sdvl_align_params &ap, vector<sdvl_align_result> *res) {
  res->resize(jobs.size());
  const sdvl_camera cam = trk_[0]->camera_->abi();
  const size_t kChunk = 4096;
  for (size_t b = 0; b < jobs.size(); b += kChunk) {
    const int n = static_cast<int>(std::min(jobs.size(), b + kChunk) - b);
    dev_->Check(sdvl_image_align_begin_stored(dev_->ctx(), n, jobs.data() + b, reloc_store_, &cam, &ap), "sdvl_image_align_begin_stored");
    dev_->Check(sdvl_image_align_end(dev_->ctx(), n, res->data() + b), "sdvl_image_align_end");
  }
}

// SDVL::Relocalize (sdvl.cc:73-89,205-238) for the trackers `lost` of a batch, inside the tabled step: the other trackers of the batch
// are not touched.  The alignment of a tracker's current frame against EVERY keyframe of its map (each started from that keyframe's
// pose, fast mode) is independent of the others: one launch for all (tracker, keyframe) pairs of all lost trackers.
//  * The keyframes' feature records do not change while a tracker is lost (no tracked step, the mapper stands still): they are packed
//    once and stay in HBM (sdvl_align_store; SDVL::RelocCache remembers where) — per frame only the job records cross the link.
//  * Fast mode gives up behind the coarsest level when the update there is still > 0.01 (image_align.cc:73-76), and on a frame that
//    shows something else nearly every keyframe does: the launch runs that level ONLY (a third of PrecomputePatches); the few jobs
//    that pass it are run again in full — same start, same arithmetic, so both passes leave what one full pass would.
// The keyframe loop then runs in the reference's order (newest first) over the results; Reproject(reloc = true) draws from rand() and
// stops at the first keyframe that gathers MinMatches, so it stays sequential per tracker — but the trackers advance together: one
// search launch per ROUND (a round = every unresolved tracker's next keyframe with error < 0.001; almost always there is one round).
// (*found)[q]: tracker lost[q] has relocalised — last_frame_ = last_kf_ = the keyframe it landed on (sdvl.cc:84-86).
void SDVLBatch::RelocalizeLost(const vector<int> &lost, FrameStats *stats, vector<char> *found) {
  const int L = static_cast<int>(lost.size());
  found->assign(L, 0);
  // ---- the keyframes' records: packed for the trackers whose map changed since (or that were not lost before)
  vector<vector<sdvl_align_feature>> fresh(L);
  vector<char> stale(L, 0);
  for (int q = 0; q < L; q++) {
    SDVL &t = *trk_[lost[q]];
    SyncStats(t);                   // the Point objects Reproject reads catch up with the device's counters
    t.map_->SetRelocalizing(true);  // sdvl.cc:80
    for (int k = 0; k < 6; k++) t.vel_[k] = 0.0;
    SDVL::RelocCache &rc = t.reloc_;
    stale[q] = (!reloc_store_ || rc.owner != this || rc.epoch != reloc_epoch_ || rc.version != t.map_->Version()) ? 1 : 0;
  }
  ParallelFor(L, [&](int q) {
    if (!stale[q]) return;
    SDVL &t = *trk_[lost[q]];
    SDVL::RelocCache &rc = t.reloc_;
    rc.kfs.clear();
    rc.begin.assign(1, 0);
    rc.T.clear();
    vector<shared_ptr<Frame>> &kfs = t.map_->GetKeyframes();
    for (auto it = kfs.rbegin(); it != kfs.rend(); it++) {
      rc.kfs.push_back(*it);
      ImageAlign::PackFeatures(**it, &fresh[q]);
      rc.begin.push_back(static_cast<int32_t>(fresh[q].size()));
      double T7[7];
      ((*it)->GetPose() * (*it)->GetPose().Inverse()).ToArray(T7);  // current_frame_->SetPose(cframe->GetPose()), then image_align.cc:66
      rc.T.insert(rc.T.end(), T7, T7 + 7);
    }
  });
  {
    size_t need = 0;
    for (int q = 0; q < L; q++)
      if (stale[q]) need += fresh[q].size();
    if (need > 0 && (!reloc_store_ || reloc_used_ + need > static_cast<size_t>(reloc_cap_))) {
      // no room behind the records in use: a new store for what is live now (the other trackers' caches are void: epoch)
      size_t live = need;
      for (int q = 0; q < L; q++)
        if (!stale[q]) {
          stale[q] = 1;  // repack: its records die with the old store
          SDVL &t = *trk_[lost[q]];
          SDVL::RelocCache &rc = t.reloc_;
          rc.kfs.clear(); rc.begin.assign(1, 0); rc.T.clear();
          vector<shared_ptr<Frame>> &kfs = t.map_->GetKeyframes();
          for (auto it = kfs.rbegin(); it != kfs.rend(); it++) {
            rc.kfs.push_back(*it);
            ImageAlign::PackFeatures(**it, &fresh[q]);
            rc.begin.push_back(static_cast<int32_t>(fresh[q].size()));
            double T7[7];
            ((*it)->GetPose() * (*it)->GetPose().Inverse()).ToArray(T7);
            rc.T.insert(rc.T.end(), T7, T7 + 7);
          }
          live += fresh[q].size();
        }
      if (reloc_store_) dev_->Check(sdvl_align_store_destroy(dev_->ctx(), reloc_store_), "sdvl_align_store_destroy");
      reloc_store_ = nullptr;
      reloc_epoch_ = g_reloc_epoch++;
      reloc_used_ = 0;
      const size_t cap = std::max<size_t>(2 * live, 65536);
      if (cap > 0x7fffffffu) throw std::runtime_error("SDVLBatch: relocalisation store too large");
      dev_->Check(sdvl_align_store_create(dev_->ctx(), static_cast<int>(cap), &reloc_store_), "sdvl_align_store_create");
      reloc_cap_ = static_cast<int>(cap);
    }
    if (need > 0) {
      vector<sdvl_align_feature> all;
      const int base = reloc_used_;
      for (int q = 0; q < L; q++) {
        if (!stale[q]) continue;
        SDVL &t = *trk_[lost[q]];
        SDVL::RelocCache &rc = t.reloc_;
        const int32_t off = base + static_cast<int32_t>(all.size());
        for (int32_t &b : rc.begin) b += off;
        all.insert(all.end(), fresh[q].begin(), fresh[q].end());
        rc.owner = this;
        rc.epoch = reloc_epoch_;
        rc.version = t.map_->Version();
      }
      dev_->Check(sdvl_align_store_write(dev_->ctx(), reloc_store_, base, static_cast<int>(all.size()), all.data()), "sdvl_align_store_write");
      reloc_used_ = base + static_cast<int>(all.size());
    }
  }
  // ---- the alignments
  vector<sdvl_align_job> jobs;
  vector<int> first(L + 1, 0);
  for (int q = 0; q < L; q++) {
    SDVL &t = *trk_[lost[q]];
    const SDVL::RelocCache &rc = t.reloc_;
    first[q] = static_cast<int>(jobs.size());
    for (size_t j = 0; j < rc.kfs.size(); j++) {
      if (rc.begin[j + 1] == rc.begin[j]) {
        std::cerr << "[ERROR] No points to track!" << std::endl;  // image_align.cc:55-58
      }
      sdvl_align_job jb;
      jb.ref = rc.kfs[j]->device();
      jb.cur = t.current_frame_->device();
      jb.feat_begin = rc.begin[j];
      jb.feat_end = rc.begin[j + 1];
      std::memcpy(jb.T, rc.T.data() + 7 * j, sizeof(jb.T));
      jobs.push_back(jb);
    }
  }
  first[L] = static_cast<int>(jobs.size());
  vector<sdvl_align_result> res;
  if (!jobs.empty()) {
    sdvl_align_params ap = AlignParams(true);
    const int min_level = ap.min_level;
    ap.min_level = ap.max_level;  // the coarsest level alone
    RelocAlign(jobs, ap, &res);
    if (min_level < ap.max_level) {
      vector<int> pass;
      for (size_t j = 0; j < jobs.size(); j++)
        if (res[j].error <= 0.01) pass.push_back(static_cast<int>(j));  // image_align.cc:73-76 did not give up: the finer levels follow
      if (!pass.empty()) {
        vector<sdvl_align_job> again;
        for (int j : pass) again.push_back(jobs[j]);
        vector<sdvl_align_result> full;
        ap.min_level = min_level;
        RelocAlign(again, ap, &full);
        for (size_t k = 0; k < pass.size(); k++) res[pass[k]] = full[k];
      }
    }
  }
  // ---- the keyframe loop, sdvl.cc:209-235
  vector<int> cursor(first.begin(), first.end() - 1);
  vector<char> open_(L, 1);
  vector<vector<sdvl_search_req>> per(L);
  vector<sdvl_search_req> reqs;
  vector<sdvl_search_res> sres;
  vector<size_t> begin(L + 1, 0);
  for (;;) {
    bool any = false;
    reqs.clear();
    for (int q = 0; q < L; q++) {
      per[q].clear();
      if (!open_[q]) continue;
      SDVL &t = *trk_[lost[q]];
      const SDVL::RelocCache &rc = t.reloc_;
      // every keyframe visited leaves its aligned pose on the frame (image_align.cc:79); those with error >= 0.001 are passed over
      while (cursor[q] < first[q + 1]) {
        const int j = cursor[q] - first[q];
        if (rc.begin[j + 1] > rc.begin[j])  // (ComputePose returns before touching the pose when frame1 has no features)
          t.current_frame_->SetPose(SE3::FromArray(res[cursor[q]].T) * rc.kfs[j]->GetPose());
        else
          t.current_frame_->SetPose(rc.kfs[j]->GetPose());
        const double err = rc.begin[j + 1] > rc.begin[j] ? res[cursor[q]].error : 1e10;
        if (err < 0.001) break;
        cursor[q]++;
      }
      if (cursor[q] >= first[q + 1]) {
        open_[q] = 0;
        continue;
      }
      const shared_ptr<Frame> &cframe = rc.kfs[cursor[q] - first[q]];
      t.feature_align_.PrepareReproject(t.current_frame_, cframe, true, &per[q]);
      any = true;
    }
    if (!any) break;
    for (int q = 0; q < L; q++) {
      begin[q] = reqs.size();
      reqs.insert(reqs.end(), per[q].begin(), per[q].end());
    }
    begin[L] = reqs.size();
    Matcher::SearchPoints(dev_, reqs, *trk_[lost[0]]->camera_, &sres);
    for (int q = 0; q < L; q++) {
      if (!open_[q] || cursor[q] >= first[q + 1]) continue;
      SDVL &t = *trk_[lost[q]];
      t.feature_align_.FinishReproject(t.current_frame_, sres.data() + begin[q]);
      t.matches_ = t.feature_align_.GetMatches();
      t.attempts_ = t.feature_align_.GetAttempts();
      if (t.matches_ >= Config::MinMatches()) {
        const shared_ptr<Frame> cframe = t.reloc_.kfs[cursor[q] - first[q]];
        t.map_->SetRelocalizing(false);  // sdvl.cc:84
        t.last_kf_ = cframe;
        t.last_frame_ = cframe;
        t.track_.valid = false;  // the table follows last_frame_
        stats[lost[q]].relocalized = 1;
        (*found)[q] = 1;
        open_[q] = 0;
      } else {
        cursor[q]++;
      }
    }
  }
}

// ProcessFrame (sdvl.cc:179-203) of ONE tracker through the per-object calls (ImageAlign::ComputePose, FeatureAlign::Reproject /
// OptimizePose), inside a tabled step: for the tracker whose last_frame no table can express (a keyframe with more features than the
// alignment kernels take) while everybody else stays on the tables.  Returns the decision (0 lost, 1 frame, 2 keyframe).
int SDVLBatch::TrackOnHost(SDVL &t, FrameStats *st) {
  st->host_path = 2;
  ImageAlign image_align;
  st->align_meas = image_align.ComputePose(t.last_frame_, t.current_frame_);
  st->align_features = static_cast<int>(t.last_frame_->GetFeatures().size());
  t.feature_align_.Reproject(t.current_frame_, t.last_frame_, t.last_kf_);
  t.matches_ = t.feature_align_.GetMatches();
  t.attempts_ = t.feature_align_.GetAttempts();
  t.feature_align_.OptimizePose(t.current_frame_);
  st->inliers = t.feature_align_.GetInliers();
  st->outliers = t.feature_align_.GetOutliers();
  {  // GetMotionModel, sdvl.cc:266-276
    const SE3 mov = t.current_frame_->GetPose() * t.last_frame_->GetPose().Inverse();
    const Vector6d vel = SE3::Log(mov);
    for (int c = 0; c < 6; c++) t.vel_[c] = 0.9 * (0.5 * vel[c] + 0.5 * t.vel_[c]);
  }
  t.CalcTrackingQuality(t.matches_, t.attempts_);
  t.track_.valid = false;  // this frame's features live on the host
  if (t.tracking_quality_ == SDVL::TRACKING_BAD) return 0;
  return (t.tracking_quality_ == SDVL::TRACKING_GOOD && t.map_->NeedKeyframe(t.current_frame_, t.matches_)) ? 2 : 1;
}

// One step of B trackers on the device-resident tables: ONE submission (alignment, detection, reprojection, search, match
// selection, pose, table update) and ONE wait.  The host keeps what only it can do: rand() (cell shuffle, RANSAC draws), the
// motion model, tracking quality, the keyframe decision and everything a keyframe sets off.
bool SDVLBatch::HandleFramesTracked(const vector<Image> &imgs, FrameStats *stats) {
  if (!persistent_ || !TrackTables() || !DevicePose() || Config::MaxRansacPoints() > 8) return false;
  const int B = static_cast<int>(trk_.size());
  for (int i = 0; i < B; i++) {
    SDVL &t = *trk_[i];
    if (t.feature_align_.MaxMatches() > FeatureAlign::kMaxDevicePoseObs || t.feature_align_.MaxMatches() < 1) return false;
    if (t.camera_ != trk_[0]->camera_) return false;
  }
  // The rows of a tracker's table live in the set of ONE batch: a tracker that was last stepped by another batch (a farm batch and its own
  // SDVL::HandleFrame batch, say) brings its counters up to date and gets its table rebuilt here, under this batch's slot
  for (int i = 0; i < B; i++) {
    SDVL::TrackState &ts = trk_[i]->track_;
    if (ts.owner != this) {
      SyncStats(*trk_[i]);
      ts.valid = false;
      ts.append_pending = false;
      ts.owner = this;
    }
    ts.slot = i;
  }
  Camera &camera = *trk_[0]->camera_;
  const int cells = trk_[0]->feature_align_.GridCells();
  if (cells > 65535) return false;
  if (!track_) {
    // a keyframe holds its matches plus what the mapper adds (one seed per free grid cell with the plane map): twice that
    // leaves room for the reference mapper's connection points; a frame that outgrows it is tracked on the host path
    int mm = 1;
    for (SDVL *t : trk_) mm = std::max(mm, t->feature_align_.MaxMatches());
    track_cap_ = std::min(4096, 2 * (mm + cells));
    track_cells_ = cells;
    dev_->Check(sdvl_track_create(dev_->ctx(), B, track_cap_, track_cap_, cells, mm, Config::MaxRansacIts(), &track_), "sdvl_track_create");
  }
  if (cells != track_cells_) return false;

  // ---- tables of the trackers whose last_frame changed behind the device's back (keyframes: seeded / mapped features).
  // First of all: a frame that cannot be expressed as a table sends the whole step to the host path, untouched.
  // (A tracker that has to relocalise — lost_frames_ >= 3, sdvl.cc:73 — gets its table further down, from the keyframe it lands on.)
  std::unique_ptr<StageClock> clk(new StageClock(ST_PREPARE));
  {
    vector<int> need;
    for (int i = 0; i < B; i++)
      if (trk_[i]->state_ == SDVL::STATE_RUNNING && !trk_[i]->track_.valid && trk_[i]->lost_frames_ < 3) need.push_back(i);
    if (!need.empty() && !UploadTables(need, nullptr)) return false;
  }

  // ---- stage 0: Frame construction, sdvl.cc:59 (pyramids now, detection behind the alignment)
  vector<shared_ptr<Frame>> frames;
  const std::function<void(int, std::function<void(int)>)> pfor = [this](int n, std::function<void(int)> fn) { ParallelFor(n, fn); };
  // the frames of a look-ahead (SetNextImages before the previous step): pyramids and detection are queued already
  bool detected_ahead = false;
  if (!ahead_frames_.empty()) {
    bool same = static_cast<int>(ahead_frames_.size()) == B && static_cast<int>(imgs.size()) == B;
    for (int i = 0; same && i < B; i++) same = imgs[i].dev_src != nullptr && imgs[i].dev_src == ahead_src_[i];
    if (same) {
      frames.swap(ahead_frames_);
      detected_ahead = true;
    }
    ahead_frames_.clear();
    ahead_src_.clear();
  }
  if (!detected_ahead) Frame::CreateBatch(&camera, &trk_[0]->orb_detector_, imgs, false, Config::NumFeatures(), &frames, &pfor);
  vector<int> run, lost;
  clk.reset(new StageClock(ST_PRELUDE));
  for (int i = 0; i < B; i++) {
    SDVL &t = *trk_[i];
    FrameStats &st = stats[i];
    st = FrameStats();
    t.current_frame_ = frames[i];
    t.current_frame_->SetID(t.frame_counter_++);
    if (t.state_ != SDVL::STATE_RUNNING) {
      // bootstrap replacement (SaveFirstFrame/SaveSecondFrame are out of scope): first frame = keyframe at first_pose
      t.current_frame_->SetPose(t.first_pose_);
      t.current_frame_->SetKeyframe();
      t.map_->AddKeyframe(t.current_frame_, false);
      t.pending_kf_ = t.current_frame_;
      t.last_frame_ = t.current_frame_;
      t.last_kf_ = t.current_frame_;
      t.state_ = SDVL::STATE_RUNNING;
      t.track_.valid = false;
      st.state = 0;
      st.keyframe = 1;
    } else if (t.lost_frames_ >= 3) {
      st.state = 2;
      lost.push_back(i);  // relocalize = lost_frames_ >= 3, sdvl.cc:73
    } else {
      st.state = 2;
      t.current_frame_->SetPose(SE3::Exp(t.vel_) * t.last_frame_->GetPose());  // SetMotionModel, sdvl.cc:278-281
      run.push_back(i);
    }
  }
  // ---- Relocalize (sdvl.cc:73-89,205-238) for the trackers that lost their map, in THIS step: their alignments against their
  // keyframes and their searches are extra launches of the same submission queue; the trackers that found a keyframe join the
  // tracked step below with a table built from that keyframe (ProcessFrame(last_frame_ = last_kf_ = the keyframe), sdvl.cc:86-93),
  // the others sit this frame out (sdvl.cc:91: !relocalize).  Nobody else's tables are touched.
  bool detected_now = false;
  vector<int> host_run;  // relocalised onto a keyframe no table can express: tracked through the per-object calls, alone
  if (!lost.empty()) {
    clk.reset(new StageClock(ST_RELOCALIZE));
    if (!detected_ahead) {  // Reproject searches the new frame's corners
      Frame::DetectBatch(frames, Config::NumFeatures());
      detected_now = true;
    }
    vector<char> found;
    RelocalizeLost(lost, stats, &found);
    vector<int> again;
    for (size_t q = 0; q < lost.size(); q++)
      if (found[q]) again.push_back(lost[q]);
    if (!again.empty()) {
      vector<char> built;
      UploadTables(again, &built);
      for (size_t q = 0; q < again.size(); q++) {
        SDVL &t = *trk_[again[q]];
        t.current_frame_->SetPose(SE3::Exp(t.vel_) * t.last_frame_->GetPose());  // SetMotionModel, sdvl.cc:278-281 (vel_ = 0)
        (built[q] ? run : host_run).push_back(again[q]);
      }
      std::sort(run.begin(), run.end());
    }
    clk.reset(new StageClock(ST_PRELUDE));
  }
  const int R = static_cast<int>(run.size());

  vector<char> decision(B, 0);  // per tracker: 0 = tracking lost / not tracked, 1 = ordinary frame, 2 = new keyframe
  vector<shared_ptr<Frame>> kfs;
  vector<int> kf_owner;
  bool filter_begun = false;
  if (R > 0) {
    // ---- the step: jobs, the cell order SelectPoints would shuffle (feature_align.cc:103), the draws SelectInliers would make
    clk.reset(new StageClock(ST_IMAGE_ALIGN));
    const int max_its = Config::MaxRansacIts();
    tr_jobs_.resize(R);
    tr_rank_.resize(static_cast<size_t>(R) * cells);
    tr_rand_.resize(static_cast<size_t>(R) * max_its);
    ParallelFor(R, [&](int k) {
      SDVL &t = *trk_[run[k]];
      sdvl_track_job &jb = tr_jobs_[k];
      jb.tracker = run[k];
      jb.feat_buf = t.track_.feat_buf;
      jb.last = t.last_frame_->device();
      jb.cur = t.current_frame_->device();
      const SE3 T = t.current_frame_->GetPose() * t.last_frame_->GetPose().Inverse();  // image_align.cc:66
      T.ToArray(jb.T);
      t.last_frame_->GetPose().ToArray(jb.last_pose);
      jb.frame_id = t.current_frame_->GetID();
      jb.max_matches = t.feature_align_.MaxMatches();
      t.feature_align_.ShuffleCellRanks(tr_rank_.data() + static_cast<size_t>(k) * cells);
      t.feature_align_.PeekRand(max_its, tr_rand_.data() + static_cast<size_t>(k) * max_its);
    });
    sdvl_track_params prm;
    prm.align = AlignParams(false);
    prm.search = SearchParams();
    prm.pose = FeatureAlign::PoseParams(camera);
    prm.cell_size = Config::CellSize();
    prm.patch_size = Config::PatchSize();
    prm.max_failed = Config::MaxFailed();
    prm.pad_ = 0;
    const sdvl_camera cam = camera.abi();
    // Round 5: a SMALL batch (a lone camera's HandleFrame above all) is a chain of launches that each fill a sliver of the chip: its
    // detection (FAST, selection: needs the pyramid only) runs on the context's side stream BESIDE the alignment instead of behind it
    // (sdvl_ctx_fork_*); a farm's batches keep one stream per group — there the other groups are the company, and a whole-context wait
    // inside a fork would block the worker's other groups (B <= 4: the batches that also spin on their waits).
    const bool fork_detect = !detected_ahead && !detected_now && B <= 4;
    if (fork_detect) dev_->Check(sdvl_ctx_fork_mark(dev_->ctx()), "sdvl_ctx_fork_mark");  // the pyramids are queued: the side chain starts here
    dev_->Check(sdvl_track_align(dev_->ctx(), track_, R, tr_jobs_.data(), tr_rank_.data(), tr_rand_.data(), &cam, &prm), "sdvl_track_align");
    if (fork_detect) {
      dev_->Check(sdvl_ctx_fork_begin(dev_->ctx()), "sdvl_ctx_fork_begin");
      try {
        Frame::DetectBatch(frames, Config::NumFeatures());
      } catch (...) {  // whatever the detection throws, the context leaves the fork
        (void)sdvl_ctx_fork_end(dev_->ctx());
        throw;
      }
      dev_->Check(sdvl_ctx_fork_end(dev_->ctx()), "sdvl_ctx_fork_end");
    } else if (!detected_ahead && !detected_now) {
      Frame::DetectBatch(frames, Config::NumFeatures());  // FAST + selection run behind the alignment
    }
    clk.reset(new StageClock(ST_SEARCH));
    dev_->Check(sdvl_track_search(dev_->ctx(), track_), "sdvl_track_search");
    // the look-ahead: the next step's pyramids and detection go behind this step's chain, ahead of its keyframe kernels
    if (!next_imgs_.empty() && R == B && static_cast<int>(next_imgs_.size()) == B) {
      bool device_images = true;
      for (const Image &im : next_imgs_) device_images = device_images && im.dev_src != nullptr && !im.transient;
      if (device_images) {
        Frame::CreateBatch(&camera, &trk_[0]->orb_detector_, next_imgs_, false, Config::NumFeatures(), &ahead_frames_, &pfor);
        Frame::DetectBatch(ahead_frames_, Config::NumFeatures());
        ahead_src_.clear();
        for (const Image &im : next_imgs_) ahead_src_.push_back(im.dev_src);
      }
    }
    tr_res_.resize(R);
    dev_->Check(sdvl_track_collect(dev_->ctx(), track_, R, tr_res_.data()), "sdvl_track_collect");

    // ---- results: pose, counters, the rand() stream, deletions, motion model, tracking quality, keyframe decision
    clk.reset(new StageClock(ST_FINISH));
    for (int k = 0; k < R; k++)
      if (tr_res_[k].status != 0) throw std::runtime_error("sdvl_track: a table capacity was exceeded on the device");
    ParallelFor(R, [&](int k) {
      const int i = run[k];
      SDVL &t = *trk_[i];
      FrameStats &st = stats[i];
      const sdvl_track_result &r = tr_res_[k];
      SDVL::TrackState &ts = t.track_;
      st.align_meas = r.align_meas;
      st.align_iters = r.align_iters;
      st.align_features = r.n_features;
      st.search_requests = r.n_requests;
      st.lk_iters = r.lk_iters;
      st.n_corners = r.n_corners;
      t.current_frame_->SetPose(SE3::FromArray(r.pose));
      t.current_frame_->SetRegistered();  // the step wrote (view, final pose) into the registry
      t.current_frame_->SetFlatFeatures(sdvl_track_features(track_, k), r.matches, ts.points);
      t.current_frame_->SetSceneDepthHint(r.scene_depth);
      const sdvl_track_point_stat *ps = sdvl_track_stats(track_, k);
      ts.stats.assign(ps, ps + ts.points->size());
      ts.stats_dirty = true;
      if (dynamic_cast<MapperMap *>(t.map_)) SyncStats(t);  // the mapper's depth filter reads and resets the failure counts
      t.feature_align_.AdvanceRand(r.n_draws);
      t.feature_align_.SetTrackedCounts(r.matches, r.attempts, r.n_inliers, r.n_outliers);
      t.matches_ = r.matches;
      t.attempts_ = r.attempts;
      st.inliers = r.n_inliers;
      st.outliers = r.n_outliers;
      if (r.n_deleted > 0)  // points that crossed MaxFailed: Map::DeletePoint (feature_align.cc:141-142), emptied in the epilogue
        for (size_t p = 0; p < ts.points->size(); p++)
          if ((ts.stats[p].status & 0x100) && !(*ts.points)[p]->ToDelete()) {
            (*ts.points)[p]->SetDeviceTrashed();  // the row carries the deletion: nothing to rebuild when the trash is emptied
            t.map_->DeletePoint((*ts.points)[p]);
          }
      {  // GetMotionModel, sdvl.cc:266-276
        const SE3 mov = t.current_frame_->GetPose() * t.last_frame_->GetPose().Inverse();
        const Vector6d vel = SE3::Log(mov);
        for (int c = 0; c < 6; c++) t.vel_[c] = 0.9 * (0.5 * vel[c] + 0.5 * t.vel_[c]);
      }
      t.CalcTrackingQuality(t.matches_, t.attempts_);
      if (t.tracking_quality_ != SDVL::TRACKING_BAD)
        decision[i] = (t.tracking_quality_ == SDVL::TRACKING_GOOD && t.map_->NeedKeyframe(t.current_frame_, t.matches_)) ? 2 : 1;
    });
  }
  // (a relocalised tracker whose keyframe no table can hold: the per-object calls, behind the batch's chain on the same stream)
  for (int i : host_run) decision[i] = static_cast<char>(TrackOnHost(*trk_[i], &stats[i]));
  vector<int> act(run);  // every tracker that executed ProcessFrame this step
  act.insert(act.end(), host_run.begin(), host_run.end());
  if (!act.empty()) {
    // the keyframes are known: queue their FilterCorners inputs now; the bookkeeping below runs meanwhile
    clk.reset(new StageClock(ST_POSE));
    {
      vector<char> fresh(B, 0);
      for (int i : act)
        if (decision[i] == 2 && !dynamic_cast<MapperMap *>(trk_[i]->map_)) fresh[i] = 1;
      for (int i = 0; i < B; i++) {
        if (trk_[i]->pending_kf_) { kfs.push_back(trk_[i]->pending_kf_); kf_owner.push_back(i); }
        else if (fresh[i]) { kfs.push_back(trk_[i]->current_frame_); kf_owner.push_back(i); }
      }
      if (!kfs.empty()) {
        StageClock fclk(ST_MAPPING);
        // bootstrap frames and the frames of trackers that sat the step out or were tracked by hand: their counts did not ride along
        if (static_cast<int>(run.size()) < B) FetchCornerCounts(frames, stats);
        Frame::FilterCornersBegin(kfs);
        filter_begun = true;
      }
    }
    vector<int> r_matches(B, 0);
    for (int k = 0; k < R; k++) r_matches[run[k]] = tr_res_[k].matches;
    ParallelFor(static_cast<int>(act.size()), [&](int a) {
      const int i = act[a];
      SDVL &t = *trk_[i];
      FrameStats &st = stats[i];
      if (decision[i] == 0) return;  // tracking lost: last_frame and its table stay
      if (decision[i] == 2) {
        // Round 4, plane-map trackers: the keyframe's ~190 matched features STAY flat records and its table stays the one the step has
        // just left on the device — what the keyframe adds (the points the map seeds in its empty cells, ~60) is appended to both
        // sides after the seeding (EpilogueAndMapper -> AppendSeeds -> sdvl_track_append).  Until round 3 every keyframe turned its
        // features into objects, linked them to their points, and the whole table was rebuilt from the objects before the next step:
        // 8 of a group-step's 12 ms of host time (MaterializeFeatures, BuildTable, malloc / mprotect under them), ~100 KB of fresh
        // memory per keyframe.  The objects appear the first time somebody asks (Frame::GetFeatures(): API callers, relocalisation,
        // the host-driven path, a table rebuild when the rows run out), the points learn of their observations then.
        const bool plane = !dynamic_cast<MapperMap *>(t.map_);
        const size_t rows = t.track_.points ? t.track_.points->size() : static_cast<size_t>(track_cap_);
        const bool room = static_cast<int>(rows) + track_cells_ <= track_cap_ && r_matches[i] + track_cells_ <= track_cap_;
        if (plane && room && t.current_frame_->HasFlatFeatures()) {
          t.current_frame_->LinkPointsOnMaterialize();
          t.track_.feat_buf ^= 1;  // the matches the step left in the other buffer are last_frame's features now
          t.track_.append_pending = true;
          t.track_.first_seed = t.current_frame_->ObjectFeatures().size();
        } else {
          // the frame becomes part of the map: its features and the points behind them turn into objects
          SyncStats(t);
          vector<shared_ptr<Feature>> &features = t.current_frame_->GetFeatures();
          for (auto it = features.begin(); it != features.end(); it++)
            if (Point *p = (*it)->GetPointRaw()) p->AddFeature(*it);
          t.track_.valid = false;  // seeding / the mapper add features: the table is rebuilt from the keyframe
        }
        t.current_frame_->ClearSceneDepthHint();  // a keyframe's features lose the points EmptyTrash deletes (map.cc:207-259)
        t.current_frame_->SetKeyframe();
        t.map_->AddKeyframe(t.current_frame_);
        t.last_kf_ = t.current_frame_;
        t.map_->LimitKeyframes(t.current_frame_);  // sdvl.cc:114
        if (plane) t.pending_kf_ = t.current_frame_;
        st.keyframe = 1;
      } else {
        t.map_->AddFrame(t.current_frame_);
        if (t.current_frame_->HasFlatFeatures()) t.track_.feat_buf ^= 1;  // the matches the step left in the other buffer are last_frame's features now
      }
      t.last_frame_ = t.current_frame_;
    });
  }
  clk.reset();
  next_imgs_.clear();  // a look-ahead belongs to ONE step, whether that step could use it (R == B tracked frames) or not (bootstrap)
  if (R == 0 && !detected_ahead && !detected_now) {  // bootstrap-only step: the new keyframes still need their corners
    // (a small batch detects on the side stream here too: the stream and its queue exist by the time a tracked frame forks)
    const bool fork0 = B <= 4;
    if (fork0) {
      dev_->Check(sdvl_ctx_fork_mark(dev_->ctx()), "sdvl_ctx_fork_mark");
      dev_->Check(sdvl_ctx_fork_begin(dev_->ctx()), "sdvl_ctx_fork_begin");
    }
    try {
      Frame::DetectBatch(frames, Config::NumFeatures());
    } catch (...) {
      if (fork0) (void)sdvl_ctx_fork_end(dev_->ctx());
      throw;
    }
    if (fork0) dev_->Check(sdvl_ctx_fork_end(dev_->ctx()), "sdvl_ctx_fork_end");
  }
  // a mapper that looks at the frames it was given (MapperMap: scene depth of every frame, feature lists of keyframes)
  // materialises them on demand; a keyframe's table is rebuilt afterwards in any case
  EpilogueAndMapper(frames, stats, &kfs, &kf_owner, filter_begun);
  // With the reference's mapper the tracker follows CANDIDATES too (Map::InitCandidates links them to the keyframe at once,
  // map.cc:379-390), and the depth filter rewrites their inverse depth, variance, position, failure count and finally
  // `fixed` after every frame (point.cc:64-100,164-178).  With the filter on the device (depth_filter_kernel) the rows are
  // patched where they lie, in the same stream, and the table stays valid; it is rebuilt from the objects when the filter ran
  // on the host or on another context (threaded mode), and whenever a point with a row died behind the device's back.
  for (int i = 0; i < B; i++) {
    SDVL &t = *trk_[i];
    MapperMap *m = dynamic_cast<MapperMap *>(t.map_);
    if (m && (m->IsThreaded() || !MapperMap::DeviceFilter())) t.track_.valid = false;
    if (t.map_->TakeTablesDirty()) t.track_.valid = false;
    // tracking lost: last_frame stays, but the mapper has had it and trashed it — its features are gone from the host
    // (MapperMap::EmptyTrash, map.cc:207-259) and the next step must not find them in the table either
    if (t.last_frame_ && t.last_frame_->FeaturesRemoved()) t.track_.valid = false;
  }
  return true;
}

// SDVL::HandleFrame (sdvl.cc:55-130) for B trackers
void SDVLBatch::HandleFrames(const vector<Image> &imgs, FrameStats *stats) {
  const int B = static_cast<int>(trk_.size());
  if (static_cast<int>(imgs.size()) != B) throw std::runtime_error("SDVLBatch::HandleFrames: one image per tracker");
  Device::SetCurrent(dev_);
  StageTimes::Active() = &stage_times;
  stage_times.steps++;
  const auto t_begin = std::chrono::steady_clock::now();
  // A transient image (level 0 aliases a ring slot the caller rewrites) is safe only while nobody reads a frame's level 0 after
  // the frame's own step.  Two configurations break that:
  //  * a mapper THREAD consumes the frames handed to AddFrame after the step, on its own stream: such a batch takes a copy of
  //    every image when the frame is built (the aliasing is refused);
  //  * SDVL.min_alignLevel 0: the next step's image alignment reads last_frame's level 0 (image_align.cc:212): last_frame copies
  //    its image out at the end of the step like a fresh keyframe does.
  bool threaded_mapper = false;
  for (SDVL *t : trk_) {
    MapperMap *m = dynamic_cast<MapperMap *>(t->map_);
    if (m && m->IsThreaded()) threaded_mapper = true;
  }
  vector<Image> owned;
  if (threaded_mapper) {
    bool any = false;
    for (const Image &im : imgs) any = any || im.transient;
    if (any) {
      owned = imgs;
      for (Image &im : owned) {
        if (im.transient) { im.transient = false; im.borrow = false; }
      }
    }
  }
  const vector<Image> &in = owned.empty() ? imgs : owned;
  if (!HandleFramesTracked(in, stats)) {
    next_imgs_.clear();  // the host-driven path takes no look-ahead, and one that was queued is dropped
    ahead_frames_.clear();
    ahead_src_.clear();
    SyncHostState();  // Feature lists and Point counters catch up with the device; the tables are rebuilt when tracking returns
    HandleFramesGeneric(in, stats);
  }
  {  // frames that became keyframes while their image sat in an input-ring slot keep a copy; everybody else never copied level 0
     // (the step's own frame pointer is gone by now: a fresh keyframe is the tracker's last_kf_, sdvl.cc:113)
    const bool keep_last = Config::MinAlignLevel() == 0;
    vector<shared_ptr<Frame>> keep;
    for (SDVL *t : trk_) {
      if (t->last_kf_ && t->last_kf_->ImageTransient()) keep.push_back(t->last_kf_);
      if (keep_last && t->last_frame_ && t->last_frame_ != t->last_kf_ && t->last_frame_->ImageTransient()) keep.push_back(t->last_frame_);
    }
    if (!keep.empty()) Frame::OwnImages(keep);
  }
  stage_times.t[ST_TOTAL] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  StageTimes::Active() = nullptr;
}

// the host-driven form, stage by stage: every request assembled here, results replayed here
void SDVLBatch::HandleFramesGeneric(const vector<Image> &imgs, FrameStats *stats) {
  const bool device_pose = DevicePose();
  const int B = static_cast<int>(trk_.size());
  for (SDVL *t : trk_) t->track_.valid = false;  // this step changes last_frame's features behind the tables' back
  const std::function<void(int, std::function<void(int)>)> pfor = [this](int n, std::function<void(int)> fn) { ParallelFor(n, fn); };

  // ---- stage 0: Frame construction (pyramid + FAST + selection + ORB), sdvl.cc:59
  vector<shared_ptr<Frame>> frames;
  // pyramids now; FAST + selection + ORB are queued BEHIND the image alignment (stage 1): alignment only needs the
  // pyramids, so its results reach the host ~0.4 ms earlier and detection overlaps the host's prepare stage
  Frame::CreateBatch(trk_[0]->camera_, &trk_[0]->orb_detector_, imgs, false, Config::NumFeatures(), &frames, &pfor);
  bool detected = false;
  for (int i = 0; i < B && !detected; i++)
    if (trk_[i]->state_ == SDVL::STATE_RUNNING && trk_[i]->lost_frames_ >= 3) {  // Relocalize searches the new frame in the prelude
      Frame::DetectBatch(frames, Config::NumFeatures());
      detected = true;
    }
  vector<int> run;  // trackers that execute ProcessFrame this step
  std::unique_ptr<StageClock> clk(new StageClock(ST_PRELUDE));
  for (int i = 0; i < B; i++) {
    SDVL &t = *trk_[i];
    FrameStats &st = stats[i];
    st = FrameStats();
    st.host_path = 1;
    t.current_frame_ = frames[i];
    t.current_frame_->SetID(t.frame_counter_++);
    if (t.state_ != SDVL::STATE_RUNNING) {
      // bootstrap replacement (SaveFirstFrame/SaveSecondFrame are out of scope): first frame = keyframe at first_pose
      t.current_frame_->SetPose(t.first_pose_);
      t.current_frame_->SetKeyframe();
      t.map_->AddKeyframe(t.current_frame_, false);  // like SaveFirstFrame / SaveSecondFrame (sdvl.cc:144,165): not queued
      t.pending_kf_ = t.current_frame_;              // seeded from the scene plane in the mapping stage
      t.last_frame_ = t.current_frame_;
      t.last_kf_ = t.current_frame_;
      t.state_ = SDVL::STATE_RUNNING;
      st.state = 0;
      st.keyframe = 1;
    } else {
      st.state = 2;
      bool relocalize = t.lost_frames_ >= 3;
      if (relocalize) {
        // Relocalize, sdvl.cc:73-89,205-238.  The alignment of the current frame against EVERY keyframe (each started
        // from that keyframe's pose, fast mode) is independent of the others: one launch with |keyframes| jobs.  The
        // keyframe loop then runs in the reference's order (newest first) over the results; Reproject stays sequential
        // because it draws from rand() and stops at the first keyframe that gathers MinMatches.
        t.map_->SetRelocalizing(true);  // sdvl.cc:80
        for (int k = 0; k < 6; k++) t.vel_[k] = 0.0;
        vector<shared_ptr<Frame>> &kfs = t.map_->GetKeyframes();
        vector<std::pair<shared_ptr<Frame>, shared_ptr<Frame>>> pairs;
        vector<SE3> start, aligned;
        for (auto it = kfs.rbegin(); it != kfs.rend(); it++) {
          pairs.push_back({*it, t.current_frame_});
          start.push_back((*it)->GetPose());
        }
        vector<int> n_meas;
        vector<double> errors;
        ImageAlign::ComputePoseBatch(pairs, true, &n_meas, &errors, nullptr, &start, &aligned);
        for (size_t j = 0; j < pairs.size(); j++) {
          const shared_ptr<Frame> &cframe = pairs[j].first;
          t.current_frame_->SetPose(aligned[j]);
          if (errors[j] >= 0.001) continue;
          t.feature_align_.Reproject(t.current_frame_, cframe, cframe, true);
          t.matches_ = t.feature_align_.GetMatches();
          t.attempts_ = t.feature_align_.GetAttempts();
          if (t.matches_ >= Config::MinMatches()) {
            t.map_->SetRelocalizing(false);  // sdvl.cc:84
            t.last_kf_ = cframe;
            t.last_frame_ = cframe;
            relocalize = false;
            st.relocalized = 1;
            break;
          }
        }
      }
      if (!relocalize) {
        t.current_frame_->SetPose(SE3::Exp(t.vel_) * t.last_frame_->GetPose());  // SetMotionModel, sdvl.cc:278-281
        run.push_back(i);
      }
    }
  }

  // ---- stage 1: ImageAlign for every running tracker, sdvl.cc:185-190
  const int R = static_cast<int>(run.size());
  clk.reset(new StageClock(ST_IMAGE_ALIGN));
  {
    vector<std::pair<shared_ptr<Frame>, shared_ptr<Frame>>> pairs;
    for (int i : run) pairs.push_back({trk_[i]->last_frame_, trk_[i]->current_frame_});
    vector<int> n_meas, iters;
    vector<double> errors;
    const std::function<void()> detect = [&]() { Frame::DetectBatch(frames, Config::NumFeatures()); };
    ImageAlign::ComputePoseBatch(pairs, false, &n_meas, &errors, &iters, nullptr, nullptr, detected ? nullptr : &detect);
    for (int k = 0; k < R; k++) {
      stats[run[k]].align_meas = n_meas[k];
      stats[run[k]].align_iters = iters[k];
      stats[run[k]].align_features = static_cast<int>(pairs[k].first->GetFeatures().size());
    }
  }

  // fresh keyframes of plane-map trackers: their FilterCorners round trip is queued as soon as the keyframe decisions are
  // known (stage 3b) and collected in stage 4, with the rest of the per-tracker bookkeeping in between
  vector<shared_ptr<Frame>> kfs;
  vector<int> kf_owner;
  bool filter_begun = false;
  // ---- stage 2: FeatureAlign::Reproject, sdvl.cc:193 — all candidates of all trackers in one launch
  {
    clk.reset(new StageClock(ST_PREPARE));
    vector<sdvl_search_req> &reqs = scratch_reqs_;  // keeps its capacity from step to step
    reqs.clear();
    vector<size_t> begin(R + 1, 0);
    bool packed = false, chain = false;
    vector<sdvl_chain_frame> chain_frames;
    vector<int> chain_obs_begin(R + 1, 0);
    FeatureAlign::PackedSink sink;
    if (threads_ <= 1 && R > 0) {
      // one host thread per batch (the farm's case): every tracker writes its requests, already in the device layout,
      // straight into the pinned staging area of one search batch
      int cap = 0;
      for (int k = 0; k < R; k++) cap += static_cast<int>(trk_[run[k]]->last_frame_->GetFeatures().size());
      sink.ctx = dev_->ctx();
      sink.cap = cap;
      // unique per DEVICE, not per SDVLBatch: frames remember the slot they got in a batch by its id, and SDVL::HandleFrame
      // makes a fresh one-tracker SDVLBatch for every frame
      sink.batch_id = ++dev_->search_batch_counter;
      // search -> match selection -> RANSAC + pose refinement as ONE submission (sdvl_search_run_chain): the device replays
      // the second half of SelectPoints itself, so the pose kernels run while this thread does the same replay for its
      // own bookkeeping (features, point statistics) instead of starting after it
      chain = device_pose && Config::MaxRansacPoints() <= 8;
      for (int k = 0; k < R && chain; k++)
        if (trk_[run[k]]->feature_align_.MaxMatches() > FeatureAlign::kMaxDevicePoseObs) chain = false;
      if (chain) {
        if (scratch_points_.size() < 3 * static_cast<size_t>(cap)) scratch_points_.resize(3 * static_cast<size_t>(cap));
        sink.points = scratch_points_.data();
      }
      dev_->Check(sdvl_search_begin(sink.ctx, cap, &sink.reqs), "sdvl_search_begin");
      for (int k = 0; k < R; k++) {
        SDVL &t = *trk_[run[k]];
        begin[k] = static_cast<size_t>(sink.count);
        t.feature_align_.PrepareReprojectPacked(t.current_frame_, t.last_frame_, false, &sink);
      }
      packed = true;
      if (sink.count == 0) chain = false;
      if (chain) {
        const int max_its = Config::MaxRansacIts();
        chain_frames.resize(R);
        chain_cand_req_.clear();
        chain_cand_first_.clear();
        chain_rand_.clear();
        for (int k = 0; k < R; k++) {
          SDVL &t = *trk_[run[k]];
          sdvl_chain_frame &cf = chain_frames[k];
          cf.cand_begin = static_cast<int32_t>(chain_cand_req_.size());
          t.feature_align_.EmitChainCandidates(static_cast<int>(begin[k]), &chain_cand_req_, &chain_cand_first_);
          cf.cand_end = static_cast<int32_t>(chain_cand_req_.size());
          cf.max_matches = t.feature_align_.MaxMatches();
          cf.rand_begin = static_cast<int32_t>(chain_rand_.size());
          t.feature_align_.PeekRand(max_its, &chain_rand_);
          t.current_frame_->GetPose().ToArray(cf.pose);
          chain_obs_begin[k] = k == 0 ? 0 : chain_obs_begin[k - 1] + chain_frames[k - 1].max_matches;
        }
      }
    } else {
      vector<vector<sdvl_search_req>> per(R);
      ParallelFor(R, [&](int k) {
        SDVL &t = *trk_[run[k]];
        t.feature_align_.PrepareReproject(t.current_frame_, t.last_frame_, false, &per[k]);
      });
      for (int k = 0; k < R; k++) {
        begin[k] = reqs.size();
        reqs.insert(reqs.end(), per[k].begin(), per[k].end());
      }
    }
    begin[R] = packed ? static_cast<size_t>(sink.count) : reqs.size();
    vector<sdvl_search_res> res;
    clk.reset(new StageClock(ST_SEARCH));
    if (packed) {
      res.resize(std::max(1, sink.count));
      const sdvl_camera cam = trk_[run[0]]->camera_->abi();
      const sdvl_search_params sp = SearchParams();
      if (chain) {
        const sdvl_pose_params pp = FeatureAlign::PoseParams(*trk_[run[0]]->camera_);
        dev_->Check(sdvl_search_run_chain(sink.ctx, sink.count, &cam, &sp, res.data(), R, chain_frames.data(),
                                          static_cast<int>(chain_cand_req_.size()), chain_cand_req_.data(), chain_cand_first_.data(),
                                          scratch_points_.data(), static_cast<int>(chain_rand_.size()), chain_rand_.data(), &pp),
                    "sdvl_search_run_chain");
      } else {
        dev_->Check(sdvl_search_run(sink.ctx, sink.count, &cam, &sp, res.data()), "sdvl_search_run");
      }
    } else if (R > 0) {
      Matcher::SearchPoints(dev_, reqs, *trk_[run[0]]->camera_, &res);
    }
    clk.reset(new StageClock(ST_FINISH));
    // ---- stage 3: replay of SelectPoints, sdvl.cc:193
    vector<FeatureAlign::PoseBatch> pb(threads_ <= 1 ? 0 : R);
    FeatureAlign::PoseBatch &all = scratch_pose_;
    all.jobs.clear(); all.obs.clear(); all.rand_idx.clear(); all.nits.clear();
    vector<int> job_of(R, -1);
    vector<char> on_device(R, 0);
    ParallelFor(R, [&](int k) {
      const int i = run[k];
      SDVL &t = *trk_[i];
      FrameStats &st = stats[i];
      st.search_requests = static_cast<int>(begin[k + 1] - begin[k]);
      for (size_t q = begin[k]; q < begin[k + 1]; q++) st.lk_iters += res[q].lk_its;
      t.feature_align_.FinishSelect(t.current_frame_, res.data() + begin[k], !chain);
      t.matches_ = t.feature_align_.GetMatches();
      t.attempts_ = t.feature_align_.GetAttempts();
      if (chain) {
        job_of[k] = k;  // its pose job is already running
      } else if (device_pose) {
        if (threads_ <= 1) {  // sequential: straight into the shared batch
          const int job = static_cast<int>(all.jobs.size());
          on_device[k] = t.feature_align_.EmitPoseJob(t.current_frame_, &all) ? 1 : 0;
          if (on_device[k]) job_of[k] = job;
        } else {
          on_device[k] = t.feature_align_.EmitPoseJob(t.current_frame_, &pb[k]) ? 1 : 0;
        }
      }
    });
    // ---- stage 3b: RANSAC + pose refinement (feature_align.cc:73-82,152-243), one launch pair for every tracker
    clk.reset(new StageClock(ST_POSE));
    vector<sdvl_pose_result> pres;
    vector<int32_t> lists;
    if (threads_ > 1)
      for (int k = 0; k < R; k++)
        if (on_device[k]) {
          job_of[k] = static_cast<int>(all.jobs.size());
          all.Append(pb[k]);
        }
    if (chain) {
      int obs_total = 0;
      for (int k = 0; k < R; k++) obs_total += chain_frames[k].max_matches;
      pres.resize(R);
      lists.resize(obs_total + 1);
      vector<int32_t> n_obs(R);
      dev_->Check(sdvl_search_chain_end(dev_->ctx(), R, pres.data(), n_obs.data(), lists.data()), "sdvl_search_chain_end");
      for (int k = 0; k < R; k++)
        if (n_obs[k] != trk_[run[k]]->feature_align_.FoundCount())
          throw std::runtime_error("SDVLBatch: the device's match selection disagrees with the host replay (" + std::to_string(n_obs[k]) + " vs " +
                                   std::to_string(trk_[run[k]]->feature_align_.FoundCount()) + " matches)");
    } else if (!all.jobs.empty()) {
      pres.resize(all.jobs.size());
      lists.resize(all.obs.size() + 1);
      const sdvl_pose_params pp = FeatureAlign::PoseParams(*trk_[run[0]]->camera_);
      dev_->Check(sdvl_pose_from_matches(dev_->ctx(), static_cast<int>(all.jobs.size()), all.jobs.data(), static_cast<int>(all.obs.size()),
                                         all.obs.data(), static_cast<int>(all.rand_idx.size()), all.rand_idx.data(),
                                         static_cast<int>(all.nits.size()), all.nits.data(), &pp, pres.data(), lists.data()),
                  "sdvl_pose_from_matches");
    }
    vector<char> decision(R, 0);  // 0 = tracking lost, 1 = ordinary frame, 2 = new keyframe
    ParallelFor(R, [&](int k) {
      const int i = run[k];
      SDVL &t = *trk_[i];
      FrameStats &st = stats[i];
      if (job_of[k] >= 0) {
        t.feature_align_.CommitPose(t.current_frame_, pres[job_of[k]], lists.data() + (chain ? chain_obs_begin[k] : all.jobs[job_of[k]].obs_begin));
      } else {
        t.feature_align_.SelectInliers(t.current_frame_);
        t.feature_align_.OptimizePose(t.current_frame_);
      }
      st.inliers = t.feature_align_.GetInliers();
      st.outliers = t.feature_align_.GetOutliers();
      {  // GetMotionModel, sdvl.cc:266-276
        const SE3 mov = t.current_frame_->GetPose() * t.last_frame_->GetPose().Inverse();
        const Vector6d vel = SE3::Log(mov);
        for (int c = 0; c < 6; c++) t.vel_[c] = 0.9 * (0.5 * vel[c] + 0.5 * t.vel_[c]);
      }
      t.CalcTrackingQuality(t.matches_, t.attempts_);
      if (t.tracking_quality_ != SDVL::TRACKING_BAD)
        decision[k] = (t.tracking_quality_ == SDVL::TRACKING_GOOD && t.map_->NeedKeyframe(t.current_frame_, t.matches_)) ? 2 : 1;
    });
    // the keyframes are known: queue their FilterCorners inputs (Shi-Tomasi, descriptors, one gather + copy) now; the
    // bookkeeping below (feature lists of the points, keyframe graph, retiring the previous frames) runs meanwhile
    if (threads_ <= 1) {
      vector<char> fresh(B, 0);
      for (int k = 0; k < R; k++)
        if (decision[k] == 2 && !dynamic_cast<MapperMap *>(trk_[run[k]]->map_)) fresh[run[k]] = 1;
      for (int i = 0; i < B; i++) {
        if (trk_[i]->pending_kf_) { kfs.push_back(trk_[i]->pending_kf_); kf_owner.push_back(i); }
        else if (fresh[i]) { kfs.push_back(trk_[i]->current_frame_); kf_owner.push_back(i); }
      }
      if (!kfs.empty()) {
        StageClock fclk(ST_MAPPING);
        FetchCornerCounts(frames, stats);  // before the filter round trip: it shares the context's result buffers
        Frame::FilterCornersBegin(kfs);
        filter_begun = true;
      }
    }
    ParallelFor(R, [&](int k) {
      const int i = run[k];
      SDVL &t = *trk_[i];
      FrameStats &st = stats[i];
      if (decision[k] != 0) {
        if (decision[k] == 2) {
          vector<shared_ptr<Feature>> &features = t.current_frame_->GetFeatures();
          for (auto it = features.begin(); it != features.end(); it++)
            if (Point *p = (*it)->GetPointRaw()) p->AddFeature(*it);
          t.current_frame_->SetKeyframe();
          t.map_->AddKeyframe(t.current_frame_);
          t.last_kf_ = t.current_frame_;
          t.map_->LimitKeyframes(t.current_frame_);  // sdvl.cc:114
          if (!dynamic_cast<MapperMap *>(t.map_)) t.pending_kf_ = t.current_frame_;  // plane map stub: seed every keyframe
          st.keyframe = 1;
        } else {
          t.map_->AddFrame(t.current_frame_);
        }
        t.last_frame_ = t.current_frame_;
      }
    });
  }

  clk.reset();
  EpilogueAndMapper(frames, stats, &kfs, &kf_owner, filter_begun);
}

// corner counts of this step's frames (they rode along with the detection kernels: no round trip once any wait on the stream
// has returned since): the statistics need them, and with them known the keyframe round trip returns rows of exactly the
// right length
void SDVLBatch::FetchCornerCounts(const vector<shared_ptr<Frame>> &frames, FrameStats *stats) {
  const int B = static_cast<int>(frames.size());
  vector<sdvl_frame *> devs(B);
  vector<int32_t> counts(B);
  for (int i = 0; i < B; i++) devs[i] = frames[i]->device();
  dev_->Check(sdvl_frames_corner_counts(dev_->ctx(), B, devs.data(), counts.data()), "sdvl_frames_corner_counts");
  for (int i = 0; i < B; i++) stats[i].n_corners = counts[i];
}

// what follows the tracking decisions in both forms of a step: the keyframes' FilterCorners round trip + seeding, the
// per-tracker epilogue, and SDVL::Mapping() of sequential mode
void SDVLBatch::EpilogueAndMapper(const vector<shared_ptr<Frame>> &frames, FrameStats *stats, vector<shared_ptr<Frame>> *kfs_io,
                                  vector<int> *kf_owner_io, bool filter_begun) {
  // ---- stage 4: mapper stand-in for fresh keyframes (sequential mode, main.cc:148-149): one K3 launch for all
  std::unique_ptr<StageClock> clk(new StageClock(ST_MAPPING));
  const int B = static_cast<int>(trk_.size());
  vector<shared_ptr<Frame>> &kfs = *kfs_io;
  vector<int> &kf_owner = *kf_owner_io;
  if (!filter_begun) FetchCornerCounts(frames, stats);
  {
    if (!filter_begun) {
      kfs.clear();
      kf_owner.clear();
      for (int i = 0; i < B; i++)
        if (trk_[i]->pending_kf_) { kfs.push_back(trk_[i]->pending_kf_); kf_owner.push_back(i); }
      Frame::FilterCornersBegin(kfs);
    }
    if (!kfs.empty()) {
      const std::function<void(int, const std::function<void(int)> &)> pfor = [this](int n, const std::function<void(int)> &fn) { ParallelFor(n, fn); };
      Frame::FilterCornersEnd(kfs, &pfor);
      vector<char> appended(kfs.size(), 0);
      ParallelFor(static_cast<int>(kfs.size()), [&](int k) {
        SDVL &t = *trk_[kf_owner[k]];
        PlaneMap *pm = dynamic_cast<PlaneMap *>(t.map_);
        if (pm) pm->SeedFromFiltered(kfs[k]);  // other Map implementations run their own mapper on AddKeyframe
        t.pending_kf_ = nullptr;
        if (t.track_.append_pending) {
          t.track_.append_pending = false;
          if (t.track_.valid && AppendSeeds(t, kfs[k])) appended[k] = 1;
          else t.track_.valid = false;  // something the rows cannot express: the table is rebuilt from the objects
        }
      });
      {  // the new rows of all keyframes of the step in ONE submission
        vector<int32_t> up_trk, up_buf, up_np, up_nf;
        tr_up_points_.clear();
        tr_up_feats_.clear();
        for (size_t k = 0; k < kfs.size(); k++) {
          if (!appended[k]) continue;
          SDVL::TrackState &ts = trk_[kf_owner[k]]->track_;
          up_trk.push_back(kf_owner[k]);
          up_buf.push_back(ts.feat_buf);
          up_np.push_back(static_cast<int32_t>(ts.up_points.size()));
          up_nf.push_back(static_cast<int32_t>(ts.up_feats.size()));
          tr_up_points_.insert(tr_up_points_.end(), ts.up_points.begin(), ts.up_points.end());
          tr_up_feats_.insert(tr_up_feats_.end(), ts.up_feats.begin(), ts.up_feats.end());
        }
        if (!up_trk.empty())
          dev_->Check(sdvl_track_append(dev_->ctx(), track_, static_cast<int>(up_trk.size()), up_trk.data(), up_buf.data(), up_np.data(), tr_up_points_.data(),
                                        up_nf.data(), tr_up_feats_.data()), "sdvl_track_append");
      }
    }
  }

  clk.reset(new StageClock(ST_EPILOGUE));
  ParallelFor(B, [&](int i) {
    SDVL &t = *trk_[i];
    FrameStats &st = stats[i];
    st.quality = static_cast<int>(t.tracking_quality_);
    st.matches = t.matches_;
    st.attempts = t.attempts_;
    t.current_frame_->GetPose().ToArray(st.pose);
    t.stats_ = st;
    t.current_frame_ = nullptr;
    t.map_->EmptyTrash();  // sdvl.cc:127
  });
  // ---- stage 5: SDVL::Mapping() of sequential mode (main.cc:148-149) for the trackers that own a real mapper: the phases of
  // Map::UpdateMap run in lock step, every phase's SearchPoint requests of ALL trackers in one K7 launch
  clk.reset(new StageClock(ST_MAPPER));
  {
    vector<MapperMap *> mm;
    {
      StageClock begin_clk(ST_MAP_BEGIN);
      for (int i = 0; i < B; i++) {
        MapperMap *m = dynamic_cast<MapperMap *>(trk_[i]->map_);
        if (m && !m->IsThreaded() && m->BeginUpdate()) mm.push_back(m);  // a started mapper thread does its own UpdateMap
      }
    }
    const int M = static_cast<int>(mm.size());
    if (M > 0) {
      const Camera &cam = *trk_[0]->camera_;
      vector<vector<sdvl_search_req>> per(M);
      vector<sdvl_search_req> reqs;
      vector<sdvl_search_res> res;
      vector<size_t> begin(M + 1, 0);
      // UpdateCandidates with the depth filter on the device: one filter state per request, outcomes beside the search results;
      // the rows of the points the trackers follow are patched in the same submission (same context, same stream)
      const bool dev_filter = MapperMap::DeviceFilter();
      vector<vector<sdvl_depth_state>> per_st(M);
      vector<sdvl_depth_state> states;
      vector<sdvl_depth_out> fout;
      const sdvl_depth_params fparams = mm[0]->FilterParams();
      auto launch = [&](bool filter = false) {  // search everything the maps emitted, leave the offsets in `begin`
        // the requests go from the maps' lists straight into the context's pinned batch (device layout, frames by slot)
        size_t total = 0;
        for (int k = 0; k < M; k++) {
          begin[k] = total;
          total += per[k].size();
        }
        begin[M] = total;
        states.clear();
        res.resize(std::max<size_t>(total, 1));   // every record is written by the call below (no clearing pass over MBs)
        fout.resize(std::max<size_t>(filter ? total : 0, 1));
        if (total > 0) {
          sdvl_ctx *ctx = dev_->ctx();
          sdvl_search_req_packed *packed = nullptr;
          dev_->Check(sdvl_search_begin(ctx, static_cast<int>(total), &packed), "sdvl_search_begin");
          size_t o = 0;
          for (int k = 0; k < M; k++) {
            for (const sdvl_search_req &r : per[k]) {
              sdvl_search_req_packed &d = packed[o++];
              d.cur = sdvl_search_slot(ctx, r.cur, r.cur_pose);
              d.ref = sdvl_search_slot(ctx, r.ref, r.ref_pose);
              if (d.cur < 0 || d.ref < 0) dev_->Check(d.cur < 0 ? d.cur : d.ref, "sdvl_search_slot");
              d.level = r.level; d.fixed = r.fixed;
              d.px[0] = r.px[0]; d.px[1] = r.px[1];
              d.bearing[0] = r.bearing[0]; d.bearing[1] = r.bearing[1]; d.bearing[2] = r.bearing[2];
              d.idepth = r.idepth; d.idepth_std = r.idepth_std;
              d.px0[0] = r.px0[0]; d.px0[1] = r.px0[1];
              std::memcpy(d.desc, r.desc, 32);
            }
            per[k].clear();
            if (filter) {
              states.insert(states.end(), per_st[k].begin(), per_st[k].end());
              per_st[k].clear();
            }
          }
          const sdvl_camera c = cam.abi();
          const sdvl_search_params sp = SearchParams();
          if (filter) {
            if (states.size() != total) throw std::runtime_error("mapper: one filter state per candidate request");
            dev_->Check(sdvl_search_run_filter(ctx, static_cast<int>(total), &c, &sp, states.data(), &fparams, track_, res.data(), fout.data()),
                        "sdvl_search_run_filter");
          } else {
            dev_->Check(sdvl_search_run(ctx, static_cast<int>(total), &c, &sp, res.data()), "sdvl_search_run");
          }
        }
      };
      std::unique_ptr<StageClock> sub(new StageClock(ST_MAP_CANDIDATES));
      for (;;) {  // UpdateCandidates, one occurrence pass at a time
        vector<char> more(M, 0);
        {
          StageClock c(ST_MAP_EMIT);
          ParallelFor(M, [&](int k) { more[k] = mm[k]->EmitCandidates(&per[k], dev_filter ? &per_st[k] : nullptr) ? 1 : 0; });
        }
        bool any = false;
        for (int k = 0; k < M; k++) any = any || more[k];
        if (!any) break;
        {
          StageClock c(ST_MAP_SEARCH);
          launch(dev_filter);
        }
        StageClock c(ST_MAP_APPLY);
        ParallelFor(M, [&](int k) { if (more[k]) mm[k]->ApplyCandidates(res.data() + begin[k], dev_filter ? fout.data() + begin[k] : nullptr); });
      }
      vector<int> kf_idx;
      for (int k = 0; k < M; k++)
        if (mm[k]->IsKeyframeUpdate()) kf_idx.push_back(k);
      sub.reset(new StageClock(ST_MAP_CONNECTIONS));
      if (!kf_idx.empty()) {
        const int K = static_cast<int>(kf_idx.size());
        ParallelFor(K, [&](int q) {
          MapperMap *m = mm[kf_idx[q]];
          m->CheckConnections();
          m->EmitConnectionsPoints(&per[kf_idx[q]]);
        });
        launch();
        vector<char> need(K, 0);
        ParallelFor(K, [&](int q) {
          MapperMap *m = mm[kf_idx[q]];
          m->ApplyConnectionsPoints(res.data() + begin[kf_idx[q]]);
          need[q] = m->PrepareInitCandidates() ? 1 : 0;
        });
        sub.reset(new StageClock(ST_MAP_INIT));
        vector<shared_ptr<Frame>> to_filter;
        for (int q = 0; q < K; q++)
          if (need[q]) to_filter.push_back(mm[kf_idx[q]]->CurrentFrame());
        if (!to_filter.empty()) {
          Frame::FilterCornersBatch(to_filter);
          ParallelFor(K, [&](int q) { if (need[q]) mm[kf_idx[q]]->EmitInitCandidates(&per[kf_idx[q]]); });
          launch();
          ParallelFor(K, [&](int q) { if (need[q]) mm[kf_idx[q]]->ApplyInitCandidates(res.data() + begin[kf_idx[q]]); });
        }
      }
      sub.reset(new StageClock(ST_MAP_FINISH));
      ParallelFor(M, [&](int k) { mm[k]->FinishUpdate(); });
      sub.reset();
    }
  }
  clk.reset();
}

}  // namespace sdvl
