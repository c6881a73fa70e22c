// sdvl_host.h — host side of the MI355X-native SDVL front-end: the reference's C++ API surface
// (Camera, Feature, Point, Frame, FastDetector, ORBDetector, ImageAlign, Matcher, FeatureAlign, SDVL) with the same
// names, argument meaning and return conventions, implemented on top of the C-ABI in include/sdvl_hip.h.
//
// MI355X-first differences (none visible to a caller that uses the reference signatures):
//   * a Frame's pyramid / corners / descriptors live in HBM; GetPyramid() / GetDescriptors() mirror to the host
//     lazily (the mapper and the UI read them);
//   * every stage has a batched form (Frame::CreateBatch, ImageAlign::ComputePoseBatch, Matcher::SearchPoints,
//     FeatureAlign::PrepareReproject/FinishReproject, SDVLBatch::HandleFrames) so that B independent sequences
//     advance one frame with ONE launch per kernel; the single-object calls are the B = 1 case;
//   * FeatureAlign evaluates Matcher::SearchPoint for ALL grid candidates in one launch and then replays the
//     reference's sequential cell-order / first-hit / max_matches logic over the results (SearchPoint is a pure
//     function of its arguments, so the outcome is identical to the one-call-at-a-time loop).
// Error convention: the reference returns bool/int and prints to cerr; so does this layer.  A failing device call
// is fatal (std::runtime_error carrying sdvl_last_error) — there is no CPU fallback.
#ifndef SDVL_HOST_H_
#define SDVL_HOST_H_

#include <array>
#include <atomic>
#include <cstring>
#include <functional>
#include <deque>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/sdvl_hip.h"
#include "config.h"
#include "se3.h"
#include "types.h"

namespace sdvl {

class Frame;
class Feature;
class Point;
class Map;

// Host memory for the frame arenas of one Device, handed out in fixed chunks and recycled most-recent-first: the chunks of
// the frames that die every step come back cache-warm.  (Mapping AND touching the pages up front was measured slower than
// taking the first-touch faults inside the run: a page the kernel has just zeroed is still in cache when the tracker
// writes its features into it.)
class ChunkPool {
 public:
  static constexpr size_t kChunk = 32 * 1024;
  ~ChunkPool();
  char *Get();
  void Put(char *c);

 private:
  void Map(size_t bytes);
  std::mutex m_;
  std::vector<char *> free_;
  std::vector<std::pair<void *, size_t>> regions_;
};

// ---------------------------------------------------------------------------------------------------------------
// Device: one sdvl_ctx (= one HIP stream) + a pool of HBM frames.  One per host thread (tracker / mapper).
class Device {
 public:
  explicit Device(int gpu = 0);
  ~Device();
  sdvl_ctx *ctx() const { return ctx_; }
  sdvl_frame *AcquireFrame(int w, int h, int levels);
  void ReleaseFrame(sdvl_frame *f, int w, int h, int levels);
  // make sure `frames` free HBM frames of this shape are pooled (keyframes keep theirs for good: size it from the
  // keyframe budget so that no hipMalloc lands on the tracking path)
  void Reserve(int w, int h, int levels, int frames);
  int CornerCap() const;  // corners_ capacity of the frames created under the current Config
  void Check(int rc, const char *what) const;
  static Device *Current();
  static Device *CurrentOrNull();
  static void SetCurrent(Device *d);
  uint64_t search_batch_counter = 0;  // ids of the packed search batches opened on this device's context
  // Frame::FilterCornersEnd's round-trip buffer: owned by the Device (= one group), because a host thread may interleave
  // several groups (fibers switch at the wait inside the call), so nothing there can be per thread
  std::vector<sdvl_filtered_corner> scratch_filtered;
  // source of Point ids for the trackers stepping on this device: ids only have to grow along one tracker's own history
  // (the mapper orders by id), and a counter shared by every group would bounce between their cores
  std::atomic<int> next_point_id{0};
  // the counter Point() draws from on threads bound to this device: its own, or another device's (a tracker's mapper thread
  // works on its own Device = its own stream, but its points belong to the tracker's id sequence)
  std::atomic<int> *point_ids = &next_point_id;
  int gpu() const { return gpu_; }
  // chunks for the frame arenas (features, points) of the trackers on this device; shared with the arenas, which may
  // outlive the device object
  std::shared_ptr<ChunkPool> chunks = std::make_shared<ChunkPool>();

 private:
  sdvl_ctx *ctx_ = nullptr;
  int gpu_ = 0;
  struct Pooled { sdvl_frame *f; int w, h, levels, cap; };
  std::vector<Pooled> pool_;
  std::mutex pool_mutex_;
  int total_frames_ = 0;
};

// glibc rand() (TYPE_3, seed 1) as a private stream: the reference draws from the process-global rand()
// (feature_align.cc:53,103,180); B trackers in one process each own the stream a lone reference process would see.
class RandStream {
 public:
  explicit RandStream(unsigned seed = 1);
  int Next();
  template <typename T>
  void Shuffle(std::vector<T> *v) {  // libstdc++ std::random_shuffle(first, last)
    for (size_t i = 1; i < v->size(); ++i) {
      const size_t j = static_cast<size_t>(Next()) % (i + 1);
      if (i != j) std::swap((*v)[i], (*v)[j]);
    }
  }

 private:
  int r_[31];
  int fi_, ri_;
};

// camera.h:34-135 (pinhole part; UndistortImage is out of scope — SURVEY §8f #2)
class Camera {
 public:
  Camera();  // from Config::GetCameraParameters(), camera.cc:28-38
  Camera(int width, int height, double fx, double fy, double u0, double v0);
  double GetWidth() const { return width_; }
  double GetHeight() const { return height_; }
  double GetFx() const { return fx_; }
  double GetFy() const { return fy_; }
  double GetU0() const { return u0_; }
  double GetV0() const { return v0_; }
  void Project(const Vector3d &p3D, Vector2d *p2D) const;
  void Unproject(const Vector2d &p2D, Vector3d *p3D) const;
  Vector2d Project(const Vector3d &p3D) const { Vector2d r; Project(p3D, &r); return r; }
  Vector3d Unproject(const Vector2d &p2D) const { Vector3d r; Unproject(p2D, &r); return r; }
  bool IsInsideImage(const Vector2i &p, int m = 0) const { return p(0) >= m && p(0) < width_ - m && p(1) >= m && p(1) < height_ - m; }
  bool IsInsideImage(const Vector2i &p, int m, int l) const {
    return p(0) >= m && p(0) < width_ / (1 << l) - m && p(1) >= m && p(1) < height_ / (1 << l) - m;
  }
  static Vector2d SimpleProject(const Vector3d &p) { return Vector2d(p(0) / p(2), p(1) / p(2)); }
  sdvl_camera abi() const { return sdvl_camera{width_, height_, fx_, fy_, u0_, v0_}; }
  // camera.cc:39-67: d0..d4 = Camera.d1..d5 of the config; as in the reference only d0 decides whether there is distortion
  void SetDistortions(double d0, double d1, double d2, double d3, double d4);
  bool HasDistortion() const { return has_distortion_; }
  sdvl_distortion distortion() const { return sdvl_distortion{{d_[0], d_[1], d_[2], d_[3], d_[4]}}; }
  // camera.cc:100-105: cv::undistort on the device of the calling thread.  `in` may live on the host or in HBM; `out`
  // is an HBM image that owns its storage (hand it to SDVL::HandleFrame / Frame like any other image).
  void UndistortImage(const Image &in, Image *out) const;

 private:
  double width_, height_, fx_, fy_, u0_, v0_;
  double d_[5] = {0, 0, 0, 0, 0};
  bool has_distortion_ = false;
};

// extra/orb_detector.h:34-56.  Descriptors come from the K4 kernel; Distance is the reference's popcount.
class ORBDetector {
 public:
  ORBDetector() {}
  // src must be a pyramid level of a Frame (it carries the HBM binding)
  bool GetDescriptor(const Image &src, const Vector2i &pos, std::vector<uchar> *desc);
  int Distance(const std::vector<uchar> &a, const std::vector<uchar> &b);
};

// extra/fast_detector.h:34-64
class FastDetector {
 public:
  FastDetector(int width, int height, bool grid = true);
  void DetectPyramid(const std::vector<Image> &pyramid, std::vector<Vector3i> *corners, int nfeatures);
  void FilterCorners(const std::vector<Image> &pyramid, const std::vector<Vector3i> &corners, std::vector<int> *indices);
  void LockCell(Vector2d p);
  void UnlockCell(Vector2d p);
  // host half of SelectPixels (quota + retainBest, fast_detector.cc:108-151) over the device's per-cell lists
  static void SelectFromCells(const sdvl_keypoint *kps, const int32_t *cell_offsets, int level_cell_begin, int wcells, int hcells,
                              int level, int level_w, int level_h, int nfeatures, std::vector<Vector3i> *pixels);
  // host half of FilterCorners given the K3 scores
  void FilterWithScores(const std::vector<Image> &pyramid, const std::vector<Vector3i> &corners, const double *scores,
                        std::vector<int> *indices);

 private:
  std::vector<std::pair<int, int>> cgrid_;
  std::vector<bool> grid_mask_;
  int grid_width_, grid_height_, cell_size_;
};

// feature.h:38-105
class Feature {
 public:
  Feature(const std::shared_ptr<Frame> &f, const Vector2d &p, int l);
  Feature(std::weak_ptr<Frame> &&f, Frame *raw, const Vector2d &p, int l);  // Frame::NewFeature: no shared_ptr round trip
  Feature(const std::shared_ptr<Frame> &f, const std::shared_ptr<Point> &ft, const Vector2d &p, int l);
  Feature(const std::shared_ptr<Frame> &f, const std::shared_ptr<Point> &ft, const Vector2d &p, const Vector3d &v, int l);
  std::shared_ptr<Frame> GetFrame() { return frame_.lock(); }
  // the frame without reference-count traffic (per-candidate loops); null once the frame is gone
  Frame *GetFrameRaw() const { return frame_.expired() ? nullptr : frame_raw_; }
  void SetFrame(const std::shared_ptr<Frame> &f) { frame_ = f; frame_raw_ = f.get(); }
  std::shared_ptr<Point> GetPoint() const { return point_; }
  Point *GetPointRaw() const { return point_.get(); }  // no reference-count traffic in the per-frame loops
  void SetPoint(const std::shared_ptr<Point> &p) { point_ = p; }
  void SetPoint(std::shared_ptr<Point> &&p) { point_ = std::move(p); }
  const Vector2d &GetPosition() const { return p2d_; }
  const Vector3d &GetVector() const { return v_; }
  void SetVector(Vector3d &v) { v_ = v; }
  int GetLevel() const { return level_; }
  // feature.h:78.  The 32 descriptor bytes live inside the feature (DescriptorData(): what the request loops read — the
  // reference's std::vector<uchar> is one more heap block and one more cache miss per search candidate); the vector the
  // reference's signature returns is made from them the first time somebody asks for it.
  const std::vector<uchar> &GetDescriptor() const {
    if (!descriptor_vec_) descriptor_vec_.reset(new std::vector<uchar>(32, 0));
    if (has_descriptor_) std::memcpy(descriptor_vec_->data(), descriptor_.data(), 32);
    return *descriptor_vec_;
  }
  const std::array<uchar, 32> &DescriptorData() const { return descriptor_; }
  void SetDescriptor(const std::vector<uchar> &d) {
    descriptor_.fill(0);
    std::memcpy(descriptor_.data(), d.data(), d.size() < 32 ? d.size() : 32);
    has_descriptor_ = true;
  }
  void SetDescriptor(const uchar *d32) {
    if (!d32) return;
    std::memcpy(descriptor_.data(), d32, 32);
    has_descriptor_ = true;
  }
  bool HasDescriptor() const { return has_descriptor_; }
  Vector2d GetLevelPosition() { return Vector2d(p2d_(0) / (1 << level_), p2d_(1) / (1 << level_)); }

 private:
  std::weak_ptr<Frame> frame_;  // the reference holds a shared_ptr (a frame<->feature cycle it never breaks)
  Frame *frame_raw_ = nullptr;
  std::shared_ptr<Point> point_;
  Vector2d p2d_;
  Vector3d v_;
  int level_;
  bool has_descriptor_;
  std::array<uchar, 32> descriptor_;
  mutable std::unique_ptr<std::vector<uchar>> descriptor_vec_;
};

// Bump allocator for the many small objects that live and die with one frame (its features, the points it seeds): one
// malloc per chunk instead of one per object, and neighbours in the frame's lists are neighbours in memory.  The arena
// counts its live objects plus one reference for the owning frame and frees itself when the last of them goes (a
// keyframe's features and points outlive the tracking step, an ordinary frame's die with it).  Like the frame's own
// lists, an arena is used by one thread at a time; only the count is atomic, since the last object may die elsewhere.
class FrameArena {
 public:
  explicit FrameArena(const std::shared_ptr<ChunkPool> &pool = nullptr) : pool_(pool), live_(1) {}
  void *Allocate(size_t bytes, size_t align) {
    live_.fetch_add(1, std::memory_order_relaxed);
    size_t at = (off_ + align - 1) / align * align;
    if (!cur_ || at + bytes > cap_) {
      if (pool_ && bytes <= ChunkPool::kChunk) {
        cur_ = pool_->Get();
        pooled_.push_back(cur_);
        cap_ = ChunkPool::kChunk;
      } else {
        cap_ = bytes > ChunkPool::kChunk ? bytes : ChunkPool::kChunk;
        heap_.emplace_back(new char[cap_]);
        cur_ = heap_.back().get();
      }
      at = 0;
    }
    off_ = at + bytes;
    return cur_ + at;
  }
  void Release() {  // one object (or the owning frame) is gone
    if (live_.fetch_sub(1, std::memory_order_acq_rel) == 1) delete this;
  }

 private:
  ~FrameArena() {
    for (char *c : pooled_) pool_->Put(c);
  }
  std::shared_ptr<ChunkPool> pool_;
  std::vector<char *> pooled_;
  std::vector<std::unique_ptr<char[]>> heap_;
  char *cur_ = nullptr;
  size_t off_ = 0, cap_ = 0;
  std::atomic<int> live_;
};

template <typename T>
struct ArenaAllocator {
  typedef T value_type;
  FrameArena *arena;
  explicit ArenaAllocator(FrameArena *a) : arena(a) {}
  template <typename U>
  ArenaAllocator(const ArenaAllocator<U> &o) : arena(o.arena) {}
  T *allocate(size_t n) { return static_cast<T *>(arena->Allocate(n * sizeof(T), alignof(T))); }
  void deallocate(T *, size_t) { arena->Release(); }
  template <typename U>
  bool operator==(const ArenaAllocator<U> &o) const { return arena == o.arena; }
  template <typename U>
  bool operator!=(const ArenaAllocator<U> &o) const { return arena != o.arena; }
};

// point.h:37-147 — the part the front-end reads or updates (the depth filter itself is map state, out of scope)
class Point {
 public:
  enum PointStatus { P_FOUND, P_NOT_FOUND, P_SEEN, P_UNSEEN, P_OUTLIER };
  Point();
  double GetInverseDepth() { return rho_; }
  double GetStd();
  std::shared_ptr<Feature> GetInitFeature() { return feature_; }
  Feature *GetInitFeatureRaw() const { return feature_.get(); }
  void SetInitFeature(const std::shared_ptr<Feature> &f) { feature_ = f; }
  int GetID() const { return id_; }
  Vector3d GetPosition() const;
  void SetPosition(const Vector3d &pos);  // point.cc:144-162
  void InitFixed(const std::shared_ptr<Feature> &f, double depth, double sigma2, const Vector3d &p3d);
  void InitCandidate(const std::shared_ptr<Feature> &f, double depth);
  std::list<std::shared_ptr<Feature>> &GetFeatures() { return features_; }
  int Score() const { return n_successful_; }
  int GetLastFrame() const { return last_frame_; }
  void SetLastFrame(int id) { last_frame_ = id; }
  PointStatus GetStatus() const { return status_; }
  void SetStatus(PointStatus s) { status_ = s; }
  bool ToDelete() const { return delete_; }
  void SetDelete() { delete_ = true; }
  void SetFixed() { fixed_ = true; }
  bool IsFixed() { return fixed_; }
  void AddFeature(const std::shared_ptr<Feature> &f) { features_.push_front(f); }
  std::shared_ptr<Feature> GetLastFeature() { return features_.front(); }
  bool Promote();
  bool Unpromote();
  // device-resident tracking tables (SDVLBatch): the counters the device advances while the point sits in a table
  int GetFailed() const { return n_failed_; }
  // (one failure more than before = one Unpromote on the device, which also counts in the depth filter's b_, point.cc:112;
  //  exact as long as the counters are collected after every frame, which SDVLBatch does whenever a mapper filters points)
  void SetTrackCounters(int n_successful, int n_failed, int last_frame, PointStatus status) {
    if (n_failed == n_failed_ + 1) b_++;
    n_successful_ = n_successful; n_failed_ = n_failed; last_frame_ = last_frame; status_ = status;
  }
  // the depth filter's state as the device kernel takes and returns it (sdvl_search_points_filter)
  void GetFilterState(sdvl_depth_state *st) const;
  void ApplyFilterOut(const sdvl_depth_out &o);
  // row of the point in the device-resident tracking tables (tracker * capacity + index), -1: none
  int TrackRow() const { return track_row_; }
  void SetTrackRow(int r) { track_row_ = r; }
  // the device knows the point is deleted (or will be after the next step): nothing to tell it when the trash is emptied
  bool DeviceTrashed() const { return dev_trashed_; }
  void SetDeviceTrashed() { dev_trashed_ = true; }
  // depth filter (point.cc:64-100,164-217): used by the mapper (MapperMap), not by the tracking path
  void Update(const std::shared_ptr<Frame> &frame, double depth, double px_error_angle);
  bool HasConverged();
  bool SeenFrom(const std::shared_ptr<Frame> &frame) const;
  // AddConnectionsPoints asks SeenFrom(current keyframe) for every point of the connected keyframes: the keyframe stamps its own
  // points once instead (a feature of the keyframe is in its point's list and vice versa)
  int SeenStamp() const { return seen_stamp_; }
  void SetSeenStamp(int frame_id) { seen_stamp_ = frame_id; }
  static void ConsumeId();  // what constructing and discarding a Point does to the id counter
  static double ComputeTau(const SE3 &pose, const Vector3d &v, double depth, double px_error_angle);
  static double PDFNormal(double mean, double sd, double x);

 private:
  // the fields the per-frame loops touch (ProjectPoints, SelectPoints) first, so that they share a cache line
  PointStatus status_;
  bool delete_;
  bool fixed_;
  int last_frame_, n_successful_, n_failed_;
  int id_;
  Vector3d p3d_;
  double rho_;
  std::shared_ptr<Feature> feature_;
  double sigma2_, a_, b_, z_range_;
  double cos_alpha_ = 1.0, last_distance_ = 1.0;
  int track_row_ = -1;
  int seen_stamp_ = -1;
  bool dev_trashed_ = false;
  std::list<std::shared_ptr<Feature>> features_;
};

// frame.h:41-173
class Frame : public std::enable_shared_from_this<Frame> {
 public:
  Frame(Camera *camera, ORBDetector *detector, const Image &img, bool corners);
  ~Frame();
  // B frames with one launch per kernel (pyramid, FAST, ORB); corners as in the ctor
  static void CreateBatch(Camera *camera, ORBDetector *detector, const std::vector<Image> &imgs, bool corners, int nfeatures,
                          std::vector<std::shared_ptr<Frame>> *out, const std::function<void(int, std::function<void(int)>)> *pfor = nullptr);

  bool IsKeyframe() { return is_keyframe_; }
  void SetKeyframe() { is_keyframe_ = true; }
  // level 0 aliases an HBM image that is valid during the frame's own step only (Image::transient)
  bool ImageTransient() const { return image_transient_; }
  void SetImageTransient(bool on) { image_transient_ = on; }
  // the frames among `frames` whose image is transient copy it into their own level 0 (one launch): keyframes outlive the ring
  static void OwnImages(const std::vector<std::shared_ptr<Frame>> &frames);
  void FilterCorners();
  static void FilterCornersBatch(const std::vector<std::shared_ptr<Frame>> &frames);
  static void FilterCornersBegin(const std::vector<std::shared_ptr<Frame>> &frames);  // the two halves of FilterCornersBatch
  static void FilterCornersEnd(const std::vector<std::shared_ptr<Frame>> &frames,
                               const std::function<void(int, const std::function<void(int)> &)> *pfor = nullptr);
  // corner detection + ORB for frames built with corners = false (CreateBatch): queues the kernels, returns at once
  static void DetectBatch(const std::vector<std::shared_ptr<Frame>> &frames, int nfeatures);
  SE3 &GetPose() { return pose_; }  // frame.h:52: callers may write through it, so the cached inverse is checked by value
  const SE3 &GetPose() const { return pose_; }
  void SetPose(const SE3 &se3) { pose_ = se3; }
  std::vector<Image> &GetPyramid();  // host mirror is filled on first call
  // Features of a frame that went through the device-resident tracking tables exist as flat records first (position, level,
  // index of the point in the tracker's table) and become Feature objects the first time somebody asks for them
  typedef std::vector<std::shared_ptr<Point>> PointTable;
  std::vector<std::shared_ptr<Feature>> &GetFeatures() {
    if (flat_) MaterializeFeatures();
    return features_;
  }
  void SetFlatFeatures(const sdvl_track_feature_out *feats, int n, const std::shared_ptr<PointTable> &points);
  bool HasFlatFeatures() const { return flat_ != nullptr; }
  bool IsRegistered() const { return registered_; }
  void SetRegistered() { registered_ = true; }
  std::vector<Vector3i> &GetCorners();  // host mirror of the HBM corner list, filled on first call
  int GetNumCorners();                  // corner count without mirroring the list
  std::vector<int> &GetFilteredCorners() { return filtered_corners_; }
  std::vector<Vector2d> &GetOutliers() { return outliers_; }
  std::vector<std::vector<uchar>> &GetDescriptors();  // host mirror of the HBM descriptors
  // What FilterCorners brought back from the device: the k-th filtered corner (x, y in level coordinates, level) and its
  // ORB descriptor (frame.cc:148-161 computes descriptors for exactly these), in the order of GetFilteredCorners()
  int NumFiltered() const { return static_cast<int>(filt_.size()); }
  Vector3i FilteredCorner(int k) const { return Vector3i(filt_[k].x, filt_[k].y, filt_[k].level); }
  const uchar *FilteredDescriptor(int k) const { return filt_[k].desc; }
  // descriptor of corner `index` (an entry of GetFilteredCorners()), null for corners FilterCorners did not keep
  const uchar *HostDescriptor(int index) const {
    for (const sdvl_filtered_corner &c : filt_)
      if (c.index == index) return c.desc;
    return nullptr;
  }
  Camera *GetCamera() const { return camera_; }
  int GetWidth() const { return width_; }
  int GetHeight() const { return height_; }
  int GetID() const { return id_; }
  void SetID(int id) { id_ = id; }
  // pose_.Inverse(), computed once per SetPose (the mapper asks for it for every candidate of every frame)
  const SE3 &GetWorldPose() const {
    if (!world_valid_ || std::memcmp(&world_of_, &pose_, sizeof(SE3)) != 0) {
      world_ = pose_.Inverse();
      world_of_ = pose_;
      world_valid_ = true;
    }
    return world_;
  }
  Vector3d GetWorldPosition() const { return GetWorldPose().GetTranslation(); }
  Vector3d GetRelativePos(const Vector3d &pos) const { return pose_ * pos; }
  void AddFeature(const std::shared_ptr<Feature> &f) { GetFeatures().push_back(f); scene_depth_hint_valid_ = false; }
  // a Feature on this frame whose storage comes from the frame's arena (same object as make_shared<Feature>(frame, ...))
  std::shared_ptr<Feature> NewFeature(const Vector2d &p, int level) {
    if (!arena_) arena_ = NewArena();
    return std::allocate_shared<Feature>(ArenaAllocator<Feature>(arena_), weak_from_this(), this, p, level);
  }
  // a Point whose storage comes from this frame's arena (the points a keyframe seeds sit next to their features)
  std::shared_ptr<Point> NewPoint() {
    if (!arena_) arena_ = NewArena();
    return std::allocate_shared<Point>(ArenaAllocator<Point>(arena_));
  }
  void AddOutlier(const Vector2d &p) { outliers_.push_back(p); }
  int GetNumFeatures() const { return flat_ ? static_cast<int>(flat_->feats.size()) : static_cast<int>(features_.size()); }
  int GetNumPoints() const;
  bool Project(const Vector3d &p3D, Vector2d *p2D);
  void CreateCorners(int levels, int nfeatures);
  void RemoveFeatures() { features_.clear(); DropFlat(); features_removed_ = true; scene_depth_hint_valid_ = false; }
  bool FeaturesRemoved() const { return features_removed_; }  // the mapper emptied the frame (Map::EmptyTrash, map.cc:207-259)
  // mapper-side state and queries (frame.h:71-87,120-136; frame.cc:70-113,181-207)
  void SetKeyframeID(int id) { kf_id_ = id; }
  int GetKeyframeID() const { return kf_id_; }
  bool IsSelected() { return selected_; }
  void SetSelected(bool v) { selected_ = v; }
  bool ToDelete() const { return delete_; }
  void SetDelete() { delete_ = true; }
  double GetSceneDepth();
  // the step computed it on the device (sdvl_track_result.scene_depth); valid until the feature list changes
  void SetSceneDepthHint(double d) { scene_depth_hint_ = d; scene_depth_hint_valid_ = true; }
  void ClearSceneDepthHint() { scene_depth_hint_valid_ = false; }
  bool IsPointVisible(const Vector3d &p);
  double DistanceTo(const Frame &frame) const;
  double DistanceTo(const Vector3d &p) const;
  void AddConnection(const std::pair<std::shared_ptr<Frame>, int> kf) { connections_.push_back(kf); }
  void GetBestConnections(std::vector<std::shared_ptr<Frame>> *connections, int n);
  sdvl_frame *device() const { return dev_; }
  Device *owner() const { return owner_; }
  int SearchSlot(sdvl_ctx *ctx, uint64_t batch_id);

 private:
  Frame() {}
  void InitCommon(Camera *camera, ORBDetector *detector, int w, int h);
  int id_ = 0;
  Camera *camera_ = nullptr;
  ORBDetector *orb_detector_ = nullptr;
  int pyramid_levels_ = 0;
  bool is_keyframe_ = false;
  bool image_transient_ = false;
  std::vector<Image> pyramid_;
  bool pyramid_on_host_ = false;
  int width_ = 0, height_ = 0;
  SE3 pose_;
  mutable SE3 world_, world_of_;
  mutable bool world_valid_ = false;
  std::vector<std::shared_ptr<Feature>> features_;
  std::vector<Vector3i> corners_;
  std::vector<int> filtered_corners_;
  std::vector<Vector2d> outliers_;
  std::vector<std::vector<uchar>> descriptors_;
  std::vector<sdvl_filtered_corner> filt_;
  bool descriptors_on_host_ = false;
  bool corners_on_host_ = true;  // false after a device-side DetectPyramid until GetCorners() mirrors the list
  sdvl_frame *dev_ = nullptr;
  Device *owner_ = nullptr;
  FrameArena *arena_ = nullptr;
  FrameArena *NewArena() const;
  // the records sit in the frame's arena (no malloc per frame: with 16 host threads the allocator's mprotect / page-fault
  // traffic was the largest single cost of a step)
  struct FlatSpan {
    const sdvl_track_feature_out *data = nullptr;
    int n = 0;
    const sdvl_track_feature_out *begin() const { return data; }
    const sdvl_track_feature_out *end() const { return data + n; }
    size_t size() const { return static_cast<size_t>(n); }
  };
  struct FlatFeatures {
    FlatSpan feats;
    std::shared_ptr<PointTable> points;
  };
  FlatFeatures flat_store_;
  FlatFeatures *flat_ = nullptr;  // &flat_store_ while the features are still flat records
  void MaterializeFeatures();
  void DropFlat();
  bool registered_ = false;  // the context's (frame, pose) registry holds this frame with its current pose
  bool features_removed_ = false;
  double scene_depth_hint_ = 0.0;
  bool scene_depth_hint_valid_ = false;
  int search_slot_ = -1;
  uint64_t search_batch_ = 0;
  int kf_id_ = 0;
  bool delete_ = false, selected_ = false;
  std::vector<std::pair<std::shared_ptr<Frame>, int>> connections_;
  static std::atomic<int> counter_;
};

// image_align.h:33-66
class ImageAlign {
 public:
  ImageAlign() {}
  int ComputePose(const std::shared_ptr<Frame> &frame1, const std::shared_ptr<Frame> &frame2, bool fast = false);
  double GetError() { return error_; }
  // n frame pairs, one launch; returns per-pair ComputePose results and errors
  static void ComputePoseBatch(const std::vector<std::pair<std::shared_ptr<Frame>, std::shared_ptr<Frame>>> &pairs, bool fast,
                               std::vector<int> *n_meas, std::vector<double> *errors, std::vector<int> *iters = nullptr,
                               const std::vector<SE3> *start_poses = nullptr, std::vector<SE3> *out_poses = nullptr,
                               const std::function<void()> *between = nullptr);
  // between: called after the alignment has been launched and before its results are awaited — device work queued there
  // (corner detection of the new frames) runs behind the alignment while the host already continues with the poses
  // start_poses[i]: pose of pairs[i].second to start from (default: its current pose); out_poses: where the aligned
  // poses go instead of into the frames — together they let ONE frame be aligned against many references at once

 private:
  double error_ = 1e10;
};

// matcher.h:39-83
class Matcher {
 public:
  explicit Matcher(int size) : patch_size_(size) {}
  bool SearchPoint(const std::shared_ptr<Frame> &frame, const std::shared_ptr<Feature> &feature, double idepth, double idepth_std,
                   bool fixed, Vector2d *px, int *flevel);
  // batched form: fills one sdvl_search_req per call site, evaluated together by SearchPoints
  static bool MakeRequest(const std::shared_ptr<Frame> &frame, const std::shared_ptr<Feature> &feature, double idepth,
                          double idepth_std, bool fixed, const Vector2d &px, sdvl_search_req *req);
  static void SearchPoints(Device *dev, const std::vector<sdvl_search_req> &reqs, const Camera &cam, std::vector<sdvl_search_res> *res);
  // the same with the mapper's depth filter behind the search (Map::UpdateCandidates); `set` (may be null) = the tracking tables
  // whose rows the filter patches
  static void SearchPointsFilter(Device *dev, const std::vector<sdvl_search_req> &reqs, const std::vector<sdvl_depth_state> &states,
                                 const Camera &cam, const sdvl_depth_params &fp, sdvl_track_set *set, std::vector<sdvl_search_res> *res,
                                 std::vector<sdvl_depth_out> *fout);

 private:
  int patch_size_;
};

// The out-of-scope back-end (map.cc) as the tracker sees it: deletion queue + keyframe decision + keyframe list.
class Map {
 public:
  virtual ~Map() {}
  // map.h:49-55,61.  The mapper thread (threaded mode, main.cc:97,120) and the tracker meet on this mutex.  It is held more
  // coarsely than in the reference — by the tracker for a whole HandleFrame, by the mapper for a whole UpdateMap — so the two
  // threads alternate on the shared objects while each drives its own sdvl_ctx / HIP stream.
  std::mutex &GetMutex() { return mutex_map_; }
  virtual void UpdateMap() {}
  virtual void Start() {}
  virtual void Stop() {}
  void DeletePoint(const std::shared_ptr<Point> &p) { points_trash_.push_back(p); }
  bool NeedKeyframe(const std::shared_ptr<Frame> &frame, int matches);  // map.cc:170-188
  virtual void AddKeyframe(const std::shared_ptr<Frame> &frame, bool search = true);
  virtual void AddFrame(const std::shared_ptr<Frame> &) {}
  virtual void LimitKeyframes(const std::shared_ptr<Frame> &) {}
  virtual void SetRelocalizing(bool) {}
  virtual void EmptyTrash();  // map.cc:207-259
  // a point with a row in the device-resident tracking tables died without the device having been told (EmptyTrash): the
  // tracker's table must be rebuilt before the next step
  bool TakeTablesDirty() { const bool d = tables_dirty_; tables_dirty_ = false; return d; }
  std::vector<std::shared_ptr<Frame>> &GetKeyframes() { return keyframes_; }
  // mapper work for a fresh keyframe (Map::InitCandidates stand-in); called outside the tracking stages
  virtual void InitCandidates(const std::shared_ptr<Frame> &) {}

 protected:
  std::vector<std::shared_ptr<Frame>> keyframes_;
  std::vector<std::shared_ptr<Point>> points_trash_;
  bool tables_dirty_ = false;
  std::shared_ptr<Frame> last_kf_;
  int last_matches_ = 0;
  std::mutex mutex_map_;
};

// Map stand-in for synthetic scenes: seeds one FIXED point per FilterCorners() corner of a keyframe, depth from a
// known scene plane n.X = d (replaces homography_init + depth filter, both out of scope).
class PlaneMap : public Map {
 public:
  PlaneMap(const Vector3d &n, double d) : n_(n), d_(d) {}
  void InitCandidates(const std::shared_ptr<Frame> &kf) override;
  void SeedFromFiltered(const std::shared_ptr<Frame> &kf);  // after Frame::FilterCorners()
  void SetPlane(const Vector3d &n, double d) { n_ = n; d_ = d; }

 private:
  Vector3d n_;
  double d_;
};

// map.h:44-139 — the reference's mapper in SEQUENTIAL mode (main.cc:148-149: SDVL::Mapping() = Map::UpdateMap() after every
// frame).  The first keyframe is still bootstrapped from the scene plane (homography_init is out of scope); every later
// keyframe goes through UpdateCandidates / CheckConnections / AddConnectionsPoints / InitCandidates, ordinary frames
// through UpdateCandidates / CheckRedundantKeyframes.  Bundle adjustment is out of scope and not run.
// Every SearchPoint the mapper would issue one at a time is emitted as a request instead (Emit*), evaluated for MANY
// trackers in one K7 launch, and the reference's sequential logic is replayed over the results (Apply*) — the same
// speculate-then-replay scheme FeatureAlign uses (SearchPoint is a pure function of its arguments).
class MapperMap : public PlaneMap {
 public:
  MapperMap(const Vector3d &n, double d, Camera *camera) : PlaneMap(n, d), camera_(camera) {}
  void AddKeyframe(const std::shared_ptr<Frame> &frame, bool search = true) override;  // map.cc:143-158
  void AddFrame(const std::shared_ptr<Frame> &frame) override { frame_queue_.push_back(frame); }
  void LimitKeyframes(const std::shared_ptr<Frame> &frame) override;                   // map.cc:190-205
  void SetRelocalizing(bool v) override { relocalizing_ = v; }
  void EmptyTrash() override;
  void InitCandidates(const std::shared_ptr<Frame> &) override {}  // the bootstrap keyframe is seeded by the driver

  // ---- Map::UpdateMap (map.cc:75-141) cut into phases; a driver calls them in this order for all its trackers
  bool BeginUpdate();                                            // pops the next frame; false = nothing to do
  // UpdateCandidates, one pass; false = no more passes.  states / fout: the depth filter runs on the device behind the search
  // (sdvl_search_points_filter) and Apply only books its outcomes; without them Apply does the arithmetic itself
  bool EmitCandidates(std::vector<sdvl_search_req> *reqs, std::vector<sdvl_depth_state> *states = nullptr);
  void ApplyCandidates(const sdvl_search_res *res, const sdvl_depth_out *fout = nullptr);
  static bool DeviceFilter();             // env SDVL_HOST_DEPTH_FILTER=1 turns it off
  static void SetDeviceFilter(bool on);
  sdvl_depth_params FilterParams() const;

  bool IsKeyframeUpdate() const { return cur_ && cur_->IsKeyframe(); }
  void CheckConnections();                                       // map.cc:500-558
  void EmitConnectionsPoints(std::vector<sdvl_search_req> *reqs);  // AddConnectionsPoints, map.cc:560-617
  void ApplyConnectionsPoints(const sdvl_search_res *res);
  bool PrepareInitCandidates();                                  // map.cc:262-283; true = the keyframe needs FilterCorners
  const std::shared_ptr<Frame> &CurrentFrame() const { return cur_; }
  void EmitInitCandidates(std::vector<sdvl_search_req> *reqs);   // after Frame::FilterCorners of CurrentFrame()
  void ApplyInitCandidates(const sdvl_search_res *res);
  void FinishUpdate();                                           // CheckRedundantKeyframes + trash for ordinary frames
  // the whole of it for one tracker (one launch per phase)
  void UpdateMap() override;
  // map.cc:49-71: the mapper thread of threaded mode.  It works on a Device of its own (one sdvl_ctx = one HIP stream per
  // host thread, include/sdvl_hip.h) on the tracker's GPU and shares the tracker's frames read-only.
  ~MapperMap() override;
  void Start() override;
  void Stop() override;
  void Run();
  long Updates() const { return updates_; }  // UpdateMap calls that found a frame to work on
  bool IsThreaded() const { return running_; }

  struct Stats { int candidates = 0, converged = 0, initialized = 0, linked = 0, connected = 0, keyframes = 0; };
  Stats GetStats() const;

 private:
  void CheckRedundantKeyframes();  // map.cc:619-690
  Camera *camera_;
  std::vector<std::shared_ptr<Point>> candidates_;
  std::deque<std::shared_ptr<Frame>> frame_queue_, keyframe_queue_;
  std::vector<std::shared_ptr<Frame>> frame_trash_, retired_;  // retired_: culled keyframes stay alive while features name them
  int num_kfs_ = 0, initial_kf_id_ = 0, last_kf_checked_ = -1;
  bool relocalizing_ = false;
  Stats stats_;
  // state of the update in flight
  std::shared_ptr<Frame> cur_;
  double depth_mean_ = 0.0;
  int pass_ = 0, req_base_ = 0;
  enum CandState { kDeleted, kInvisible, kTooClose, kSearch };
  struct CandWork { size_t index; int req; CandState state; };  // position in candidates_, request (relative) or -1, decision before the search
  std::vector<CandWork> cand_work_;
  std::vector<int> occurrence_;                            // per candidates_ entry: which occurrence of its point it is
  std::vector<std::pair<std::shared_ptr<Point>, int>> acp_work_;  // AddConnectionsPoints: point, request or -1
  std::vector<std::shared_ptr<Frame>> best_kfs_;
  std::vector<int> ic_req_;                                // InitCandidates: request of (kf k, filtered corner c) or -1
  std::vector<std::shared_ptr<Feature>> ic_feature_;
  struct KfScan { double x, y; bool live; };
  std::vector<KfScan> kf_scan_;                            // InitCandidates: a connected keyframe's features as the seed loop scans them
  std::vector<sdvl_search_req> ic_proto_;                  // InitCandidates: the request of filtered corner c without its keyframe
  std::thread thread_;
  std::atomic<bool> running_{false};
  std::atomic<long> updates_{0};
  Device *tracker_device_ = nullptr;  // whose GPU and point-id sequence the mapper thread joins
};

typedef std::pair<std::shared_ptr<Point>, Vector2d> PointInfo;
typedef std::list<PointInfo> GridCell;

// feature_align.h:42-116.  Same public interface and arithmetic; internally the per-frame working set is flat
// (observation records + index lists) instead of lists of shared_ptr, because this code is what the host CPU spends
// its time on once the kernels are batched.
class FeatureAlign {
 public:
  FeatureAlign(Map *map, Camera *camera, int max_matches);  // feature_align.h:46: draws from a rand() stream of its own
  // the same with the caller's stream (B trackers in one process each own the stream a lone reference process would see)
  FeatureAlign(Map *map, Camera *camera, int max_matches, RandStream *rng);
  ~FeatureAlign();
  void Reproject(const std::shared_ptr<Frame> &frame, const std::shared_ptr<Frame> &last_frame, const std::shared_ptr<Frame> &last_kf,
                 bool reloc = false);
  bool OptimizePose(const std::shared_ptr<Frame> &frame);
  int GetMatches() { return matches_; }
  int GetAttempts() { return num_attempts_; }
  int GetInliers() const { return static_cast<int>(inliers_.size()); }
  int GetOutliers() const { return static_cast<int>(outliers_.size()); }
  // batched form of Reproject: project + shuffle + emit every candidate request; then replay over the results
  void PrepareReproject(const std::shared_ptr<Frame> &frame, const std::shared_ptr<Frame> &last_frame, bool reloc,
                        std::vector<sdvl_search_req> *reqs);
  void FinishReproject(const std::shared_ptr<Frame> &frame, const sdvl_search_res *res);
  // packed form of PrepareReproject: the requests are written, already in the device layout, into an open
  // sdvl_search_begin batch (many trackers share one batch; res of sdvl_search_run + the tracker's first index go to Finish*)
  struct PackedSink {
    sdvl_ctx *ctx = nullptr;
    sdvl_search_req_packed *reqs = nullptr;
    int count = 0, cap = 0;
    uint64_t batch_id = 0;
    double *points = nullptr;  // optional [cap][3]: Point::GetPosition() of every request (sdvl_search_run_chain)
  };
  // the candidates of the last PrepareReproject in SelectPoints order, for sdvl_search_run_chain: request index (made
  // global with `req_offset`) or -1, and the index of the first candidate of the same cell
  void EmitChainCandidates(int req_offset, std::vector<int32_t> *cand_req, std::vector<int32_t> *cand_first) const;
  int MaxMatches() const { return max_matches_; }
  int FoundCount() const { return static_cast<int>(found_.size()); }
  // the next `n` values of the tracker's rand() stream, without advancing it
  void PeekRand(int n, std::vector<int32_t> *out) const;
  void PeekRand(int n, int32_t *out) const;
  void PrepareReprojectPacked(const std::shared_ptr<Frame> &frame, const std::shared_ptr<Frame> &last_frame, bool reloc, PackedSink *sink);
  // FinishReproject without SelectInliers: the pose stage then runs either on the host (SelectInliers + OptimizePose)
  // or batched on the device: EmitPoseJob for every tracker, ONE sdvl_pose_from_matches, CommitPose for every tracker.
  struct PoseBatch {
    std::vector<sdvl_pose_job> jobs;
    std::vector<sdvl_pose_obs> obs;
    std::vector<int32_t> rand_idx, nits;
    void Append(const PoseBatch &o);
  };
  static constexpr int kMaxDevicePoseObs = 1024;
  void FinishSelect(const std::shared_ptr<Frame> &frame, const sdvl_search_res *res, bool build_obs = true);
  bool EmitPoseJob(const std::shared_ptr<Frame> &frame, PoseBatch *batch);  // false: too many matches, use the host path
  void CommitPose(const std::shared_ptr<Frame> &frame, const sdvl_pose_result &r, const int32_t *lists);
  static sdvl_pose_params PoseParams(const Camera &cam);
  // device-resident tables (SDVLBatch::HandleFramesTracked): the host keeps what only it can do — rand()
  void ShuffleCellRanks(uint16_t *rank_of_cell);  // random_shuffle(cell_order_) of SelectPoints (feature_align.cc:103) -> rank per cell
  int GridCells() const { return grid_width_ * grid_height_; }
  void AdvanceRand(int n);                        // the draws SelectInliers made on the device
  void SetTrackedCounts(int matches, int attempts, int inliers, int outliers);
  void SelectInliers(const std::shared_ptr<Frame> &frame);  // host RANSAC (feature_align.cc:152-216)

 private:
  struct CellEntry { int src; Vector2d p; int score; };      // src = index into last_frame->GetFeatures()
  struct Candidate { int src; int req; };                     // req < 0: SearchPoint not evaluated
  struct Obs {                                                // one matched feature of the current frame
    double ax, ay;       // Camera::SimpleProject(feature->GetVector())
    double px, py, pz;   // point->GetPosition()
    double inv_cov;      // 1 / (1 << level)
  };
  void OptimizePoseOnce(const std::shared_ptr<Frame> &frame);
  bool RescueOutliers(const std::shared_ptr<Frame> &frame);
  void RemoveOutliers(const std::shared_ptr<Frame> &frame);
  int CheckReprojectionError(const std::vector<int> &idx, const SE3 &se3, double threshold, std::vector<int> *inliers, std::vector<int> *outliers);
  void ProjectPoints(const std::shared_ptr<Frame> &frame, const std::shared_ptr<Frame> &last_frame);
  void PrepareReprojectImpl(const std::shared_ptr<Frame> &frame, const std::shared_ptr<Frame> &last_frame, bool reloc,
                            std::vector<sdvl_search_req> *reqs, PackedSink *sink);
  bool ConvergePose(const SE3 &frame_pose, const int *idx, int n, SE3 *se3);
  static double GetTukeyValue(double x);

  Map *map_;
  Camera *camera_;
  std::unique_ptr<RandStream> own_rng_;
  RandStream *rng_;
  int cell_size_, max_matches_, grid_width_, grid_height_;
  std::vector<std::vector<CellEntry>> grid_;
  std::vector<int> cell_order_;
  std::vector<Candidate> plan_;             // candidates of all visited cells, in visiting order
  std::vector<int> plan_begin_;             // per visited cell: first candidate (size = cells + 1)
  int req_base_ = 0;                        // index of this tracker's first request in the caller's vector
  std::shared_ptr<Frame> last_frame_;       // source of the projected points of the current Prepare/Finish pair
  std::vector<Feature *> found_;  // features created on the current frame (fs_found); the frame's list keeps them alive
  std::vector<Obs> obs_;                    // parallel to found_
  std::vector<int> inliers_, outliers_;     // indices into found_
  std::vector<double> errors_;              // scratch of ConvergePose
  int matches_, num_attempts_;
  bool relocalizing_;
  static constexpr double KMADNorm = 1.4826;
  static constexpr double KTukeyC = 4.6851 * 4.6851;
};

// wall-clock per stage of SDVLBatch::HandleFrames, accumulated (seconds); index = StageId
enum StageId { ST_UPLOAD_PYR = 0, ST_FAST, ST_SELECT, ST_CORNERS_ORB, ST_PRELUDE, ST_IMAGE_ALIGN, ST_PREPARE, ST_SEARCH, ST_FINISH, ST_POSE, ST_MAPPING,
               ST_EPILOGUE, ST_MAPPER, ST_TOTAL, ST_MAP_CANDIDATES, ST_MAP_CONNECTIONS, ST_MAP_INIT, ST_MAP_FINISH, ST_MAP_BEGIN, ST_MAP_EMIT, ST_MAP_SEARCH, ST_MAP_APPLY, ST_COUNT };
struct StageTimes {
  double t[ST_COUNT] = {0};
  long steps = 0;
  static StageTimes *&Active();  // stage clock of the batch currently running on this thread (may be null)
};

struct FrameStats {
  int state = 0, quality = 0, matches = 0, attempts = 0, inliers = 0, outliers = 0, n_corners = 0, align_meas = 0, keyframe = 0,
      relocalized = 0;
  double pose[7] = {1, 0, 0, 0, 0, 0, 0};
  // traffic accounting: features / GN evaluations of the alignment job, SearchPoint requests / LK iterations
  int align_features = 0, align_iters = 0, search_requests = 0, lk_iters = 0;
};

class SDVLBatch;

// sdvl.h:36-108.  The two-frame homography bootstrap is out of scope: the first frame becomes a keyframe at
// `first_pose` and the Map seeds its points (PlaneMap for synthetic scenes).
class SDVL {
 public:
  enum State { STATE_FIRST_FRAME, STATE_SECOND_FRAME, STATE_RUNNING };
  enum TrackingQuality { TRACKING_GOOD, TRACKING_INSUFFICIENT, TRACKING_BAD };
  // sdvl.h:49: the tracker owns its map (the reference's mapper, MapperMap).  The two-frame homography bootstrap is out of
  // scope, so the first keyframe's points come from a scene plane: z = Config::MapScale() in the first camera's frame (the
  // depth the reference's initialisation normalises the map to) unless SetBootstrapPlane says otherwise.  A Device for the
  // calling thread is created if none is bound yet (GPU 0, or SDVL_GPU).
  explicit SDVL(Camera *camera);
  // additions: a caller-owned map (PlaneMap stub or MapperMap) and first pose — what SDVLBatch's trackers are built with
  SDVL(Camera *camera, Map *map, const SE3 &first_pose = SE3());
  ~SDVL();
  void SetBootstrapPlane(const Vector3d &n, double d);
  // sdvl.h:52-54 (the UI's queries)
  void GetCameraTrail(std::vector<std::pair<SE3, bool>> *positions);
  void GetPoints(std::vector<Vector3d> *positions);
  void GetLastFeatures(std::vector<Vector3i> *positions);
  // sdvl.h:58-60: Start / Stop the mapper thread (threaded mode, main.cc:120); Mapping() = one mapper step (sequential mode)
  void Start() { map_->Start(); }
  void Stop() { map_->Stop(); }
  void Mapping();
  bool HandleFrame(const Image &img);
  SE3 GetPose() const;
  TrackingQuality GetTrackingQuality() const { return tracking_quality_; }
  bool HasMap() { return state_ == STATE_RUNNING; }
  const FrameStats &LastStats() const { return stats_; }
  Map *GetMap() { return map_; }

 private:
  friend class SDVLBatch;
  void CalcTrackingQuality(int matches, int attempts);
  std::unique_ptr<Device> own_device_;  // SDVL(Camera*) on a thread without a Device
  std::unique_ptr<Map> own_map_;        // SDVL(Camera*): the reference's `Map map_` member
  Camera *camera_;
  Map *map_;
  ORBDetector orb_detector_;
  RandStream rng_;
  State state_;
  std::shared_ptr<Frame> current_frame_, last_frame_, last_kf_, pending_kf_;
  TrackingQuality tracking_quality_;
  Vector6d vel_;
  FeatureAlign feature_align_;
  int lost_frames_, matches_, attempts_;
  int frame_counter_;
  SE3 first_pose_;
  FrameStats stats_;
  bool relocalize_pending_ = false;
  // device-resident tracking table of this tracker (SDVLBatch owns the set): valid = the table holds last_frame_'s features
  struct TrackState {
    bool valid = false;
    int feat_buf = 0;
    int slot = 0;                                     // the tracker's index in its batch = which table of the set is its own
    std::shared_ptr<Frame::PointTable> points;        // table index -> Point
    std::vector<sdvl_track_point_stat> stats;         // newest counters of `points`; the Point objects lag behind
    bool stats_dirty = false;
    // rows of a rebuilt table on their way to the device (every tracker fills its own: the rebuilds of a step run in parallel)
    std::vector<sdvl_track_point> up_points;
    std::vector<sdvl_track_feature> up_feats;
    std::vector<Frame *> up_register;                 // keyframes the registry does not know yet
  } track_;
};

// B independent trackers stepping together, one launch per kernel per stage (MI355X-first driver)
class SDVLBatch {
 public:
  SDVLBatch(Device *dev, const std::vector<SDVL *> &trackers, int host_threads);
  ~SDVLBatch();
  // imgs[i] feeds tracker i; stats[i] receives its FrameStats
  void HandleFrames(const std::vector<Image> &imgs, FrameStats *stats);
  // pose stage (RANSAC + refinement) on the device (default) or with the host implementation; process-wide switch,
  // also set by SDVL_POSE_HOST=1 in the environment.  Both produce the same decisions (tests/test_gpu_tracker.py).
  static void SetDevicePose(bool on);
  static bool DevicePose();
  // Device-resident tracking tables (default for a persistent batch on one host thread): last_frame's features and points
  // stay in HBM, a step is ONE submission and ONE wait, the host keeps rand(), the motion model and the keyframe logic.
  // Off (SDVL_NO_TRACK_TABLES=1 or SetTrackTables(false)): the host assembles every request as before.  Same results.
  static void SetTrackTables(bool on);
  static bool TrackTables();
  // Point objects / Feature lists catch up with the device (called on its own whenever the batch leaves the tracked path)
  void SyncHostState();

 private:
  void ParallelFor(int n, const std::function<void(int)> &fn);
 public:
  StageTimes stage_times;
 private:
  Device *dev_;
  std::vector<SDVL *> trk_;
  int threads_;
  std::vector<sdvl_search_req> scratch_reqs_;  // per-step request / pose batches, reused so that they never reallocate
  FeatureAlign::PoseBatch scratch_pose_;
  std::vector<double> scratch_points_;  // sdvl_search_run_chain inputs, same idea
  std::vector<int32_t> chain_cand_req_, chain_cand_first_, chain_rand_;
  // ---- device-resident tables
  bool HandleFramesTracked(const std::vector<Image> &imgs, FrameStats *stats);  // false: not applicable this step
  void HandleFramesGeneric(const std::vector<Image> &imgs, FrameStats *stats);
  bool BuildTable(SDVL &t);
  void SyncStats(SDVL &t);
  void FetchCornerCounts(const std::vector<std::shared_ptr<Frame>> &frames, FrameStats *stats);
  void EpilogueAndMapper(const std::vector<std::shared_ptr<Frame>> &frames, FrameStats *stats, std::vector<std::shared_ptr<Frame>> *kfs,
                         std::vector<int> *kf_owner, bool filter_begun);
  sdvl_track_set *track_ = nullptr;
  int track_cells_ = 0, track_cap_ = 0;
  bool persistent_ = true;  // SDVL::HandleFrame's one-shot batches never build tables
  friend class SDVL;
  std::vector<sdvl_track_job> tr_jobs_;
  std::vector<uint16_t> tr_rank_;
  std::vector<int32_t> tr_rand_;
  std::vector<sdvl_track_result> tr_res_;
  std::vector<sdvl_track_point> tr_up_points_;
  std::vector<sdvl_track_feature> tr_up_feats_;
};

}  // namespace sdvl

#endif  // SDVL_HOST_H_
