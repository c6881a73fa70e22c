// frontend.h — the hot path's classes and nothing else: Device (context + frame pool), Frame (frame.h:41-173), Feature (feature.h:38-105),
// FastDetector (extra/fast_detector.h), ORBDetector (extra/orb_detector.h), ImageAlign (image_align.h), Matcher (matcher.h),
// FeatureAlign (feature_align.h), with the reference's names, argument meaning and return conventions, implemented in frontend.cc on top
// of the C-ABI (include/sdvl_hip.h).  Camera, Point and Map come from frontend_deps.h (the reference tree's own classes); SDVL, the map
// stand-ins and the batch driver are in sdvl_host.h / standalone.cc.
#ifndef SDVL_FRONTEND_H_
#define SDVL_FRONTEND_H_

#include "frontend_deps.h"

namespace sdvl {

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
  // (on a frame whose tracked features are still flat records the new feature goes BEHIND them: MaterializeFeatures keeps that order)
  void AddFeature(const std::shared_ptr<Feature> &f) { features_.push_back(f); scene_depth_hint_valid_ = false; }
  // Round 4: a keyframe of the tracked path keeps its matched features flat; the points learn of their new observation
  // (Point::AddFeature, sdvl.cc:109-111 / feature_align.cc) when — if ever — somebody asks for the Feature objects
  void LinkPointsOnMaterialize() { link_on_materialize_ = true; }
  // positions of all features, flat or not, without turning flat records into objects (FastDetector::LockCell, fast_detector.cc:48-51)
  template <class Fn>
  void ForEachFeaturePosition(Fn fn) const {
    if (flat_)
      for (const sdvl_track_feature_out &f : flat_->feats) fn(f.px[0], f.px[1]);
    for (const std::shared_ptr<Feature> &f : features_) fn(f->GetPosition()(0), f->GetPosition()(1));
  }
  // the Feature objects appended behind the flat records (a keyframe's seeded features), without materialising anything
  const std::vector<std::shared_ptr<Feature>> &ObjectFeatures() const { return features_; }
  int NumFlatFeatures() const { return flat_ ? static_cast<int>(flat_->feats.size()) : 0; }
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
  int GetNumFeatures() const { return (flat_ ? static_cast<int>(flat_->feats.size()) : 0) + static_cast<int>(features_.size()); }
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
  bool link_on_materialize_ = false;
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
  // what ComputePose reads of frame1's features, as the device's records, appended to `feats`
  static void PackFeatures(Frame &frame1, std::vector<sdvl_align_feature> *feats);

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
               ST_EPILOGUE, ST_MAPPER, ST_TOTAL, ST_MAP_CANDIDATES, ST_MAP_CONNECTIONS, ST_MAP_INIT, ST_MAP_FINISH, ST_MAP_BEGIN, ST_MAP_EMIT, ST_MAP_SEARCH, ST_MAP_APPLY, ST_RELOCALIZE, ST_COUNT };
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
  // which form of the step carried this tracker: 0 = the device-resident tables (one submission), 1 = the whole batch went through the
  // host-driven form (HandleFramesGeneric), 2 = this tracker alone was tracked through the per-object calls inside a tabled step
  int host_path = 0;
};

}  // namespace sdvl

#endif  // SDVL_FRONTEND_H_
