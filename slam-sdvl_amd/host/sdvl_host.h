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
// Round 4: the header is in three parts — frontend_deps.h (Camera, Point, Map: the reference tree's own classes), frontend.h (the hot
// path's classes) and, below, what only the standalone build needs: the map stand-ins, SDVL and the batch driver.
#ifndef SDVL_HOST_H_
#define SDVL_HOST_H_

#include "frontend.h"

namespace sdvl {

// Map stand-in for synthetic scenes: seeds one FIXED point per FilterCorners() corner of a keyframe, depth from a
// known scene plane n.X = d (replaces homography_init + depth filter, both out of scope).
class PlaneMap : public Map {
 public:
  PlaneMap(const Vector3d &n, double d) : n_(n), d_(d) {}
  void InitCandidates(const std::shared_ptr<Frame> &kf) override;
  void SeedFromFiltered(const std::shared_ptr<Frame> &kf);  // after Frame::FilterCorners()
  void SetPlane(const Vector3d &n, double d) { n_ = n; d_ = d; }
  // Round 5: the stub honours SDVL.max_keyframes like the reference's map (map.cc:190-205,692-706: once the list is full, the
  // keyframe furthest from the new one goes to the trash) — and, being the owner of the points it seeded there, deletes the points
  // whose FIRST observation lies in the culled keyframe with it (the reference keeps a culled keyframe's image alive for as long as
  // any point names it; here the HBM frame goes back to the pool, so whole sequences run in bounded memory).  The oracle's plane
  // map does the same (oracle/ref_tracker.h); with the reference's cfg values (max_keyframes 1000) S-A never gets there.
  void LimitKeyframes(const std::shared_ptr<Frame> &frame) override;
  void EmptyTrash() override;

 private:
  Vector3d n_;
  double d_;
  std::vector<std::shared_ptr<Frame>> culled_;
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
  // Round 5, an addition for callers that READ a sequence (main.cc's Video.type 1: images from disk are there before they are needed):
  // name the image the NEXT HandleFrame call will be given (a device image, e.g. what Camera::UndistortImage returns, kept alive and
  // unchanged until that call).  Its pyramid and corner detection are queued behind the coming call's chain and run while the host
  // finishes that call.  Same results; an unused or mismatching look-ahead is dropped.
  void SetNextImage(const Image &next);
  SE3 GetPose() const;
  TrackingQuality GetTrackingQuality() const { return tracking_quality_; }
  bool HasMap() { return state_ == STATE_RUNNING; }
  const FrameStats &LastStats() const { return stats_; }
  Map *GetMap() { return map_; }
  // wall clock per host stage of this tracker's own HandleFrame calls (null before the first call); diagnostic
  const StageTimes *HandleFrameStageTimes() const;

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
  // Round 5: HandleFrame() steps through a batch of one that LIVES with the tracker, so a lone camera gets the device-resident
  // tracking tables too — a tracked frame is one submission and one wait instead of the host-driven stage-by-stage form
  // (SDVL_HANDLEFRAME_ONE_SHOT=1: a fresh batch per call, rounds 1-4)
  std::unique_ptr<SDVLBatch> self_batch_;
  std::vector<Image> next_image_;  // SetNextImage: handed to the batch by the HandleFrame call it belongs to
  void SyncSelfBatch();
  // device-resident tracking table of this tracker (SDVLBatch owns the set): valid = the table holds last_frame_'s features
  struct TrackState {
    bool valid = false;
    int feat_buf = 0;
    int slot = 0;                                     // the tracker's index in its batch = which table of the set is its own
    const void *owner = nullptr;                      // the SDVLBatch whose set holds that table (a tracker may be stepped by a farm batch AND through HandleFrame)
    std::shared_ptr<Frame::PointTable> points;        // table index -> Point
    std::vector<sdvl_track_point_stat> stats;         // newest counters of `points`; the Point objects lag behind
    bool stats_dirty = false;
    // rows of a rebuilt table on their way to the device (every tracker fills its own: the rebuilds of a step run in parallel)
    std::vector<sdvl_track_point> up_points;
    std::vector<sdvl_track_feature> up_feats;
    std::vector<Frame *> up_register;                 // keyframes the registry does not know yet
    // round 4: the step made the frame a keyframe and left the table current on the device: the rows of the points the map seeds on
    // it are APPENDED (sdvl_track_append) instead of the table being rebuilt from objects; first_seed = the keyframe's object
    // features before the seeding (what comes behind are the seeds)
    bool append_pending = false;
    size_t first_seed = 0;
  } track_;
  // Relocalize (sdvl.cc:205-238) aligns every new frame against the same keyframes: their feature records live in the batch's
  // sdvl_align_store from the first lost frame until the map changes (Map::Version)
  struct RelocCache {
    const void *owner = nullptr;               // the SDVLBatch whose store holds the records
    unsigned long long version = ~0ull;        // Map::Version() the records were packed at
    unsigned long long epoch = 0;              // the store's epoch (a store that was reset has lost them)
    std::vector<std::shared_ptr<Frame>> kfs;   // newest first, the order Relocalize visits them in
    std::vector<int32_t> begin;                // kfs.size() + 1 record offsets in the store
    std::vector<double> T;                     // 7 per keyframe: start pose * keyframe pose^-1 (image_align.cc:66 with frame2 at the keyframe's pose)
  } reloc_;
};

// B independent trackers stepping together, one launch per kernel per stage (MI355X-first driver)
class SDVLBatch {
 public:
  SDVLBatch(Device *dev, const std::vector<SDVL *> &trackers, int host_threads);
  ~SDVLBatch();
  // imgs[i] feeds tracker i; stats[i] receives its FrameStats
  void HandleFrames(const std::vector<Image> &imgs, FrameStats *stats);
  // Round 4: a caller that already holds the images of the NEXT step (a sequence from disk, frames resident in HBM) names them before
  // the current step: their pyramids and corner detection — all that depends on the image alone, 45 % of a step's vector instructions —
  // are queued right behind the current step's search / pose chain and run while the host waits for that chain and does its keyframe
  // bookkeeping; the next HandleFrames with these images picks the frames up.  Same kernels on the same inputs: same results.
  // The images must stay valid until that call (device images are aliased).  An unused look-ahead is dropped.
  void SetNextImages(const std::vector<Image> &next);
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
  std::vector<Image> next_imgs_;                        // SetNextImages: what the next step will be given
  std::vector<std::shared_ptr<Frame>> ahead_frames_;    // their frames, pyramids and detection queued
  std::vector<const void *> ahead_src_;                 // the images they were made from
  void HandleFramesGeneric(const std::vector<Image> &imgs, FrameStats *stats);
  bool BuildTable(SDVL &t);
  bool UploadTables(const std::vector<int> &need, std::vector<char> *built);
  void RelocalizeLost(const std::vector<int> &lost, FrameStats *stats, std::vector<char> *found);
  int TrackOnHost(SDVL &t, FrameStats *st);
  bool AppendSeeds(SDVL &t, const std::shared_ptr<Frame> &kf);  // rows of the points seeded on kf -> track_.up_points / up_feats
  void SyncStats(SDVL &t);
  void FetchCornerCounts(const std::vector<std::shared_ptr<Frame>> &frames, FrameStats *stats);
  void EpilogueAndMapper(const std::vector<std::shared_ptr<Frame>> &frames, FrameStats *stats, std::vector<std::shared_ptr<Frame>> *kfs,
                         std::vector<int> *kf_owner, bool filter_begun);
  sdvl_track_set *track_ = nullptr;
  // keyframe feature records of the trackers that are relocalising (SDVL::RelocCache): one bump-allocated store per batch
  sdvl_align_store *reloc_store_ = nullptr;
  int reloc_cap_ = 0, reloc_used_ = 0;
  unsigned long long reloc_epoch_ = 0;  // 0: no store yet
  void RelocAlign(const std::vector<sdvl_align_job> &jobs, const sdvl_align_params &ap, std::vector<sdvl_align_result> *res);
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
