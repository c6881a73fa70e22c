// frontend_deps.h — the classes the hot path's classes (frontend.h) are written AGAINST and do not implement: Camera (camera.h:34-135),
// Point (point.h:37-147) and the tracker's view of Map (map.h:49-61).  In the reference's tree these are its own camera.h / point.h /
// map.h, implemented by its own camera.cc / point.cc / map.cc; here standalone.cc implements them, and host/minimal_deps.cc is a
// second, independent implementation of exactly the members frontend.cc leaves undefined —
//     Camera::Project, Camera::Unproject, Point::GetPosition, Point::GetStd, Point::Promote, Point::Unpromote
// (all six are members of the reference's classes with the reference's signatures) — against which `make frontend_link_check` links
// frontend.o without standalone.cc (tests/test_frontend_split.py keeps the list honest with nm).
#ifndef SDVL_FRONTEND_DEPS_H_
#define SDVL_FRONTEND_DEPS_H_

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
  void DeletePoint(const std::shared_ptr<Point> &p) { points_trash_.push_back(p); version_++; }
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
  // (an addition) changes whenever the keyframe list or a point behind a keyframe's feature changed (added / culled keyframe, deleted
  // point, a mapper update): what SDVLBatch's relocalisation cache of keyframe feature records is checked against
  unsigned long long Version() const { return version_; }
  void Touch() { version_++; }
  // mapper work for a fresh keyframe (Map::InitCandidates stand-in); called outside the tracking stages
  virtual void InitCandidates(const std::shared_ptr<Frame> &) {}

 protected:
  std::vector<std::shared_ptr<Frame>> keyframes_;
  std::vector<std::shared_ptr<Point>> points_trash_;
  bool tables_dirty_ = false;
  unsigned long long version_ = 0;
  std::shared_ptr<Frame> last_kf_;
  int last_matches_ = 0;
  std::mutex mutex_map_;
};

}  // namespace sdvl

#endif  // SDVL_FRONTEND_DEPS_H_
