// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_math.h header).  PARITY UNPINNED.
//
// ref_tracker.h — the per-frame driver around the kernels: Frame / Feature / Point data model,
// FeatureAlign (grid reprojection, RANSAC, pose refinement) and the SDVL::HandleFrame state machine,
// executed strictly sequentially on one core exactly as the reference tracker thread does.
//   Frame:        /root/reference/frame.cc:34-56,94-163 ; frame.h:41-173
//   Feature:      /root/reference/feature.cc:28-56
//   Point (read side + Promote/Unpromote): /root/reference/point.cc:105-142
//   FeatureAlign: /root/reference/feature_align.cc:33-433
//   SDVL:         /root/reference/sdvl.cc:55-130,179-281 ; Map::NeedKeyframe map.cc:170-188
// The mapper / initialiser (map.cc, homography_init.cc — out of scope) is replaced by a
// "plane map stub": keyframes seed fixed points on their FilterCorners() corners with depth taken from
// a known scene plane.  rand() is the glibc TYPE_3 generator with seed 1, one private stream per tracker.
#ifndef SDVL_ORACLE_REF_TRACKER_H_
#define SDVL_ORACLE_REF_TRACKER_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#include "ref_align.h"
#include "ref_detect.h"
#include "ref_math.h"
#include "ref_orb.h"

namespace sdvlref {

// glibc rand(): random_r TYPE_3 (x^31 + x^3 + 1 additive feedback), srand(1) default state.
struct GlibcRand {
  std::vector<int32_t> ring;  // 31-entry additive-feedback state
  int fi, ri;
  explicit GlibcRand(uint32_t seed = 1) { Seed(seed); }
  void Seed(uint32_t seed) {
    if (seed == 0) seed = 1;
    ring.assign(31, 0);
    ring[0] = static_cast<int32_t>(seed);
    for (int i = 1; i < 31; i++) {
      const long hi = ring[i - 1] / 127773;
      const long lo = ring[i - 1] % 127773;
      long word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      ring[i] = static_cast<int32_t>(word);
    }
    fi = 3;
    ri = 0;
    for (int i = 0; i < 310; i++) Next();
  }
  int Next() {
    uint32_t val = static_cast<uint32_t>(ring[fi]) + static_cast<uint32_t>(ring[ri]);
    ring[fi] = static_cast<int32_t>(val);
    const int result = static_cast<int>(val >> 1);
    if (++fi >= 31) fi = 0;
    if (++ri >= 31) ri = 0;
    return result;
  }
};

// libstdc++ std::random_shuffle(first,last): for i in [1,n): swap(v[i], v[rand() % (i+1)])
template <typename T>
inline void RandomShuffle(std::vector<T> *v, GlibcRand *rng) {
  for (size_t i = 1; i < v->size(); ++i) {
    const size_t j = static_cast<size_t>(rng->Next()) % (i + 1);
    if (i != j) std::swap((*v)[i], (*v)[j]);
  }
}

struct RFrame;
struct RPoint;

struct RFeature {
  RFrame *frame = nullptr;  // owning keyframes are kept alive by the tracker
  std::shared_ptr<RPoint> point;
  Vec2 p{0, 0};
  Vec3 v{0, 0, 1};
  int level = 0;
  uint8_t desc[32] = {0};
  bool has_desc = false;
};

enum PointStatus { P_FOUND, P_NOT_FOUND, P_SEEN, P_UNSEEN, P_OUTLIER };

struct RFrame {
  int id = 0;
  std::vector<std::vector<uint8_t>> pyr_data;
  std::vector<Image> pyr;
  SE3 pose;
  bool is_keyframe = false;
  std::vector<Corner> corners;
  std::vector<std::vector<uint8_t>> descriptors;
  std::vector<int> filtered;
  std::vector<std::shared_ptr<RFeature>> features;

  Vec3 WorldPosition() const { return pose.Inverse().t; }
  int NumPoints() const {
    int c = 0;
    for (auto &f : features) if (f && f->point) c++;
    return c;
  }
};

struct RPoint {
  int id = 0;
  int status = P_NOT_FOUND;
  bool del = false;
  int last_frame = -1;
  int n_successful = 0, n_failed = 0;
  double rho = 1.0, sigma2 = 1.0;
  bool fixed = false;
  Vec3 p3d{0, 0, 0};
  std::shared_ptr<RFeature> init_feature;
  std::vector<std::shared_ptr<RFeature>> features;

  // point.cc:128-142
  Vec3 GetPosition() const {
    if (fixed) return p3d;
    const SE3 w = init_feature->frame->pose.Inverse();
    return w * ((1.0 / rho) * init_feature->v);
  }
  double GetStd() const { return std::sqrt(sigma2); }
};

struct ScenePlane {  // n . X = d in world coordinates (map stub only)
  Vec3 n{0, 0, 1};
  double d = 2.0;
};

struct FrameStats {
  int state = 0;          // 0 first frame, 2 running
  int quality = 0;        // 0 good, 1 insufficient, 2 bad
  int matches = 0, attempts = 0;
  int inliers = 0, outliers = 0;
  int n_corners = 0;
  int align_meas = 0;
  int keyframe = 0;
  int relocalized = 0;
  double pose[7] = {1, 0, 0, 0, 0, 0, 0};
};

struct Tracker {
  Params prm;
  Camera cam;
  ScenePlane plane;
  GlibcRand rng;
  OrbDetector orb;
  int frame_counter = 0, point_counter = 0;
  bool running = false;
  int lost_frames = 0;
  int quality = 0;
  double vel[6] = {0, 0, 0, 0, 0, 0};
  std::shared_ptr<RFrame> last_frame, last_kf, current;
  std::shared_ptr<RFrame> map_last_kf;  // Map::last_kf_ (map.cc:157), distinct from SDVL::last_kf_ after a relocalisation
  std::vector<std::shared_ptr<RFrame>> keyframes;
  std::vector<std::shared_ptr<RPoint>> points_trash;
  int last_matches = 0;  // Map::last_matches_
  // FeatureAlign state
  int grid_width, grid_height;
  std::vector<std::vector<std::pair<std::shared_ptr<RPoint>, Vec2>>> grid;
  std::vector<int> cell_order;
  int matches = 0, attempts = 0;
  bool relocalizing = false;
  std::vector<std::shared_ptr<RFeature>> inliers, outliers;
  SE3 init_pose;

  Tracker(const Params &p, const Camera &c, const ScenePlane &pl, const SE3 &first_pose)
      : prm(p), cam(c), plane(pl), rng(1), orb(p.orb_size), init_pose(first_pose) {
    // FeatureAlign ctor, feature_align.cc:33-54
    grid_width = static_cast<int>(std::ceil(static_cast<double>(cam.width) / prm.cell_size));
    grid_height = static_cast<int>(std::ceil(static_cast<double>(cam.height) / prm.cell_size));
    const int size = grid_width * grid_height;
    grid.resize(size);
    for (int i = 0; i < size; ++i) cell_order.push_back(i);
    RandomShuffle(&cell_order, &rng);
  }

  // Frame::Frame, frame.cc:34-56 (+ CreatePyramid :114-120, CreateCorners :122-131)
  std::shared_ptr<RFrame> MakeFrame(const uint8_t *img, int stride, bool corners) {
    auto f = std::make_shared<RFrame>();
    f->id = frame_counter++;
    const int W = static_cast<int>(cam.width), H = static_cast<int>(cam.height);
    f->pyr_data.resize(prm.pyramid_levels);
    f->pyr.resize(prm.pyramid_levels);
    f->pyr_data[0].resize(static_cast<size_t>(W) * H);
    for (int y = 0; y < H; y++) std::memcpy(&f->pyr_data[0][static_cast<size_t>(y) * W], img + static_cast<size_t>(y) * stride, W);
    f->pyr[0] = Image{f->pyr_data[0].data(), W, H, W};
    for (int i = 1; i < prm.pyramid_levels; i++) {
      const int w = f->pyr[i - 1].cols / 2, h = f->pyr[i - 1].rows / 2;
      f->pyr_data[i].resize(static_cast<size_t>(w) * h);
      PyrDown(f->pyr[i - 1], f->pyr_data[i].data(), w);
      f->pyr[i] = Image{f->pyr_data[i].data(), w, h, w};
    }
    if (corners) CreateCorners(f.get(), prm.num_features);
    return f;
  }
  void CreateCorners(RFrame *f, int nfeatures) {
    DetectPyramid(f->pyr, prm, nfeatures, &f->corners);
    if (prm.use_orb) f->descriptors.assign(f->corners.size(), {});
  }

  // Frame::FilterCorners, frame.cc:133-163
  void FilterCorners(RFrame *f) {
    CornerGrid det(static_cast<int>(cam.width), static_cast<int>(cam.height), prm);
    for (auto &ft : f->features) det.LockCell(ft->p.x, ft->p.y);
    det.FilterCorners(f->pyr, f->corners, prm, &f->filtered);
    if (prm.use_orb) {
      for (int index : f->filtered) {
        const Corner &c = f->corners[index];
        if (f->descriptors[index].empty()) {
          f->descriptors[index].resize(32);
          orb.GetDescriptor(f->pyr[c.level], c.x, c.y, f->descriptors[index].data());
        }
      }
    }
  }

  // Map stub (replaces Map::InitCandidates map.cc:262-400 + the depth filter): one fixed point per filtered
  // corner, depth from the scene plane along the feature bearing, in the keyframe's own estimated pose.
  void SeedPoints(const std::shared_ptr<RFrame> &kf) {
    FilterCorners(kf.get());
    const SE3 world = kf->pose.Inverse();
    const Mat3 Rw = world.Rotation();
    for (int index : kf->filtered) {
      const Corner &c = kf->corners[index];
      const int scale = (1 << c.level);
      auto ft = std::make_shared<RFeature>();
      ft->frame = kf.get();
      ft->p = Vec2{static_cast<double>(c.x * scale), static_cast<double>(c.y * scale)};
      ft->v = cam.Unproject(ft->p);
      ft->level = c.level;
      if (prm.use_orb) {
        std::memcpy(ft->desc, kf->descriptors[index].data(), 32);
        ft->has_desc = true;
      }
      const Vec3 ray = MatVec(Rw, ft->v);
      const double denom = Dot(plane.n, ray);
      if (!(std::fabs(denom) > 1e-9)) continue;
      const double s = (plane.d - Dot(plane.n, world.t)) / denom;
      if (!(s > 0.05)) continue;
      auto pt = std::make_shared<RPoint>();
      pt->id = point_counter++;
      pt->init_feature = ft;
      pt->rho = 1.0 / s;
      pt->sigma2 = (0.05 * pt->rho) * (0.05 * pt->rho);
      pt->fixed = true;
      pt->p3d = world * (s * ft->v);
      ft->point = pt;
      kf->features.push_back(ft);
      pt->features.insert(pt->features.begin(), ft);
    }
  }

  // SDVL::HandleFrame, sdvl.cc:55-130
  FrameStats HandleFrame(const uint8_t *img, int stride) {
    FrameStats st;
    current = MakeFrame(img, stride, true);
    st.n_corners = static_cast<int>(current->corners.size());
    if (!running) {
      current->pose = init_pose;
      current->is_keyframe = true;
      keyframes.push_back(current);
      map_last_kf = current;
      SeedPoints(current);
      last_frame = current;
      last_kf = current;
      running = true;
      st.state = 0;
      st.keyframe = 1;
      last_matches = 0;
    } else {
      st.state = 2;
      bool relocalize = lost_frames >= 3;
      if (relocalize) {
        for (int i = 0; i < 6; i++) vel[i] = 0.0;
        if (Relocalize(&last_kf)) {
          last_frame = last_kf;
          relocalize = false;
          st.relocalized = 1;
        }
      }
      if (!relocalize) {
        // SetMotionModel, sdvl.cc:278-281
        current->pose = SE3Exp(vel) * last_frame->pose;
        st.align_meas = ProcessFrame(last_frame, last_kf);
        // GetMotionModel, sdvl.cc:266-276
        {
          const SE3 mov = current->pose * last_frame->pose.Inverse();
          double v[6];
          SE3Log(mov, v);
          for (int i = 0; i < 6; i++) vel[i] = 0.9 * (0.5 * v[i] + 0.5 * vel[i]);
        }
        CalcTrackingQuality(matches, attempts);
        if (quality != 2) {
          if (quality == 0 && NeedKeyframe(current, matches)) {
            for (auto &ft : current->features)
              if (ft->point) ft->point->features.insert(ft->point->features.begin(), ft);
            current->is_keyframe = true;
            keyframes.push_back(current);
            map_last_kf = current;
            last_kf = current;
            SeedPoints(current);  // sequential-mode mapper work, outside the reference's timing window
            st.keyframe = 1;
          }
          last_frame = current;
        }
      }
    }
    st.quality = quality;
    st.matches = matches;
    st.attempts = attempts;
    st.inliers = static_cast<int>(inliers.size());
    st.outliers = static_cast<int>(outliers.size());
    st.pose[0] = current->pose.q0; st.pose[1] = current->pose.q1; st.pose[2] = current->pose.q2; st.pose[3] = current->pose.q3;
    st.pose[4] = current->pose.t.x; st.pose[5] = current->pose.t.y; st.pose[6] = current->pose.t.z;
    current = nullptr;
    EmptyTrash();
    return st;
  }

  // Map::NeedKeyframe, map.cc:170-188
  bool NeedKeyframe(const std::shared_ptr<RFrame> &frame, int) {
    const int npoints = frame->NumPoints();
    const bool enough_its = (frame->id - map_last_kf->id) >= prm.min_keyframe_its;
    const bool lost_many = npoints < last_matches * prm.lost_ratio;
    const bool lost_some = npoints < last_matches * 0.9;
    last_matches = std::max(last_matches, npoints);
    if ((enough_its && lost_some) || lost_many) {
      last_matches = npoints;
      return true;
    }
    return false;
  }

  // Map::EmptyTrash (points part), map.cc:207-259
  void EmptyTrash() {
    for (auto &p : points_trash) {
      for (auto &f : p->features) f->point = nullptr;
      p->features.clear();
      p->del = true;
    }
    points_trash.clear();
  }

  // SDVL::CalcTrackingQuality, sdvl.cc:240-264
  void CalcTrackingQuality(int m, int a) {
    const double ratio = (a == 0) ? 0.0 : static_cast<double>(m) / static_cast<double>(a);
    if (ratio > 0.2) { quality = 0; lost_frames = 0; return; }
    if (m < prm.min_matches) { quality = 2; lost_frames++; return; }
    lost_frames = 0;
    quality = 1;
  }

  static void AlignFeaturesOf(const RFrame &f1, std::vector<AlignFeature> *out) {
    const Vec3 first_pos = f1.WorldPosition();
    for (auto &ft : f1.features) {
      AlignFeature a;
      a.px = ft->p.x; a.py = ft->p.y; a.f = ft->v;
      a.valid = (ft->point && !ft->point->del) ? 1 : 0;
      a.depth = a.valid ? Norm(ft->point->GetPosition() - first_pos) : 0.0;
      out->push_back(a);
    }
  }

  // SDVL::ProcessFrame, sdvl.cc:179-203
  int ProcessFrame(const std::shared_ptr<RFrame> &lastf, const std::shared_ptr<RFrame> &lastkf) {
    int n = 0;
    {
      ImageAlign ia;
      std::vector<AlignFeature> feats;
      AlignFeaturesOf(*lastf, &feats);
      if (!feats.empty()) {
        SE3 T = current->pose * lastf->pose.Inverse();
        n = ia.ComputePose(lastf->pyr, current->pyr, feats, cam, prm, &T, false);
        current->pose = T * lastf->pose;
      }
    }
    Reproject(current, lastf, lastkf, false);
    OptimizePoseAll(current);
    return n;
  }

  // SDVL::Relocalize, sdvl.cc:205-238
  bool Relocalize(std::shared_ptr<RFrame> *lkf) {
    for (auto it = keyframes.rbegin(); it != keyframes.rend(); it++) {
      std::shared_ptr<RFrame> cframe = *it;
      current->pose = cframe->pose;
      ImageAlign ia;
      std::vector<AlignFeature> feats;
      AlignFeaturesOf(*cframe, &feats);
      if (feats.empty()) continue;  // ComputePose returns 0 with error_ = 1e10
      SE3 T = current->pose * cframe->pose.Inverse();
      ia.ComputePose(cframe->pyr, current->pyr, feats, cam, prm, &T, true);
      current->pose = T * cframe->pose;
      if (ia.error >= 0.001) continue;
      Reproject(current, cframe, cframe, true);
      if (matches >= prm.min_matches) {
        *lkf = cframe;
        return true;
      }
    }
    return false;
  }

  // FeatureAlign::Reproject, feature_align.cc:59-71
  void Reproject(const std::shared_ptr<RFrame> &frame, const std::shared_ptr<RFrame> &lastf, const std::shared_ptr<RFrame> &lastkf, bool reloc) {
    std::vector<std::shared_ptr<RFeature>> selected;
    inliers.clear();
    outliers.clear();
    relocalizing = reloc;
    SelectPoints(frame, lastf, lastkf, &selected);
    SelectInliers(frame, selected, &inliers, &outliers);
  }

  // FeatureAlign::OptimizePose(frame), feature_align.cc:73-82
  void OptimizePoseAll(const std::shared_ptr<RFrame> &frame) {
    OptimizePose(frame, &inliers, &outliers);
    if (RescueOutliers(frame, &inliers, &outliers)) OptimizePose(frame, &inliers, &outliers);
    RemoveOutliers(frame, &outliers);
  }

  bool FrameProject(const SE3 &pose, const Vec3 &p3d, Vec2 *p2d) const {
    const Vec3 rel = pose * p3d;
    if (rel.z < 0.0) return false;
    *p2d = cam.Project(rel);
    return true;
  }

  // feature_align.cc:285-339
  void ProjectPoints(const std::shared_ptr<RFrame> &frame, const std::shared_ptr<RFrame> &lastf) {
    matches = 0;
    attempts = 0;
    for (auto &c : grid) c.clear();
    for (auto &ft : lastf->features) {
      if (!ft) continue;
      std::shared_ptr<RPoint> point = ft->point;
      if (!point || point->del) continue;
      if (frame->id == point->last_frame) continue;
      ProjectPoint(frame, point);
      if (!relocalizing) point->last_frame = frame->id;
    }
  }
  bool ProjectPoint(const std::shared_ptr<RFrame> &frame, const std::shared_ptr<RPoint> &point) {
    Vec2 p;
    if (!FrameProject(frame->pose, point->GetPosition(), &p)) { point->status = P_UNSEEN; return false; }
    if (!cam.IsInsideImage(static_cast<int>(p.x), static_cast<int>(p.y), prm.patch_size)) { point->status = P_UNSEEN; return false; }
    const int k = static_cast<int>(p.y / prm.cell_size) * grid_width + static_cast<int>(p.x / prm.cell_size);
    grid.at(k).push_back(std::make_pair(point, p));
    point->status = P_SEEN;
    return true;
  }

  // feature_align.cc:88-150
  void SelectPoints(const std::shared_ptr<RFrame> &frame, const std::shared_ptr<RFrame> &lastf, const std::shared_ptr<RFrame> &,
                    std::vector<std::shared_ptr<RFeature>> *fs_found) {
    Matcher matcher(prm.patch_size, prm, cam);
    ProjectPoints(frame, lastf);
    matches = 0;
    attempts = 0;
    RandomShuffle(&cell_order, &rng);
    const int size = static_cast<int>(grid.size());
    SearchCur cur{&frame->pyr, frame->pose, &frame->corners, &frame->descriptors};
    for (int i = 0; i < size && matches < prm.max_matches; i++) {
      bool found = false;
      auto &cell = grid.at(cell_order[i]);
      std::stable_sort(cell.begin(), cell.end(),
                       [](const std::pair<std::shared_ptr<RPoint>, Vec2> &a, const std::pair<std::shared_ptr<RPoint>, Vec2> &b) {
                         return a.first->n_successful > b.first->n_successful;
                       });
      for (auto it = cell.begin(); it != cell.end() && !found; it++) {
        std::shared_ptr<RPoint> point = it->first;
        if (point->del) continue;
        std::shared_ptr<RFeature> feature = point->init_feature;
        if (!feature) continue;
        attempts++;
        Vec2 pos = it->second;
        int level = 0;
        cur.pose = frame->pose;
        SearchRef ref;
        ref.ref_pyr = &feature->frame->pyr;
        ref.ref_pose = feature->frame->pose;
        ref.px = feature->p.x; ref.py = feature->p.y; ref.f = feature->v; ref.level = feature->level;
        std::memcpy(ref.desc, feature->desc, 32);
        found = matcher.SearchPoint(&cur, ref, point->rho, point->GetStd(), point->fixed, &pos, &level);
        if (found) {
          if (!relocalizing) {
            point->n_successful++;  // Promote, point.cc:105-109
            point->n_failed = 0;
            auto nf = std::make_shared<RFeature>();
            nf->frame = frame.get();
            nf->p = pos;
            nf->v = cam.Unproject(pos);
            nf->level = level;
            nf->point = point;
            frame->features.push_back(nf);
            point->status = P_FOUND;
            fs_found->push_back(nf);
          }
          matches++;
        } else {
          if (!relocalizing) {
            point->n_failed++;  // Unpromote, point.cc:111-118 (b_ is map state, not read by the path)
            if (point->n_failed > prm.max_failed) points_trash.push_back(point);
            point->status = P_NOT_FOUND;
          }
        }
      }
    }
  }

  static Vec2 SimpleProject(const Vec3 &p) { return {p.x / p.z, p.y / p.z}; }

  // feature_align.cc:258-283
  int CheckReprojectionError(const std::vector<std::shared_ptr<RFeature>> &features, const SE3 &se3, double threshold,
                             std::vector<std::shared_ptr<RFeature>> *in, std::vector<std::shared_ptr<RFeature>> *out) {
    int valids = 0;
    for (auto &ft : features) {
      std::shared_ptr<RPoint> point = ft->point;
      if (!point) continue;
      const Vec3 pos = se3 * point->GetPosition();
      const Vec2 a = SimpleProject(ft->v), b = SimpleProject(pos);
      double ex = a.x - b.x, ey = a.y - b.y;
      const double sqrt_inv_cov = 1.0 / (1 << ft->level);
      ex *= sqrt_inv_cov;
      ey *= sqrt_inv_cov;
      if (std::sqrt(ex * ex + ey * ey) <= threshold) {
        valids++;
        if (in) in->push_back(ft);
      } else {
        if (out) out->push_back(ft);
      }
    }
    return valids;
  }

  static double TukeyValue(double x) {  // feature_align.cc:423-431
    const double kTukeyC = 4.6851 * 4.6851;
    const double x_square = x * x;
    if (x_square <= kTukeyC) {
      const double tmp = 1.0 - x_square / kTukeyC;
      return tmp * tmp;
    }
    return 0.0;
  }

  // feature_align.cc:341-421
  bool ConvergePose(const std::shared_ptr<RFrame> &frame, const std::vector<std::shared_ptr<RFeature>> &features, SE3 *se3) {
    const double kMADNorm = 1.4826;
    SE3 last_se3 = frame->pose;
    *se3 = last_se3;
    double chi2 = 0.0;
    std::vector<double> errors;
    for (auto &ft : features) {
      std::shared_ptr<RPoint> point = ft->point;
      if (!point) continue;
      const Vec3 pos = (*se3) * point->GetPosition();
      const Vec2 a = SimpleProject(ft->v), b = SimpleProject(pos);
      double ex = a.x - b.x, ey = a.y - b.y;
      const double s = 1.0 / (1 << ft->level);
      ex *= s; ey *= s;
      errors.push_back(std::sqrt(ex * ex + ey * ey));
    }
    if (errors.empty()) return false;
    // GetMedianVector, extra/utils.cc:215-220
    auto mid = errors.begin() + static_cast<long>(std::floor(errors.size() / 2));
    std::nth_element(errors.begin(), mid, errors.end());
    double scale = kMADNorm * (*mid);
    for (int i = 0; i < prm.max_optim_pose_its; i++) {
      double A[6][6], b[6];
      for (int r = 0; r < 6; r++) { b[r] = 0.0; for (int c = 0; c < 6; c++) A[r][c] = 0.0; }
      double new_chi2 = 0.0;
      if (i == 5) scale = 0.85 / cam.fx;
      for (auto &ft : features) {
        std::shared_ptr<RPoint> point = ft->point;
        if (!point) continue;
        const Vec3 pos = (*se3) * point->GetPosition();
        double J[2][6];
        Jacobian3DToPlane(pos, J);
        const Vec2 pa = SimpleProject(ft->v), pb = SimpleProject(pos);
        double ex = pa.x - pb.x, ey = pa.y - pb.y;
        const double sqrt_inv_cov = 1.0 / (1 << ft->level);
        ex *= sqrt_inv_cov; ey *= sqrt_inv_cov;
        for (int c = 0; c < 6; c++) { J[0][c] *= sqrt_inv_cov; J[1][c] *= sqrt_inv_cov; }
        const double weight = TukeyValue(std::sqrt(ex * ex + ey * ey) / scale);
        for (int r = 0; r < 6; r++) {
          for (int c = 0; c < 6; c++) A[r][c] += (J[0][r] * J[0][c] + J[1][r] * J[1][c]) * weight;
          b[r] -= (J[0][r] * ex + J[1][r] * ey) * weight;
        }
        new_chi2 += (ex * ex + ey * ey) * weight;
      }
      double dT[6];
      LdltSolve6(A, b, dT);
      if ((i > 0 && new_chi2 > chi2) || std::isnan(dT[0])) {
        *se3 = last_se3;
        break;
      }
      const SE3 T_new = SE3Exp(dT) * (*se3);
      last_se3 = *se3;
      *se3 = T_new;
      chi2 = new_chi2;
      if (AbsMax6(dT) <= 1e-10) break;
    }
    return true;
  }

  // feature_align.cc:152-216
  void SelectInliers(const std::shared_ptr<RFrame> &frame, std::vector<std::shared_ptr<RFeature>> &fs_found,
                     std::vector<std::shared_ptr<RFeature>> *in, std::vector<std::shared_ptr<RFeature>> *out) {
    in->clear();
    out->clear();
    if (fs_found.empty()) return;
    const int size = static_cast<int>(fs_found.size());
    const int npoints = std::min(prm.max_ransac_points, size);
    std::vector<std::shared_ptr<RFeature>> selected;
    SE3 se3, best_se3;
    const double sprob = 0.99;
    int nits = prm.max_ransac_its;
    int best_supporters = 0;
    int it = 0;
    const double thr = prm.inlier_error_threshold / cam.fx;
    while (it < nits) {
      selected.clear();
      const int index = rng.Next() % size;
      for (int i = 0; i < npoints; i++) selected.push_back(fs_found.at((index + i) % size));
      if (!ConvergePose(frame, selected, &se3)) { it++; continue; }
      const int supporters = CheckReprojectionError(fs_found, se3, thr, nullptr, nullptr);
      if (supporters > best_supporters) {
        best_supporters = supporters;
        best_se3 = se3;
        const double epsilon = 1.0 - (static_cast<double>(supporters) / static_cast<double>(size));
        double tmp = 1.0 - epsilon;
        for (int k = 1; k < npoints; k++) tmp *= tmp;
        if (tmp < 1e-5) nits = prm.max_ransac_its;
        else nits = std::min(prm.max_ransac_its, static_cast<int>(std::log(1.0 - sprob) / std::log(1.0 - tmp)));
      }
      it++;
    }
    CheckReprojectionError(fs_found, best_se3, thr, in, out);
  }

  // feature_align.cc:218-230
  void OptimizePose(const std::shared_ptr<RFrame> &frame, std::vector<std::shared_ptr<RFeature>> *features,
                    std::vector<std::shared_ptr<RFeature>> *out) {
    SE3 se3 = frame->pose;
    if (!ConvergePose(frame, *features, &se3)) return;
    frame->pose = se3;
    std::vector<std::shared_ptr<RFeature>> cfeatures = *features;
    features->clear();
    CheckReprojectionError(cfeatures, frame->pose, prm.inlier_error_threshold / cam.fx, features, out);
  }

  // feature_align.cc:232-243
  bool RescueOutliers(const std::shared_ptr<RFrame> &frame, std::vector<std::shared_ptr<RFeature>> *in,
                      std::vector<std::shared_ptr<RFeature>> *out) {
    const int init_inliers = static_cast<int>(in->size());
    std::vector<std::shared_ptr<RFeature>> cfeatures = *out;
    out->clear();
    CheckReprojectionError(cfeatures, frame->pose, 2 * prm.inlier_error_threshold / cam.fx, in, out);
    return static_cast<int>(in->size()) > init_inliers;
  }

  // feature_align.cc:245-256
  void RemoveOutliers(const std::shared_ptr<RFrame> &, std::vector<std::shared_ptr<RFeature>> *out) {
    for (auto &ft : *out) {
      std::shared_ptr<RPoint> p = ft->point;
      if (!p) continue;
      ft->point = nullptr;
      p->status = P_NOT_FOUND;
    }
  }
};

}  // namespace sdvlref

#endif  // SDVL_ORACLE_REF_TRACKER_H_
