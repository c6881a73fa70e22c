// ORACLE — TEST INFRASTRUCTURE ONLY.  extern "C" surface of the CPU restatement (see sdvl_oracle.h).
// PARITY UNPINNED (see ref_math.h).
#include "sdvl_oracle.h"

#include <cstring>
#include <vector>

#include "ref_align.h"
#include "ref_detect.h"
#include "ref_math.h"
#include "ref_orb.h"
#include "ref_tracker.h"
#include "ref_undistort.h"

using namespace sdvlref;

namespace {

Params ToParams(const sdvl_ref_params *p) {
  Params q;
  q.pyramid_levels = p->pyramid_levels; q.cell_size = p->cell_size; q.max_fast_levels = p->max_fast_levels;
  q.fast_threshold = p->fast_threshold; q.num_features = p->num_features; q.use_orb = p->use_orb;
  q.orb_size = p->orb_size; q.patch_size = p->patch_size; q.max_align_its = p->max_align_its;
  q.search_size = p->search_size; q.align_patch_size = p->align_patch_size; q.max_align_level = p->max_align_level;
  q.min_align_level = p->min_align_level; q.max_img_align_its = p->max_img_align_its;
  q.min_feature_score = p->min_feature_score; q.max_matches = p->max_matches; q.min_matches = p->min_matches;
  q.max_failed = p->max_failed; q.max_optim_pose_its = p->max_optim_pose_its; q.max_ransac_points = p->max_ransac_points;
  q.max_ransac_its = p->max_ransac_its; q.min_keyframe_its = p->min_keyframe_its;
  q.inlier_error_threshold = p->inlier_error_threshold; q.lost_ratio = p->lost_ratio;
  return q;
}

struct Pyramid {
  std::vector<std::vector<uint8_t>> data;
  std::vector<Image> lv;
  Pyramid(const uint8_t *img, int w, int h, int stride, int levels) {
    data.resize(levels);
    lv.resize(levels);
    data[0].resize(static_cast<size_t>(w) * h);
    for (int y = 0; y < h; y++) std::memcpy(&data[0][static_cast<size_t>(y) * w], img + static_cast<size_t>(y) * stride, w);
    lv[0] = Image{data[0].data(), w, h, w};
    for (int i = 1; i < levels; i++) {
      const int cw = lv[i - 1].cols / 2, ch = lv[i - 1].rows / 2;
      data[i].resize(static_cast<size_t>(cw) * ch);
      PyrDown(lv[i - 1], data[i].data(), cw);
      lv[i] = Image{data[i].data(), cw, ch, cw};
    }
  }
};

SE3 ToSE3(const double *T) {
  SE3 s;
  s.q0 = T[0]; s.q1 = T[1]; s.q2 = T[2]; s.q3 = T[3];
  s.t = Vec3{T[4], T[5], T[6]};
  return s;
}
void FromSE3(const SE3 &s, double *T) {
  T[0] = s.q0; T[1] = s.q1; T[2] = s.q2; T[3] = s.q3; T[4] = s.t.x; T[5] = s.t.y; T[6] = s.t.z;
}
Camera ToCam(const double *cam, int w, int h) {
  Camera c;
  c.width = w; c.height = h; c.fx = cam[0]; c.fy = cam[1]; c.u0 = cam[2]; c.v0 = cam[3];
  return c;
}

}  // namespace

extern "C" {

void sdvl_ref_default_params(sdvl_ref_params *p) {
  Params q;
  p->pyramid_levels = q.pyramid_levels; p->cell_size = q.cell_size; p->max_fast_levels = q.max_fast_levels;
  p->fast_threshold = q.fast_threshold; p->num_features = q.num_features; p->use_orb = q.use_orb;
  p->orb_size = q.orb_size; p->patch_size = q.patch_size; p->max_align_its = q.max_align_its;
  p->search_size = q.search_size; p->align_patch_size = q.align_patch_size; p->max_align_level = q.max_align_level;
  p->min_align_level = q.min_align_level; p->max_img_align_its = q.max_img_align_its;
  p->min_feature_score = q.min_feature_score; p->max_matches = q.max_matches; p->min_matches = q.min_matches;
  p->max_failed = q.max_failed; p->max_optim_pose_its = q.max_optim_pose_its; p->max_ransac_points = q.max_ransac_points;
  p->max_ransac_its = q.max_ransac_its; p->min_keyframe_its = q.min_keyframe_its;
  p->inlier_error_threshold = q.inlier_error_threshold; p->lost_ratio = q.lost_ratio;
}

int sdvl_ref_pyr_down(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dst_stride) {
  PyrDown(Image{src, w, h, stride}, dst, dst_stride);
  return 0;
}

int sdvl_ref_fast(const uint8_t *img, int w, int h, int stride, int thr, int nonmax, int cap, int32_t *out_xys) {
  std::vector<KeyPoint> kps;
  Fast9_16(Image{img, w, h, stride}, &kps, thr, nonmax != 0);
  for (size_t i = 0; i < kps.size() && static_cast<int>(i) < cap; i++) {
    out_xys[3 * i] = static_cast<int>(kps[i].x);
    out_xys[3 * i + 1] = static_cast<int>(kps[i].y);
    out_xys[3 * i + 2] = static_cast<int>(kps[i].response);
  }
  return static_cast<int>(kps.size());
}

int sdvl_ref_fast_cells(const uint8_t *img, int w, int h, int stride, const sdvl_ref_params *p, int cap,
                        int32_t *out_xys, int32_t *cell_offsets, uint8_t *ran_out) {
  std::vector<std::vector<KeyPoint>> cells;
  std::vector<uint8_t> ran;
  int wc, hc;
  FastCells(Image{img, w, h, stride}, ToParams(p), &cells, &ran, &wc, &hc);
  int n = 0;
  for (int c = 0; c < wc * hc; c++) {
    cell_offsets[c] = n;
    ran_out[c] = ran[c];
    for (const auto &k : cells[c]) {
      if (n < cap) {
        out_xys[3 * n] = static_cast<int>(k.x);
        out_xys[3 * n + 1] = static_cast<int>(k.y);
        out_xys[3 * n + 2] = static_cast<int>(k.response);
      }
      n++;
    }
  }
  cell_offsets[wc * hc] = n;
  return n;
}

int sdvl_ref_detect_pyramid(const uint8_t *img, int w, int h, int stride, const sdvl_ref_params *p, int nfeatures,
                            int cap, int32_t *corners) {
  const Params q = ToParams(p);
  Pyramid pyr(img, w, h, stride, q.pyramid_levels);
  std::vector<Corner> cs;
  DetectPyramid(pyr.lv, q, nfeatures, &cs);
  for (size_t i = 0; i < cs.size() && static_cast<int>(i) < cap; i++) {
    corners[3 * i] = cs[i].x; corners[3 * i + 1] = cs[i].y; corners[3 * i + 2] = cs[i].level;
  }
  return static_cast<int>(cs.size());
}

int sdvl_ref_retain_best(uint32_t *packed, int len, int n_points) {
  std::vector<KeyPoint> kps(len);
  for (int i = 0; i < len; i++)
    kps[i] = KeyPoint{static_cast<float>(packed[i] & 0xFFF), static_cast<float>((packed[i] >> 12) & 0xFFF), static_cast<float>(packed[i] >> 24)};
  RetainBest(&kps, n_points);
  for (size_t i = 0; i < kps.size(); i++)
    packed[i] = static_cast<uint32_t>(kps[i].x) | (static_cast<uint32_t>(kps[i].y) << 12) | (static_cast<uint32_t>(kps[i].response) << 24);
  return static_cast<int>(kps.size());
}

double sdvl_ref_shi_tomasi(const uint8_t *img, int w, int h, int stride, int x, int y) {
  return ShiTomasiScore(Image{img, w, h, stride}, x, y);
}

int sdvl_ref_filter_corners(const uint8_t *img, int w, int h, int stride, const sdvl_ref_params *p, int n,
                            const int32_t *corners, int n_locked, const double *locked, int cap, int32_t *indices) {
  const Params q = ToParams(p);
  Pyramid pyr(img, w, h, stride, q.pyramid_levels);
  std::vector<Corner> cs(n);
  for (int i = 0; i < n; i++) cs[i] = Corner{corners[3 * i], corners[3 * i + 1], corners[3 * i + 2]};
  CornerGrid g(w, h, q);
  for (int i = 0; i < n_locked; i++) g.LockCell(locked[2 * i], locked[2 * i + 1]);
  std::vector<int> idx;
  g.FilterCorners(pyr.lv, cs, q, &idx);
  for (size_t i = 0; i < idx.size() && static_cast<int>(i) < cap; i++) indices[i] = idx[i];
  return static_cast<int>(idx.size());
}

void sdvl_ref_orb_describe(const uint8_t *img, int w, int h, int stride, int n, const int32_t *xy, uint8_t *desc,
                           float *angles_deg) {
  OrbDetector orb(31);
  const Image im{img, w, h, stride};
  for (int i = 0; i < n; i++) {
    orb.GetDescriptor(im, xy[2 * i], xy[2 * i + 1], desc + 32 * i);
    if (angles_deg) angles_deg[i] = orb.GetOrientation(im, xy[2 * i], xy[2 * i + 1]);
  }
}

int sdvl_ref_orb_distance(const uint8_t *a, const uint8_t *b) { return OrbDetector::Distance(a, b); }

int sdvl_ref_image_align(const uint8_t *img1, const uint8_t *img2, int w, int h, const sdvl_ref_params *p,
                         const double *cam, int n, const double *px, const double *bearing, const double *depth,
                         const uint8_t *valid, double *T_io, int fast, double *error, double *chi2, int *its) {
  const Params q = ToParams(p);
  Pyramid p1(img1, w, h, w, q.pyramid_levels), p2(img2, w, h, w, q.pyramid_levels);
  std::vector<AlignFeature> feats(n);
  for (int i = 0; i < n; i++) {
    feats[i].px = px[2 * i]; feats[i].py = px[2 * i + 1];
    feats[i].f = Vec3{bearing[3 * i], bearing[3 * i + 1], bearing[3 * i + 2]};
    feats[i].depth = depth[i];
    feats[i].valid = valid[i];
  }
  ImageAlign ia;
  SE3 T = ToSE3(T_io);
  const int r = ia.ComputePose(p1.lv, p2.lv, feats, ToCam(cam, w, h), q, &T, fast != 0);
  FromSE3(T, T_io);
  if (error) *error = ia.error;
  if (chi2) *chi2 = ia.chi2;
  if (its) for (int i = 0; i < 8; i++) its[i] = ia.its_per_level[i];
  return r;
}

int sdvl_ref_search_point(const uint8_t *ref_img, const uint8_t *cur_img, int w, int h, const sdvl_ref_params *p,
                          const double *cam, const double *ref_pose, const double *cur_pose, const double *feat_px,
                          const double *feat_bearing, int feat_level, const uint8_t *feat_desc, double idepth,
                          double idepth_std, int fixed, int n_corners, const int32_t *corners, double *px_io,
                          int *out_level, uint8_t *out_border_patch, int *out_slevel) {
  const Params q = ToParams(p);
  Pyramid pr(ref_img, w, h, w, q.pyramid_levels), pc(cur_img, w, h, w, q.pyramid_levels);
  std::vector<Corner> cs(n_corners);
  for (int i = 0; i < n_corners; i++) cs[i] = Corner{corners[3 * i], corners[3 * i + 1], corners[3 * i + 2]};
  std::vector<std::vector<uint8_t>> descs(n_corners);
  const Camera c = ToCam(cam, w, h);
  Matcher m(q.patch_size, q, c);
  SearchCur cur{&pc.lv, ToSE3(cur_pose), &cs, &descs};
  SearchRef ref;
  ref.ref_pyr = &pr.lv;
  ref.ref_pose = ToSE3(ref_pose);
  ref.px = feat_px[0]; ref.py = feat_px[1];
  ref.f = Vec3{feat_bearing[0], feat_bearing[1], feat_bearing[2]};
  ref.level = feat_level;
  std::memcpy(ref.desc, feat_desc, 32);
  Vec2 px{px_io[0], px_io[1]};
  int level = -1;
  const bool found = m.SearchPoint(&cur, ref, idepth, idepth_std, fixed != 0, &px, &level);
  px_io[0] = px.x; px_io[1] = px.y;
  if (out_level) *out_level = level;
  if (out_border_patch) std::memcpy(out_border_patch, m.border_patch.data(), m.border_patch.size());
  if (out_slevel) {
    Mat2 A;
    m.WarpMatrixAffine(Vec2{ref.px, ref.py}, ref.f, 1.0 / idepth, ToSE3(cur_pose) * ToSE3(ref_pose).Inverse(), feat_level, &A);
    *out_slevel = m.GetSearchLevel(A);
  }
  return found ? 1 : 0;
}

int sdvl_ref_align_patch(const uint8_t *img, int w, int h, int stride, const uint8_t *border_patch,
                         const uint8_t *patch, double *px_io, int max_its) {
  Params q;
  q.max_align_its = max_its;
  Camera c;
  Matcher m(8, q, c);
  Vec2 px{px_io[0], px_io[1]};
  const bool ok = m.AlignPatch(Image{img, w, h, stride}, border_patch, patch, &px);
  px_io[0] = px.x; px_io[1] = px.y;
  return ok ? 1 : 0;
}

void sdvl_ref_se3_exp(const double *u6, double *T7) { FromSE3(SE3Exp(u6), T7); }
void sdvl_ref_se3_log(const double *T7, double *u6) { SE3Log(ToSE3(T7), u6); }
void sdvl_ref_se3_mul(const double *A7, const double *B7, double *C7) { FromSE3(ToSE3(A7) * ToSE3(B7), C7); }
void sdvl_ref_se3_inv(const double *A7, double *B7) { FromSE3(ToSE3(A7).Inverse(), B7); }
void sdvl_ref_ldlt_solve6(const double *A36, const double *b6, double *x6) {
  double A[6][6];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) A[i][j] = A36[6 * i + j];
  LdltSolve6(A, b6, x6);
}
void sdvl_ref_rand_stream(unsigned seed, int n, int *out) {
  GlibcRand r(seed);
  for (int i = 0; i < n; i++) out[i] = r.Next();
}

void *sdvl_ref_tracker_create(const sdvl_ref_params *p, int w, int h, const double *cam, const double *plane4,
                              const double *first_pose7) {
  ScenePlane pl;
  pl.n = Vec3{plane4[0], plane4[1], plane4[2]};
  pl.d = plane4[3];
  return new Tracker(ToParams(p), ToCam(cam, w, h), pl, ToSE3(first_pose7));
}
void sdvl_ref_tracker_destroy(void *t) { delete static_cast<Tracker *>(t); }
// SDVL.max_keyframes for the plane-map stub too (Tracker::PlaneLimitKeyframes); the reference's default is 100 (config.cc:63),
// its cfg files say 1000
void sdvl_ref_tracker_set_max_keyframes(void *t, int max_keyframes) { static_cast<sdvlref::Tracker *>(t)->max_keyframes = max_keyframes; }

// switch the tracker to the reference's mapper in sequential mode (SDVL::Mapping after every frame, main.cc:148-149)
void sdvl_ref_tracker_use_mapper(void *t, int on, int max_search_keyframes, int max_keyframes, double map_scale, double scale_min_dist) {
  Tracker *tr = static_cast<Tracker *>(t);
  tr->use_mapper = on != 0;
  tr->max_search_keyframes = max_search_keyframes;
  tr->max_keyframes = max_keyframes;
  tr->map_scale = map_scale;
  tr->scale_min_dist = scale_min_dist;
}
// candidates, converged (cumulative), initialized (cumulative), linked (cumulative), connections (cumulative), keyframes
void sdvl_ref_tracker_map_stats(void *t, int *out6) {
  const Tracker *tr = static_cast<Tracker *>(t);
  out6[0] = static_cast<int>(tr->candidates.size()); out6[1] = tr->map_stats.converged; out6[2] = tr->map_stats.initialized;
  out6[3] = tr->map_stats.linked; out6[4] = tr->map_stats.connected; out6[5] = static_cast<int>(tr->keyframes.size());
}
// positions of the points the mapper itself created (not the plane-bootstrapped ones) that are still alive:
// out[i] = x y z converged(0/1).  Returns the number written (<= cap).
int sdvl_ref_tracker_mapper_points(void *t, int cap, double *out) {
  Tracker *tr = static_cast<Tracker *>(t);
  std::set<const RPoint *> seen;
  int n = 0;
  for (auto &kf : tr->keepalive)
    for (auto &ft : kf->features) {
      if (!ft || !ft->point || ft->point->del) continue;
      const RPoint *p = ft->point.get();
      if (p->init_feature->frame->id <= tr->initial_kf_id) continue;  // bootstrap points come from the plane itself
      if (!seen.insert(p).second) continue;
      if (n >= cap) return n;
      const Vec3 pos = p->GetPosition();
      out[4 * n] = pos.x; out[4 * n + 1] = pos.y; out[4 * n + 2] = pos.z; out[4 * n + 3] = p->fixed ? 1.0 : 0.0;
      n++;
    }
  return n;
}
// GetDepthFromTriangulation (extra/utils.cc:193-205) and the depth-filter helpers, for direct checks
int sdvl_ref_triangulate(const double *pose7, const double *v_ref3, const double *v_cur3, double *depth) {
  return Tracker::DepthFromTriangulation(ToSE3(pose7), Vec3{v_ref3[0], v_ref3[1], v_ref3[2]}, Vec3{v_cur3[0], v_cur3[1], v_cur3[2]}, depth) ? 1 : 0;
}
double sdvl_ref_pdf_normal(double mean, double sd, double x) { return Tracker::PDFNormal(mean, sd, x); }
double sdvl_ref_compute_tau(const double *pose7, const double *v3, double depth, double px_error_angle) {
  return Tracker::ComputeTau(ToSE3(pose7), Vec3{v3[0], v3[1], v3[2]}, depth, px_error_angle);
}
int sdvl_ref_depth_filter(void *t, const double *cur_pose7, const double *ref_pose7, const double *bearing3, int found, const double *px2,
                          double depth_mean, double *st) {
  Tracker *tr = static_cast<Tracker *>(t);
  // the objects the loop body touches: the candidate, its first observation on the reference keyframe, the current frame
  RFrame ref_frame, cur_frame;
  ref_frame.pose = ToSE3(ref_pose7);
  cur_frame.pose = ToSE3(cur_pose7);
  auto feature = std::make_shared<RFeature>();
  feature->frame = &ref_frame;
  feature->v = Vec3{bearing3[0], bearing3[1], bearing3[2]};
  RPoint point;
  point.init_feature = feature;
  point.rho = st[0]; point.sigma2 = st[1]; point.a = st[2]; point.b = st[3]; point.z_range = st[4];
  point.cos_alpha = st[5]; point.last_distance = st[6];
  point.p3d = Vec3{st[7], st[8], st[9]};
  point.fixed = st[10] != 0.0;
  point.n_failed = static_cast<int>(st[11]);
  int outcome = 1;
  const double px_error_angle = std::atan(1.0 / (2.0 * tr->cam.fx)) * 2.0;
  if (!found) {  // map.cc:454-458
    outcome = tr->PointUnpromote(&point) ? 0x100 : 0;
  } else {       // map.cc:459-497, statement for statement as Tracker::UpdateCandidates has it
    const SE3 pose = cur_frame.pose * feature->frame->pose.Inverse();
    const Vec3 v3d = tr->cam.Unproject(Vec2{px2[0], px2[1]});
    double depth = 0.0;
    bool go = Tracker::DepthFromTriangulation(pose, feature->v, v3d, &depth);
    if (go) {
      const Vec3 p3d = feature->frame->pose.Inverse() * (depth * feature->v);
      const double cos_alpha = Tracker::Parallax(feature->frame->WorldPosition(), cur_frame.WorldPosition(), p3d);
      if (cos_alpha >= 0.999999) go = false;
    }
    if (go && (depth < tr->map_scale * tr->scale_min_dist || depth < depth_mean * tr->scale_min_dist)) go = false;
    if (go) {
      tr->PointUpdate(&point, cur_frame, depth, px_error_angle);
      outcome = Tracker::PointHasConverged(&point) ? 3 : 2;
    }
  }
  st[0] = point.rho; st[1] = point.sigma2; st[2] = point.a; st[3] = point.b; st[4] = point.z_range;
  st[5] = point.cos_alpha; st[6] = point.last_distance;
  st[7] = point.p3d.x; st[8] = point.p3d.y; st[9] = point.p3d.z;
  st[10] = point.fixed ? 1.0 : 0.0;
  st[11] = point.n_failed;
  return outcome;
}
int sdvl_ref_tracker_handle_frame(void *t, const uint8_t *img, int stride, sdvl_ref_frame_stats *out) {
  Tracker *tr = static_cast<Tracker *>(t);
  const FrameStats s = tr->HandleFrame(img, stride);
  if (tr->use_mapper) tr->UpdateMap();  // sequential mode: the mapper runs inline, outside the tracking time window
  out->state = s.state; out->quality = s.quality; out->matches = s.matches; out->attempts = s.attempts;
  out->inliers = s.inliers; out->outliers = s.outliers; out->n_corners = s.n_corners; out->align_meas = s.align_meas;
  out->keyframe = s.keyframe; out->relocalized = s.relocalized;
  for (int i = 0; i < 7; i++) out->pose[i] = s.pose[i];
  return 0;
}

// FeatureAlign::SelectInliers + OptimizePose (feature_align.cc:73-82,152-243) on a caller-given match list.
// obs[n][6] = {ax, ay, px, py, pz, level}: feature bearing (ax, ay, 1), fixed 3D point, pyramid level.
// rand_seed / rand_skip position the glibc stream; n_draws returns how many rand() calls the RANSAC loop made.
// in_idx / out_idx receive the final inlier / outlier lists as indices into obs (reference order).
int sdvl_ref_pose_from_matches(const sdvl_ref_params *p, int w, int h, const double *cam, int n, const double *obs,
                               unsigned rand_seed, int rand_skip, double *pose7_io, int *n_draws, int *n_in, int *in_idx,
                               int *n_out, int *out_idx) {
  ScenePlane pl;
  Tracker t(ToParams(p), ToCam(cam, w, h), pl, SE3());
  t.rng.Seed(rand_seed);
  for (int i = 0; i < rand_skip; i++) t.rng.Next();
  auto frame = std::make_shared<RFrame>();
  frame->pose = ToSE3(pose7_io);
  std::vector<std::shared_ptr<RFeature>> found;
  for (int i = 0; i < n; i++) {
    auto pt = std::make_shared<RPoint>();
    pt->fixed = true;
    pt->p3d = Vec3{obs[6 * i + 2], obs[6 * i + 3], obs[6 * i + 4]};
    auto ft = std::make_shared<RFeature>();
    ft->frame = frame.get();
    ft->point = pt;
    ft->v = Vec3{obs[6 * i], obs[6 * i + 1], 1.0};
    ft->level = static_cast<int>(obs[6 * i + 5]);
    found.push_back(ft);
  }
  GlibcRand before = t.rng;
  t.SelectInliers(frame, found, &t.inliers, &t.outliers);
  // count the draws: advance the saved copy until it matches the live stream
  int draws = 0;
  while (!(before.fi == t.rng.fi && before.ri == t.rng.ri && before.ring == t.rng.ring) && draws <= t.prm.max_ransac_its) {
    before.Next();
    draws++;
  }
  *n_draws = draws;
  t.OptimizePose(frame, &t.inliers, &t.outliers);
  if (t.RescueOutliers(frame, &t.inliers, &t.outliers)) t.OptimizePose(frame, &t.inliers, &t.outliers);
  auto index_of = [&](const std::shared_ptr<RFeature> &f) {
    for (int i = 0; i < n; i++) if (found[i] == f) return i;
    return -1;
  };
  *n_in = static_cast<int>(t.inliers.size());
  *n_out = static_cast<int>(t.outliers.size());
  for (int i = 0; i < *n_in; i++) in_idx[i] = index_of(t.inliers[i]);
  for (int i = 0; i < *n_out; i++) out_idx[i] = index_of(t.outliers[i]);
  FromSE3(frame->pose, pose7_io);
  return 0;
}

// Camera::UndistortImage = cv::undistort (camera.cc:100-105): cam4 = fx fy u0 v0, dist5 = d0..d4 (k1 k2 p1 p2 k3)
int sdvl_ref_undistort(const uint8_t *img, int w, int h, int stride, const double *cam4, const double *dist5, uint8_t *out) {
  if (dist5[0] == 0.0) {  // Camera::SetDistortions only tests d0 (camera.cc:46): no distortion -> in.clone()
    for (int y = 0; y < h; y++) std::memcpy(out + static_cast<size_t>(y) * w, img + static_cast<size_t>(y) * stride, w);
    return 0;
  }
  Undistort(img, w, h, stride, cam4, dist5, out);
  return 1;
}
// the 32x32x4 bilinear weight table of cv::remap (shorts, scale 32768)
void sdvl_ref_remap_weights(int16_t *out4096) { std::memcpy(out4096, BilinearTabI(), sizeof(int16_t) * 4096); }

}  // extern "C"
