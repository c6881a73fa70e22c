// api_surface_check.cc — the reference's public API, call by call, using ONLY signatures that exist in the reference's
// headers (sdvl.h:49-69, frame.h:45-139, feature.h:42-94, point.h:50-110, camera.h:36-101, config.h:59-62,
// extra/fast_detector.h:36-46, extra/orb_detector.h:36-42, image_align.h:38-43, matcher.h:41-46, feature_align.h:46-57,
// map.h:44-84): an unchanged sdvl.cc / map.cc / main.cc / ui would compile against sdvl_host.h the same way.  The only
// stand-ins are the types of types.h / se3.h (Image for cv::Mat, Vec for Eigen vectors: neither library is in this image).
// What each call returns is checked against the batched path (same corners / descriptors) or against the motion the frames
// were rendered with.  Exit code 0 = every check passed; each check prints one line.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <vector>

#include "sdvl_host.h"
#undef SDVL_HD
#include "../csrc/sdvl_synth.h"

extern "C" int sdvl_synth_render_host(const sdvl_synth_view *view, int width, int height, uint8_t *out, int stride);

using namespace sdvl;
using std::shared_ptr;
using std::vector;

static int g_failed = 0;
static void Check(bool ok, const char *what, double value = 0.0) {
  std::printf("%s  %s  (%.6g)\n", ok ? "ok  " : "FAIL", what, value);
  if (!ok) g_failed++;
}

static const double kCam[4] = {517.3, 516.5, 318.6, 255.3};
static const int W = 640, H = 480;

static SE3 PoseOf(int k) {
  Vector6d xi;
  const double tw[6] = {0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005};
  for (int q = 0; q < 6; q++) xi.v[q] = tw[q] * k;
  return SE3::Exp(xi);
}

static vector<uint8_t> Render(int k) {
  const SE3 T = PoseOf(k);
  sdvl_synth_view v = {};
  v.fx = kCam[0]; v.fy = kCam[1]; v.u0 = kCam[2]; v.v0 = kCam[3];
  const M3 R = T.GetRotation();
  for (int q = 0; q < 9; q++) v.R[q] = R.m[q];
  const Vector3d t = T.GetTranslation();
  for (int q = 0; q < 3; q++) v.t[q] = t(q);
  v.plane[0] = 0; v.plane[1] = 0; v.plane[2] = 1; v.plane[3] = 2.0;
  v.seed = 20260001;
  v.frame_id = static_cast<uint32_t>(k);
  vector<uint8_t> px(static_cast<size_t>(W) * H);
  sdvl_synth_render_host(&v, W, H, px.data(), W);
  return px;
}

static double PoseDiff(const SE3 &a, const SE3 &b) {
  double pa[7], pb[7], m = 0.0;
  a.ToArray(pa);
  b.ToArray(pb);
  for (int q = 0; q < 7; q++) m = std::fmax(m, std::fabs(pa[q] - pb[q]));
  return m;
}

int main() {
  // main.cc:60-66: the configuration comes from a file (config_tum_f1.cfg's values; map_scale = the scene's depth, which
  // is what the bootstrap normalises the first keyframe's points to)
  const char *cfg_path = "/tmp/sdvl_api_surface_check.cfg";
  {
    std::ofstream f(cfg_path);
    f << "Camera.width: 640\nCamera.height: 480\nCamera.fx: 517.3\nCamera.fy: 516.5\nCamera.u0: 318.6\nCamera.v0: 255.3\n"
         "Camera.d1: 0\nCamera.d2: 0\nCamera.d3: 0\nCamera.d4: 0\nCamera.d5: 0\n"
         "SDVL.cell_size: 32\nSDVL.min_avg_shift: 5\nSDVL.max_matches: 200\nSDVL.max_keyframes: 1000\nSDVL.use_orb: 1\n"
         "SDVL.fast_threshold: 10\nSDVL.lost_ratio: 0.7\nSDVL.num_features: 1000\nSDVL.map_scale: 2.0\n";
  }
  Config &config = Config::GetInstance();
  if (!config.ReadParameters(cfg_path)) return 2;
  try {
    Camera *camera = new Camera();                       // main.cc:72, camera.cc:28-38
    Check(camera->GetWidth() == W && camera->GetFx() == kCam[0], "Camera(): parameters of the configuration file", camera->GetFx());
    const vector<uint8_t> px0 = Render(0), px3 = Render(3);
    const Image img0(H, W, CV_8UC1, px0.data()), img3(H, W, CV_8UC1, px3.data());

    // ---- sdvl.h: SDVL(Camera*) owns its map; the first HandleFrame makes the first keyframe
    SDVL *handler = new SDVL(camera);                    // main.cc:88 (creates the thread's Device: there is no CPU path)
    Image imgu;
    camera->UndistortImage(img0, &imgu);                 // main.cc:133 (d1 == 0: a copy, camera.cc:46)
    Check(handler->HandleFrame(imgu), "SDVL::HandleFrame(frame 0)");
    handler->Mapping();                                  // main.cc:148-149 (sequential mode)
    Check(handler->HasMap(), "SDVL::HasMap() after the first keyframe");
    vector<std::pair<SE3, bool>> trail;
    handler->GetCameraTrail(&trail);
    Check(trail.size() == 1 && PoseDiff(trail[0].first, SE3()) < 1e-12, "GetCameraTrail(): one keyframe at the origin", trail.size());
    vector<Vector3d> cloud;
    handler->GetPoints(&cloud);
    bool on_plane = cloud.size() >= 200;
    for (const Vector3d &p : cloud) on_plane = on_plane && std::fabs(p(2) - 2.0) < 1e-9;
    Check(on_plane, "GetPoints(): two entries per map point, all on the bootstrap plane z = map_scale", cloud.size());

    {  // (scope: these frames and this map go before the tracker, whose Device they live on)
    // ---- frame.h: Frame(camera, detector, img, corners) as sdvl.cc:59 calls it, and its accessors
    ORBDetector orb;
    shared_ptr<Frame> f0 = std::make_shared<Frame>(camera, &orb, img0, true);
    const vector<Vector3i> corners = f0->GetCorners();
    Check(corners.size() >= 900 && corners.size() <= 1100, "Frame(..., corners=true): GetCorners() holds ~NumFeatures corners", corners.size());
    vector<Image> &pyr = f0->GetPyramid();
    bool pyr_ok = pyr.size() == 5 && pyr[0].cols == W && pyr[4].cols == W / 16 && pyr[0].data != nullptr;
    for (int i = 0; pyr_ok && i < W * H; i += 997) pyr_ok = pyr[0].data[i] == px0[i];
    Check(pyr_ok, "GetPyramid(): 5 levels on the host, level 0 is the image");
    f0->GetPose() = PoseOf(0);                            // frame.h:52: the non-const accessor is writable
    Check(PoseDiff(f0->GetWorldPose(), PoseOf(0).Inverse()) < 1e-15, "GetPose() (non-const) writes through; GetWorldPose() follows");
    Check(f0->GetWidth() == W && f0->GetHeight() == H && f0->GetCamera() == camera && !f0->IsKeyframe(), "GetWidth / GetHeight / GetCamera / IsKeyframe");

    // ---- extra/fast_detector.h
    {
      FastDetector det(W, H, false);
      vector<Vector3i> again;
      det.DetectPyramid(pyr, &again, Config::NumFeatures());
      bool same = again.size() == corners.size();
      for (size_t i = 0; same && i < again.size(); i++) same = again[i](0) == corners[i](0) && again[i](1) == corners[i](1) && again[i](2) == corners[i](2);
      Check(same, "FastDetector::DetectPyramid(pyramid) == the corners the Frame constructor produced, same order", again.size());
      FastDetector grid(W, H);
      grid.LockCell(Vector2d(100.0, 100.0));
      vector<int> idx, idx_free;
      grid.FilterCorners(pyr, corners, &idx);
      grid.UnlockCell(Vector2d(100.0, 100.0));
      FastDetector grid2(W, H);
      grid2.FilterCorners(pyr, corners, &idx_free);
      bool locked_empty = true;
      for (int i : idx) locked_empty = locked_empty && !(corners[i](0) * (1 << corners[i](2)) / 32 == 3 && corners[i](1) * (1 << corners[i](2)) / 32 == 3);
      Check(!idx.empty() && idx.size() <= idx_free.size() && locked_empty, "FastDetector::FilterCorners / LockCell: no corner from the locked cell", idx.size());
    }
    // ---- Frame::FilterCorners (map.cc:283) + descriptors, extra/orb_detector.h
    f0->FilterCorners();
    const vector<int> filtered = f0->GetFilteredCorners();
    Check(filtered.size() >= 150 && filtered.size() <= 300, "Frame::FilterCorners(): one corner per free 32-px cell above MinFeatureScore", filtered.size());
    const vector<vector<uchar>> &descs = f0->GetDescriptors();
    Check(descs.size() == corners.size() && descs[filtered[0]].size() == 32, "GetDescriptors(): 32 bytes per corner", descs.size());
    {
      int tested = 0, equal = 0, self0 = 0;
      for (size_t q = 0; q < filtered.size() && tested < 40; q += 5) {
        const Vector3i cnr = corners[filtered[q]];
        vector<uchar> d;
        if (!orb.GetDescriptor(pyr[cnr(2)], Vector2i(cnr(0), cnr(1)), &d)) continue;
        tested++;
        equal += d == descs[filtered[q]];
        self0 += orb.Distance(d, descs[filtered[q]]) == 0;
      }
      Check(tested >= 20 && equal == tested && self0 == tested, "ORBDetector::GetDescriptor == the frame's descriptor, Distance(d, d) == 0", tested);
      Check(orb.Distance(descs[filtered[0]], descs[filtered[1]]) > 0, "Distance of two different corners > 0", orb.Distance(descs[filtered[0]], descs[filtered[1]]));
    }
    // ---- feature.h / point.h: a keyframe with fixed points on the scene plane, made the way homography_init.cc:137-168
    //      makes the initial map (Point(), Feature(frame, point, px, vector, level), InitCandidate, AddFeature, SetFixed)
    Map map;                                              // map.h:46
    f0->SetKeyframe();
    map.AddKeyframe(f0, false);                           // sdvl.cc:144
    const SE3 world = f0->GetWorldPose();
    for (int index : filtered) {
      const Vector3i c = corners[index];
      const Vector2d px(c(0) * (1 << c(2)), c(1) * (1 << c(2)));
      const Vector3d v = camera->Unproject(px);
      const Vector3d ray = world * v;                     // the pose is ~identity: direction ~ v, origin ~ 0
      const Vector3d org = world.GetTranslation();
      const double s = (2.0 - org(2)) / (ray(2) - org(2));
      shared_ptr<Point> pt = std::make_shared<Point>();
      shared_ptr<Feature> ft = std::make_shared<Feature>(f0, pt, px, v, c(2));
      ft->SetDescriptor(descs[index]);
      pt->InitCandidate(ft, s);
      pt->SetFixed();
      pt->SetPosition(world * Vector3d(s * v(0), s * v(1), s * v(2)));
      f0->AddFeature(ft);
      pt->AddFeature(ft);
    }
    Check(f0->GetNumFeatures() == static_cast<int>(filtered.size()) && f0->GetNumPoints() == f0->GetNumFeatures(), "AddFeature / GetNumFeatures / GetNumPoints", f0->GetNumFeatures());
    {
      const shared_ptr<Feature> &ft = f0->GetFeatures()[0];
      const vector<uchar> &d = ft->GetDescriptor();       // feature.h:78
      Check(ft->HasDescriptor() && d.size() == 32 && d == descs[filtered[0]] && ft->GetFrame() == f0 && ft->GetPoint()->GetInitFeature() == ft,
            "Feature::GetDescriptor() is a std::vector<uchar> of 32 bytes; GetFrame / GetPoint / GetInitFeature link up", d.size());
    }

    // ---- image_align.h: ImageAlign::ComputePose(kf, frame) (sdvl.cc:189)
    shared_ptr<Frame> f3 = std::make_shared<Frame>(camera, &orb, img3, true);
    f3->SetPose(f0->GetPose());  // no motion model: start from the keyframe's pose
    ImageAlign ia;
    const int n_meas = ia.ComputePose(f0, f3);
    const double d_ia = PoseDiff(f3->GetPose(), PoseOf(3));
    Check(n_meas >= 100 && d_ia < 2e-3, "ImageAlign::ComputePose(kf, frame) recovers the rendered motion (pose error)", d_ia);
    Check(ia.GetError() < 1e-3, "ImageAlign::GetError() small after convergence", ia.GetError());

    // ---- matcher.h: Matcher::SearchPoint for single features (map.cc:326 style)
    {
      Matcher matcher(Config::PatchSize());
      int tried = 0, found = 0, close = 0;
      const SE3 Ttrue = PoseOf(3);
      for (const auto &ft : f0->GetFeatures()) {
        if (tried >= 60) break;
        shared_ptr<Point> pt = ft->GetPoint();
        if (!pt) continue;
        const Vector3d pc = Ttrue * pt->GetPosition();
        const Vector2d truth(kCam[2] + kCam[0] * pc(0) / pc(2), kCam[3] + kCam[1] * pc(1) / pc(2));
        if (truth(0) < 40 || truth(1) < 40 || truth(0) > W - 40 || truth(1) > H - 40) continue;
        tried++;
        Vector2d px(truth(0) + 0.8, truth(1) - 0.6);  // a slightly wrong prediction, as after image alignment
        int level = -1;
        if (matcher.SearchPoint(f3, ft, pt->GetInverseDepth(), pt->GetStd(), pt->IsFixed(), &px, &level)) {
          found++;
          const double dx = px(0) - truth(0), dy = px(1) - truth(1);
          close += std::sqrt(dx * dx + dy * dy) < 1.0;
        }
      }
      Check(tried >= 40 && found >= tried * 6 / 10 && close >= found * 9 / 10, "Matcher::SearchPoint: most points found, within 1 px of their true projection", found);
    }
    // ---- feature_align.h: FeatureAlign(map, camera, max_matches), Reproject + OptimizePose (sdvl.cc:193,200)
    {
      FeatureAlign fa(&map, camera, Config::MaxMatches());
      fa.Reproject(f3, f0, f0);
      Check(fa.GetMatches() >= 100 && fa.GetAttempts() >= fa.GetMatches(), "FeatureAlign::Reproject: >= 100 matches", fa.GetMatches());
      const bool ok = fa.OptimizePose(f3);
      const double d_fa = PoseDiff(f3->GetPose(), PoseOf(3));
      Check(ok && d_fa < 5e-4, "FeatureAlign::OptimizePose refines the pose (pose error)", d_fa);
      Check(f3->GetNumFeatures() >= 100, "the matched features were added to the frame", f3->GetNumFeatures());
      map.EmptyTrash();                                   // sdvl.cc:127
    }
    }
    // ---- sdvl.h again: the tracker follows the sequence on its own; the UI's queries
    bool tracked = true;
    for (int k = 1; k <= 4; k++) {
      const vector<uint8_t> px = Render(k);
      tracked = handler->HandleFrame(Image(H, W, CV_8UC1, px.data())) && tracked;
      handler->Mapping();
    }
    Check(tracked && handler->GetTrackingQuality() == SDVL::TRACKING_GOOD, "SDVL::HandleFrame(frames 1..4): tracking quality GOOD");
    const double d_track = PoseDiff(handler->GetPose(), PoseOf(4).Inverse());  // GetPose() = the camera's pose in the world, sdvl.cc:351-356
    Check(d_track < 1e-3, "SDVL::GetPose() follows the rendered trajectory (pose error)", d_track);
    vector<Vector3i> last;
    handler->GetLastFeatures(&last);
    bool inside = last.size() >= 100;
    for (const Vector3i &p : last) inside = inside && p(0) >= 0 && p(0) < W && p(1) >= 0 && p(1) < H && p(2) >= 0 && p(2) <= 4;
    Check(inside, "GetLastFeatures(): (x, y, status) of the last frame's features", last.size());
    delete handler;
    delete camera;
  } catch (const std::exception &e) {
    std::cerr << "api_surface_check: " << e.what() << std::endl;
    return 1;
  }
  std::printf("%d check(s) failed\n", g_failed);
  return g_failed == 0 ? 0 : 3;
}
