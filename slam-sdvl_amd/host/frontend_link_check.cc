// frontend_link_check.cc — the hot path's classes WITHOUT the rest of the host layer.  Includes frontend.h only and is linked as
//     frontend.cc + minimal_deps.cc + config.cc + this file  (+ libsdvl_hip.so, libsdvl_synth.so)
// — no standalone.cc (Camera / Point / Map / SDVL / SDVLBatch), no mapper.cc, no capi.cc: what INTEGRATION.md route A puts next to
// the reference's own sdvl.cc / map.cc / point.cc / camera.cc / config.cc.  That the link succeeds is the proof that the front end
// has no hidden dependency; on a GPU the program also tracks one frame through the reference's per-object calls with the minimal
// Camera / Point (Frame ctor -> FilterCorners -> ImageAlign::ComputePose -> Matcher::SearchPoint -> FeatureAlign::Reproject +
// OptimizePose, sdvl.cc:59,189,193,200) and checks the pose against the rendered motion.  Without a GPU it says so and exits 0.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <memory>
#include <vector>

#include "frontend.h"
#undef SDVL_HD
#include "../csrc/sdvl_synth.h"

extern "C" int sdvl_synth_render_host(const sdvl_synth_view *view, int width, int height, uint8_t *out, int stride);

using namespace sdvl;
using std::shared_ptr;
using std::vector;

namespace {
int g_failed = 0;
void Check(bool ok, const char *what, double value = 0.0) {
  std::printf("%s  %s  (%.6g)\n", ok ? "ok  " : "FAIL", what, value);
  if (!ok) g_failed++;
}
const double kCam[4] = {517.3, 516.5, 318.6, 255.3};
const int W = 640, H = 480;

SE3 PoseOf(int k) {
  Vector6d xi;
  const double tw[6] = {0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005};
  for (int q = 0; q < 6; q++) xi.v[q] = tw[q] * k;
  return SE3::Exp(xi);
}
vector<uint8_t> Render(int k) {
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
double PoseDiff(const SE3 &a, const SE3 &b) {
  double pa[7], pb[7], m = 0.0;
  a.ToArray(pa);
  b.ToArray(pb);
  for (int q = 0; q < 7; q++) m = std::fmax(m, std::fabs(pa[q] - pb[q]));
  return m;
}
}  // namespace

int main() {
  Config &config = Config::GetInstance();
  const struct { const char *k; double v; } over[] = {{"SDVL.cell_size", 32}, {"SDVL.min_avg_shift", 5}, {"SDVL.max_matches", 200}, {"SDVL.max_keyframes", 1000},
                                                      {"SDVL.use_orb", 1}, {"SDVL.fast_threshold", 10}, {"SDVL.lost_ratio", 0.7}, {"SDVL.num_features", 1000}};
  for (const auto &o : over)
    if (!config.SetParameter(o.k, o.v)) return 2;
  std::unique_ptr<Device> dev;
  try {
    dev.reset(new Device(0));
  } catch (const std::exception &e) {
    std::printf("frontend_link_check: linked against minimal_deps.cc; no GPU here (%s): the calls were not run\n", e.what());
    return 0;
  }
  try {
    Camera camera(W, H, kCam[0], kCam[1], kCam[2], kCam[3]);
    Map map;
    ORBDetector orb;
    const vector<uint8_t> px0 = Render(0), px3 = Render(3);
    const Image img0(H, W, CV_8UC1, px0.data()), img3(H, W, CV_8UC1, px3.data());
    shared_ptr<Frame> f0 = std::make_shared<Frame>(&camera, &orb, img0, true);   // sdvl.cc:59
    const vector<Vector3i> corners = f0->GetCorners();
    Check(corners.size() >= 900 && corners.size() <= 1100, "Frame(camera, detector, img, corners = true): ~NumFeatures corners", corners.size());
    f0->SetPose(PoseOf(0));
    f0->FilterCorners();                                                          // map.cc:283
    const vector<int> filtered = f0->GetFilteredCorners();
    const vector<vector<uchar>> &descs = f0->GetDescriptors();
    Check(filtered.size() >= 150 && filtered.size() <= 300, "Frame::FilterCorners(): one corner per 32-px cell", filtered.size());
    f0->SetKeyframe();
    const SE3 world = f0->GetWorldPose();
    for (int index : filtered) {                                                  // homography_init.cc:137-168: the initial map, on the plane z = 2
      const Vector3i c = corners[index];
      const Vector2d px(c(0) * (1 << c(2)), c(1) * (1 << c(2)));
      const Vector3d v = camera.Unproject(px);
      const Vector3d ray = world * v, org = world.GetTranslation();
      const double s = (2.0 - org(2)) / (ray(2) - org(2));
      shared_ptr<Point> pt = std::make_shared<Point>();
      shared_ptr<Feature> ft = std::make_shared<Feature>(f0, pt, px, v, c(2));
      ft->SetDescriptor(descs[index]);
      pt->InitFixed(ft, s, (0.05 / s) * (0.05 / s), world * Vector3d(s * v(0), s * v(1), s * v(2)));
      f0->AddFeature(ft);
      pt->AddFeature(ft);
    }
    Check(f0->GetNumPoints() == static_cast<int>(filtered.size()), "features with fixed points (minimal Point) on the keyframe", f0->GetNumPoints());
    shared_ptr<Frame> f3 = std::make_shared<Frame>(&camera, &orb, img3, true);
    f3->SetPose(f0->GetPose());
    ImageAlign ia;
    const int n_meas = ia.ComputePose(f0, f3);                                     // sdvl.cc:189
    Check(n_meas >= 100 && PoseDiff(f3->GetPose(), PoseOf(3)) < 2e-3, "ImageAlign::ComputePose recovers the rendered motion (pose error)",
          PoseDiff(f3->GetPose(), PoseOf(3)));
    {
      Matcher matcher(Config::PatchSize());
      int tried = 0, found = 0;
      for (const auto &ft : f0->GetFeatures()) {
        if (tried >= 40) break;
        shared_ptr<Point> pt = ft->GetPoint();
        const Vector3d pc = PoseOf(3) * pt->GetPosition();
        Vector2d px = camera.Project(pc);
        if (px(0) < 40 || px(1) < 40 || px(0) > W - 40 || px(1) > H - 40) continue;
        tried++;
        int level = -1;
        found += matcher.SearchPoint(f3, ft, pt->GetInverseDepth(), pt->GetStd(), pt->IsFixed(), &px, &level) ? 1 : 0;   // map.cc:326 style
      }
      Check(tried >= 30 && found >= tried * 6 / 10, "Matcher::SearchPoint finds most points", found);
    }
    {
      FeatureAlign fa(&map, &camera, Config::MaxMatches());
      fa.Reproject(f3, f0, f0);                                                    // sdvl.cc:193
      Check(fa.GetMatches() >= 100, "FeatureAlign::Reproject: >= 100 matches", fa.GetMatches());
      const bool ok = fa.OptimizePose(f3);                                         // sdvl.cc:200
      Check(ok && PoseDiff(f3->GetPose(), PoseOf(3)) < 5e-4, "FeatureAlign::OptimizePose refines the pose (pose error)", PoseDiff(f3->GetPose(), PoseOf(3)));
    }
  } catch (const std::exception &e) {
    std::cerr << "frontend_link_check: " << e.what() << std::endl;
    return 1;
  }
  std::printf("%d check(s) failed\n", g_failed);
  return g_failed == 0 ? 0 : 3;
}
