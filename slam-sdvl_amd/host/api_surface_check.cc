// api_surface_check.cc — the reference's per-object calls, one at a time, as its own callers make them (SURVEY §8b:
// sdvl.cc:59,189,193,200; map.cc:283,326): Frame(camera, detector, img, corners) / GetCorners / FilterCorners /
// GetDescriptors / GetPyramid, FastDetector::DetectPyramid, ORBDetector::GetDescriptor / Distance, ImageAlign::ComputePose,
// Matcher::SearchPoint, FeatureAlign::Reproject / OptimizePose — the batched forms have their own parity tests; this checks
// that the single-call surface (what a maintainer's unchanged sdvl.cc / map.cc use) gives the same answers and recovers the
// motion the frames were rendered with.  Exit code 0 = every check passed; each check prints one line.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <memory>
#include <vector>

#include "sdvl_host.h"
#undef SDVL_HD
#include "../csrc/sdvl_synth.h"

extern "C" int sdvl_synth_render_host(const sdvl_synth_view *view, int width, int height, uint8_t *out, int stride);

using namespace sdvl;

static int g_failed = 0;
static void Check(bool ok, const char *what, double value = 0.0) {
  std::printf("%s  %s  (%.6g)\n", ok ? "ok  " : "FAIL", what, value);
  if (!ok) g_failed++;
}

static SE3 PoseOf(int k) {
  Vector6d xi;
  const double tw[6] = {0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005};
  for (int q = 0; q < 6; q++) xi.v[q] = tw[q] * k;
  return SE3::Exp(xi);
}

static std::vector<uint8_t> Render(int k, int W, int H, const double *cam4) {
  const SE3 T = PoseOf(k);
  sdvl_synth_view v;
  v.fx = cam4[0]; v.fy = cam4[1]; v.u0 = cam4[2]; v.v0 = cam4[3];
  const M3 R = T.GetRotation();
  for (int q = 0; q < 9; q++) v.R[q] = R.m[q];
  const Vector3d t = T.GetTranslation();
  for (int q = 0; q < 3; q++) v.t[q] = t(q);
  v.plane[0] = 0; v.plane[1] = 0; v.plane[2] = 1; v.plane[3] = 2.0;
  v.seed = 20260001;
  v.frame_id = static_cast<uint32_t>(k);
  std::vector<uint8_t> px(static_cast<size_t>(W) * H);
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
  const int W = 640, H = 480;
  const double cam4[4] = {517.3, 516.5, 318.6, 255.3};
  Config &c = Config::GetInstance();
  c.SetParameter("SDVL.cell_size", 32); c.SetParameter("SDVL.max_matches", 200); c.SetParameter("SDVL.use_orb", 1);
  c.SetParameter("SDVL.fast_threshold", 10); c.SetParameter("SDVL.num_features", 1000);
  try {
    Device dev(0);
    Device::SetCurrent(&dev);
    Camera camera(W, H, cam4[0], cam4[1], cam4[2], cam4[3]);
    ORBDetector orb;
    const std::vector<uint8_t> px0 = Render(0, W, H, cam4), px3 = Render(3, W, H, cam4);
    Image img0, img3;
    img0.data = px0.data(); img0.cols = W; img0.rows = H; img0.step = W;
    img3.data = px3.data(); img3.cols = W; img3.rows = H; img3.step = W;

    // ---- Frame::Frame(camera, detector, img, corners) as sdvl.cc:59 calls it, and its accessors
    std::shared_ptr<Frame> f0 = std::make_shared<Frame>(&camera, &orb, img0, true);
    const std::vector<Vector3i> corners = f0->GetCorners();
    Check(corners.size() >= 900 && corners.size() <= 1100, "Frame(..., corners=true): GetCorners() holds ~NumFeatures corners", corners.size());
    Check(f0->GetNumCorners() == static_cast<int>(corners.size()), "GetNumCorners() agrees with the mirrored list", f0->GetNumCorners());
    std::vector<Image> &pyr = f0->GetPyramid();
    bool pyr_ok = pyr.size() == 5 && pyr[0].cols == W && pyr[4].cols == W / 16 && pyr[0].data != nullptr;
    for (int i = 0; pyr_ok && i < W * H; i += 997) pyr_ok = pyr[0].data[i] == px0[i];
    Check(pyr_ok, "GetPyramid(): 5 levels on the host, level 0 is the image");

    // ---- FastDetector::DetectPyramid on the frame's pyramid gives the frame's own corner list (fast_detector.cc:154-175)
    {
      FastDetector det(W, H, false);
      std::vector<Vector3i> again;
      det.DetectPyramid(pyr, &again, Config::NumFeatures());
      bool same = again.size() == corners.size();
      for (size_t i = 0; same && i < again.size(); i++) same = again[i](0) == corners[i](0) && again[i](1) == corners[i](1) && again[i](2) == corners[i](2);
      Check(same, "FastDetector::DetectPyramid(pyramid) == the corners Frame::CreateCorners produced, same order", again.size());
    }
    // ---- Frame::FilterCorners (map.cc:283) + descriptors, ORBDetector::GetDescriptor / Distance
    f0->FilterCorners();
    const std::vector<int> filtered = f0->GetFilteredCorners();
    Check(filtered.size() >= 150 && filtered.size() <= 300, "FilterCorners(): one corner per free 32-px cell above MinFeatureScore", filtered.size());
    const std::vector<std::vector<uchar>> &descs = f0->GetDescriptors();
    Check(descs.size() == corners.size() && descs[filtered[0]].size() == 32, "GetDescriptors(): 32 bytes per corner", descs.size());
    {
      int tested = 0, equal = 0, self0 = 0;
      for (size_t q = 0; q < filtered.size() && tested < 40; q += 5) {
        const Vector3i cnr = corners[filtered[q]];
        std::vector<uchar> d;
        if (!orb.GetDescriptor(pyr[cnr(2)], Vector2i(cnr(0), cnr(1)), &d)) continue;
        tested++;
        equal += d == descs[filtered[q]];
        self0 += orb.Distance(d, descs[filtered[q]]) == 0;
      }
      Check(tested >= 20 && equal == tested && self0 == tested, "ORBDetector::GetDescriptor == the frame's descriptor, Distance(d, d) == 0", tested);
      Check(orb.Distance(descs[filtered[0]], descs[filtered[1]]) > 0, "Distance of two different corners > 0", orb.Distance(descs[filtered[0]], descs[filtered[1]]));
    }
    // ---- a keyframe with points on the scene plane (the bootstrap stand-in), then the tracker's calls on a second frame
    PlaneMap map(Vector3d(0, 0, 1), 2.0);
    f0->SetPose(SE3());
    f0->SetKeyframe();
    map.AddKeyframe(f0, false);
    map.SeedFromFiltered(f0);
    Check(f0->GetNumFeatures() == static_cast<int>(filtered.size()), "PlaneMap::SeedFromFiltered: one feature + point per filtered corner", f0->GetNumFeatures());

    std::shared_ptr<Frame> f3 = std::make_shared<Frame>(&camera, &orb, img3, true);
    f3->SetID(1);
    f3->SetPose(f0->GetPose());  // no motion model: start from the keyframe's pose
    ImageAlign ia;
    const int n_meas = ia.ComputePose(f0, f3);  // sdvl.cc:189
    const double d_ia = PoseDiff(f3->GetPose(), PoseOf(3));
    Check(n_meas >= 100 && d_ia < 2e-3, "ImageAlign::ComputePose(kf, frame) recovers the rendered motion (pose error)", d_ia);
    Check(ia.GetError() < 1e-3, "ImageAlign::GetError() small after convergence", ia.GetError());

    // ---- Matcher::SearchPoint for single features (map.cc:326 style): found, within a pixel of the true projection
    {
      Matcher matcher(Config::PatchSize());
      int tried = 0, found = 0, close = 0;
      const SE3 Ttrue = PoseOf(3);
      for (const auto &ft : f0->GetFeatures()) {
        if (tried >= 60) break;
        Point *pt = ft->GetPointRaw();
        if (!pt) continue;
        const Vector3d P = pt->GetPosition();
        const Vector3d pc = Ttrue * P;
        const Vector2d truth(cam4[2] + cam4[0] * pc(0) / pc(2), cam4[3] + cam4[1] * pc(1) / pc(2));
        if (truth(0) < 40 || truth(1) < 40 || truth(0) > W - 40 || truth(1) > H - 40) continue;
        tried++;
        Vector2d px(truth(0) + 0.8, truth(1) - 0.6);  // a slightly wrong prediction, as after image alignment
        int level = -1;
        if (matcher.SearchPoint(f3, ft, pt->GetInverseDepth(), pt->GetStd(), true, &px, &level)) {
          found++;
          const double dx = px(0) - truth(0), dy = px(1) - truth(1);
          close += std::sqrt(dx * dx + dy * dy) < 1.0;
        }
      }
      Check(tried >= 40 && found >= tried * 6 / 10 && close >= found * 9 / 10, "Matcher::SearchPoint: most points found, within 1 px of their true projection", found);
    }
    // ---- FeatureAlign::Reproject + OptimizePose (sdvl.cc:193,200)
    {
      RandStream rng(1);
      FeatureAlign fa(&map, &camera, Config::MaxMatches(), &rng);
      fa.Reproject(f3, f0, f0);
      Check(fa.GetMatches() >= 100 && fa.GetAttempts() >= fa.GetMatches(), "FeatureAlign::Reproject: >= 100 matches", fa.GetMatches());
      const bool ok = fa.OptimizePose(f3);
      const double d_fa = PoseDiff(f3->GetPose(), PoseOf(3));
      Check(ok && d_fa < 5e-4, "FeatureAlign::OptimizePose refines the pose (pose error)", d_fa);
      Check(f3->GetNumFeatures() >= 100, "the matched features were added to the frame", f3->GetNumFeatures());
    }
  } catch (const std::exception &e) {
    std::cerr << "api_surface_check: " << e.what() << std::endl;
    return 1;
  }
  std::printf("%d check(s) failed\n", g_failed);
  return g_failed == 0 ? 0 : 3;
}
