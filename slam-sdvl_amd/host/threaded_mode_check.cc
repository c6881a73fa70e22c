// threaded_mode_check.cc — main.cc:89-171 in its DEFAULT mode (`sequential = false`, main.cc:97): handler->Start() runs the
// mapper on a thread of its own (map.cc:49-71) while the main thread tracks.  Here that means two host threads inside the
// path at once, each with its own sdvl_ctx / HIP stream, sharing the tracker's frames read-only (SURVEY §8b Threading).
// The reference is not deterministic in this mode (the mapper sees a frame whenever it gets to it), so the checks are: every
// frame tracked, the pose follows the rendered trajectory, the mapper thread did its work (frames consumed, candidates
// created and converged), and the map ends up close to what sequential mode builds from the same frames.
//   threaded_mode_check [n_frames]      prints one line per mode; exit code 0 = all checks passed
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <thread>
#include <vector>

#include "sdvl_host.h"
#undef SDVL_HD
#include "../csrc/sdvl_synth.h"

extern "C" int sdvl_synth_render_host(const sdvl_synth_view *view, int width, int height, uint8_t *out, int stride);

using namespace sdvl;

static const double kCam[4] = {517.3, 516.5, 318.6, 255.3};
static const int W = 640, H = 480;

static SE3 PoseOf(int k) {
  Vector6d xi;
  const double tw[6] = {0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005};
  for (int q = 0; q < 6; q++) xi.v[q] = tw[q] * k;
  return SE3::Exp(xi);
}

static void Render(int k, std::vector<uint8_t> *px) {
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
  px->resize(static_cast<size_t>(W) * H);
  sdvl_synth_render_host(&v, W, H, px->data(), W);
}

struct Outcome {
  int tracked = 0, keyframes = 0;
  double pose_err = 0.0;
  MapperMap::Stats map;
  long updates = 0;
  size_t cloud = 0;
};

static Outcome Run(Camera *camera, int n_frames, bool sequential) {
  Outcome out;
  SDVL *handler = new SDVL(camera);            // main.cc:118
  if (!sequential) handler->Start();           // main.cc:119-120
  std::vector<uint8_t> px;
  for (int k = 0; k < n_frames; k++) {
    Render(k, &px);
    Image img(H, W, CV_8UC1, px.data()), imgu;
    camera->UndistortImage(img, &imgu);        // main.cc:133
    handler->HandleFrame(imgu);                // main.cc:136
    if (k > 0 && handler->GetTrackingQuality() != SDVL::TRACKING_BAD) out.tracked++;
    if (sequential) handler->Mapping();        // main.cc:148-149
    else std::this_thread::sleep_for(std::chrono::milliseconds(6));  // main.cc:157-158 paces the loop: the mapper thread gets its turn
  }
  double pa[7], pb[7];
  handler->GetPose().ToArray(pa);
  PoseOf(n_frames - 1).Inverse().ToArray(pb);
  for (int q = 0; q < 7; q++) out.pose_err = std::fmax(out.pose_err, std::fabs(pa[q] - pb[q]));
  if (!sequential) {
    std::this_thread::sleep_for(std::chrono::milliseconds(50));  // let the mapper drain its queue
    handler->Stop();                           // main.cc:165-166
  }
  std::vector<std::pair<SE3, bool>> trail;
  handler->GetCameraTrail(&trail);
  out.keyframes = static_cast<int>(trail.size());
  std::vector<Vector3d> cloud;
  handler->GetPoints(&cloud);
  out.cloud = cloud.size();
  MapperMap *m = dynamic_cast<MapperMap *>(handler->GetMap());
  out.map = m->GetStats();
  out.updates = m->Updates();
  delete handler;
  return out;
}

int main(int argc, char **argv) {
  const int n_frames = argc > 1 ? std::atoi(argv[1]) : 40;
  const char *cfg_path = "/tmp/sdvl_threaded_mode_check.cfg";
  {
    std::ofstream f(cfg_path);
    f << "Camera.width: 640\nCamera.height: 480\nCamera.fx: 517.3\nCamera.fy: 516.5\nCamera.u0: 318.6\nCamera.v0: 255.3\nCamera.d1: 0\n"
         "SDVL.cell_size: 32\nSDVL.min_avg_shift: 5\nSDVL.max_matches: 200\nSDVL.max_keyframes: 1000\nSDVL.use_orb: 1\n"
         "SDVL.fast_threshold: 10\nSDVL.lost_ratio: 0.7\nSDVL.num_features: 1000\nSDVL.map_scale: 2.0\n";
  }
  if (!Config::GetInstance().ReadParameters(cfg_path)) return 2;
  int failed = 0;
  try {
    Camera camera;
    const Outcome seq = Run(&camera, n_frames, true);
    const Outcome thr = Run(&camera, n_frames, false);
    for (const auto *o : {&seq, &thr})
      std::printf("%s tracked=%d keyframes=%d pose_err=%.3g candidates=%d converged=%d initialized=%d mapper_updates=%ld cloud=%zu\n",
                  o == &seq ? "sequential" : "threaded  ", o->tracked, o->keyframes, o->pose_err, o->map.candidates, o->map.converged, o->map.initialized,
                  o->updates, o->cloud);
    auto check = [&](bool ok, const char *what) {
      std::printf("%s  %s\n", ok ? "ok  " : "FAIL", what);
      if (!ok) failed++;
    };
    check(seq.tracked == n_frames - 1 && thr.tracked == n_frames - 1, "every frame tracked in both modes");
    check(seq.pose_err < 2e-3 && thr.pose_err < 2e-3, "the pose follows the rendered trajectory in both modes");
    check(thr.updates >= (n_frames - 1) / 2 && thr.updates <= seq.updates, "the mapper THREAD consumed frames (it may skip ordinary frames when a keyframe is queued, map.cc:93-100)");
    check(thr.map.initialized > 0 && thr.map.initialized >= seq.map.initialized / 2, "the mapper thread created candidates on its own context");
    check(std::abs(thr.keyframes - seq.keyframes) <= 2 + seq.keyframes / 4, "keyframe count close to sequential mode's");
    check(thr.cloud >= seq.cloud / 2, "map size comparable to sequential mode's");
  } catch (const std::exception &e) {
    std::cerr << "threaded_mode_check: " << e.what() << std::endl;
    return 1;
  }
  return failed == 0 ? 0 : 3;
}
