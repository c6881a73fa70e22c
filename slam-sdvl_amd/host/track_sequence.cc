// track_sequence.cc — the per-frame loop of the reference's main.cc:89-171 on the MI355X front-end, in plain C++ against
// sdvl_host.h: read (or render) a frame, undistort it, SDVL::HandleFrame(), print the pose.  It is the "maintainer's
// view" of the drop-in: nothing here knows about HIP; the classes are the reference's (Camera, Config, SDVL, Map).
//
//   track_sequence --synthetic N [--seed S]                      N frames of the S-A scene, rendered on the host
//   track_sequence --list frames.txt                             one binary PGM (P5, 8 bit) path per line, e.g. a TUM / EuRoC list
//   common:  [--config file.cfg] [--size W H] [--cam fx fy u0 v0] [--dist d0 d1 d2 d3 d4] [--plane nx ny nz d] [--mapper]
//
// The two-frame homography bootstrap is out of scope: the first frame becomes a keyframe whose points are
// seeded on the plane n.X = d (world = first camera), which is exact for the synthetic scene and a stand-in for real data.
// Output: one line per frame  "k state quality matches attempts inliers  qw qx qy qz tx ty tz"  and the tracked frames/s of
// the HandleFrame calls alone (the window of main.cc:136-138).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "sdvl_host.h"
#undef SDVL_HD  // sdvl_math.h and sdvl_synth.h each define their own host/device qualifier macro
#include "../csrc/sdvl_synth.h"

extern "C" int sdvl_synth_render_host(const sdvl_synth_view *view, int width, int height, uint8_t *out, int stride);

using namespace sdvl;

namespace {

bool ReadPGM(const std::string &path, int *w, int *h, std::vector<uint8_t> *px) {
  std::ifstream f(path, std::ios::binary);
  std::string magic;
  if (!(f >> magic) || magic != "P5") return false;
  int vals[3], n = 0;
  while (n < 3) {  // width, height, maxval with '#' comments in between
    f >> std::ws;
    if (f.peek() == '#') { std::string skip; std::getline(f, skip); continue; }
    if (!(f >> vals[n])) return false;
    n++;
  }
  if (vals[2] != 255) return false;
  f.get();  // the single whitespace byte after maxval
  *w = vals[0];
  *h = vals[1];
  px->resize(static_cast<size_t>(vals[0]) * vals[1]);
  f.read(reinterpret_cast<char *>(px->data()), static_cast<std::streamsize>(px->size()));
  return static_cast<size_t>(f.gcount()) == px->size();
}

}  // namespace

int main(int argc, char **argv) {
  int W = 640, H = 480, n_synth = 0;
  unsigned seed = 20260001;
  double cam4[4] = {517.3, 516.5, 318.6, 255.3}, dist[5] = {0, 0, 0, 0, 0}, plane[4] = {0, 0, 1, 2.0};
  std::string list, cfg;
  bool mapper = false, size_given = false, cam_given = false, dist_given = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&](int k) { if (i + k >= argc) { std::cerr << "missing value after " << a << std::endl; std::exit(2); } };
    if (a == "--synthetic") { need(1); n_synth = std::atoi(argv[++i]); }
    else if (a == "--seed") { need(1); seed = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10)); }
    else if (a == "--list") { need(1); list = argv[++i]; }
    else if (a == "--config") { need(1); cfg = argv[++i]; }
    else if (a == "--size") { need(2); W = std::atoi(argv[++i]); H = std::atoi(argv[++i]); size_given = true; }
    else if (a == "--cam") { need(4); for (int k = 0; k < 4; k++) cam4[k] = std::atof(argv[++i]); cam_given = true; }
    else if (a == "--dist") { need(5); for (int k = 0; k < 5; k++) dist[k] = std::atof(argv[++i]); dist_given = true; }
    else if (a == "--plane") { need(4); for (int k = 0; k < 4; k++) plane[k] = std::atof(argv[++i]); }
    else if (a == "--mapper") mapper = true;
    else { std::cerr << "unknown argument " << a << std::endl; return 2; }
  }
  if ((n_synth > 0) == !list.empty()) { std::cerr << "give either --synthetic N or --list file" << std::endl; return 2; }

  // main.cc:60-75: configuration file, then the TUM overrides the reference's config_tum_f1.cfg carries
  Config &c = Config::GetInstance();
  c.SetParameter("SDVL.cell_size", 32); c.SetParameter("SDVL.max_matches", 200); c.SetParameter("SDVL.use_orb", 1);
  c.SetParameter("SDVL.fast_threshold", 10); c.SetParameter("SDVL.num_features", 1000); c.SetParameter("SDVL.min_avg_shift", 5);
  c.SetParameter("SDVL.max_keyframes", 1000); c.SetParameter("SDVL.lost_ratio", 0.7);
  if (!cfg.empty()) {
    if (!c.ReadParameters(cfg)) { std::cerr << "cannot read " << cfg << std::endl; return 2; }
    // the camera block of the file (main.cc:72: Camera() reads it) unless --size / --cam / --dist said otherwise
    const CameraParameters &cp = Config::GetCameraParameters();
    if (!size_given) { W = cp.width; H = cp.height; }
    if (!cam_given) { cam4[0] = cp.fx; cam4[1] = cp.fy; cam4[2] = cp.u0; cam4[3] = cp.v0; }
    if (!dist_given) { dist[0] = cp.d1; dist[1] = cp.d2; dist[2] = cp.d3; dist[3] = cp.d4; dist[4] = cp.d5; }
  }

  std::vector<std::string> files;
  if (!list.empty()) {
    std::ifstream lf(list);
    for (std::string line; std::getline(lf, line);)
      if (!line.empty() && line[0] != '#') files.push_back(line);
    if (files.empty()) { std::cerr << "no frames in " << list << std::endl; return 2; }
  }
  const int n_frames = n_synth > 0 ? n_synth : static_cast<int>(files.size());

  try {
    Device dev(0);             // one per thread that enters the path; fails loudly without an MI355X
    Device::SetCurrent(&dev);
    Camera camera(W, H, cam4[0], cam4[1], cam4[2], cam4[3]);
    camera.SetDistortions(dist[0], dist[1], dist[2], dist[3], dist[4]);  // camera.cc:39-67
    std::unique_ptr<Map> map;
    if (mapper) map.reset(new MapperMap(Vector3d(plane[0], plane[1], plane[2]), plane[3], &camera));
    else map.reset(new PlaneMap(Vector3d(plane[0], plane[1], plane[2]), plane[3]));
    SDVL sdvl(&camera, map.get());

    std::vector<uint8_t> px(static_cast<size_t>(W) * H);
    double busy = 0.0;
    int tracked = 0;
    for (int k = 0; k < n_frames; k++) {
      if (n_synth > 0) {  // T_k = Exp(k * xi), SURVEY §8d
        Vector6d xi;
        const double tw[6] = {0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005};
        for (int q = 0; q < 6; q++) xi.v[q] = tw[q] * k;
        const SE3 T = SE3::Exp(xi);
        sdvl_synth_view v;
        v.fx = cam4[0]; v.fy = cam4[1]; v.u0 = cam4[2]; v.v0 = cam4[3];
        const M3 R = T.GetRotation();
        for (int q = 0; q < 9; q++) v.R[q] = R.m[q];
        const Vector3d t = T.GetTranslation();
        for (int q = 0; q < 3; q++) v.t[q] = t(q);
        for (int q = 0; q < 4; q++) v.plane[q] = plane[q];
        v.seed = seed;
        v.frame_id = static_cast<uint32_t>(k);
        sdvl_synth_render_host(&v, W, H, px.data(), W);
      } else {
        int w = 0, h = 0;
        if (!ReadPGM(files[k], &w, &h, &px) || w != W || h != H) {
          std::cerr << "cannot read " << files[k] << " as a " << W << "x" << H << " binary PGM" << std::endl;
          return 3;
        }
      }
      Image img;
      img.data = px.data(); img.cols = W; img.rows = H; img.step = W;
      Image imgu;
      camera.UndistortImage(img, &imgu);                      // main.cc:133
      const auto t0 = std::chrono::steady_clock::now();
      sdvl.HandleFrame(imgu);                                 // main.cc:136-138
      if (mapper) sdvl.Mapping();                             // sequential mode, main.cc:148-149
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const FrameStats &st = sdvl.LastStats();
      if (k > 0) { busy += dt; tracked += st.quality != 2; }  // frame 0 is the bootstrap keyframe
      std::printf("%d %d %d %d %d %d  %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", k, st.state, st.quality, st.matches, st.attempts, st.inliers,
                  st.pose[0], st.pose[1], st.pose[2], st.pose[3], st.pose[4], st.pose[5], st.pose[6]);
    }
    std::fprintf(stderr, "%d tracked frames in %.3f s of HandleFrame = %.1f tracked frames/s (one sequence, one stream)\n", tracked, busy,
                 busy > 0 ? tracked / busy : 0.0);
  } catch (const std::exception &e) {
    std::cerr << "track_sequence: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
