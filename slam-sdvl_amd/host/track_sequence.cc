// track_sequence.cc — the per-frame loop of the reference's main.cc:89-171 on the MI355X front-end, in plain C++ against
// sdvl_host.h: read (or render) a frame, undistort it, SDVL::HandleFrame(), print the pose.  It is the "maintainer's
// view" of the drop-in: nothing here knows about HIP; the classes are the reference's (Camera, Config, SDVL, Map).
//
//   track_sequence --synthetic N [--seed S]                      N frames of the S-A scene, rendered on the host
//   track_sequence --list frames.txt                             one binary PGM (P5, 8 bit) path per line, e.g. a TUM / EuRoC list
//   common:  [--config file.cfg] [--size W H] [--cam fx fy u0 v0] [--dist d0 d1 d2 d3 d4] [--plane nx ny nz d] [--mapper]
//   round 5 (bench.py's latency legs):  [--texture plane|camera]  [--trackers N]  N cameras = N host threads, each with its own Device
//            (= HIP stream), Camera, Map and SDVL, all fed the same frames;  [--batch]  with --trackers N: the N cameras step together through
//            ONE sdvl::SDVLBatch::HandleFrames call per frame on one thread and one stream (the batched form of the same call);  [--prerender]  the synthetic frames are rendered on the GPU
//            into page-locked host memory before the loop (the window of main.cc:136-138 never contained the rendering anyway; this keeps
//            a 300-frame run short);  [--pageable]  with --prerender: plain malloc'ed frames, what an unregistered cv::Mat is;
//            [--lookahead]  the loop undistorts frame k + 1 before it tracks frame k and names it with SDVL::SetNextImage (a sequence from
//            disk is there before it is needed): its pyramid and corners are built while the host finishes frame k;
//            [--set SDVL.key value]  a configuration value (after --config);  [--quiet]  no per-frame lines;  [--json]  one JSON line with the rates;  [--profile]  per-kernel dispatch time and host stages
//
// The two-frame homography bootstrap is out of scope: the first frame becomes a keyframe whose points are
// seeded on the plane n.X = d (world = first camera), which is exact for the synthetic scene and a stand-in for real data.
// Output: one line per frame  "k state quality matches attempts inliers  qw qx qy qz tx ty tz"  and the tracked frames/s of
// the HandleFrame calls alone (the window of main.cc:136-138).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
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
  bool prerender = false, pageable = false, quiet = false, json = false, profile = false, batch = false, lookahead = false;
  int n_trackers = 1;
  std::vector<std::pair<std::string, double>> sets;  // --set SDVL.key value, applied after the configuration file
  unsigned texture = SDVL_TEXTURE_PLANE_NOISE;
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
    else if (a == "--texture") { need(1); const std::string t = argv[++i]; if (t == "camera") texture = SDVL_TEXTURE_CAMERA; else if (t != "plane") { std::cerr << "unknown texture " << t << std::endl; return 2; } }
    else if (a == "--trackers") { need(1); n_trackers = std::max(1, std::atoi(argv[++i])); }
    else if (a == "--batch") batch = true;
    else if (a == "--lookahead") lookahead = true;
    else if (a == "--prerender") prerender = true;
    else if (a == "--pageable") pageable = true;
    else if (a == "--quiet") quiet = true;
    else if (a == "--json") json = true;
    else if (a == "--profile") profile = true;
    else if (a == "--set") { need(2); sets.emplace_back(argv[i + 1], std::atof(argv[i + 2])); i += 2; }
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

  for (const auto &kv : sets)
    if (!c.SetParameter(kv.first, kv.second)) { std::cerr << "unknown parameter " << kv.first << std::endl; return 2; }

  std::vector<std::string> files;
  if (!list.empty()) {
    std::ifstream lf(list);
    for (std::string line; std::getline(lf, line);)
      if (!line.empty() && line[0] != '#') files.push_back(line);
    if (files.empty()) { std::cerr << "no frames in " << list << std::endl; return 2; }
  }
  const int n_frames = n_synth > 0 ? n_synth : static_cast<int>(files.size());

  const double tw[6] = {0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005};
  auto view_of = [&](int k) {  // T_k = Exp(k * xi), SURVEY §8d
    Vector6d xi;
    for (int q = 0; q < 6; q++) xi.v[q] = tw[q] * k;
    const SE3 T = SE3::Exp(xi);
    sdvl_synth_view v = {};
    v.fx = cam4[0]; v.fy = cam4[1]; v.u0 = cam4[2]; v.v0 = cam4[3];
    const M3 R = T.GetRotation();
    for (int q = 0; q < 9; q++) v.R[q] = R.m[q];
    const Vector3d t = T.GetTranslation();
    for (int q = 0; q < 3; q++) v.t[q] = t(q);
    for (int q = 0; q < 4; q++) v.plane[q] = plane[q];
    v.seed = seed;
    v.frame_id = static_cast<uint32_t>(k);
    v.texture = texture;
    for (int q = 0; q < 5; q++) v.dist[q] = dist[q];  // --dist: the frames are what a camera with that lens records (camera.UndistortImage undoes it)
    return v;
  };

  try {
    const size_t frame_bytes = static_cast<size_t>(W) * H;
    // --prerender: all frames of the synthetic sequence, rendered by the device generator (the same bytes as the host one) into
    // page-locked (or, --pageable, plain) host memory before anything is timed
    uint8_t *pool = nullptr;
    bool pool_pinned = false;
    std::unique_ptr<Device> render_dev;
    if (prerender && n_synth > 0) {
      render_dev.reset(new Device(0));
      sdvl_ctx *rc = render_dev->ctx();
      if (pageable) {
        pool = static_cast<uint8_t *>(std::malloc(frame_bytes * n_frames));
        if (!pool) throw std::runtime_error("out of memory for the frame pool");
      } else {
        void *pp = nullptr;
        render_dev->Check(sdvl_host_alloc_pinned(rc, static_cast<int64_t>(frame_bytes * n_frames), &pp), "sdvl_host_alloc_pinned");
        pool = static_cast<uint8_t *>(pp);
        pool_pinned = true;
      }
      void *dbuf = nullptr;
      const int chunk = 32;
      render_dev->Check(sdvl_device_malloc(rc, static_cast<int64_t>(frame_bytes * chunk), &dbuf), "sdvl_device_malloc");
      for (int k0 = 0; k0 < n_frames; k0 += chunk) {
        const int n = std::min(chunk, n_frames - k0);
        std::vector<sdvl_synth_view> views;
        for (int k = k0; k < k0 + n; k++) views.push_back(view_of(k));
        render_dev->Check(sdvl_synth_render(rc, n, views.data(), W, H, dbuf, static_cast<int64_t>(frame_bytes)), "sdvl_synth_render");
        render_dev->Check(sdvl_device_download(rc, dbuf, static_cast<int64_t>(frame_bytes * n), pool + frame_bytes * k0), "sdvl_device_download");
      }
      render_dev->Check(sdvl_device_free(rc, dbuf), "sdvl_device_free");
    }

    struct PerTracker { double busy = 0.0; int tracked = 0; std::vector<float> ms; std::string err; };
    std::vector<PerTracker> per(n_trackers);
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::chrono::steady_clock::time_point t_go;
    auto run_tracker = [&](int id) {
      PerTracker &me = per[id];
      try {
        Device dev(0);             // one per thread that enters the path; fails loudly without an MI355X
        Device::SetCurrent(&dev);
        Camera camera(W, H, cam4[0], cam4[1], cam4[2], cam4[3]);
        camera.SetDistortions(dist[0], dist[1], dist[2], dist[3], dist[4]);  // camera.cc:39-67
        std::unique_ptr<Map> map;
        if (mapper) map.reset(new MapperMap(Vector3d(plane[0], plane[1], plane[2]), plane[3], &camera));
        else map.reset(new PlaneMap(Vector3d(plane[0], plane[1], plane[2]), plane[3]));
        SDVL sdvl(&camera, map.get());
        std::vector<uint8_t> px(pool ? 0 : frame_bytes);
        me.ms.reserve(n_frames);
        Image ahead;
        bool ahead_valid = false;
        if (n_trackers > 1) {  // all cameras start together
          ready.fetch_add(1);
          while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        }
        for (int k = 0; k < n_frames; k++) {
          uint8_t *data = px.data();
          if (pool) {
            data = pool + frame_bytes * k;
          } else if (n_synth > 0) {
            const sdvl_synth_view v = view_of(k);
            sdvl_synth_render_host(&v, W, H, px.data(), W);
          } else {
            int w = 0, h = 0;
            if (!ReadPGM(files[k], &w, &h, &px) || w != W || h != H) {
              me.err = "cannot read " + files[k] + " as a binary PGM of the configured size";
              return;
            }
            data = px.data();
          }
          Image img;
          img.data = data; img.cols = W; img.rows = H; img.step = W;
          Image imgu;
          if (lookahead && pool && ahead_valid) {
            imgu = ahead;                                          // undistorted one iteration ago
          } else {
            camera.UndistortImage(img, &imgu);                      // main.cc:133
          }
          ahead_valid = false;
          if (lookahead && pool && k + 1 < n_frames) {              // the next frame is there already: undistort it now, name it
            Image nxt;
            nxt.data = pool + frame_bytes * (k + 1); nxt.cols = W; nxt.rows = H; nxt.step = W;
            camera.UndistortImage(nxt, &ahead);
            ahead_valid = true;
            if (k > 0) sdvl.SetNextImage(ahead);                   // (frame 0 is the bootstrap: no tracked step to queue behind)
          }
          if (k == 1 && profile) dev.Check(sdvl_ctx_timing_enable(dev.ctx(), 1), "sdvl_ctx_timing_enable");
          const auto t0 = std::chrono::steady_clock::now();
          sdvl.HandleFrame(imgu);                                 // main.cc:136-138
          if (mapper) sdvl.Mapping();                             // sequential mode, main.cc:148-149
          const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          const FrameStats &st = sdvl.LastStats();
          if (k > 0) { me.busy += dt; me.tracked += st.quality != 2; me.ms.push_back(static_cast<float>(dt * 1e3)); }  // frame 0 is the bootstrap keyframe
          if (!quiet && id == 0)
            std::printf("%d %d %d %d %d %d  %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", k, st.state, st.quality, st.matches, st.attempts, st.inliers,
                        st.pose[0], st.pose[1], st.pose[2], st.pose[3], st.pose[4], st.pose[5], st.pose[6]);
        }
        if (profile && id == 0) {
          char names[32][32];
          double ms[32];
          int64_t launches[32];
          int n = 0;
          dev.Check(sdvl_ctx_timing_get(dev.ctx(), 32, names, ms, launches, &n), "sdvl_ctx_timing_get");
          const int nf = std::max(1, n_frames - 1);
          double sum = 0.0;
          for (int q = 0; q < n; q++) sum += ms[q];
          std::fprintf(stderr, "kernel dispatch time per tracked frame (HIP events on the tracker's stream, %d frames): total %.1f us\n", nf, sum / nf * 1e3);
          for (int q = 0; q < n; q++)
            std::fprintf(stderr, "  %-24s %8.1f us/frame  %6.2f launches/frame  %7.1f us/launch\n", names[q], ms[q] / nf * 1e3,
                         static_cast<double>(launches[q]) / nf, launches[q] ? ms[q] / launches[q] * 1e3 : 0.0);
          if (const StageTimes *stt = sdvl.HandleFrameStageTimes()) {
            static const char *stage_names[] = {"upload_pyr", "fast", "select", "corners_orb", "prelude", "image_align", "prepare", "search", "finish", "pose",
                                                "mapping", "epilogue", "mapper", "total"};
            std::fprintf(stderr, "host stages per HandleFrame (wall, %ld calls):\n", stt->steps);
            for (int q = 0; q <= ST_TOTAL; q++)
              if (stt->t[q] > 0) std::fprintf(stderr, "  %-12s %8.1f us\n", stage_names[q], stt->t[q] / std::max(1L, stt->steps) * 1e6);
          }
        }
      } catch (const std::exception &e) {
        me.err = e.what();
      }
    };
    // --batch: the N cameras are frames of ONE SDVLBatch::HandleFrames call (one thread, one stream, one launch per kernel for all N)
    auto run_batch = [&]() {
      PerTracker &me = per[0];
      try {
        Device dev(0);
        Device::SetCurrent(&dev);
        Camera camera(W, H, cam4[0], cam4[1], cam4[2], cam4[3]);
        camera.SetDistortions(dist[0], dist[1], dist[2], dist[3], dist[4]);
        std::vector<std::unique_ptr<Map>> maps;
        std::vector<std::unique_ptr<SDVL>> trackers;
        std::vector<SDVL *> raw;
        for (int i = 0; i < n_trackers; i++) {
          if (mapper) maps.emplace_back(new MapperMap(Vector3d(plane[0], plane[1], plane[2]), plane[3], &camera));
          else maps.emplace_back(new PlaneMap(Vector3d(plane[0], plane[1], plane[2]), plane[3]));
          trackers.emplace_back(new SDVL(&camera, maps.back().get()));
          raw.push_back(trackers.back().get());
        }
        {
          SDVLBatch b(&dev, raw, 1);
          std::vector<FrameStats> st(n_trackers);
          std::vector<uint8_t> px(pool ? 0 : frame_bytes);
          for (int k = 0; k < n_frames; k++) {
            uint8_t *data = px.data();
            if (pool) data = pool + frame_bytes * k;
            else if (n_synth > 0) { const sdvl_synth_view v = view_of(k); sdvl_synth_render_host(&v, W, H, px.data(), W); }
            else { int w = 0, h = 0; if (!ReadPGM(files[k], &w, &h, &px) || w != W || h != H) { me.err = "cannot read " + files[k]; return; } data = px.data(); }
            Image img;
            img.data = data; img.cols = W; img.rows = H; img.step = W;
            std::vector<Image> imgs(n_trackers);
            for (int i = 0; i < n_trackers; i++) camera.UndistortImage(img, &imgs[i]);  // every camera its own frame in HBM (main.cc:133, outside the window)
            const auto t0 = std::chrono::steady_clock::now();
            b.HandleFrames(imgs, st.data());
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (k > 0) {
              me.busy += dt;
              for (int i = 0; i < n_trackers; i++) me.tracked += st[i].quality != 2;
              me.ms.push_back(static_cast<float>(dt * 1e3));
            }
            if (!quiet)
              std::printf("%d %d %d %d %d %d  %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", k, st[0].state, st[0].quality, st[0].matches, st[0].attempts, st[0].inliers,
                          st[0].pose[0], st[0].pose[1], st[0].pose[2], st[0].pose[3], st[0].pose[4], st[0].pose[5], st[0].pose[6]);
          }
        }
        trackers.clear();
        maps.clear();
      } catch (const std::exception &e) {
        me.err = e.what();
      }
    };
    double wall = 0.0;
    if (batch) {
      const auto w0 = std::chrono::steady_clock::now();
      run_batch();
      wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    } else if (n_trackers == 1) {
      const auto w0 = std::chrono::steady_clock::now();
      run_tracker(0);
      wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    } else {
      std::vector<std::thread> th;
      for (int i = 0; i < n_trackers; i++) th.emplace_back(run_tracker, i);
      while (ready.load() < n_trackers) {
        bool failed = false;
        for (const PerTracker &p : per) failed = failed || !p.err.empty();
        if (failed) break;
        std::this_thread::yield();
      }
      const auto w0 = std::chrono::steady_clock::now();
      go.store(true, std::memory_order_release);
      for (std::thread &t : th) t.join();
      wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    }
    for (const PerTracker &p : per)
      if (!p.err.empty()) { std::cerr << "track_sequence: " << p.err << std::endl; return 1; }
    if (pool) {
      if (pool_pinned) render_dev->Check(sdvl_host_free_pinned(render_dev->ctx(), pool), "sdvl_host_free_pinned");
      else std::free(pool);
    }
    int tracked = 0;
    double busy_max = 0.0, busy_sum = 0.0;
    std::vector<float> all_ms;
    for (const PerTracker &p : per) {
      tracked += p.tracked;
      busy_max = std::max(busy_max, p.busy);
      busy_sum += p.busy;
      all_ms.insert(all_ms.end(), p.ms.begin(), p.ms.end());
    }
    std::sort(all_ms.begin(), all_ms.end());
    const auto pct = [&](double q) { return all_ms.empty() ? 0.0 : static_cast<double>(all_ms[std::min(all_ms.size() - 1, static_cast<size_t>(q * all_ms.size()))]); };
    // one camera: tracked frames over the summed HandleFrame time (the window of main.cc:136-138).  N cameras: every camera's own rate is
    // its tracked frames over ITS summed HandleFrame time; the job's rate is all tracked frames over the slowest camera's summed time
    const double total = busy_max > 0 ? tracked / busy_max : 0.0;
    const double per_camera = batch ? total / n_trackers : (busy_sum > 0 ? tracked / busy_sum : 0.0);
    std::fprintf(stderr, "%d tracked frames in %.3f s of HandleFrame = %.1f tracked frames/s (%d sequence(s), one stream each; per camera %.1f)\n", tracked,
                 busy_max, total, n_trackers, per_camera);
    if (json)
      std::printf("{\"trackers\": %d, \"frames\": %d, \"tracked\": %d, \"frames_per_s\": %.2f, \"frames_per_s_per_camera\": %.2f, \"ms_per_frame_p50\": %.4f, "
                  "\"ms_per_frame_p95\": %.4f, \"ms_per_frame_max\": %.4f, \"wall_s\": %.3f, \"input\": \"%s\", \"texture\": \"%s\", \"width\": %d, \"height\": %d, "
                  "\"mapper\": %s, \"api\": \"%s\"}\n",
                  n_trackers, n_frames, tracked, total, per_camera, pct(0.5), pct(0.95), all_ms.empty() ? 0.0 : static_cast<double>(all_ms.back()), wall,
                  pool ? (pool_pinned ? "page-locked host memory" : "pageable host memory") : "host memory of the loop (pageable)",
                  texture == SDVL_TEXTURE_CAMERA ? "camera" : "plane", W, H, mapper ? "true" : "false",
                  batch ? "sdvl::SDVLBatch::HandleFrames, one call per frame for all cameras (one thread, one stream)"
                  : lookahead ? "SDVL::SetNextImage + SDVL::HandleFrame per frame (the next frame of the sequence named one call ahead)"
                        : "SDVL::HandleFrame per frame and camera (host/track_sequence.cc, the loop of main.cc:126-159; one thread and stream per camera)");
  } catch (const std::exception &e) {
    std::cerr << "track_sequence: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
