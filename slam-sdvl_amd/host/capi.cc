// capi.cc — flat C entry points over the host layer (sdvl_host.h) for the Python harness (tests, smoke, bench):
// B independent SDVL trackers on one MI355X stepping together through sdvl::SDVLBatch.
#include <dlfcn.h>
#include <malloc.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>
#include <ucontext.h>
#include <cxxabi.h>

#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <atomic>
#include <algorithm>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sdvl_host.h"

using namespace sdvl;

namespace {
struct Batch {
  Device *dev;
  std::unique_ptr<Camera> cam;
  std::vector<std::unique_ptr<PlaneMap>> maps;
  std::vector<std::unique_ptr<SDVL>> trackers;
  std::unique_ptr<SDVLBatch> batch;
  bool raw_input = false;  // sdvlh_batch_set_distortion: the images of a step are RAW camera frames (Image::raw)
  int w, h;
  std::string err;
};
thread_local std::string g_err;
bool g_use_mapper = false;  // batches created from now on own the reference's mapper instead of the plane map stub
}  // namespace

extern "C" {

struct sdvlh_frame_stats {
  int state, quality, matches, attempts, inliers, outliers, n_corners, align_meas, keyframe, relocalized;
  double pose[7];
  int align_features, align_iters, search_requests, lk_iters;
  int host_path;
};

const char *sdvlh_last_error() { return g_err.c_str(); }

void *sdvlh_device_create(int gpu) {
  try {
    return new Device(gpu);
  } catch (const std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
void sdvlh_device_destroy(void *d) { delete static_cast<Device *>(d); }
void *sdvlh_device_ctx(void *d) { return static_cast<Device *>(d)->ctx(); }

int sdvlh_config_set(const char *key, double value) { return Config::GetInstance().SetParameter(key, value) ? 0 : -1; }
int sdvlh_config_read(const char *filename) { return Config::GetInstance().ReadParameters(filename) ? 0 : -1; }
void sdvlh_config_reset() { Config::GetInstance().Reset(); }
// the values a test compares with its own copies of the reference's configuration files: camera block (11) followed by the
// SDVL.* keys those files set: cell_size, min_avg_shift, max_matches, max_keyframes, use_orb, fast_threshold, lost_ratio,
// num_features, min_matches
void sdvlh_config_snapshot(double *out20) {
  const CameraParameters &c = Config::GetCameraParameters();
  const double v[20] = {static_cast<double>(c.width), static_cast<double>(c.height), c.fx, c.fy, c.u0, c.v0, c.d1, c.d2, c.d3, c.d4, c.d5,
                        static_cast<double>(Config::CellSize()), static_cast<double>(Config::MinAvgShift()), static_cast<double>(Config::MaxMatches()),
                        static_cast<double>(Config::MaxKeyframes()), Config::UseORB() ? 1.0 : 0.0, static_cast<double>(Config::FastThreshold()),
                        Config::LostRatio(), static_cast<double>(Config::NumFeatures()), static_cast<double>(Config::MinMatches())};
  for (int i = 0; i < 20; i++) out20[i] = v[i];
}
// test hook: the host build of the sin/cos the ORB kernels use (csrc/sdvl_math.h), same IEEE arithmetic as on the device
void sdvlh_sincos_2pi(double x, double *sn, double *cs) { sincos_2pi(x, sn, cs); }

void *sdvlh_batch_create(void *device, int B, int w, int h, const double *cam4, const double *plane4, const double *first_poses7,
                         int host_threads) {
  try {
    Batch *b = new Batch();
    b->dev = static_cast<Device *>(device);
    Device::SetCurrent(b->dev);
    b->w = w;
    b->h = h;
    b->cam.reset(new Camera(w, h, cam4[0], cam4[1], cam4[2], cam4[3]));
    std::vector<SDVL *> raw;
    for (int i = 0; i < B; i++) {
      if (g_use_mapper) b->maps.emplace_back(new MapperMap(Vector3d(plane4[0], plane4[1], plane4[2]), plane4[3], b->cam.get()));
      else b->maps.emplace_back(new PlaneMap(Vector3d(plane4[0], plane4[1], plane4[2]), plane4[3]));
      b->trackers.emplace_back(new SDVL(b->cam.get(), b->maps.back().get(), SE3::FromArray(first_poses7 + 7 * i)));
      raw.push_back(b->trackers.back().get());
    }
    b->batch.reset(new SDVLBatch(b->dev, raw, host_threads));
    return b;
  } catch (const std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}

// The camera of the batch gets a lens (Camera::SetDistortions, camera.cc:39-67; dist5 = Camera.d1..d5 of the cfg files) and the images of
// every later step are taken as RAW camera frames: Camera::UndistortImage (main.cc:133) runs inside the step, fused into the frames'
// upload.  dist5[0] == 0 (the reference's test, camera.cc:46) turns it off again.
int sdvlh_batch_set_distortion(void *bp, const double *dist5) {
  Batch *b = static_cast<Batch *>(bp);
  if (!b || !dist5) return -1;
  b->cam->SetDistortions(dist5[0], dist5[1], dist5[2], dist5[3], dist5[4]);
  b->raw_input = b->cam->HasDistortion();
  return 0;
}

// map mode of the batches created AFTER the call: 0 = plane map stub (every keyframe seeded from the scene plane),
// 1 = the reference's mapper in sequential mode (map.cc; the first keyframe is still bootstrapped from the plane)
void sdvlh_set_mapper(int on) { g_use_mapper = on != 0; }
int sdvlh_batch_map_stats(void *bp, int i, int *out6) {
  Batch *b = static_cast<Batch *>(bp);
  MapperMap *m = dynamic_cast<MapperMap *>(b->maps[i].get());
  if (!m) return -1;
  const MapperMap::Stats s = m->GetStats();
  out6[0] = s.candidates; out6[1] = s.converged; out6[2] = s.initialized; out6[3] = s.linked; out6[4] = s.connected; out6[5] = s.keyframes;
  return 0;
}

void sdvlh_batch_destroy(void *bp) {
  Batch *b = static_cast<Batch *>(bp);
  if (!b) return;
  Device::SetCurrent(b->dev);
  b->batch.reset();
  b->trackers.clear();
  b->maps.clear();
  delete b;
}

static int step(Batch *b, std::vector<Image> &imgs, sdvlh_frame_stats *out) {
  try {
    std::vector<FrameStats> st(imgs.size());
    if (b->raw_input)
      for (Image &im : imgs) im.raw = true;
    b->batch->HandleFrames(imgs, st.data());
    for (size_t i = 0; i < imgs.size(); i++) {
      const FrameStats &s = st[i];
      sdvlh_frame_stats &o = out[i];
      o.state = s.state; o.quality = s.quality; o.matches = s.matches; o.attempts = s.attempts; o.inliers = s.inliers;
      o.outliers = s.outliers; o.n_corners = s.n_corners; o.align_meas = s.align_meas; o.keyframe = s.keyframe;
      o.relocalized = s.relocalized;
      std::memcpy(o.pose, s.pose, sizeof(o.pose));
      o.align_features = s.align_features; o.align_iters = s.align_iters; o.search_requests = s.search_requests; o.lk_iters = s.lk_iters;
      o.host_path = s.host_path;
    }
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// Camera::UndistortImage of the host layer (camera.cc:100-105) for one host image; the result comes back to the host
int sdvlh_camera_undistort(void *device, int w, int h, const double *cam4, const double *dist5, const uint8_t *in, int stride, uint8_t *out) {
  try {
    Device *dev = static_cast<Device *>(device);
    Device::SetCurrent(dev);
    Camera cam(w, h, cam4[0], cam4[1], cam4[2], cam4[3]);
    cam.SetDistortions(dist5[0], dist5[1], dist5[2], dist5[3], dist5[4]);
    Image u;
    cam.UndistortImage(Image::Wrap(in, w, h, stride), &u);
    dev->Check(sdvl_device_download(dev->ctx(), u.dev_src, static_cast<int64_t>(w) * h, out), "sdvl_device_download");
    return cam.HasDistortion() ? 1 : 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// device-resident tracking tables on (default) / off for the batches stepping from now on; process-wide
void sdvlh_set_track_tables(int on) { SDVLBatch::SetTrackTables(on != 0); }
int sdvlh_track_tables() { return SDVLBatch::TrackTables() ? 1 : 0; }
// the mapper's depth filter on the device (default) or on the host, for the mapper updates from now on; process-wide
void sdvlh_set_device_filter(int on) { MapperMap::SetDeviceFilter(on != 0); }
int sdvlh_device_filter() { return MapperMap::DeviceFilter() ? 1 : 0; }

// test hook: a digest of the map state of tracker i after SyncHostState — the points reachable from its keyframes' features
// (each once): count, sum of Score(), sum of failure counts, deleted ones, sum of inverse depths, sum of GetStd()
int sdvlh_batch_point_digest(void *bp, int i, double *out6) {
  Batch *b = static_cast<Batch *>(bp);
  Device::SetCurrent(b->dev);
  b->batch->SyncHostState();
  std::set<Point *> seen;
  for (int k = 0; k < 6; k++) out6[k] = 0.0;
  for (auto &kf : b->maps[i]->GetKeyframes())
    for (auto &ft : kf->GetFeatures()) {
      Point *p = ft ? ft->GetPointRaw() : nullptr;
      if (!p || !seen.insert(p).second) continue;
      out6[0] += 1.0;
      out6[1] += p->Score();
      out6[2] += p->GetFailed();
      out6[3] += p->ToDelete() ? 1.0 : 0.0;
      out6[4] += p->GetInverseDepth();
      out6[5] += p->GetStd();
    }
  return 0;
}

// pose stage on the device (default) or with the host implementation; process-wide
void sdvlh_set_device_pose(int on) { SDVLBatch::SetDevicePose(on != 0); }
int sdvlh_device_pose() { return SDVLBatch::DevicePose() ? 1 : 0; }

// accumulated wall time per SDVLBatch stage (seconds, sdvl::StageId order); returns the number of steps; reset != 0 clears
long sdvlh_batch_stage_times(void *bp, double *out, int cap, int reset) {
  Batch *b = static_cast<Batch *>(bp);
  StageTimes &st = b->batch->stage_times;
  for (int i = 0; i < cap && i < ST_COUNT; i++) out[i] = st.t[i];
  const long n = st.steps;
  if (reset) st = StageTimes();
  return n;
}

// imgs: B host pointers (row stride `stride`)
int sdvlh_batch_step_host(void *bp, const uint8_t *const *imgs, int stride, sdvlh_frame_stats *out) {
  Batch *b = static_cast<Batch *>(bp);
  std::vector<Image> v;
  for (size_t i = 0; i < b->trackers.size(); i++) v.push_back(Image::Wrap(imgs[i], b->w, b->h, stride));
  return step(b, v, out);
}

// imgs: B device pointers (frames already in HBM).  The images are aliased, not copied: they must stay valid while the
// trackers may still read them (the last frame and every keyframe keep their level 0).
int sdvlh_batch_step_device(void *bp, const void *const *dev_imgs, int stride, sdvlh_frame_stats *out) {
  Batch *b = static_cast<Batch *>(bp);
  std::vector<Image> v;
  for (size_t i = 0; i < b->trackers.size(); i++) v.push_back(Image::WrapDevice(dev_imgs[i], b->w, b->h, stride, true));
  return step(b, v, out);
}

// Look-ahead (SDVLBatch::SetNextImages): the device images the NEXT sdvlh_batch_step_device call will be given; their pyramids and
// corner detection are queued behind the coming step's search / pose chain.  They must stay valid AND UNCHANGED until that call:
// the look-ahead is matched to the next step's images by device address alone, so a buffer that is rewritten in between would be
// tracked from pyramids of its old content.  The look-ahead belongs to the one coming step: if that step cannot use it (a bootstrap
// step, the host-driven path) it is dropped.
int sdvlh_batch_set_next_device(void *bp, const void *const *dev_imgs, int stride) {
  Batch *b = static_cast<Batch *>(bp);
  try {
    std::vector<Image> v;
    if (dev_imgs)
      for (size_t i = 0; i < b->trackers.size(); i++) v.push_back(Image::WrapDevice(dev_imgs[i], b->w, b->h, stride, true));
    if (b->raw_input)
      for (Image &im : v) im.raw = true;
    b->batch->SetNextImages(v);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// the same for images in a buffer the caller will overwrite (an input ring): every frame copies its image into its own level 0
int sdvlh_batch_step_device_copy(void *bp, const void *const *dev_imgs, int stride, sdvlh_frame_stats *out) {
  Batch *b = static_cast<Batch *>(bp);
  std::vector<Image> v;
  for (size_t i = 0; i < b->trackers.size(); i++) v.push_back(Image::WrapDevice(dev_imgs[i], b->w, b->h, stride, false));
  return step(b, v, out);
}

// the same without the copy for the frames that never need one (round 3): the frames alias the ring slot for the step that tracks
// them; those that become keyframes copy their image out at the end of the step (Frame::OwnImages).  The caller must not refill
// the slot before work queued on the batch's stream has passed (sdvl_ctx_prefetch_images orders itself behind that stream).
int sdvlh_batch_step_device_transient(void *bp, const void *const *dev_imgs, int stride, sdvlh_frame_stats *out) {
  Batch *b = static_cast<Batch *>(bp);
  std::vector<Image> v;
  for (size_t i = 0; i < b->trackers.size(); i++) v.push_back(Image::WrapDevice(dev_imgs[i], b->w, b->h, stride, true, true));
  return step(b, v, out);
}

// ---------------------------------------------------------------------------------------------------------------
// Farm: G groups x Bg sequences on ONE GPU.  Every group owns a host thread, an sdvl::Device (= sdvl_ctx = HIP stream
// + staging + frame pool) and an SDVLBatch; groups free-run through their steps, so the host stages of one group
// overlap the kernels and PCIe copies of the others (one context per host thread, as include/sdvl_hip.h prescribes).
void sdvlh_farm_destroy(void *fp);

struct Farm {
  int gpu, G, Bg, w, h;
  std::vector<void *> devices;
  std::vector<void *> batches;
  std::string err;
  std::mutex m;
  std::condition_variable cv_work;
  // the current run
  int n_steps = 0, stride = 0;
  const void *const *dev_frames = nullptr;
  sdvlh_frame_stats *out = nullptr;
  std::vector<int> done;
  std::vector<char> busy;
  bool failed = false;

  int fibers_per_worker = 1;  // > 1: a worker interleaves that many group-steps, switching at every GPU wait
  bool host_input = false;    // the frame pointers of a run are HOST pointers (pinned): every step uploads its frames
  // host input travels ahead: while a group computes step s, its copy stream carries the images of the next steps into the free
  // slots of the group's input ring (kRingSlots x Bg frames of HBM); the frames then copy their image out of the ring (HBM -> HBM)
  bool input_ring = true;
  // Round 3: ONE copy stream for the whole farm, filled by a feeder thread of its own (sdvl_feed), and a ring of three steps per
  // group.  Before, every group's worker thread issued its own prefetch on its own copy stream: hipMemcpyAsync of 79 MB kept the
  // worker for a good part of the transfer (15 of the 34 ms of a group-step were spent outside HandleFrames), and the 16 groups'
  // transfers ran side by side and shared the link, so the one a group was waiting for was slowed by the ones nobody needed yet
  // (38-47 GB/s of the 57 the link gives; a ring of three was slower still).  The feeder issues the transfers one after the
  // other, the group that is furthest behind first; the workers only exchange events with it.
  int kRingSlots = 3;
  sdvl_feed *feed = nullptr;
  double feed_call_s = 0.0, feed_wait_s = 0.0;  // the feeder's time inside sdvl_feed_images / waiting for a free slot, last run
  double work_wait_s = 0.0;                     // the workers' time waiting for their step's transfer to be issued (under m)
  double arrive_wait_s = 0.0;                   // ... and for it to arrive (host-side polls, summed over groups)
  double feed_throttle_s = 0.0;                 // the feeder's time waiting for its own earlier transfers (at most 2 queued)
  long late_acquires = 0;                       // group-steps that began while their transfer was still under way
  // SDVL_FARM_TIMELINE=<file>: one line per group-step (group, step, enter, transfer issued, step returned, late) and per transfer
  // (group, step, call begin, call end) of a host-fed run, seconds since the run began
  struct StepMark { int g, s; double t_enter, t_issued, t_end; int late; double st[12]; int kf; };
  struct FeedMark { int g, s; double t0, t1; };
  std::vector<std::vector<StepMark>> step_marks;  // [group]
  std::vector<FeedMark> feed_marks;
  std::vector<std::vector<double>> mark_prev;
  std::chrono::steady_clock::time_point run_t0;
  double Since() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - run_t0).count(); }
  long feed_calls = 0;
  std::vector<int> issued;          // [group] steps whose images the feeder has queued
  std::condition_variable cv_feed;  // feeder <-> workers (under m)
  std::vector<void *> ring;
  std::vector<int> ring_ticket;  // [group][slot]: the prefetch that filled the slot

  // next (group, step) for a worker: the idle group that is furthest behind.  Returns -1 when nothing is left, -2 when
  // every remaining group is busy elsewhere (only with `may_block` false; otherwise it waits for one to come free).
  int TakeGroup(bool may_block) {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      if (failed) return -1;
      int best = -1, remaining = 0;
      for (int k = 0; k < G; k++) {
        if (done[k] < n_steps) remaining++;
        if (!busy[k] && done[k] < n_steps && (best < 0 || done[k] < done[best])) best = k;
      }
      if (remaining == 0) return -1;
      if (best >= 0) { busy[best] = 1; return best; }
      if (!may_block) return -2;
      cv_work.wait(lk);
    }
  }

  int StepGroup(int g) {
    const int total = G * Bg;
    const int s = done[g];  // only the owner of a busy group reads or writes its counter
    const size_t off = static_cast<size_t>(s) * total + static_cast<size_t>(g) * Bg;
    if (host_input && input_ring && feed) {
      sdvl_ctx *ctx = static_cast<sdvl_ctx *>(sdvlh_device_ctx(devices[g]));
      {
        std::unique_lock<std::mutex> lk(m);
        const auto tw = std::chrono::steady_clock::now();
        cv_feed.wait(lk, [&] { return issued[g] > s || failed; });
        work_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
        if (failed) return -1;
      }
      const double t_issued = Since();
      const int slot = g * kRingSlots + s % kRingSlots;
      const bool late = sdvl_feed_slot_arrived(feed, slot) == 0;  // statistics only: this step starts behind a transfer still under way
      if (late) {
        std::lock_guard<std::mutex> lk(m);
        late_acquires++;
      }
      // The step is submitted only once its images are in HBM: a stream that waits for a transfer ON THE DEVICE holds a barrier
      // packet in its hardware queue, and the streams that share the queue (4 queues for 16 groups) stall behind it although their
      // own images arrived long ago.
      if (late) {
        const auto tw = std::chrono::steady_clock::now();
        for (;;) {
          const int a = sdvl_feed_slot_arrived(feed, slot);
          if (a != 0) {
            if (a < 0) { g_err = std::string("input ring: ") + sdvl_feed_last_error(feed); return -1; }
            break;
          }
          timespec ts{0, 50000};
          nanosleep(&ts, nullptr);
        }
        const double dw = std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
        std::lock_guard<std::mutex> lk(m);
        arrive_wait_s += dw;
      }
      const double t_enter = Since();
      if (sdvl_ctx_feed_acquire(ctx, feed, slot) != SDVL_OK) {  // this step's kernels start behind its images
        g_err = std::string("input ring: ") + sdvl_last_error(ctx);
        return -1;
      }
      const size_t fb = static_cast<size_t>(w) * h;
      std::vector<void *> dst(Bg);
      uint8_t *base = static_cast<uint8_t *>(ring[g]) + static_cast<size_t>(s % kRingSlots) * Bg * fb;
      for (int i = 0; i < Bg; i++) dst[i] = base + static_cast<size_t>(i) * fb;
      const int rc = sdvlh_batch_step_device_transient(batches[g], dst.data(), w, out + off);
      // the kernels queued up to here (the keyframes' copy out of the slot included) are the slot's last readers
      if (rc == 0 && sdvl_ctx_feed_release(ctx, feed, slot) != SDVL_OK) {
        g_err = std::string("input ring: ") + sdvl_last_error(ctx);
        return -1;
      }
      if (!step_marks.empty()) {
        StepMark k{g, s, t_enter, t_issued, Since(), late ? 1 : 0, {0}, 0};
        double now_t[ST_COUNT];
        sdvlh_batch_stage_times(batches[g], now_t, ST_COUNT, 0);
        if (mark_prev.size() != static_cast<size_t>(G)) mark_prev.assign(G, std::vector<double>(ST_COUNT, 0.0));
        for (int i = 0; i < 12; i++) { k.st[i] = now_t[i] - mark_prev[g][i]; }
        mark_prev[g].assign(now_t, now_t + ST_COUNT);
        for (int i = 0; i < Bg; i++) k.kf += out[off + i].keyframe ? 1 : 0;
        step_marks[g].push_back(k);
      }
      return rc;
    }
    if (host_input) return sdvlh_batch_step_host(batches[g], reinterpret_cast<const uint8_t *const *>(dev_frames + off), stride, out + off);
    // frames resident in HBM: the group knows its next images — their pyramids and detection are queued a step ahead (SDVL_NO_LOOKAHEAD=1: A/B)
    static const bool lookahead = getenv("SDVL_NO_LOOKAHEAD") == nullptr;
    if (lookahead && s + 1 < n_steps && sdvlh_batch_set_next_device(batches[g], dev_frames + off + total, stride) != 0) return -1;
    return sdvlh_batch_step_device(batches[g], dev_frames + off, stride, out + off);
  }

  // the message of a failed group-step: the calling thread's own last error (thread-local), never another thread's
  static std::string StepError() { return sdvlh_last_error(); }

  void FinishGroup(int g, int rc) {
    {
      std::lock_guard<std::mutex> lk(m);
      if (rc != 0 && !failed) { failed = true; err = StepError(); }  // the first error is the cause; later ones are its echoes
      done[g]++;
      busy[g] = 0;
    }
    cv_work.notify_all();
    cv_feed.notify_all();
  }

  // The feeder: queues the images of (group, step) pairs on the farm's one copy stream, always for the group whose next step is
  // the earliest among those with a free ring slot (step s may go into its slot once step s - kRingSlots is complete).
  void RunFeeder() {
    const size_t fb = static_cast<size_t>(w) * h;
    const int total = G * Bg;
    std::vector<void *> dst(Bg);
    std::deque<int> in_flight;  // slots whose transfer has been queued and not yet seen complete, oldest first
    constexpr int feed_in_flight = 2;  // at most two transfers queued on the device (one: the link idles between them; more: no gain)
    for (;;) {
      int g = -1, s = 0;
      {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          if (failed) return;
          bool left = false;
          g = -1;
          for (int k = 0; k < G; k++) {
            if (issued[k] >= n_steps) continue;
            left = true;
            if (issued[k] - done[k] < kRingSlots && (g < 0 || issued[k] < issued[g])) g = k;
          }
          if (!left) return;
          if (g >= 0) break;
          const auto tw = std::chrono::steady_clock::now();
          cv_feed.wait(lk);
          feed_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
        }
        s = issued[g];
      }
      // At most kFeedInFlight transfers are queued on the device at any time.  Every queued transfer is handed to a DMA engine at
      // once and sits there waiting for its predecessor; with a ring's worth of them queued (48) every engine of the GPU holds one,
      // and the small DMA copies of the compute streams (FilterCorners records, 1 MB) wait behind 79-MB transfers for tens of ms.
      while (static_cast<int>(in_flight.size()) >= feed_in_flight) {
        const int a = sdvl_feed_slot_arrived(feed, in_flight.front());
        if (a < 0) {
          std::lock_guard<std::mutex> lk(m);
          if (!failed) { failed = true; err = std::string("input ring: ") + sdvl_feed_last_error(feed); }
          cv_feed.notify_all();
          return;
        }
        if (a == 1) { in_flight.pop_front(); continue; }
        const auto tw = std::chrono::steady_clock::now();
        timespec ts{0, 50000};
        nanosleep(&ts, nullptr);
        feed_throttle_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
      }
      const auto tc = std::chrono::steady_clock::now();
      const double tm0 = Since();
      uint8_t *base = static_cast<uint8_t *>(ring[g]) + static_cast<size_t>(s % kRingSlots) * Bg * fb;
      for (int i = 0; i < Bg; i++) dst[i] = base + static_cast<size_t>(i) * fb;
      const size_t o = static_cast<size_t>(s) * total + static_cast<size_t>(g) * Bg;
      const int rc = sdvl_feed_images(feed, g * kRingSlots + s % kRingSlots, Bg, reinterpret_cast<const uint8_t *const *>(dev_frames + o), stride, w, h,
                                      dst.data());
      feed_call_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
      feed_calls++;
      in_flight.push_back(g * kRingSlots + s % kRingSlots);
      if (!step_marks.empty()) feed_marks.push_back(FeedMark{g, s, tm0, Since()});
      {
        std::lock_guard<std::mutex> lk(m);
        if (rc != SDVL_OK && !failed) { failed = true; err = std::string("input ring: ") + sdvl_feed_last_error(feed); }
        issued[g]++;
      }
      cv_feed.notify_all();
      cv_work.notify_all();
    }
  }

  // A worker repeatedly takes the idle group that is furthest behind and executes its next step, so a descheduled or
  // throttled thread delays one group-step, not a whole group.
  // With one worker per group every worker owns its group for the whole run: the group's host objects stay in that
  // thread's caches and allocator arena (measured +8 % over handing group-steps out dynamically).  With fewer workers
  // than groups a worker repeatedly takes the idle group that is furthest behind, so that a descheduled or throttled
  // thread delays one group-step, not a whole group.
  void RunShare(int worker, int n_workers) {
    if (n_workers == G) {
      for (int s = 0; s < n_steps; s++) {
        {
          std::lock_guard<std::mutex> lk(m);
          if (failed) return;
        }
        const bool marks_here = !step_marks.empty() && !(host_input && input_ring && feed);
        const double tb = marks_here ? Since() : 0.0;
        const int rc = StepGroup(worker);
        if (marks_here) step_marks[worker].push_back(StepMark{worker, s, tb, tb, Since(), 0, {0}, 0});
        {
          std::lock_guard<std::mutex> lk(m);
          done[worker]++;
          if (rc != 0 && !failed) { failed = true; err = StepError(); }
        }
        cv_feed.notify_all();
        if (rc != 0) return;
      }
      return;
    }
    for (;;) {
      const int g = TakeGroup(true);
      if (g < 0) return;
      FinishGroup(g, StepGroup(g));
    }
  }

  void RunShareFibers(int worker, int n_workers);
};

// ---- cooperative group-steps -------------------------------------------------------------------------------------
// A group-step spends ~40 % of its time blocked on its own stream (alignment, search, pose, filter results).  With
// fibers_per_worker > 1 a worker thread runs several group-steps on separate stacks (ucontext); the library's wait hook
// (sdvl_ctx_set_wait_hook) switches to another fiber whenever a step would sleep, and the thread only sleeps — a 25 us
// nanosleep between polls, no spinning: the CPU quota is the scarce resource — when every fiber is waiting for the GPU.
namespace {
// a fiber's stack: 1 MB with an inaccessible page below it, so that an overflow (deep recursion in the mapper or the HIP
// runtime) faults instead of silently overwriting the heap
struct FiberStack {
  static constexpr size_t kBytes = 1 << 20, kGuard = 4096;
  char *base = nullptr;
  FiberStack() {
    void *p = mmap(nullptr, kBytes + kGuard, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (p == MAP_FAILED) throw std::bad_alloc();
    base = static_cast<char *>(p);
    mprotect(base, kGuard, PROT_NONE);
  }
  ~FiberStack() { if (base) munmap(base, kBytes + kGuard); }
  FiberStack(const FiberStack &) = delete;
  FiberStack &operator=(const FiberStack &) = delete;
  char *data() const { return base + kGuard; }
  size_t size() const { return kBytes; }
};

struct Fiber {
  ucontext_t ctx;
  FiberStack stack;
  Farm *farm = nullptr;
  int group = -1;      // -1 = free
  int rc = 0;
  bool finished = true, whole_run = false;
  sdvl_ctx *waiting_on = nullptr;
  // per-thread state of the host layer, saved across switches
  Device *tls_device = nullptr;
  StageTimes *tls_stage = nullptr;
};
struct FiberScheduler {
  ucontext_t main;
  Fiber *current = nullptr;
};
thread_local FiberScheduler *g_sched = nullptr;

void FiberWaitHook(void *, sdvl_ctx *ctx) {
  FiberScheduler *s = g_sched;
  if (!s || !s->current) {  // not on a fiber (plain batch use): sleep like the default wait
    sdvl_ctx_wait_block(ctx);
    return;
  }
  Fiber *f = s->current;
  f->waiting_on = ctx;
  f->tls_device = Device::Current();
  f->tls_stage = StageTimes::Active();
  swapcontext(&f->ctx, &s->main);
  Device::SetCurrent(f->tls_device);
  StageTimes::Active() = f->tls_stage;
  f->waiting_on = nullptr;
}

void FiberMain(unsigned lo, unsigned hi) {
  Fiber *f = reinterpret_cast<Fiber *>((static_cast<uintptr_t>(hi) << 32) | lo);
  if (f->whole_run) {  // the fiber owns its group: all steps, back to back
    f->rc = 0;
    for (int s = 0; s < f->farm->n_steps && f->rc == 0; s++) {
      f->rc = f->farm->StepGroup(f->group);
      {
        std::lock_guard<std::mutex> lk(f->farm->m);
        f->farm->done[f->group]++;
      }
      f->farm->cv_feed.notify_all();
    }
  } else {
    f->rc = f->farm->StepGroup(f->group);
  }
  f->finished = true;
  // returning ends the context: uc_link takes the thread back to the scheduler
}
}  // namespace

void Farm::RunShareFibers(int worker, int n_workers) {
  FiberScheduler sched;
  g_sched = &sched;
  prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);  // a 25 us nanosleep should cost ~30 us, not 25 + the default 50 us slack
  std::vector<Fiber> fibers(fibers_per_worker);
  for (Fiber &f : fibers) f.farm = this;
  int idle_polls = 0;
  auto start = [&](Fiber &f, int g, bool whole_run) {
    f.group = g;
    f.finished = false;
    f.whole_run = whole_run;
    f.waiting_on = nullptr;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = &sched.main;
    const uintptr_t p = reinterpret_cast<uintptr_t>(&f);
    makecontext(&f.ctx, reinterpret_cast<void (*)()>(FiberMain), 2, static_cast<unsigned>(p & 0xFFFFFFFFu), static_cast<unsigned>(p >> 32));
  };
  // as many groups as fibers in the farm: every fiber owns one group for the whole run (same reason as in RunShare)
  const bool owned = n_workers * fibers_per_worker == G;
  if (owned)
    for (int i = 0; i < fibers_per_worker; i++) start(fibers[i], worker * fibers_per_worker + i, true);
  bool no_more = owned;
  for (;;) {
    int active = 0;
    for (Fiber &f : fibers) {  // give every free fiber a group-step
      if (f.group < 0 && !no_more) {
        const int g = TakeGroup(false);
        if (g == -1) no_more = true;
        if (g >= 0) start(f, g, false);
      }
      if (f.group >= 0) active++;
    }
    if (active == 0) {
      if (no_more) break;
      const int g = TakeGroup(true);  // everything left is busy on other workers: wait for one (or for the end)
      if (g < 0) break;
      FinishGroup(g, StepGroup(g));   // no fiber needed: nothing else to overlap with right now
      continue;
    }
    bool progressed = false;
    for (Fiber &f : fibers) {
      if (f.group < 0) continue;
      if (f.waiting_on && !sdvl_ctx_wait_done(f.waiting_on)) continue;  // its stream is still busy
      sched.current = &f;
      swapcontext(&sched.main, &f.ctx);
      sched.current = nullptr;
      progressed = true;
      if (f.finished) {
        if (f.whole_run) {
          if (f.rc != 0) { std::lock_guard<std::mutex> lk(m); if (!failed) { failed = true; err = StepError(); } }
        } else {
          FinishGroup(f.group, f.rc);
        }
        f.group = -1;
      }
    }
    if (!progressed) {  // every fiber waits for the GPU: sleep a poll interval (any of their streams may finish first)
      const struct timespec ts = {0, 25000};
      nanosleep(&ts, nullptr);
      if (++idle_polls % 2048 == 0) {  // ~every 100 ms without progress: a faulted stream never reaches its mark
        for (Fiber &f : fibers)
          if (f.group >= 0 && f.waiting_on && sdvl_ctx_health(f.waiting_on) != SDVL_OK) {
            std::lock_guard<std::mutex> lk(m);
            if (!failed) {
              failed = true;
              err = std::string("GPU fault while a group-step was waiting: ") + sdvl_last_error(f.waiting_on);
            }
          }
        bool stop;
        { std::lock_guard<std::mutex> lk(m); stop = failed; }
        if (stop) break;  // the fibers' stacks are abandoned: the run is over
      }
    } else {
      idle_polls = 0;
    }
  }
  g_sched = nullptr;
}

void *sdvlh_farm_create(int gpu, int G, int Bg, int w, int h, const double *cam4, const double *plane4, const double *first_poses7,
                        int host_threads_per_group) {
  // G host threads allocate and free small objects all the time while their keyframes keep ~100 KB each for good: with the
  // default settings every per-thread malloc arena grows in 128 KB steps (one mprotect each, all serialised on the process's
  // address-space lock together with the page faults) and gives memory back as eagerly.  Grow in 32 MB steps, never trim.
  mallopt(M_TOP_PAD, 32 << 20);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_MMAP_THRESHOLD, 16 << 20);
  Farm *f = new Farm();
  f->gpu = gpu; f->G = G; f->Bg = Bg; f->w = w; f->h = h;
  f->ring.assign(G, nullptr);
  f->ring_ticket.assign(static_cast<size_t>(G) * f->kRingSlots, -1);
  for (int g = 0; g < G; g++) {
    void *d = sdvlh_device_create(gpu);
    if (!d) { sdvlh_farm_destroy(f); return nullptr; }
    f->devices.push_back(d);
    void *b = sdvlh_batch_create(d, Bg, w, h, cam4, plane4, first_poses7 + static_cast<size_t>(7) * g * Bg, host_threads_per_group);
    if (!b) { sdvlh_farm_destroy(f); return nullptr; }
    f->batches.push_back(b);
  }
  return f;
}

// n > 1: every worker thread interleaves n group-steps and switches between them at GPU waits (0 / 1 = one at a time)
void sdvlh_farm_set_fibers(void *fp, int n) {
  Farm *f = static_cast<Farm *>(fp);
  f->fibers_per_worker = n > 1 ? n : 1;
  for (void *d : f->devices)
    sdvl_ctx_set_wait_hook(static_cast<Device *>(d)->ctx(), f->fibers_per_worker > 1 ? FiberWaitHook : nullptr, nullptr);
}

// the frame pointers handed to sdvlh_farm_run are host pointers (SDVL::HandleFrame(const cv::Mat&) takes a host image,
// sdvl.cc:55-59): every group uploads its frames on its own stream inside the step
void sdvlh_farm_set_host_input(void *fp, int on) {
  Farm *f = static_cast<Farm *>(fp);
  f->host_input = on != 0;
  if (!f->host_input || !f->input_ring) return;
  // the input rings are allocated here, not inside the first host-fed step (hipMalloc synchronises the device)
  const size_t fb = static_cast<size_t>(f->w) * f->h;
  for (int g = 0; g < f->G; g++)
    if (!f->ring[g]) {
      sdvl_ctx *ctx = static_cast<sdvl_ctx *>(sdvlh_device_ctx(f->devices[g]));
      if (sdvl_device_malloc(ctx, static_cast<int64_t>(f->kRingSlots * fb * f->Bg), &f->ring[g]) != SDVL_OK) f->ring[g] = nullptr;  // StepGroup retries and reports
    }
}
// (the feed itself is created by the first host-fed run)
// the input ring of host-fed runs on (default) / off (every step uploads its own frames on its own stream before it computes)
void sdvlh_farm_set_input_ring(void *fp, int on) { static_cast<Farm *>(fp)->input_ring = on != 0; }

// the feeder thread of the last host-fed run: seconds inside sdvl_feed_images, seconds waiting for a free ring slot, transfers
// + seconds the workers waited for their step's transfer to be issued and to arrive (summed over groups), group-steps that began
// before their images had arrived, seconds the feeder waited for its own earlier transfers (at most two are queued on the device)
void sdvlh_farm_feed_stats(void *fp, double *out6) {
  Farm *f = static_cast<Farm *>(fp);
  out6[0] = f->feed_call_s;
  out6[1] = f->feed_wait_s;
  out6[2] = static_cast<double>(f->feed_calls);
  out6[3] = f->work_wait_s + f->arrive_wait_s;
  out6[4] = static_cast<double>(f->late_acquires);
  out6[5] = f->feed_throttle_s;
}

void sdvlh_farm_destroy(void *fp) {
  Farm *f = static_cast<Farm *>(fp);
  if (!f) return;
  for (void *b : f->batches) sdvlh_batch_destroy(b);
  if (f->feed) sdvl_feed_destroy(f->feed);  // waits for its stream: nothing writes the rings after this
  for (size_t g = 0; g < f->ring.size() && g < f->devices.size(); g++)
    if (f->ring[g]) sdvl_device_free(static_cast<sdvl_ctx *>(sdvlh_device_ctx(f->devices[g])), f->ring[g]);
  for (void *d : f->devices) sdvlh_device_destroy(d);
  delete f;
}

// pool `frames_per_group` free HBM frames in every group (keyframe budget), so the run allocates nothing
int sdvlh_farm_reserve(void *fp, int frames_per_group) {
  Farm *f = static_cast<Farm *>(fp);
  try {
    for (void *d : f->devices) static_cast<Device *>(d)->Reserve(f->w, f->h, Config::PyramidLevels(), frames_per_group);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

void *sdvlh_farm_ctx(void *fp, int g) { return sdvlh_device_ctx(static_cast<Farm *>(fp)->devices[g]); }
void *sdvlh_farm_batch(void *fp, int g) { return static_cast<Farm *>(fp)->batches[g]; }


// ---- SDVL_PROFILE=1: a sampling profile of the host threads during sdvlh_farm_run (the boxes have no perf): SIGPROF
// every millisecond of process CPU time, the handler records the call stack, the run prints self / inclusive shares.
namespace {
constexpr int kProfDepth = 24, kProfMax = 400000;
struct ProfSample { int n; void *pc[kProfDepth]; };
ProfSample *g_prof = nullptr;
std::atomic<int> g_prof_n{0};
void ProfHandler(int) {
  const int i = g_prof_n.fetch_add(1);
  if (i >= kProfMax) return;
  g_prof[i].n = backtrace(g_prof[i].pc, kProfDepth);
}
std::string ProfName(void *pc) {
  Dl_info info;
  if (dladdr(pc, &info) && info.dli_sname) {
    int st = 0;
    char *d = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &st);
    std::string r = (st == 0 && d) ? d : info.dli_sname;
    free(d);
    if (r.size() > 90) r.resize(90);
    return r;
  }
  if (dladdr(pc, &info) && info.dli_fname) {
    std::string f = info.dli_fname;
    const size_t k = f.rfind('/');
    return "[" + (k == std::string::npos ? f : f.substr(k + 1)) + "]";
  }
  return "[?]";
}
void ProfStart() {
  if (!g_prof) g_prof = new ProfSample[kProfMax];
  g_prof_n = 0;
  void *warm[4];
  backtrace(warm, 4);  // loads libgcc outside the handler
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = ProfHandler;
  sa.sa_flags = SA_RESTART;
  sigaction(SIGPROF, &sa, nullptr);
}
// one CPU-time timer per worker thread (a process-wide ITIMER_PROF fires once per scheduler tick for the whole process)
timer_t ProfThreadStart() {
  struct sigevent sev;
  memset(&sev, 0, sizeof(sev));
  sev.sigev_notify = SIGEV_THREAD_ID;
  sev.sigev_signo = SIGPROF;
  sev._sigev_un._tid = static_cast<pid_t>(syscall(SYS_gettid));
  timer_t t = nullptr;
  if (timer_create(CLOCK_THREAD_CPUTIME_ID, &sev, &t) != 0) return nullptr;
  struct itimerspec its = {{0, 1000000}, {0, 1000000}};
  timer_settime(t, 0, &its, nullptr);
  return t;
}
void ProfThreadStop(timer_t t) {
  if (t) timer_delete(t);
}
void ProfStop() {
  const int n = std::min(g_prof_n.load(), kProfMax);
  std::map<std::string, int> self, incl, own;
  for (int i = 0; i < n; i++) {
    std::set<std::string> seen;
    bool mine = false;
    for (int d = 2; d < g_prof[i].n; d++) {  // 0 = handler, 1 = signal trampoline
      const std::string name = ProfName(g_prof[i].pc[d]);
      if (d == 2) self[name]++;
      if (seen.insert(name).second) incl[name]++;
      if (!mine && (name.compare(0, 4, "sdvl") == 0 || name.compare(0, 5, "void ") == 0 || name.compare(0, 5, "std::") == 0)) {
        // innermost frame of this repo's code (or a libstdc++ template instantiated in it): everything below it, HIP / HSA
        // / libc included, is charged to it
        if (name.compare(0, 4, "sdvl") == 0) { own[name]++; mine = true; }
      }
    }
    if (!mine) own[g_prof[i].n > 2 ? "(other threads) " + ProfName(g_prof[i].pc[g_prof[i].n - 1]) : "(empty)"]++;
  }
  auto dump = [&](const char *title, std::map<std::string, int> &m, int top) {
    std::vector<std::pair<int, std::string>> v;
    for (auto &kv : m) v.push_back({kv.second, kv.first});
    std::sort(v.rbegin(), v.rend());
    fprintf(stderr, "---- %s (%d samples of 1 ms CPU)\n", title, n);
    for (int i = 0; i < top && i < static_cast<int>(v.size()); i++) fprintf(stderr, "%6.2f%%  %s\n", 100.0 * v[i].first / std::max(1, n), v[i].second.c_str());
  };
  dump("charged to the innermost sdvl function (callees in HIP / HSA / libc included)", own, 40);
  dump("self", self, 25);
  dump("inclusive", incl, 30);
}
}  // namespace

// Runs n_steps steps with `workers` host threads (0 = one per group).
// dev_frames[(step * G*Bg) + g*Bg + i] = device pointer of the frame of sequence (g, i) at that step (row stride =
// `stride`); out[(step * G*Bg) + g*Bg + i] receives its stats.  Returns 0, or -1 with sdvlh_last_error.
// The threads are created per run on purpose: measured on the MI355X hosts, freshly forked threads are spread over idle
// cores by the scheduler, while long-lived workers woken through a condition variable are pulled next to their waker
// and run ~6 % slower (80.0k vs 75.6k tracked frames/s at 120 steps); creating 16 threads costs well under a millisecond.
int sdvlh_farm_run(void *fp, int n_steps, const void *const *dev_frames, int stride, sdvlh_frame_stats *out, int workers) {
  Farm *f = static_cast<Farm *>(fp);
  const int W = workers > 0 ? (workers < f->G ? workers : f->G) : f->G;
  {
    std::lock_guard<std::mutex> lk(f->m);
    f->n_steps = n_steps; f->stride = stride; f->dev_frames = dev_frames; f->out = out;
    f->done.assign(f->G, 0);
    f->busy.assign(f->G, 0);
    f->issued.assign(f->G, 0);
    f->feed_call_s = f->feed_wait_s = f->work_wait_s = f->arrive_wait_s = f->feed_throttle_s = 0.0;
    f->late_acquires = 0;
    f->feed_calls = 0;
    f->failed = false;
    f->step_marks.clear();
    f->feed_marks.clear();
    if (std::getenv("SDVL_FARM_TIMELINE")) f->step_marks.assign(f->G, {});
    f->run_t0 = std::chrono::steady_clock::now();
  }
  std::thread feeder;
  if (f->host_input && f->input_ring) {
    bool rings = true;
    for (int g = 0; g < f->G; g++) rings = rings && f->ring[g] != nullptr;
    if (!rings) { g_err = "input ring: the rings were not allocated (sdvlh_farm_set_host_input)"; return -1; }
    if (!f->feed && sdvl_feed_create(f->gpu, f->G * f->kRingSlots, &f->feed) != SDVL_OK) { g_err = "input ring: sdvl_feed_create failed"; return -1; }
    feeder = std::thread([f] { f->RunFeeder(); });
  }
  static const bool profile = std::getenv("SDVL_PROFILE") != nullptr;
  if (profile) ProfStart();
  std::vector<std::thread> threads;
  for (int w = 0; w < W; w++)
    threads.emplace_back([f, w, W] {
      timer_t pt = profile ? ProfThreadStart() : nullptr;
      if (f->fibers_per_worker > 1) f->RunShareFibers(w, W);
      else f->RunShare(w, W);
      ProfThreadStop(pt);
    });
  for (auto &t : threads) t.join();
  if (feeder.joinable()) {
    f->cv_feed.notify_all();
    feeder.join();
  }
  if (profile) ProfStop();
  if (!f->step_marks.empty()) {
    if (FILE *fp = std::fopen(std::getenv("SDVL_FARM_TIMELINE"), "a")) {
      std::fprintf(fp, "# run: %d groups x %d steps\n", f->G, n_steps);
      for (const auto &v : f->step_marks)
        for (const auto &k : v) {
          std::fprintf(fp, "step %d %d %.6f %.6f %.6f %d kf %d stages", k.g, k.s, k.t_enter, k.t_issued, k.t_end, k.late, k.kf);
          for (int i = 0; i < 12; i++) std::fprintf(fp, " %.4f", k.st[i]);
          std::fprintf(fp, "\n");
        }
      for (const auto &k : f->feed_marks) std::fprintf(fp, "feed %d %d %.6f %.6f\n", k.g, k.s, k.t0, k.t1);
      std::fclose(fp);
    }
  }
  if (f->failed) { g_err = f->err; return -1; }
  return 0;
}

}  // extern "C"
