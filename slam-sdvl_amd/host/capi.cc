// capi.cc — flat C entry points over the host layer (sdvl_host.h) for the Python harness (tests, smoke, bench):
// B independent SDVL trackers on one MI355X stepping together through sdvl::SDVLBatch.
#include <cstring>
#include <memory>
#include <string>
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
  int w, h;
  std::string err;
};
thread_local std::string g_err;
}  // namespace

extern "C" {

struct sdvlh_frame_stats {
  int state, quality, matches, attempts, inliers, outliers, n_corners, align_meas, keyframe, relocalized;
  double pose[7];
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
      b->maps.emplace_back(new PlaneMap(Vector3d(plane4[0], plane4[1], plane4[2]), plane4[3]));
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

void sdvlh_batch_destroy(void *bp) {
  Batch *b = static_cast<Batch *>(bp);
  if (!b) return;
  Device::SetCurrent(b->dev);
  b->batch.reset();
  b->trackers.clear();
  b->maps.clear();
  delete b;
}

static int step(Batch *b, const std::vector<Image> &imgs, sdvlh_frame_stats *out) {
  try {
    std::vector<FrameStats> st(imgs.size());
    b->batch->HandleFrames(imgs, st.data());
    for (size_t i = 0; i < imgs.size(); i++) {
      const FrameStats &s = st[i];
      sdvlh_frame_stats &o = out[i];
      o.state = s.state; o.quality = s.quality; o.matches = s.matches; o.attempts = s.attempts; o.inliers = s.inliers;
      o.outliers = s.outliers; o.n_corners = s.n_corners; o.align_meas = s.align_meas; o.keyframe = s.keyframe;
      o.relocalized = s.relocalized;
      std::memcpy(o.pose, s.pose, sizeof(o.pose));
    }
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// imgs: B host pointers (row stride `stride`)
int sdvlh_batch_step_host(void *bp, const uint8_t *const *imgs, int stride, sdvlh_frame_stats *out) {
  Batch *b = static_cast<Batch *>(bp);
  std::vector<Image> v;
  for (size_t i = 0; i < b->trackers.size(); i++) v.push_back(Image::Wrap(imgs[i], b->w, b->h, stride));
  return step(b, v, out);
}

// imgs: B device pointers (frames already in HBM)
int sdvlh_batch_step_device(void *bp, const void *const *dev_imgs, int stride, sdvlh_frame_stats *out) {
  Batch *b = static_cast<Batch *>(bp);
  std::vector<Image> v;
  for (size_t i = 0; i < b->trackers.size(); i++) v.push_back(Image::WrapDevice(dev_imgs[i], b->w, b->h, stride));
  return step(b, v, out);
}

}  // extern "C"
