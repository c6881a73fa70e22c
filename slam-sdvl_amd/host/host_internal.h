// host_internal.h — what the two translation units of the host layer share besides sdvl_host.h (not installed):
//   frontend.cc    Device, Frame, Feature, FastDetector, ORBDetector, ImageAlign, Matcher, FeatureAlign — the hot path's classes
//   standalone.cc  Camera, Point, Map / PlaneMap, SDVL, SDVLBatch — what the reference's own sdvl.cc / map.cc / point.cc /
//                  camera.cc provide in its tree (INTEGRATION.md route A)
#ifndef SDVL_HOST_INTERNAL_H_
#define SDVL_HOST_INTERNAL_H_

#include <chrono>

#include "frontend.h"

namespace sdvl {

// time of a host stage of the batch driver (StageTimes::Active() is thread-local; no clock runs when it is null)
struct StageClock {
  int id;
  std::chrono::steady_clock::time_point t0;
  explicit StageClock(int i) : id(i), t0(std::chrono::steady_clock::now()) {}
  ~StageClock() {
    if (StageTimes *s = StageTimes::Active()) s->t[id] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
};

// the Config getters the kernels' parameter blocks are filled from (frontend.cc)
sdvl_align_params AlignParams(bool fast);
sdvl_search_params SearchParams();

}  // namespace sdvl

#endif  // SDVL_HOST_INTERNAL_H_
