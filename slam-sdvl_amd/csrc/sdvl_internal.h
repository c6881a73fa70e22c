// sdvl_internal.h — context / frame records behind the C-ABI of include/sdvl_hip.h (not installed).
#ifndef SDVL_INTERNAL_H_
#define SDVL_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/sdvl_hip.h"

// Device view of one frame: everything a kernel needs, passed by value inside job records.
struct FrameView {
  uint8_t *level[SDVL_MAX_LEVELS];  // level[l] row-major, row stride = lw[l]
  int32_t lw[SDVL_MAX_LEVELS], lh[SDVL_MAX_LEVELS];
  int32_t levels;
  int32_t n_corners;    // host copy of the corner count, -1 = only known on the device (after sdvl_detect_corners)
  int32_t *corner_hdr;  // device: [4] = {count, 0, 0, 0}, immediately followed by `corners`
  int32_t *corners;     // device: [SDVL_MAX_CORNERS][4]: x, y, level, pad
  uint8_t *desc;        // device: [SDVL_MAX_CORNERS][32]
};

struct sdvl_frame {
  FrameView v;
  uint8_t *base;      // one allocation: pyramid | corners | descriptors | cell lists
  size_t bytes;
  int width, height;
  uint32_t *cell_kps;    // [total_cells][SDVL_CELL_KP_CAP] packed (x | y<<12 | score<<24), level coordinates
  int32_t *cell_counts;  // [total_cells]
  int32_t *level_corners;  // [4][SDVL_MAX_CORNERS][4] per-level output segments of the selection kernel
  int32_t *level_counts;   // [4]
  int max_cells;
  int desc_valid;
};

struct KernelTimer {
  std::string name;
  double ms = 0.0;
  int64_t launches = 0;
};

struct sdvl_ctx {
  int device;
  hipStream_t stream;
  std::string err;
  // pinned + device staging, grown on demand
  void *h_stage = nullptr; size_t h_stage_bytes = 0;
  void *d_stage = nullptr; size_t d_stage_bytes = 0;
  void *h_out = nullptr;   size_t h_out_bytes = 0;
  void *d_out = nullptr;   size_t d_out_bytes = 0;
  void *d_work = nullptr;  size_t d_work_bytes = 0;
  // corner counts of the last sdvl_detect_corners batch, written by the pack kernel: one D2H serves all frames
  void *d_counts = nullptr; size_t d_counts_bytes = 0;
  std::vector<sdvl_frame *> detect_frames;
  hipEvent_t wait_event = nullptr;  // created with hipEventBlockingSync | hipEventDisableTiming
  // per-kernel timing (HIP events on `stream`)
  int timing = 0;
  std::vector<KernelTimer> timers;
  struct Pending { int timer; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
};

#define SDVL_HIP_CHECK(ctx, expr)                                                            \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
      return SDVL_ERR_HIP;                                                                   \
    }                                                                                        \
  } while (0)

#define SDVL_REQUIRE(ctx, cond, msg)          \
  do {                                        \
    if (!(cond)) {                            \
      (ctx)->err = std::string(msg);          \
      return SDVL_ERR_INVALID;                \
    }                                         \
  } while (0)

int sdvl_ensure(sdvl_ctx *ctx, void **p, size_t *cur, size_t need, bool pinned);
// wait for everything queued on ctx->stream WITHOUT spinning: hipEventBlockingSync event + hipEventSynchronize.
hipError_t sdvl_stream_wait(sdvl_ctx *ctx);
// host copy of a frame's corner count; fetches it (blocking) when only the device knows it
int sdvl_frame_count_host(sdvl_ctx *ctx, sdvl_frame *f, int *n);
int sdvl_timer_begin(sdvl_ctx *ctx, const char *name);  // returns pending index or -1
void sdvl_timer_end(sdvl_ctx *ctx, int pending);

struct ScopedKernelTimer {
  sdvl_ctx *ctx;
  int idx;
  ScopedKernelTimer(sdvl_ctx *c, const char *name) : ctx(c), idx(sdvl_timer_begin(c, name)) {}
  ~ScopedKernelTimer() { sdvl_timer_end(ctx, idx); }
};

#endif  // SDVL_INTERNAL_H_
