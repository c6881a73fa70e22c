// sdvl_ctx.hip — context, staging buffers, per-kernel HIP-event timing, frame allocation and transfers.
#include <sys/prctl.h>
#include <time.h>

#include "sdvl_internal.h"
#include <chrono>
#include "sdvl_search_types.h"

// HIP's current device is per THREAD and starts at 0: every host thread that works for a context of GPU n (farm workers,
// fibers, pool helpers, a user's mapper thread) must select that GPU before it allocates or launches, or its hipMalloc /
// hipHostMalloc land on GPU 0.  Not cached: other code on the thread (torch) may select devices too; hipSetDevice is a
// thread-local store and is only called where memory is allocated and once per step of a host thread.
hipError_t sdvl_bind_device(const sdvl_ctx *ctx) { return hipSetDevice(ctx->device); }

int sdvl_ensure(sdvl_ctx *ctx, void **p, size_t *cur, size_t need, bool pinned) {
  if (*cur >= need && *p) return SDVL_OK;
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  size_t want = need + need / 2 + 4096;
  if (*p) {
    SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
    if (pinned) SDVL_HIP_CHECK(ctx, hipHostFree(*p));
    else SDVL_HIP_CHECK(ctx, hipFree(*p));
    *p = nullptr;
    *cur = 0;
  }
  if (pinned) SDVL_HIP_CHECK(ctx, hipHostMalloc(p, want, hipHostMallocDefault));
  else SDVL_HIP_CHECK(ctx, hipMalloc(p, want));
  *cur = want;
  return SDVL_OK;
}

int sdvl_registry_reserve(sdvl_ctx *ctx, int extra) {
  const int need = ctx->registry_next + extra - static_cast<int>(ctx->registry_free.size());
  if (need <= ctx->registry_cap) return SDVL_OK;
  int cap = ctx->registry_cap ? ctx->registry_cap : 1024;
  while (cap < need) cap *= 2;
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  void *fresh = nullptr;
  SDVL_HIP_CHECK(ctx, hipMalloc(&fresh, sizeof(SearchFramePose) * static_cast<size_t>(cap)));
  if (ctx->d_registry) {
    SDVL_HIP_CHECK(ctx, hipMemcpyAsync(fresh, ctx->d_registry, sizeof(SearchFramePose) * static_cast<size_t>(ctx->registry_cap), hipMemcpyDeviceToDevice,
                                       ctx->stream));
    SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));  // nothing queued may still read the old table
    SDVL_HIP_CHECK(ctx, hipFree(ctx->d_registry));
  }
  ctx->d_registry = fresh;
  ctx->registry_cap = cap;
  return SDVL_OK;
}

namespace {
typedef unsigned int push_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stage_push_kernel(const push_u32x4 *__restrict__ src, push_u32x4 *__restrict__ dst, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = __builtin_nontemporal_load(src + i);
}
}  // namespace

hipError_t sdvl_push(sdvl_ctx *ctx, void *dst_dev, const void *src_staged, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  const uint8_t *s8 = static_cast<const uint8_t *>(src_staged), *ring = static_cast<const uint8_t *>(ctx->h_stage);
  // inside the ring allocations are 256-byte granules on both sides: copying whole 16-byte units never leaves them
  const bool staged = ring && s8 >= ring && s8 + bytes <= ring + ctx->h_stage_bytes && ((reinterpret_cast<uintptr_t>(s8) | reinterpret_cast<uintptr_t>(dst_dev)) & 15u) == 0;
  if (!staged || bytes > (static_cast<size_t>(64) << 20)) return hipMemcpyAsync(dst_dev, src_staged, bytes, hipMemcpyHostToDevice, ctx->stream);
  const int n16 = static_cast<int>((bytes + 15) >> 4);
  const int blocks = n16 <= 256 ? 1 : (n16 >= 256 * 64 ? 64 : (n16 + 255) / 256);
  hipLaunchKernelGGL(stage_push_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const push_u32x4 *>(s8), static_cast<push_u32x4 *>(dst_dev), n16);
  return hipGetLastError();
}

// The other direction: results that a kernel left in device memory go to a pinned host buffer through a kernel's stores (posted PCIe
// writes) instead of a DMA command — the tracked step's last sizeable DMA copy (the FilterCorners records, ~1 MB per group-step) was
// what waited behind parked image transfers.  Both pointers 16-byte aligned, whole 16-byte units are copied.
namespace {
__global__ __launch_bounds__(256) void stage_pull_kernel(const push_u32x4 *__restrict__ src, push_u32x4 *__restrict__ dst_host, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) __builtin_nontemporal_store(src[i], dst_host + i);
}
}  // namespace

hipError_t sdvl_pull(sdvl_ctx *ctx, void *dst_host_pinned, const void *src_dev, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  const bool ok = ((reinterpret_cast<uintptr_t>(dst_host_pinned) | reinterpret_cast<uintptr_t>(src_dev)) & 15u) == 0;
  if (!ok || bytes > (static_cast<size_t>(64) << 20)) return hipMemcpyAsync(dst_host_pinned, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
  const int n16 = static_cast<int>((bytes + 15) >> 4);
  const int blocks = n16 <= 256 ? 1 : (n16 >= 256 * 64 ? 64 : (n16 + 255) / 256);
  hipLaunchKernelGGL(stage_pull_kernel, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const push_u32x4 *>(src_dev), static_cast<push_u32x4 *>(dst_host_pinned), n16);
  return hipGetLastError();
}

int sdvl_stage_alloc(sdvl_ctx *ctx, size_t bytes, void **h, void **d) {
  const size_t need = (bytes + 255) / 256 * 256;
  const size_t cap = ctx->h_stage_bytes < ctx->d_stage_bytes ? ctx->h_stage_bytes : ctx->d_stage_bytes;
  if (!ctx->h_stage || !ctx->d_stage || ctx->stage_off + need > cap) {
    SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));  // everything staged so far has been consumed; the ring restarts
    if (need > cap || !ctx->h_stage || !ctx->d_stage) {
      const size_t want = need * 2 + (static_cast<size_t>(4) << 20);
      int rc = sdvl_ensure(ctx, &ctx->h_stage, &ctx->h_stage_bytes, want, true);
      if (!rc) rc = sdvl_ensure(ctx, &ctx->d_stage, &ctx->d_stage_bytes, want, false);
      if (rc) return rc;
    }
  }
  *h = static_cast<uint8_t *>(ctx->h_stage) + ctx->stage_off;
  *d = static_cast<uint8_t *>(ctx->d_stage) + ctx->stage_off;
  ctx->stage_off += need;
  return SDVL_OK;
}

hipError_t sdvl_mark_record(sdvl_ctx *ctx, int kind, uint32_t *ticket) {
  const uint32_t seq = ++ctx->flag_seq;
  *ticket = seq;
  (void)kind;
  if (!ctx->h_flag) {
    void *p = nullptr;
    hipError_t e = sdvl_bind_device(ctx);
    if (e != hipSuccess) return e;
    e = hipHostMalloc(&p, 64, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    memset(p, 0, 64);
    ctx->h_flag = static_cast<volatile uint32_t *>(p);
  }
  // the stream itself writes the sequence number once everything before it has completed: no event object, and the
  // waiting thread polls a cache line instead of calling into the runtime
  return hipStreamWriteValue32(ctx->stream, const_cast<uint32_t *>(ctx->h_flag), seq, 0);
}

// 1 = the mark has been reached, 0 = not yet, < 0 = the stream reported an error
static int mark_reached(sdvl_ctx *ctx, int kind, uint32_t ticket, hipError_t *err) {
  *err = hipSuccess;
  (void)kind;
  const uint32_t seen = __atomic_load_n(const_cast<const uint32_t *>(ctx->h_flag), __ATOMIC_ACQUIRE);
  return static_cast<int32_t>(seen - ticket) >= 0 ? 1 : 0;
}

// hipEventSynchronize / hipStreamSynchronize busy-wait here even for hipEventBlockingSync events (measured: CPU time = wall
// time inside the wait), and CPU is what the host side is short of.  So: poll and sleep in between; with the timer slack
// of the thread at 1 us a 25 us nanosleep costs ~30 us.  Every ~100 ms the stream is asked whether it is still healthy,
// so that a faulted kernel surfaces as an error instead of a hang.
hipError_t sdvl_mark_wait(sdvl_ctx *ctx, int kind, uint32_t ticket) {
  constexpr int poll_ns = 25000;
  static thread_local bool slack_set = false;
  if (!slack_set) {
    prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    slack_set = true;
  }
  const struct timespec ts = {0, poll_ns > 0 ? poll_ns : 1000};
  hipError_t err = hipSuccess;
  int polls = 0;
  if (ctx->wait_hook) {
    ctx->waiting = 1;
    ctx->waiting_ticket = ticket;
    ctx->waiting_kind = kind;
  }
  // a context that waits for a lone camera's chain (0.2 ms) spins first: a 25-us sleep wakes 30 us late on average half a sleep after the
  // mark, 5 % of such a step; farms (9-ms waits, CPU short) never spin
  const bool spin = ctx->wait_spin_us > 0 && !ctx->wait_hook;
  const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin ? ctx->wait_spin_us : 0);
  for (;;) {
    const int r = mark_reached(ctx, kind, ticket, &err);
    if (r != 0) break;
    if (ctx->wait_hook) ctx->wait_hook(ctx->wait_user, ctx);
    else if (spin && std::chrono::steady_clock::now() < spin_until) __builtin_ia32_pause();
    else nanosleep(&ts, nullptr);
    if (++polls % 4096 == 0) {
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipSuccess && q != hipErrorNotReady) { err = q; break; }
    }
  }
  ctx->waiting = 0;
  return err;
}

hipError_t sdvl_stream_wait(sdvl_ctx *ctx) {
  if (ctx->forked) {  // a whole-context wait inside a fork (the staging ring ran out): both chains must have drained
    hipError_t es = hipStreamSynchronize(ctx->main_stream);
    if (es != hipSuccess) return es;
  }
  uint32_t t = 0;
  hipError_t e = sdvl_mark_record(ctx, SDVL_MARK_STREAM, &t);
  if (e != hipSuccess) return e;
  e = sdvl_mark_wait(ctx, SDVL_MARK_STREAM, t);
  if (e == hipSuccess) {
    ctx->stage_off = 0;
    ctx->wait_gen++;
  }
  return e;
}

extern "C" int sdvl_ctx_set_wait_spin(sdvl_ctx *ctx, int microseconds) {
  if (!ctx || microseconds < 0) return SDVL_ERR_INVALID;
  ctx->wait_spin_us = microseconds;
  return SDVL_OK;
}

extern "C" int sdvl_ctx_set_wait_hook(sdvl_ctx *ctx, void (*hook)(void *user, sdvl_ctx *ctx), void *user) {
  if (!ctx) return SDVL_ERR_INVALID;
  ctx->wait_hook = hook;
  ctx->wait_user = user;
  return SDVL_OK;
}

// 1 = everything queued before the wait in flight has completed, 0 = still running (never blocks)
extern "C" int sdvl_ctx_wait_done(sdvl_ctx *ctx) {
  if (!ctx || !ctx->waiting) return 1;
  hipError_t err;
  return mark_reached(ctx, ctx->waiting_kind, ctx->waiting_ticket, &err) != 0 ? 1 : 0;
}

// 0 while the context's stream is healthy (idle or busy), SDVL_ERR_HIP once it has reported a fault: a scheduler that polls
// sdvl_ctx_wait_done asks this now and then, so that a faulted kernel ends the run instead of hanging it
extern "C" int sdvl_ctx_health(sdvl_ctx *ctx) {
  if (!ctx) return SDVL_ERR_INVALID;
  const hipError_t q = hipStreamQuery(ctx->stream);
  if (q == hipSuccess || q == hipErrorNotReady) return SDVL_OK;
  ctx->err = std::string("stream fault: ") + hipGetErrorString(q);
  return SDVL_ERR_HIP;
}

extern "C" int sdvl_ctx_counters(sdvl_ctx *ctx, int64_t *out4) {
  if (!ctx || !out4) return SDVL_ERR_INVALID;
  for (int i = 0; i < 4; i++) out4[i] = ctx->counters[i];
  out4[2] = ctx->undist_maps_built;
  return SDVL_OK;
}

// sleep (no spinning) until the wait in flight has completed
extern "C" int sdvl_ctx_wait_block(sdvl_ctx *ctx) {
  if (!ctx || !ctx->waiting) return SDVL_OK;
  const struct timespec ts = {0, 25000};
  hipError_t err = hipSuccess;
  while (mark_reached(ctx, ctx->waiting_kind, ctx->waiting_ticket, &err) == 0) nanosleep(&ts, nullptr);
  return err == hipSuccess ? SDVL_OK : SDVL_ERR_HIP;
}

int sdvl_frame_fix_header(sdvl_ctx *ctx, sdvl_frame *f) {
  if (f->hdr_stale) {
    SDVL_HIP_CHECK(ctx, hipMemsetAsync(f->v.corner_hdr, 0, 16, ctx->stream));
    f->hdr_stale = 0;
  }
  return SDVL_OK;
}

int sdvl_frame_count_host(sdvl_ctx *ctx, sdvl_frame *f, int *n) {
  if (f->hdr_stale) f->v.n_corners = 0;
  if (f->v.n_corners < 0) {
    int rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, 64, true);
    if (rc) return rc;
    SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, f->v.corner_hdr, 16, hipMemcpyDeviceToHost, ctx->stream));
    SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
    f->v.n_corners = static_cast<const int32_t *>(ctx->h_out)[0];
  }
  *n = f->v.n_corners;
  return SDVL_OK;
}

bool sdvl_timer_events(sdvl_ctx *ctx, const char *name, hipEvent_t *a, hipEvent_t *b) {
  *a = nullptr;
  *b = nullptr;
  if (!ctx->timing) return false;
  if (!ctx->timing_only.empty() && ctx->timing_only != name) return false;
  int t = -1;
  for (size_t i = 0; i < ctx->timers.size(); i++)
    if (ctx->timers[i].name == name) { t = static_cast<int>(i); break; }
  if (t < 0) {
    KernelTimer kt;
    kt.name = name;
    ctx->timers.push_back(kt);
    t = static_cast<int>(ctx->timers.size()) - 1;
  }
  sdvl_ctx::Pending p;
  p.timer = t;
  for (hipEvent_t *e : {&p.a, &p.b}) {
    if (!ctx->free_events.empty()) {
      *e = ctx->free_events.back();
      ctx->free_events.pop_back();
    } else if (hipEventCreate(e) != hipSuccess) {
      return false;
    }
  }
  ctx->pending.push_back(p);
  *a = p.a;
  *b = p.b;
  return true;
}

static void sdvl_timer_collect(sdvl_ctx *ctx) {
  for (auto &p : ctx->pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->timers[p.timer].ms += ms;
      ctx->timers[p.timer].launches += 1;
    }
    ctx->free_events.push_back(p.a);
    ctx->free_events.push_back(p.b);
  }
  ctx->pending.clear();
}

// ---- batched upload: the GPU pulls the images out of pinned host memory itself ---------------------------------------------
namespace {
struct UploadJob {
  const uint8_t *src;  // device-visible address of the image in pinned host memory
  uint8_t *dst;        // level 0 of the frame
};
constexpr int kPrefetchChunks = 4;  // the prefetch runs beside a step's kernels: few workgroups, each lane many loads in flight
constexpr int kUploadChunks = 30;  // workgroups per image: 256 images x 30 = 7680 workgroups, every lane keeps 4 x 16 B reads in flight

__global__ __launch_bounds__(256) void frames_upload_kernel(const UploadJob *__restrict__ jobs, int width, int height, int stride) {
  const UploadJob job = jobs[blockIdx.y];
  const int tid = threadIdx.x + blockIdx.x * 256, nthr = 256 * gridDim.x;
  if (stride == width && (reinterpret_cast<uintptr_t>(job.src) & 15u) == 0 && ((static_cast<size_t>(width) * height) & 15u) == 0) {
    // contiguous image: 16-byte units, four independent loads per lane and round (reads cross the bus: latency is long)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *s = reinterpret_cast<const u32x4 *>(job.src);
    u32x4 *d = reinterpret_cast<u32x4 *>(job.dst);
    const int n16 = static_cast<int>((static_cast<size_t>(width) * height) >> 4);
    int i = tid;
    for (; i + 3 * nthr < n16; i += 4 * nthr) {
      const u32x4 a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + nthr);
      const u32x4 c = __builtin_nontemporal_load(s + i + 2 * nthr), e = __builtin_nontemporal_load(s + i + 3 * nthr);
      d[i] = a; d[i + nthr] = b; d[i + 2 * nthr] = c; d[i + 3 * nthr] = e;
    }
    for (; i < n16; i += nthr) d[i] = __builtin_nontemporal_load(s + i);
  } else {  // padded rows or odd alignment: row by row, 4-byte units where the row allows it
    for (int r = blockIdx.x; r < height; r += gridDim.x) {
      const uint8_t *s = job.src + static_cast<size_t>(r) * stride;
      uint8_t *d = job.dst + static_cast<size_t>(r) * width;
      if ((reinterpret_cast<uintptr_t>(s) & 3u) == 0 && (reinterpret_cast<uintptr_t>(d) & 3u) == 0 && (width & 3) == 0) {
        for (int x = threadIdx.x; x < (width >> 2); x += 256) reinterpret_cast<uint32_t *>(d)[x] = reinterpret_cast<const uint32_t *>(s)[x];
      } else {
        for (int x = threadIdx.x; x < width; x += 256) d[x] = s[x];
      }
    }
  }
}
struct OwnRec {
  int32_t id, pad_;
  const uint8_t *level0;
};
// the registry record of a frame that has taken its image into its own storage: level 0 points there from now on
__global__ __launch_bounds__(64) void registry_level0_kernel(const OwnRec *__restrict__ recs, int n, SearchFramePose *__restrict__ registry) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) registry[recs[i].id].f.level[0] = recs[i].level0;
}
}  // namespace

extern "C" {

int sdvl_ctx_create(int device, sdvl_ctx **out) {
  if (!out) return SDVL_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SDVL_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return SDVL_ERR_INVALID;
  sdvl_ctx *ctx = new sdvl_ctx();
  ctx->device = device;
  if (sdvl_bind_device(ctx) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return SDVL_ERR_HIP;
  }
  *out = ctx;
  return SDVL_OK;
}

int sdvl_ctx_destroy(sdvl_ctx *ctx) {
  if (!ctx) return SDVL_ERR_INVALID;
  (void)sdvl_bind_device(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  sdvl_timer_collect(ctx);
  for (hipEvent_t e : ctx->free_events) (void)hipEventDestroy(e);
  if (ctx->h_flag) (void)hipHostFree(const_cast<uint32_t *>(ctx->h_flag));
  if (ctx->d_nits) (void)hipFree(ctx->d_nits);
  if (ctx->d_registry) (void)hipFree(ctx->d_registry);
  if (ctx->d_fast_table) (void)hipFree(ctx->d_fast_table);
  if (ctx->d_undist_map) (void)hipFree(ctx->d_undist_map);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->h_search) (void)hipHostFree(ctx->h_search);
  if (ctx->d_search) (void)hipFree(ctx->d_search);
  if (ctx->h_out) (void)hipHostFree(ctx->h_out);
  if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
  if (ctx->d_stage) (void)hipFree(ctx->d_stage);
  if (ctx->d_out) (void)hipFree(ctx->d_out);
  if (ctx->d_work) (void)hipFree(ctx->d_work);
  if (ctx->d_detect) (void)hipFree(ctx->d_detect);
  if (ctx->d_counts) (void)hipFree(ctx->d_counts);
  for (void *sl : ctx->slabs) (void)hipFree(sl);
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->forked) ctx->stream = ctx->main_stream;
  if (ctx->side_stream) {
    (void)hipStreamSynchronize(ctx->side_stream);
    (void)hipStreamDestroy(ctx->side_stream);
  }
  if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
  if (ctx->join_event) (void)hipEventDestroy(ctx->join_event);
  for (hipEvent_t e : ctx->copy_event)
    if (e) (void)hipEventDestroy(e);
  if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
  for (void *j : ctx->h_prefetch_jobs)
    if (j) (void)hipHostFree(j);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return SDVL_OK;
}

const char *sdvl_last_error(const sdvl_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int sdvl_ctx_bind_thread(sdvl_ctx *ctx) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  return SDVL_OK;
}

int sdvl_ctx_device(const sdvl_ctx *ctx) { return ctx ? ctx->device : SDVL_ERR_INVALID; }

int sdvl_pointer_device(sdvl_ctx *ctx, const void *p, int *device) {
  if (!ctx || !p || !device) return SDVL_ERR_INVALID;
  hipPointerAttribute_t a;
  SDVL_HIP_CHECK(ctx, hipPointerGetAttributes(&a, p));
  *device = a.device;
  return SDVL_OK;
}

int sdvl_ctx_scratch_device(sdvl_ctx *ctx, int *device) {
  if (!ctx || !device) return SDVL_ERR_INVALID;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, 256, false);
  if (rc) return rc;
  return sdvl_pointer_device(ctx, ctx->d_out, device);
}

int sdvl_ctx_synchronize(sdvl_ctx *ctx) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  return SDVL_OK;
}

void *sdvl_ctx_stream(sdvl_ctx *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

int sdvl_ctx_timing_enable(sdvl_ctx *ctx, int on) {
  if (!ctx) return SDVL_ERR_INVALID;
  ctx->timing = on;
  return SDVL_OK;
}

// Dispatch events are not free: a launch with start / stop events costs the host ~12 us instead of ~4, and with every dispatch of
// every stream carrying them the farm tracked ~10 % fewer frames per second (round 4: 345 k against 385 k).  A caller that needs ONE
// kernel's launch durations over a long run (bench.py: the roofline's kernel over the timed region) names it here; null or "" = all.
int sdvl_ctx_timing_only(sdvl_ctx *ctx, const char *name) {
  if (!ctx) return SDVL_ERR_INVALID;
  ctx->timing_only = name ? name : "";
  return SDVL_OK;
}

int sdvl_ctx_timing_get(sdvl_ctx *ctx, int cap, char (*names)[32], double *ms, int64_t *launches, int *n) {
  if (!ctx || !n) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  sdvl_timer_collect(ctx);
  int k = 0;
  for (const auto &t : ctx->timers) {
    if (k >= cap) break;
    if (names) { strncpy(names[k], t.name.c_str(), 31); names[k][31] = 0; }
    if (ms) ms[k] = t.ms;
    if (launches) launches[k] = t.launches;
    k++;
  }
  *n = k;
  return SDVL_OK;
}

int sdvl_ctx_timing_reset(sdvl_ctx *ctx) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  sdvl_timer_collect(ctx);
  for (auto &t : ctx->timers) { t.ms = 0.0; t.launches = 0; }
  return SDVL_OK;
}

// ---- frames -------------------------------------------------------------------------------------------------

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }


namespace {
// Round 3: a frame holds only what stays useful for as long as the frame lives — a keyframe lives for the rest of the run:
//   level0 | level1 | ... (u8, 256-B aligned, 64 B slack) | corner header {count,0,0,0} + corners[cap] (x,y,level,pad int32)
//   | descriptors[cap][32] | bin offsets [bin_cells + 1] | bin entries [cap] (8 B)
// cap = the context's corner capacity when the frame was created (SDVL_MAX_CORNERS unless sdvl_ctx_set_corner_capacity says less).
// 640x480, cap 6144: 0.76 MB (1.43 MB in round 2, which kept 0.7 MB of detection scratch per frame); cap 1536: 0.50 MB.
struct FrameLayout {
  size_t level_off[SDVL_MAX_LEVELS], corners_off, desc_off, bin_start_off, bin_entries_off, bytes;
  int lw[SDVL_MAX_LEVELS], lh[SDVL_MAX_LEVELS], bin_gw, bin_cells, corner_cap;
};

bool frame_layout(int width, int height, int levels, int corner_cap, FrameLayout *L) {
  size_t off = 0;
  int w = width, h = height;
  for (int l = 0; l < levels; l++) {
    if (w < 1 || h < 1) return false;
    L->lw[l] = w;
    L->lh[l] = h;
    L->level_off[l] = off;
    off = align_up(off + static_cast<size_t>(w) * h + 64, 256);  // +64: slack so that word loads may overrun a row end
    w /= 2;
    h /= 2;
  }
  L->corner_cap = corner_cap;
  L->corners_off = off;  // 16-byte header {count,0,0,0} + corner records, written by ONE copy
  off = align_up(off + sizeof(int32_t) * 4 * (static_cast<size_t>(corner_cap) + 1), 256);
  L->desc_off = off;
  off = align_up(off + static_cast<size_t>(32) * corner_cap, 256);
  L->bin_gw = (width + 31) / 32;
  const int cells = L->bin_gw * ((height + 31) / 32);
  L->bin_cells = cells <= 4096 ? cells : 0;  // larger grids: no bins, the searches scan the list
  L->bin_start_off = off;
  off = align_up(off + sizeof(int32_t) * (static_cast<size_t>(L->bin_cells) + 1), 256);
  L->bin_entries_off = off;
  off = align_up(off + sizeof(uint2) * static_cast<size_t>(L->bin_cells > 0 ? corner_cap : 0), 256);
  L->bytes = off;
  return true;
}

sdvl_frame *frame_bind(const FrameLayout &L, int width, int height, int levels, uint8_t *base, int in_slab) {
  sdvl_frame *f = new sdvl_frame();
  memset(&f->v, 0, sizeof(f->v));
  f->width = width;
  f->height = height;
  f->base = base;
  f->bytes = L.bytes;
  f->in_slab = in_slab;
  for (int l = 0; l < levels; l++) {
    f->v.level[l] = base + L.level_off[l];
    f->v.lw[l] = L.lw[l];
    f->v.lh[l] = L.lh[l];
  }
  f->own_level0 = f->v.level[0];
  f->hdr_stale = 0;
  f->v.levels = levels;
  f->v.n_corners = 0;
  f->v.corner_hdr = reinterpret_cast<int32_t *>(base + L.corners_off);
  f->v.corners = f->v.corner_hdr + 4;
  f->v.desc = base + L.desc_off;
  f->corner_cap = L.corner_cap;
  f->desc_valid = 0;
  f->bins_valid = 0;
  f->bin_gw = L.bin_gw;
  f->bin_cells = L.bin_cells;
  f->bin_start = reinterpret_cast<int32_t *>(base + L.bin_start_off);
  f->bin_entries = reinterpret_cast<uint2 *>(base + L.bin_entries_off);
  f->reg_id = -1;
  f->home = nullptr;
  return f;
}
}  // namespace

int sdvl_ctx_set_corner_capacity(sdvl_ctx *ctx, int max_corners) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, max_corners >= 64 && max_corners <= SDVL_MAX_CORNERS, "corner capacity must lie in [64, SDVL_MAX_CORNERS]");
  ctx->corner_cap = max_corners;
  return SDVL_OK;
}

int64_t sdvl_frame_footprint_cap(int width, int height, int levels, int max_corners) {
  FrameLayout L;
  if (width < 16 || height < 16 || levels < 1 || levels > SDVL_MAX_LEVELS || max_corners < 64 || max_corners > SDVL_MAX_CORNERS ||
      !frame_layout(width, height, levels, max_corners, &L))
    return -1;
  return static_cast<int64_t>(L.bytes);
}

int sdvl_frame_create(sdvl_ctx *ctx, int width, int height, int levels, sdvl_frame **out) {
  return sdvl_frame_create_many(ctx, width, height, levels, 1, out);
}

int64_t sdvl_frame_footprint(int width, int height, int levels) { return sdvl_frame_footprint_cap(width, height, levels, SDVL_MAX_CORNERS); }

// n frames out of ONE allocation (hipMalloc costs ~0.2 ms and synchronises; a tracker farm turns frames into keyframes
// all the time).  The slab belongs to the context and is released with it; sdvl_frame_destroy only drops the handle.

int sdvl_frame_create_many(sdvl_ctx *ctx, int width, int height, int levels, int n, sdvl_frame **out) {
  if (!ctx || !out || n <= 0) return SDVL_ERR_INVALID;
  for (int i = 0; i < n; i++) out[i] = nullptr;
  SDVL_REQUIRE(ctx, width >= 16 && height >= 16 && width <= 4095 && height <= 4095, "frame size out of range");
  SDVL_REQUIRE(ctx, levels >= 1 && levels <= SDVL_MAX_LEVELS, "pyramid levels out of range");
  FrameLayout L;
  SDVL_REQUIRE(ctx, frame_layout(width, height, levels, ctx->corner_cap, &L), "image too small for the pyramid depth");
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  {
    const int rc = sdvl_registry_reserve(ctx, n);
    if (rc) return rc;
  }
  uint8_t *base = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&base), L.bytes * static_cast<size_t>(n));
  if (e != hipSuccess) {
    ctx->err = std::string("hipMalloc(frames): ") + hipGetErrorString(e);
    return SDVL_ERR_HIP;
  }
  const int in_slab = n > 1 ? 1 : 0;
  if (in_slab) ctx->slabs.push_back(base);
  for (int i = 0; i < n; i++) {
    out[i] = frame_bind(L, width, height, levels, base + L.bytes * static_cast<size_t>(i), in_slab);
    out[i]->home = ctx;
    if (!ctx->registry_free.empty()) {
      out[i]->reg_id = ctx->registry_free.back();
      ctx->registry_free.pop_back();
    } else {
      out[i]->reg_id = ctx->registry_next++;
    }
    e = hipMemsetAsync(out[i]->v.corner_hdr, 0, 16, ctx->stream);
    if (e != hipSuccess) {
      ctx->err = std::string("hipMemsetAsync(frame): ") + hipGetErrorString(e);
      return SDVL_ERR_HIP;
    }
  }
  return SDVL_OK;
}

int sdvl_frame_destroy(sdvl_ctx *ctx, sdvl_frame *f) {
  if (!ctx || !f) return SDVL_ERR_INVALID;
  if (!f->in_slab) {
    SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
    SDVL_HIP_CHECK(ctx, hipFree(f->base));
  }
  if (f->home == ctx && f->reg_id >= 0) ctx->registry_free.push_back(f->reg_id);
  delete f;
  return SDVL_OK;
}

int sdvl_frame_upload(sdvl_ctx *ctx, sdvl_frame *f, const uint8_t *img, int stride) {
  if (!ctx || !f || !img) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, stride >= f->width, "stride smaller than width");
  // hipMemcpy2DAsync from pageable memory stages internally; it returns once the source has been consumed.
  f->v.level[0] = f->own_level0;
  if (stride == f->width)  // contiguous: one linear copy (pinned sources go out as a single DMA)
    SDVL_HIP_CHECK(ctx, hipMemcpyAsync(f->v.level[0], img, static_cast<size_t>(f->width) * f->height, hipMemcpyHostToDevice, ctx->stream));
  else
    SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(f->v.level[0], f->width, img, stride, f->width, f->height, hipMemcpyHostToDevice, ctx->stream));
  f->hdr_stale = 1;
  f->v.n_corners = 0;
  f->desc_valid = 0;
  f->bins_valid = 0;
  return SDVL_OK;
}

int sdvl_frames_upload(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const uint8_t *const *imgs, int stride) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !imgs))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  const int width = frames[0]->width, height = frames[0]->height;
  SDVL_REQUIRE(ctx, stride >= width, "stride smaller than width");
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  // which images can the GPU read where they are?
  std::vector<const uint8_t *> mapped(n, nullptr);
  int n_mapped = 0;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] && imgs[i], "null frame or image");
    SDVL_REQUIRE(ctx, frames[i]->width == width && frames[i]->height == height, "sdvl_frames_upload: frames of one shape");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, imgs[i]) == hipSuccess && (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeDevice) &&
        attr.devicePointer) {  // pinned host memory, or an image that already sits in HBM (input ring): the same gather
      mapped[i] = static_cast<const uint8_t *>(attr.devicePointer);
      n_mapped++;
    } else {
      (void)hipGetLastError();  // an unregistered pointer is not an error here
    }
  }
  if (n_mapped > 0) {
    void *hs = nullptr, *ds = nullptr;
    const int rc = sdvl_stage_alloc(ctx, sizeof(UploadJob) * static_cast<size_t>(n_mapped), &hs, &ds);
    if (rc) return rc;
    UploadJob *hj = static_cast<UploadJob *>(hs);
    int k = 0;
    for (int i = 0; i < n; i++)
      if (mapped[i]) hj[k++] = UploadJob{mapped[i], frames[i]->own_level0};
    SDVL_HIP_CHECK(ctx, sdvl_push(ctx, ds, hs, sizeof(UploadJob) * static_cast<size_t>(n_mapped)));
    SDVL_LAUNCH(ctx, "frames_upload", frames_upload_kernel, dim3(kUploadChunks, n_mapped), dim3(256), static_cast<const UploadJob *>(ds), width,
                height, stride);
    SDVL_HIP_CHECK(ctx, hipGetLastError());
  }
  for (int i = 0; i < n; i++) {
    sdvl_frame *f = frames[i];
    f->v.level[0] = f->own_level0;
    if (!mapped[i]) {
      if (stride == width)
        SDVL_HIP_CHECK(ctx, hipMemcpyAsync(f->v.level[0], imgs[i], static_cast<size_t>(width) * height, hipMemcpyHostToDevice, ctx->stream));
      else
        SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(f->v.level[0], width, imgs[i], stride, width, height, hipMemcpyHostToDevice, ctx->stream));
    }
    f->hdr_stale = 1;
    f->v.n_corners = 0;
    f->desc_valid = 0;
    f->bins_valid = 0;
  }
  return SDVL_OK;
}

// ---- input ring -----------------------------------------------------------------------------------------------------------
int sdvl_ctx_prefetch_images(sdvl_ctx *ctx, int n, const uint8_t *const *imgs, int stride, int width, int height, void *const *dev_dst) {
  if (!ctx || n <= 0 || !imgs || !dev_dst || width <= 0 || height <= 0) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, stride >= width, "stride smaller than width");
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  if (!ctx->copy_stream) {
    SDVL_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t &e : ctx->copy_event) SDVL_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  if (ctx->prefetch_jobs_cap < static_cast<size_t>(n)) {
    SDVL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));  // nobody reads the old lists any more
    for (void *&j : ctx->h_prefetch_jobs) {
      if (j) (void)hipHostFree(j);
      j = nullptr;
      SDVL_HIP_CHECK(ctx, hipHostMalloc(&j, sizeof(UploadJob) * static_cast<size_t>(n), hipHostMallocDefault));
    }
    ctx->prefetch_jobs_cap = static_cast<size_t>(n);
  }
  // Dense images that follow each other in memory on both sides (a capture buffer, a batch laid out frame after frame) go as
  // ONE copy per run: the DMA engines reach the link rate with transfers of tens of MB (57 GB/s measured against ~30 GB/s for
  // 300 KB pieces or for a kernel pulling the bytes itself), and they take no compute unit from the step that is running.
  const size_t fb = static_cast<size_t>(width) * height;
  const unsigned ticket = ctx->prefetch_count++;
  // the job list and the event of four calls ago are reused: that prefetch must have completed (nothing obliges a caller to fence
  // every prefetch, and the gather kernel reads its list in place from host memory)
  if (ticket >= 4u) SDVL_HIP_CHECK(ctx, hipEventSynchronize(ctx->copy_event[ticket & 3u]));
  UploadJob *hj = static_cast<UploadJob *>(ctx->h_prefetch_jobs[ticket & 3u]);
  // the destinations may still be read by work queued on the context's stream (frames that alias an input-ring slot, and the
  // keyframes' copy out of it, sdvl_frames_own_images): the prefetch starts behind everything queued there so far
  if (!ctx->order_event) SDVL_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->order_event, hipEventDisableTiming));
  SDVL_HIP_CHECK(ctx, hipEventRecord(ctx->order_event, ctx->stream));
  SDVL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->order_event, 0));
  int n_mapped = 0;
  for (int i = 0; i < n;) {
    SDVL_REQUIRE(ctx, imgs[i] && dev_dst[i], "null image or destination");
    int run = 1;
    if (stride == width)
      while (i + run < n && imgs[i + run] == imgs[i] + run * fb && dev_dst[i + run] == static_cast<uint8_t *>(dev_dst[i]) + run * fb) run++;
    if (run > 1 || stride == width) {
      SDVL_HIP_CHECK(ctx, hipMemcpyAsync(dev_dst[i], imgs[i], fb * run, hipMemcpyHostToDevice, ctx->copy_stream));
    } else {
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, imgs[i]) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
        hj[n_mapped++] = UploadJob{static_cast<const uint8_t *>(attr.devicePointer), static_cast<uint8_t *>(dev_dst[i])};  // padded rows, pinned: gather
      } else {
        (void)hipGetLastError();
        SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(dev_dst[i], width, imgs[i], stride, width, height, hipMemcpyHostToDevice, ctx->copy_stream));
      }
    }
    i += run;
  }
  if (n_mapped > 0) {
    hipPointerAttribute_t ja;
    SDVL_HIP_CHECK(ctx, hipPointerGetAttributes(&ja, hj));
    hipLaunchKernelGGL(frames_upload_kernel, dim3(kPrefetchChunks, n_mapped), dim3(256), 0, ctx->copy_stream,
                       static_cast<const UploadJob *>(ja.devicePointer), width, height, stride);
    SDVL_HIP_CHECK(ctx, hipGetLastError());
  }
  SDVL_HIP_CHECK(ctx, hipEventRecord(ctx->copy_event[ticket & 3u], ctx->copy_stream));
  return static_cast<int>(ticket & 0x3FFFFFFFu);
}

int sdvl_ctx_prefetch_fence(sdvl_ctx *ctx, int ticket) {
  if (!ctx || ticket < 0) return SDVL_ERR_INVALID;
  const unsigned next = ctx->prefetch_count & 0x3FFFFFFFu, t = static_cast<unsigned>(ticket);
  SDVL_REQUIRE(ctx, ctx->copy_stream && ((next - t) & 0x3FFFFFFFu) >= 1 && ((next - t) & 0x3FFFFFFFu) <= 4,
               "sdvl_ctx_prefetch_fence: ticket is not one of the last four prefetches");
  SDVL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->copy_event[t & 3u], 0));
  return SDVL_OK;
}

// ---- feed: a copy stream of its own, driven by a thread of its own ---------------------------------------------------------
// A feed is ONE stream that ONE caller thread (a farm's feeder) fills in the order the images will be needed; consumers (contexts,
// each on its own thread) only exchange events with it:
//   feeder    sdvl_feed_images(slot, ...)       transfer into the slot's buffers, behind the slot's last release
//   consumer  sdvl_ctx_feed_acquire(ctx, slot)  ctx's stream waits for that transfer
//             sdvl_ctx_feed_release(ctx, slot)  what ctx has queued so far was the last reader of the slot's buffers
// The caller orders the three calls of a slot among its threads (a mutex / condition variable); the library orders the streams.
struct sdvl_feed {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ready, released;
  // the feeder thread (sdvl_feed_images) and the consumers' threads (sdvl_feed_slot_arrived) both report here: the message of a
  // failed call is kept per calling thread, so that one thread's failure is neither torn nor replaced by another's
  void fail(const std::string &m) const;
};
thread_local std::string t_feed_err;
void sdvl_feed::fail(const std::string &m) const { t_feed_err = m; }

int sdvl_feed_create(int device, int n_slots, sdvl_feed **out) {
  if (!out || n_slots <= 0) return SDVL_ERR_INVALID;
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) return SDVL_ERR_NO_DEVICE;
  sdvl_feed *f = new sdvl_feed();
  f->device = device;
  // A stream of another priority class gets a hardware queue of its own: the markers that follow every transfer (hipEventRecord
  // behind an SDMA copy is a barrier packet that waits for the copy's signal) would otherwise sit in a queue shared with compute
  // streams and stall them for the duration of every transfer.
  int lo = 0, hi = 0;
  bool ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(&f->stream, hipStreamNonBlocking, hi) == hipSuccess;
  f->ready.assign(n_slots, nullptr);
  f->released.assign(n_slots, nullptr);
  for (int i = 0; i < n_slots && ok; i++)
    ok = hipEventCreateWithFlags(&f->ready[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&f->released[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    sdvl_feed_destroy(f);
    return SDVL_ERR_HIP;
  }
  *out = f;
  return SDVL_OK;
}

int sdvl_feed_destroy(sdvl_feed *f) {
  if (!f) return SDVL_ERR_INVALID;
  (void)hipSetDevice(f->device);
  if (f->stream) {
    (void)hipStreamSynchronize(f->stream);
    (void)hipStreamDestroy(f->stream);
  }
  for (hipEvent_t e : f->ready)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : f->released)
    if (e) (void)hipEventDestroy(e);
  delete f;
  return SDVL_OK;
}

const char *sdvl_feed_last_error(const sdvl_feed *f) { return f ? t_feed_err.c_str() : "null feed"; }

#define SDVL_FEED_CHECK(f, expr)                                        \
  do {                                                                  \
    hipError_t e_ = (expr);                                             \
    if (e_ != hipSuccess) {                                             \
      (f)->fail(std::string(#expr) + ": " + hipGetErrorString(e_));     \
      return SDVL_ERR_HIP;                                              \
    }                                                                   \
  } while (0)

int sdvl_feed_images(sdvl_feed *f, int slot, int n, const uint8_t *const *imgs, int stride, int width, int height, void *const *dev_dst) {
  if (!f || n <= 0 || !imgs || !dev_dst || width <= 0 || height <= 0 || stride < width) return SDVL_ERR_INVALID;
  if (slot < 0 || slot >= static_cast<int>(f->ready.size())) return SDVL_ERR_INVALID;
  SDVL_FEED_CHECK(f, hipSetDevice(f->device));
  SDVL_FEED_CHECK(f, hipStreamWaitEvent(f->stream, f->released[slot], 0));  // never recorded yet: no wait
  const size_t fb = static_cast<size_t>(width) * height;
  for (int i = 0; i < n;) {
    if (!imgs[i] || !dev_dst[i]) { f->fail("null image or destination"); return SDVL_ERR_INVALID; }
    int run = 1;  // dense images that follow each other on both sides travel as ONE transfer
    if (stride == width)
      while (i + run < n && imgs[i + run] == imgs[i] + run * fb && dev_dst[i + run] == static_cast<uint8_t *>(dev_dst[i]) + run * fb) run++;
    if (stride == width) SDVL_FEED_CHECK(f, hipMemcpyAsync(dev_dst[i], imgs[i], fb * run, hipMemcpyHostToDevice, f->stream));
    else SDVL_FEED_CHECK(f, hipMemcpy2DAsync(dev_dst[i], width, imgs[i], stride, width, height, hipMemcpyHostToDevice, f->stream));
    i += run;
  }
  SDVL_FEED_CHECK(f, hipEventRecord(f->ready[slot], f->stream));
  return SDVL_OK;
}

int sdvl_feed_slot_arrived(sdvl_feed *f, int slot) {
  if (!f || slot < 0 || slot >= static_cast<int>(f->ready.size())) return SDVL_ERR_INVALID;
  const hipError_t e = hipEventQuery(f->ready[slot]);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  f->fail(std::string("hipEventQuery: ") + hipGetErrorString(e));
  return SDVL_ERR_HIP;
}

int sdvl_ctx_feed_acquire(sdvl_ctx *ctx, sdvl_feed *f, int slot) {
  if (!ctx || !f || slot < 0 || slot >= static_cast<int>(f->ready.size())) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, ctx->device == f->device, "feed and context of one GPU");
  SDVL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, f->ready[slot], 0));
  return SDVL_OK;
}

int sdvl_ctx_feed_release(sdvl_ctx *ctx, sdvl_feed *f, int slot) {
  if (!ctx || !f || slot < 0 || slot >= static_cast<int>(f->released.size())) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, hipEventRecord(f->released[slot], ctx->stream));
  return SDVL_OK;
}

int sdvl_frame_set_image_device(sdvl_ctx *ctx, sdvl_frame *f, const void *dev_img, int stride) {
  if (!ctx || !f || !dev_img) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, stride >= f->width, "stride smaller than width");
  f->v.level[0] = f->own_level0;
  SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(f->v.level[0], f->width, dev_img, stride, f->width, f->height,
                                       hipMemcpyDeviceToDevice, ctx->stream));
  f->hdr_stale = 1;
  f->v.n_corners = 0;
  f->desc_valid = 0;
  f->bins_valid = 0;
  return SDVL_OK;
}

int sdvl_frame_borrow_image_device(sdvl_ctx *ctx, sdvl_frame *f, const void *dev_img) {
  if (!ctx || !f || !dev_img) return SDVL_ERR_INVALID;
  f->v.level[0] = static_cast<uint8_t *>(const_cast<void *>(dev_img));  // read-only use: kernels never write level 0
  f->hdr_stale = 1;
  f->v.n_corners = 0;
  f->desc_valid = 0;
  f->bins_valid = 0;
  return SDVL_OK;
}

// Frames whose level 0 aliases a caller image (sdvl_frame_borrow_image_device: an input-ring slot, a capture buffer) and that must
// outlive it — the frames that have just become keyframes (reference patches for SearchPoint come from the keyframe of a point's
// first observation, matcher.cc:52,87) — copy the image into their own level 0: ONE gather launch, and the level-0 pointer of
// their registry records follows.  Frames that already own their image are skipped.
int sdvl_frames_own_images(sdvl_ctx *ctx, int n, sdvl_frame *const *frames) {
  if (!ctx || n < 0 || (n > 0 && !frames)) return SDVL_ERR_INVALID;
  int m = 0;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] != nullptr, "null frame");
    SDVL_REQUIRE(ctx, frames[i]->width == frames[0]->width && frames[i]->height == frames[0]->height, "sdvl_frames_own_images: frames of one shape");
    if (frames[i]->v.level[0] != frames[i]->own_level0) m++;
  }
  if (m == 0) return SDVL_OK;
  void *hs = nullptr, *ds = nullptr;
  const size_t jb = (sizeof(UploadJob) * static_cast<size_t>(m) + 255) / 256 * 256;
  const int rc = sdvl_stage_alloc(ctx, jb + sizeof(OwnRec) * static_cast<size_t>(m), &hs, &ds);
  if (rc) return rc;
  UploadJob *hj = static_cast<UploadJob *>(hs);
  OwnRec *hr = reinterpret_cast<OwnRec *>(static_cast<uint8_t *>(hs) + jb);
  int k = 0, n_reg = 0;
  for (int i = 0; i < n; i++) {
    sdvl_frame *f = frames[i];
    if (f->v.level[0] == f->own_level0) continue;
    hj[k++] = UploadJob{f->v.level[0], f->own_level0};
    if (f->home == ctx && f->reg_id >= 0) hr[n_reg++] = OwnRec{f->reg_id, 0, f->own_level0};
    f->v.level[0] = f->own_level0;
  }
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, ds, hs, jb + sizeof(OwnRec) * static_cast<size_t>(m)));
  SDVL_LAUNCH(ctx, "frames_own", frames_upload_kernel, dim3(kUploadChunks, m), dim3(256), static_cast<const UploadJob *>(ds), frames[0]->width,
              frames[0]->height, frames[0]->width);
  if (n_reg > 0 && ctx->d_registry)
    hipLaunchKernelGGL(registry_level0_kernel, dim3((n_reg + 63) / 64), dim3(64), 0, ctx->stream,
                       reinterpret_cast<const OwnRec *>(static_cast<uint8_t *>(ds) + jb), n_reg, static_cast<SearchFramePose *>(ctx->d_registry));
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

int sdvl_frame_download_level(sdvl_ctx *ctx, const sdvl_frame *f, int level, uint8_t *out, int stride) {
  if (!ctx || !f || !out) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, level >= 0 && level < f->v.levels, "level out of range");
  SDVL_REQUIRE(ctx, stride >= f->v.lw[level], "stride smaller than level width");
  SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(out, stride, f->v.level[level], f->v.lw[level], f->v.lw[level], f->v.lh[level],
                                       hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  return SDVL_OK;
}

int sdvl_frame_set_corners(sdvl_ctx *ctx, sdvl_frame *f, int n, const int32_t *xyl) {
  if (!ctx || !f || (n > 0 && !xyl)) return SDVL_ERR_INVALID;
  if (n < 0 || n > f->corner_cap) {
    ctx->err = "too many corners for one frame (its corner capacity, at most SDVL_MAX_CORNERS)";
    return SDVL_ERR_CAPACITY;
  }
  // validate on the host: every corner must lie inside its level image (kernels index with it)
  for (int i = 0; i < n; i++) {
    const int x = xyl[3 * i], y = xyl[3 * i + 1], l = xyl[3 * i + 2];
    SDVL_REQUIRE(ctx, l >= 0 && l < f->v.levels && x >= 0 && y >= 0 && x < f->v.lw[l] && y < f->v.lh[l],
                 "corner outside its pyramid level");
  }
  {
    const size_t bytes = sizeof(int32_t) * 4 * (n + 1);
    void *hs = nullptr, *dsx = nullptr;
    int rc = sdvl_stage_alloc(ctx, bytes, &hs, &dsx);
    if (rc) return rc;
    int32_t *st = static_cast<int32_t *>(hs);
    st[0] = n; st[1] = 0; st[2] = 0; st[3] = 0;
    for (int i = 0; i < n; i++) {
      st[4 * (i + 1)] = xyl[3 * i]; st[4 * (i + 1) + 1] = xyl[3 * i + 1]; st[4 * (i + 1) + 2] = xyl[3 * i + 2]; st[4 * (i + 1) + 3] = 0;
    }
    SDVL_HIP_CHECK(ctx, sdvl_push(ctx, f->v.corner_hdr, st, bytes));
    SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  }
  f->hdr_stale = 0;
  f->v.n_corners = n;
  f->desc_valid = 0;
  f->bins_valid = 0;
  return SDVL_OK;
}

int sdvl_frames_set_corners(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const int32_t *counts, const int32_t *xyl) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !counts))) return SDVL_ERR_INVALID;
  size_t total = 0;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] != nullptr, "null frame");
    if (counts[i] < 0 || counts[i] > frames[i]->corner_cap) {
      ctx->err = "too many corners for one frame (its corner capacity, at most SDVL_MAX_CORNERS)";
      return SDVL_ERR_CAPACITY;
    }
    total += counts[i];
  }
  if (n == 0) return SDVL_OK;
  if (total > 0 && !xyl) return SDVL_ERR_INVALID;
  {
    size_t k = 0;
    for (int i = 0; i < n; i++) {
      const sdvl_frame *f = frames[i];
      for (int j = 0; j < counts[i]; j++, k++) {
        const int x = xyl[3 * k], y = xyl[3 * k + 1], l = xyl[3 * k + 2];
        SDVL_REQUIRE(ctx, l >= 0 && l < f->v.levels && x >= 0 && y >= 0 && x < f->v.lw[l] && y < f->v.lh[l],
                     "corner outside its pyramid level");
      }
    }
  }
  const size_t bytes = sizeof(int32_t) * 4 * (total + n);
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_stage_alloc(ctx, bytes, &hs, &dsx);
  if (rc) return rc;
  int32_t *st = static_cast<int32_t *>(hs);
  size_t k = 0, w = 0;
  for (int i = 0; i < n; i++) {
    int32_t *rec = st + 4 * w;  // {count,0,0,0} header + records: one copy per frame
    rec[0] = counts[i]; rec[1] = 0; rec[2] = 0; rec[3] = 0;
    for (int j = 0; j < counts[i]; j++, k++) {
      rec[4 * (j + 1)] = xyl[3 * k]; rec[4 * (j + 1) + 1] = xyl[3 * k + 1]; rec[4 * (j + 1) + 2] = xyl[3 * k + 2]; rec[4 * (j + 1) + 3] = 0;
    }
    SDVL_HIP_CHECK(ctx, hipMemcpyAsync(frames[i]->v.corner_hdr, rec, sizeof(int32_t) * 4 * (counts[i] + 1), hipMemcpyHostToDevice, ctx->stream));
    frames[i]->v.n_corners = counts[i];
    frames[i]->hdr_stale = 0;
    frames[i]->desc_valid = 0;
    frames[i]->bins_valid = 0;
    w += counts[i] + 1;
  }
  // no synchronisation: later launches on the same stream are ordered behind the copies; the pinned staging
  // buffer is only rewritten after the next wait every entry point performs before reuse
  return SDVL_OK;
}

int sdvl_frame_num_corners(const sdvl_frame *f) { return f ? f->v.n_corners : SDVL_ERR_INVALID; }
int sdvl_frame_corner_capacity(const sdvl_frame *f) { return f ? f->corner_cap : SDVL_ERR_INVALID; }

int sdvl_frame_download_corners(sdvl_ctx *ctx, sdvl_frame *f, int cap, int32_t *xyl, int *n_out) {
  if (!ctx || !f || !n_out) return SDVL_ERR_INVALID;
  int n = 0;
  int rc = sdvl_frame_count_host(ctx, f, &n);
  if (rc) return rc;
  *n_out = n;
  if (n == 0) return SDVL_OK;
  if (!xyl) return SDVL_ERR_INVALID;
  if (n > cap) {
    ctx->err = "corner output capacity smaller than the corner count";
    return SDVL_ERR_CAPACITY;
  }
  const size_t bytes = sizeof(int32_t) * 4 * n;
  rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, bytes, true);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, f->v.corners, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  const int32_t *src = static_cast<const int32_t *>(ctx->h_out);
  for (int i = 0; i < n; i++) { xyl[3 * i] = src[4 * i]; xyl[3 * i + 1] = src[4 * i + 1]; xyl[3 * i + 2] = src[4 * i + 2]; }
  return SDVL_OK;
}

int sdvl_device_malloc(sdvl_ctx *ctx, int64_t bytes, void **out) {
  if (!ctx || !out || bytes <= 0) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  SDVL_HIP_CHECK(ctx, hipMalloc(out, static_cast<size_t>(bytes)));
  return SDVL_OK;
}

int sdvl_device_free(sdvl_ctx *ctx, void *p) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  SDVL_HIP_CHECK(ctx, hipFree(p));
  return SDVL_OK;
}

int sdvl_device_download(sdvl_ctx *ctx, const void *dev, int64_t bytes, void *host) {
  if (!ctx || !dev || !host || bytes <= 0) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(host, dev, static_cast<size_t>(bytes), hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  return SDVL_OK;
}

// ---- fork / join: a side chain behind an earlier point of the main stream ---------------------------------------------------------
int sdvl_ctx_fork_mark(sdvl_ctx *ctx) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, !ctx->forked, "sdvl_ctx_fork_mark inside a fork");
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  if (!ctx->side_stream) {
    SDVL_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    SDVL_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
    SDVL_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->join_event, hipEventDisableTiming));
  }
  SDVL_HIP_CHECK(ctx, hipEventRecord(ctx->fork_event, ctx->stream));
  ctx->fork_marked = 1;
  return SDVL_OK;
}

int sdvl_ctx_fork_begin(sdvl_ctx *ctx) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, ctx->fork_marked && !ctx->forked, "sdvl_ctx_fork_begin without sdvl_ctx_fork_mark");
  SDVL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->fork_event, 0));
  ctx->main_stream = ctx->stream;
  ctx->stream = ctx->side_stream;
  ctx->forked = 1;
  ctx->fork_marked = 0;
  return SDVL_OK;
}

int sdvl_ctx_fork_end(sdvl_ctx *ctx) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, ctx->forked, "sdvl_ctx_fork_end without sdvl_ctx_fork_begin");
  ctx->stream = ctx->main_stream;
  ctx->forked = 0;
  SDVL_HIP_CHECK(ctx, hipEventRecord(ctx->join_event, ctx->side_stream));
  SDVL_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->join_event, 0));
  return SDVL_OK;
}

int sdvl_host_alloc_pinned(sdvl_ctx *ctx, int64_t bytes, void **out) {
  if (!ctx || !out || bytes <= 0) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  SDVL_HIP_CHECK(ctx, hipHostMalloc(out, static_cast<size_t>(bytes), hipHostMallocMapped | hipHostMallocPortable));
  return SDVL_OK;
}

int sdvl_host_free_pinned(sdvl_ctx *ctx, void *p) {
  if (!ctx) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  SDVL_HIP_CHECK(ctx, hipHostFree(p));
  return SDVL_OK;
}

int sdvl_host_register(sdvl_ctx *ctx, void *p, int64_t bytes) {
  if (!ctx || !p || bytes <= 0) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  SDVL_HIP_CHECK(ctx, hipHostRegister(p, static_cast<size_t>(bytes), hipHostRegisterMapped | hipHostRegisterPortable));
  return SDVL_OK;
}

int sdvl_host_unregister(sdvl_ctx *ctx, void *p) {
  if (!ctx || !p) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  SDVL_HIP_CHECK(ctx, hipHostUnregister(p));
  return SDVL_OK;
}

}  // extern "C"
