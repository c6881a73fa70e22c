// sdvl_internal.h — context / frame records behind the C-ABI of include/sdvl_hip.h (not installed).
#ifndef SDVL_INTERNAL_H_
#define SDVL_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <hip/hip_ext.h>
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
  int32_t *corners;     // device: [corner_cap][4]: x, y, level, pad
  uint8_t *desc;        // device: [corner_cap][32]
};

struct sdvl_frame {
  FrameView v;
  uint8_t *base;      // one allocation: pyramid | corner header + list | descriptors | bin offsets | bin entries
  size_t bytes;
  int width, height;
  int corner_cap;        // corners the resident list / descriptor block / bin entries hold (sdvl_ctx_set_corner_capacity)
  int desc_valid;
  // the corner list binned by 32-px cell of level-0 coordinates (written by the detection's pack kernel): searches visit the
  // cells around a point instead of scanning every corner.  Round 3: part of the RESIDENT block; the detection's per-cell lists
  // are scratch of the context (sdvl_ctx::d_detect), no longer part of a frame
  int32_t *bin_start;    // [bin_cells + 1]
  uint2 *bin_entries;    // [n_corners] {packed corner x | y<<12 | level<<24, index in the corner list}
  int bin_gw, bin_cells; // grid; bin_cells == 0: the frame size does not fit the binning kernel
  int bins_valid;        // the bins describe the current corner list (sdvl_detect_corners); cleared with the corners
  int in_slab;           // storage belongs to a slab owned by the context (sdvl_frame_create_many)
  int hdr_stale;         // device corner header still holds the count of a previous image (reset lazily)
  uint8_t *own_level0;   // the frame's own level-0 storage (level[0] may point at a borrowed caller image)
  // slot of this frame in its creating context's device-resident (frame, pose) registry: device-resident tracking tables
  // (sdvl_track.hip) name frames by this index instead of carrying their views through every launch
  int reg_id;
  sdvl_ctx *home;
};

struct KernelTimer {
  std::string name;
  double ms = 0.0;
  int64_t launches = 0;
};

struct sdvl_ctx {
  int device;
  hipStream_t stream;
  // fast_cells_kernel's per-cell geometry (level, clipped ROI) for the last frame shape / grid seen, resident in HBM
  void *d_fast_table = nullptr;
  long long fast_table_key = -1;
  int fast_table_cells = 0;
  // Camera::UndistortImage's map (sdvl_undistort.hip): one 32-bit word per pixel for the last (size, intrinsics, distortion) seen
  void *d_undist_map = nullptr;
  size_t undist_map_bytes = 0;
  double undist_key[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  void *undist_quads = nullptr;  // the per-quad records behind the per-pixel words, same allocation
  long long undist_maps_built = 0;
  // input ring (sdvl_ctx_prefetch_images / _fence): a second stream that carries the NEXT step's images while this one computes
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_event[4] = {nullptr, nullptr, nullptr, nullptr};  // one per prefetch in flight (ticket & 3)
  hipEvent_t order_event = nullptr;  // marks the compute stream when a prefetch is issued: the copy stream starts behind it
  void *h_prefetch_jobs[4] = {nullptr, nullptr, nullptr, nullptr};  // pinned job lists, read by the gather kernel where they are
  size_t prefetch_jobs_cap = 0;
  unsigned prefetch_count = 0;
  // Round 5: a side stream for work that depends on an earlier point of the main stream only (sdvl_ctx_fork_mark / _begin / _end):
  // a lone camera's corner detection needs the pyramid, not the image alignment queued behind it, so the two chains run side by side
  hipStream_t side_stream = nullptr, main_stream = nullptr;
  hipEvent_t fork_event = nullptr, join_event = nullptr;
  int fork_marked = 0, forked = 0;
  std::string err;
  // pinned + device staging, grown on demand
  void *h_stage = nullptr; size_t h_stage_bytes = 0;
  void *d_stage = nullptr; size_t d_stage_bytes = 0;
  void *h_out = nullptr;   size_t h_out_bytes = 0;
  void *d_out = nullptr;   size_t d_out_bytes = 0;
  void *d_work = nullptr;  size_t d_work_bytes = 0;
  // detection scratch (sdvl_detect.hip): per batch slot the per-cell FAST lists, counts and the selection's intermediate lists;
  // dead once the batch's select_pack kernel has run
  void *d_detect = nullptr; size_t d_detect_bytes = 0;
  size_t pack_lds_limit = 0;            // dynamic LDS select_pack_kernel has been allowed so far
  int corner_cap = SDVL_MAX_CORNERS;    // capacity of the corner list of frames created from now on
  size_t stage_off = 0;  // bump pointer into h_stage/d_stage; reset by every sdvl_stream_wait
  // A search batch (sdvl_search_begin .. sdvl_search_run) is filled by the caller over time and must survive every wait in
  // between, so it does not live in the ring (whose bump pointer any sdvl_stream_wait resets) but in buffers of its own
  void *h_search = nullptr; size_t h_search_bytes = 0;
  void *d_search = nullptr; size_t d_search_bytes = 0;
  uint64_t search_busy_gen = ~0ull;  // wait_gen when the last batch was queued: its buffers are in use until a later wait
  // (frame, pose) registry: one SearchFramePose record per frame created on this context, index = sdvl_frame::reg_id
  void *d_registry = nullptr;
  int registry_cap = 0, registry_next = 0;
  std::vector<int> registry_free;
  // corner counts of the last sdvl_detect_corners batch, written by the pack kernel: one D2H serves all frames
  void *d_counts = nullptr; size_t d_counts_bytes = 0;
  std::vector<sdvl_frame *> detect_frames;
  // the counts of that batch also travel to this pinned array right behind the pack kernel; they have landed as soon as
  // any later wait on the stream has returned (wait_gen > counts_gen), so asking for them then costs no round trip
  void *h_counts = nullptr; size_t h_counts_bytes = 0;
  uint64_t wait_gen = 0, counts_gen = ~0ull;
  // which path the tracked steps' searches took (sdvl_ctx_counters): [0] jobs searched through the corner bins, [1] jobs whose search
  // scans the whole corner list (no bin layout for the frame size, corners set by hand, SDVL_TRACK_NO_BINS)
  long long counters[4] = {0, 0, 0, 0};
  std::vector<void *> slabs;  // bulk frame storage, released with the context
  // Waiting for a point of the stream: a 32-bit sequence number written by the stream itself (hipStreamWriteValue32) into
  // pinned host memory, polled by the waiting thread with plain loads
  volatile uint32_t *h_flag = nullptr;
  uint32_t flag_seq = 0;
  int wait_spin_us = 0;  // sdvl_ctx_set_wait_spin: poll without sleeping for this long before the sleeping polls (a lone camera's 0.2-ms waits)
  // cooperative waits: when set, sdvl_stream_wait polls the event and calls the hook while the stream is still busy, so a
  // host thread that drives several contexts can run another one's host stage instead of sleeping
  uint32_t align_ticket = 0;  // marks the result copy of sdvl_image_align_begin
  int align_pending = 0;
  // sdvl_search_run_chain: the search results have landed at chain_ticket; the pose results follow at the stream's tail
  uint32_t chain_ticket = 0;
  // sdvl_filter_inputs_begin in flight: its mark and the row geometry _end needs
  uint32_t filter_ticket = 0;
  int filter_pending = 0, filter_ccap = 0, filter_desc = 0;
  size_t filter_sc_bytes = 0, filter_row = 0;
  int chain_pending = 0;             // trackers of the chained batch in flight
  size_t chain_host_off = 0;         // where its pose results start in h_out
  int chain_obs_total = 0;
  // iteration budgets of SelectInliers for every match count 0..nits_max_size (row s at s*(s+1)/2), resident in HBM
  void *d_nits = nullptr;
  std::vector<int32_t> nits_host;
  int nits_points = -1, nits_its = -1, nits_max_size = -1;
  int waiting = 0;                   // a cooperative wait is polling `waiting_ticket` (sdvl_ctx_wait_done / _block)
  uint32_t waiting_ticket = 0;
  int waiting_kind = 0;
  void (*wait_hook)(void *user, sdvl_ctx *ctx) = nullptr;
  void *wait_user = nullptr;
  // per-kernel timing (HIP events on `stream`)
  int timing = 0;
  std::string timing_only;  // when not empty: only launches of this name get dispatch events (sdvl_ctx_timing_only)
  std::vector<KernelTimer> timers;
  struct Pending { int timer; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
};

// device record of one pose job (sdvl_pose.hip); sdvl_search_run_chain fills these on the device
struct PoseJobDev {
  int obs_begin, n_obs;
  int rand_begin, nits_begin;
  double pose[7];
  double pad_;
};
// queues pose_hypotheses + pose_refine for n_jobs device-resident jobs (no copies, no wait); d_hyp = n_jobs * max_ransac_its
// records of sdvl_pose_hyp_bytes() each
size_t sdvl_pose_hyp_bytes();
// max_obs: the most observations any of the jobs can hold (the device knows the actual counts; the supporter kernel's grid covers this many)
int sdvl_pose_enqueue_device(sdvl_ctx *ctx, int n_jobs, const PoseJobDev *d_jobs, const sdvl_pose_obs *d_obs, const int32_t *d_rand,
                             const int32_t *d_nits, const sdvl_pose_params *p, void *d_hyp, sdvl_pose_result *d_res, int32_t *d_lists, int max_obs,
                             int batch_size);  // batch_size: see sdvl_image_align_track_enqueue

// sdvl_image_align.hip: queue the alignment of n_jobs pairs; features from the host (`features`) or resident (`d_features`);
// results to `d_results` (device) or, when null, to the context's result buffers (see sdvl_image_align_begin)
int sdvl_image_align_enqueue(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, int n_features, const sdvl_align_feature *features,
                             const sdvl_align_feature *d_features, const sdvl_camera *cam, const sdvl_align_params *p,
                             sdvl_align_result *d_results);

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

// room for `extra` more frames in the registry (grows by reallocation + device copy behind a stream wait)
int sdvl_registry_reserve(sdvl_ctx *ctx, int extra);
// select the context's GPU on the calling thread (HIP's current device is per thread); cached, one compare when already bound
hipError_t sdvl_bind_device(const sdvl_ctx *ctx);
int sdvl_ensure(sdvl_ctx *ctx, void **p, size_t *cur, size_t need, bool pinned);
// `bytes` of pinned host staging + its device mirror (same offset in h_stage / d_stage).  Allocations made since the
// last sdvl_stream_wait never overlap, so a call can fill its records while earlier copies are still in flight;
// only when the ring is exhausted does this wait for the stream.
int sdvl_stage_alloc(sdvl_ctx *ctx, size_t bytes, void **h, void **d);
// Records staged in the context's pinned ring (sdvl_stage_alloc) -> device memory, queued on ctx->stream.  Round 3: a small KERNEL
// pulls them over the bus instead of a hipMemcpyAsync: a DMA copy command waits in its engine's queue behind whatever that engine
// is doing — with a farm's 79 MB image transfers under way every job-record copy (a few KB, in front of every kernel) queued up
// behind ~1.4 ms of somebody else's images, and the groups' steps ran one after the other (host-fed 115-135 k tracked frames/s with
// the transfers running, 234 k with the same steps and the transfers skipped).  Sources outside the staging ring fall back to the
// DMA copy.  SDVL_STAGE_DMA=1: always the DMA copy (A/B).
hipError_t sdvl_push(sdvl_ctx *ctx, void *dst_dev, const void *src_staged, size_t bytes);
hipError_t sdvl_pull(sdvl_ctx *ctx, void *dst_host_pinned, const void *src_dev, size_t bytes);
// wait for everything queued on ctx->stream WITHOUT spinning: a mark (sdvl_mark_record) + sleeping polls (sdvl_mark_wait).
hipError_t sdvl_stream_wait(sdvl_ctx *ctx);
// a point of the stream to wait for later: everything queued before the mark has completed once the wait returns; work
// queued after it may still be running.  kind: 0 = whole-stream waits, 1 = image alignment results, 2 = chained search,
// 3 = keyframe filter inputs
enum { SDVL_MARK_STREAM = 0, SDVL_MARK_ALIGN = 1, SDVL_MARK_CHAIN = 2, SDVL_MARK_FILTER = 3 };
hipError_t sdvl_mark_record(sdvl_ctx *ctx, int kind, uint32_t *ticket);
hipError_t sdvl_mark_wait(sdvl_ctx *ctx, int kind, uint32_t ticket);
// host copy of a frame's corner count; fetches it (blocking) when only the device knows it
int sdvl_frame_count_host(sdvl_ctx *ctx, sdvl_frame *f, int *n);
// make the device corner header match the host view before a kernel reads it (after an image change without detection)
int sdvl_frame_fix_header(sdvl_ctx *ctx, sdvl_frame *f);

// Kernel launch with optional DISPATCH timing: hipExtLaunchKernelGGL attaches the start/stop events to the kernel's
// own AQL packet, so the measured span is the dispatch itself (what rocprofv3 --kernel-trace reports), not the queueing
// behind other streams that share a hardware queue.
bool sdvl_timer_events(sdvl_ctx *ctx, const char *name, hipEvent_t *a, hipEvent_t *b);
// (Small results go from the kernels straight into the context's pinned host buffers: host-coherent memory is mapped into the
// device's address space; the stores are posted PCIe writes, visible to the host once the kernel has completed, i.e. before the
// sequence number the stream writes behind it — no device buffer + D2H copy packet.)

#define SDVL_LAUNCH(ctx, name, kernel, grid, block, ...)                                                   \
  do {                                                                                                     \
    hipEvent_t ev_a_ = nullptr, ev_b_ = nullptr;                                                           \
    sdvl_timer_events((ctx), (name), &ev_a_, &ev_b_);                                                      \
    hipExtLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, ev_a_, ev_b_, 0, __VA_ARGS__);            \
  } while (0)

#endif  // SDVL_INTERNAL_H_
