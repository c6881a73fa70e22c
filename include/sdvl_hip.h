/* sdvl_hip.h — C-ABI of the MI355X-native SDVL tracking front-end (libsdvl_hip.so, gfx950).
 *
 * The reference (JdeRobot/slam-SDVL) has no FFI: its hot path is reached through C++ member calls.  Each entry
 * point below names the reference call it replaces (file:line under the reference tree); INTEGRATION.md shows
 * the glue a maintainer adds inside frame.cc / image_align.cc / matcher.cc.  Plain pointers and sizes only;
 * every function returns 0 on success or a negative sdvl_status, never throws, never frees caller memory.
 *
 * Everything is batched: one call handles n frames / alignment jobs / search requests of any number of
 * independent sequences, so that a single launch fills the 256 CUs.  Frames live in HBM behind opaque handles
 * (pyramid + corner list + ORB descriptors); only small records cross PCIe.
 *
 * Poses are 7 doubles (qw,qx,qy,qz,tx,ty,tz) of Frame::pose_ (world -> camera), extra/se3.h:32-78.
 * Threading: one sdvl_ctx per host thread (tracker / mapper); frames may be shared read-only across contexts
 * of the same device once the creating context has been synchronised. */
#ifndef SDVL_HIP_H_
#define SDVL_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdvl_ctx sdvl_ctx;
typedef struct sdvl_frame sdvl_frame; /* device-resident Frame: pyramid_, corners_, descriptors_ (frame.h:155-165) */

enum sdvl_status {
  SDVL_OK = 0,
  SDVL_ERR_INVALID = -1,   /* bad argument (null pointer, size out of range) */
  SDVL_ERR_HIP = -2,       /* a HIP runtime call failed; see sdvl_last_error */
  SDVL_ERR_CAPACITY = -3,  /* an output or a per-frame capacity would overflow */
  SDVL_ERR_NO_DEVICE = -4  /* no gfx950 device visible */
};

#define SDVL_MAX_LEVELS 8
#define SDVL_MAX_CORNERS 6144        /* per frame: num_features plus the ties retainBest keeps (config C: 4000 + ~2 %) */
#define SDVL_MAX_ALIGN_FEATURES 2048 /* features per image-alignment job */
#define SDVL_CELL_KP_CAP 176         /* a 32x32 ROI holds at most 13*13 = 169 NMS-surviving corners */

typedef struct sdvl_camera { /* camera.h:34-135 pinhole part */
  double width, height, fx, fy, u0, v0;
} sdvl_camera;

typedef struct sdvl_detect_params { /* Config getters used by FastDetector (config.cc:55-85) */
  int cell_size;       /* CellSize, 32 */
  int max_fast_levels; /* MaxFastLevels, 3 */
  int fast_threshold;  /* FastThreshold, 10 */
  int margin;          /* 4+ORBSize/2 (ORB) or 1+PatchSize/2, fast_detector.cc:63-66 */
} sdvl_detect_params;

typedef struct sdvl_keypoint { /* one cv::KeyPoint of a per-cell cv::FAST call, image coordinates of its level */
  uint16_t x, y;
  uint8_t score; /* response */
  uint8_t level;
  uint16_t cell; /* row-major cell index inside the level */
} sdvl_keypoint;

typedef struct sdvl_align_feature { /* what ImageAlign reads of one frame1 feature, image_align.cc:147-160,219-236 */
  double px, py;      /* Feature::GetPosition() */
  double fx, fy, fz;  /* Feature::GetVector() */
  double depth;       /* |point->GetPosition() - frame1->GetWorldPosition()| */
  int32_t valid;      /* feature->GetPoint() && !ToDelete() */
  int32_t pad_;
} sdvl_align_feature;

typedef struct sdvl_align_job {
  const sdvl_frame *ref; /* frame1 */
  const sdvl_frame *cur; /* frame2 */
  int32_t feat_begin, feat_end; /* range in the features array */
  double T[7];           /* frame2.pose * frame1.pose^-1 (image_align.cc:66) */
} sdvl_align_job;

typedef struct sdvl_align_params { /* Config getters used by ImageAlign */
  int max_level, min_level; /* MaxAlignLevel 4, MinAlignLevel 2 */
  int max_its;              /* MaxImgAlignIts 30 */
  int patch_size;           /* AlignPatchSize 4 (only 4 is supported) */
  int fast;                 /* ComputePose(..., fast) relocalisation early-out */
} sdvl_align_params;

typedef struct sdvl_align_result {
  double T[7];      /* refined frame2.pose * frame1.pose^-1 */
  double error;     /* ImageAlign::GetError() */
  double chi2;
  int32_t n_meas;   /* n_meas_ / patch_area = ComputePose return value */
  int32_t its[SDVL_MAX_LEVELS]; /* accepted Gauss-Newton steps per level */
  int32_t stop;
  int32_t iters_run; /* ComputeResiduals evaluations over all levels (accepted + rejected), for traffic accounting */
  int32_t pad_;
} sdvl_align_result;

typedef struct sdvl_search_params { /* Config getters used by Matcher */
  int patch_size;      /* PatchSize 8 (only 8 is supported: one wave64 per 8x8 patch) */
  int max_align_its;   /* MaxAlignIts 10 */
  int search_size;     /* SearchSize 6 */
  int max_fast_levels; /* MaxFastLevels 3 */
  int margin;          /* corner margin, matcher.cc:131-134 */
  int use_orb;         /* UseORB (1: Hamming on ORB descriptors; 0: ZMSSD on 8x8 patches) */
  int lk_tree_sums;    /* 0: AlignPatch's three 64-term sums in the reference's sequential order (offsets bit-identical to the
                          CPU path); 1: wave butterfly sums (__shfl_xor tree): same iteration, different rounding order —
                          offsets agree within the path's 1e-4 tolerance, convergence decisions may differ on borderline patches */
  int pad_;
} sdvl_search_params;

typedef struct sdvl_search_req { /* arguments of Matcher::SearchPoint, matcher.cc:45-46 */
  const sdvl_frame *cur;  /* frame */
  const sdvl_frame *ref;  /* feature->GetFrame() */
  double cur_pose[7], ref_pose[7];
  double px[2];           /* feature->GetPosition() */
  double bearing[3];      /* feature->GetVector() */
  double idepth, idepth_std;
  double px0[2];          /* *px on entry (search centre for fixed points) */
  int32_t level;          /* feature->GetLevel() */
  int32_t fixed;
  uint8_t desc[32];       /* feature->GetDescriptor() */
} sdvl_search_req;

typedef struct sdvl_search_res {
  double px[2];     /* *px on return */
  int32_t found;    /* return value */
  int32_t level;    /* *flevel */
  int32_t best_corner; /* index of the winning corner in the current frame, -1 if none */
  int32_t stage;    /* 0 rejected before patch, 1 no corner matched, 2 LK not converged, 3 found */
  int32_t lk_its;   /* AlignPatch iterations executed */
  int32_t slevel;   /* GetSearchLevel() */
} sdvl_search_res;

/* ---- context ---------------------------------------------------------------------------------------------- */
int sdvl_ctx_create(int device, sdvl_ctx **out);
int sdvl_ctx_destroy(sdvl_ctx *ctx);
const char *sdvl_last_error(const sdvl_ctx *ctx);
int sdvl_ctx_synchronize(sdvl_ctx *ctx);
/* HIP's current device is per host thread (and starts at 0): a thread that did not create the context selects the context's
 * GPU with this before its first call (the reference's mapper thread, map.cc:55-71; worker threads of a batch driver).
 * Allocation sites inside the library select it themselves; this makes launches and copies of the thread follow. */
int sdvl_ctx_bind_thread(sdvl_ctx *ctx);
int sdvl_ctx_device(const sdvl_ctx *ctx);
/* diagnostics: the GPU a device pointer lives on; the GPU the context's scratch buffers were allocated on (must equal
 * sdvl_ctx_device for every thread that ever grew them) */
int sdvl_pointer_device(sdvl_ctx *ctx, const void *p, int *device);
int sdvl_ctx_scratch_device(sdvl_ctx *ctx, int *device);
void *sdvl_ctx_stream(sdvl_ctx *ctx); /* the hipStream_t every launch of this context goes to */
/* per-kernel device time (HIP events on the context stream) accumulated since the last reset; names/ms/launches */
int sdvl_ctx_timing_enable(sdvl_ctx *ctx, int on);
/* time only the launches of the kernel called `name` (as sdvl_ctx_timing_get reports it); NULL or "" = every launch.  Dispatch
 * events cost the host ~8 us per launch and, on every dispatch of a 16-stream farm, ~10 % of its throughput */
int sdvl_ctx_timing_only(sdvl_ctx *ctx, const char *name);
int sdvl_ctx_timing_get(sdvl_ctx *ctx, int cap, char (*names)[32], double *ms, int64_t *launches, int *n);
int sdvl_ctx_timing_reset(sdvl_ctx *ctx);

/* ---- Frame: pyramid + corners (frame.cc:34-56) -------------------------------------------------------------- */
int sdvl_frame_create(sdvl_ctx *ctx, int width, int height, int levels, sdvl_frame **out);
/* n frames backed by one allocation (hipMalloc is slow and synchronises); the storage is released with the context */
int sdvl_frame_create_many(sdvl_ctx *ctx, int width, int height, int levels, int n, sdvl_frame **out);
int sdvl_frame_destroy(sdvl_ctx *ctx, sdvl_frame *f);
/* pyramid_[0] = img (frame.cc:116): host image -> HBM (async on the context stream, staged through pinned memory) */
/* HBM bytes one frame of this shape occupies for as long as it lives (pyramid + corner list + descriptors + corner bins); -1 for
 * an invalid shape.  Round 3: the detection's per-cell lists are scratch of the context (sdvl_detect_scratch_bytes per frame of the
 * largest batch), no longer part of every frame: a keyframe keeps 0.76 MB at 640x480 instead of 1.43 MB, 0.50 MB with a corner
 * capacity of 1536. */
int64_t sdvl_frame_footprint(int width, int height, int levels);                       /* at the default capacity, SDVL_MAX_CORNERS */
int64_t sdvl_frame_footprint_cap(int width, int height, int levels, int max_corners);
/* corners_ capacity of the frames this context creates from now on (Frame::corners_ holds num_features plus the ties retainBest
 * keeps, fast_detector.cc:147-148: 2 x num_features is ample); detection into a frame whose list overflows reports
 * SDVL_ERR_CAPACITY through sdvl_frames_corner_counts, as for SDVL_MAX_CORNERS */
int sdvl_ctx_set_corner_capacity(sdvl_ctx *ctx, int max_corners);
int sdvl_frame_upload(sdvl_ctx *ctx, sdvl_frame *f, const uint8_t *img, int stride);
/* the same for n frames of one shape in ONE submission.  Images in pinned (device-mapped) host memory — hipHostMalloc,
 * hipHostRegister, torch pin_memory — are pulled over the bus by one gather kernel that reads the host pages directly (no DMA
 * descriptor per image: 256 images of 300 KB each cost 256 copy launches otherwise and reach about half the link rate);
 * pageable images fall back to one staged copy each.  The images must stay valid until the stream has passed this point. */
int sdvl_frames_upload(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const uint8_t *const *imgs, int stride);
/* Input ring for callers that know their next images early (a camera driver's queue, a farm of sequences): the images travel
 * on a second stream of the context while the current step computes.
 *   sdvl_ctx_prefetch_images  n host images (dense rows that follow each other in memory on both sides: ONE DMA per run; padded
 *                             rows in pinned memory: a gather kernel; pageable: staged copies) -> dev_dst[i] (width x height, dense
 *                             rows), queued on the context's copy stream; returns at once with a ticket (>= 0) or a negative status
 *   sdvl_ctx_prefetch_fence   work queued on the context's stream from now on starts after the prefetch with this ticket (one of
 *                             the last four)
 * The caller owns the destination buffers and must not prefetch into one that queued work may still read. */
int sdvl_ctx_prefetch_images(sdvl_ctx *ctx, int n, const uint8_t *const *imgs, int stride, int width, int height, void *const *dev_dst);
int sdvl_ctx_prefetch_fence(sdvl_ctx *ctx, int ticket);
/* Feed: ONE copy stream for several contexts of a GPU (a farm of tracker groups, one context per host thread), filled by ONE thread
 * of the caller's in the order the images will be needed.  hipMemcpyAsync of tens of MB keeps its calling thread for much of the
 * transfer, and transfers issued by 16 threads onto 16 streams share the link — in one queue they follow each other at the link's
 * rate.  n_slots buffers sets (caller-owned HBM) take turns:
 *   sdvl_feed_images       feeder thread: n host images -> dev_dst[i] (dense rows that follow each other on both sides: one DMA per
 *                          run), behind the slot's last sdvl_ctx_feed_release
 *   sdvl_ctx_feed_acquire  consumer (the thread that drives ctx): ctx's stream waits for the slot's transfer
 *   sdvl_ctx_feed_release  consumer: the work ctx has queued so far is the last reader of the slot's buffers
 * The caller orders the three calls of a slot among its threads; the library orders the streams. */
typedef struct sdvl_feed sdvl_feed;
int sdvl_feed_create(int device, int n_slots, sdvl_feed **out);
int sdvl_feed_destroy(sdvl_feed *f);
/* the message of the CALLING thread's last failed sdvl_feed_* call (the feeder and the consumers report independently) */
const char *sdvl_feed_last_error(const sdvl_feed *f);
/* 1: the slot's last transfer (sdvl_feed_images) has arrived in HBM, 0: still under way (a consumer that acquires it now will wait).
 * Only meaningful after a sdvl_feed_images for that slot: a slot that was never fed also reads 1 (nothing is under way). */
int sdvl_feed_slot_arrived(sdvl_feed *f, int slot);
int sdvl_feed_images(sdvl_feed *f, int slot, int n, const uint8_t *const *imgs, int stride, int width, int height, void *const *dev_dst);
int sdvl_ctx_feed_acquire(sdvl_ctx *ctx, sdvl_feed *f, int slot);
int sdvl_ctx_feed_release(sdvl_ctx *ctx, sdvl_feed *f, int slot);
/* same, image already in HBM (device pointer) */
int sdvl_frame_set_image_device(sdvl_ctx *ctx, sdvl_frame *f, const void *dev_img, int stride);
/* same without the copy: level 0 aliases the caller's HBM image (row stride == width), which must stay valid and
 * unmodified for as long as the frame is used; the next upload / set_image returns the frame to its own storage */
int sdvl_frame_borrow_image_device(sdvl_ctx *ctx, sdvl_frame *f, const void *dev_img);
/* n frames (one shape) whose level 0 aliases a caller image take the image into their own storage: one gather launch, queued on the
 * context's stream.  For frames that must outlive the caller's buffer — a frame that has become a keyframe while its image sat in
 * an input ring (Map::AddKeyframe keeps the Frame, map.cc:143-158; SearchPoint reads its pyramid for as long as its points live).
 * Frames that own their image already are skipped. */
int sdvl_frames_own_images(sdvl_ctx *ctx, int n, sdvl_frame *const *frames);
/* Frame::CreatePyramid, frame.cc:114-120: levels 1..L-1 by cv::pyrDown for n frames */
int sdvl_pyramid_build(sdvl_ctx *ctx, int n, sdvl_frame *const *frames);
/* host mirror of GetPyramid()[level] (read by the mapper / UI) */
int sdvl_frame_download_level(sdvl_ctx *ctx, const sdvl_frame *f, int level, uint8_t *out, int stride);

/* The cv::FAST(roi, kps, thr, true) calls of FastDetector::SelectPixels, fast_detector.cc:79-106, for every cell
 * of levels 0..max_fast_levels-1 of n frames.  out_kps[i*cap ..] holds frame i's keypoints grouped by
 * (level, cell) in row-major cell order and cv::FAST scan order inside a cell; out_cell_offsets[i*(ncells+1) ..]
 * are exclusive offsets over the frame's concatenated cells (ncells = sum over levels, see sdvl_fast_num_cells).
 * The quota / retainBest selection (fast_detector.cc:108-151) stays on the host. */
int sdvl_fast_num_cells(int width, int height, const sdvl_detect_params *p, int *cells_per_level, int *total);
/* bytes of context scratch one frame of a detection batch needs while the batch is in flight (per-cell lists, selection lists) */
int64_t sdvl_detect_scratch_bytes(int width, int height, const sdvl_detect_params *p);
int sdvl_fast_cells(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const sdvl_detect_params *p, int cap,
                    sdvl_keypoint *out_kps, int32_t *out_cell_offsets);

/* FastDetector::DetectPyramid (fast_detector.cc:154-175) entirely on the device for n frames: per-cell FAST, the
 * quota loop and cv::KeyPointsFilter::retainBest in libstdc++'s nth_element/partition order, levels concatenated
 * into corners_.  Asynchronous: nothing returns to the host, the corner count stays in HBM. */
int sdvl_detect_corners(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const sdvl_detect_params *p, int nfeatures);
/* corner counts of n frames (blocking, one round trip); SDVL_ERR_CAPACITY if a device capacity overflowed */
int sdvl_frames_corner_counts(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int32_t *counts);
/* host mirror of corners_ (GetCorners()): xyl = [n][3], *n_out = count */
int sdvl_frame_download_corners(sdvl_ctx *ctx, sdvl_frame *f, int cap, int32_t *xyl, int *n_out);
/* diagnostic: cv::KeyPointsFilter::retainBest on packed keypoints (response in the top byte) run by the device code;
 * cooperative = 0: one lane, 1: the whole workgroup with barriers (len <= 4096), 2: a group of 8 lanes — the form the
 * selection kernel uses per cell (len <= SDVL_CELL_KP_CAP), 3: one wave — the form it uses per level (len <= 4096) */
int sdvl_retain_best(sdvl_ctx *ctx, uint32_t *packed, int len, int n_points, int cooperative, int *out_len);

/* corners_ of a frame (x, y in level coordinates, level), fast_detector.cc:151 -> HBM; descriptors are invalidated */
int sdvl_frame_set_corners(sdvl_ctx *ctx, sdvl_frame *f, int n, const int32_t *xyl);
/* same for n frames in one transfer: counts[i] corners of frame i, concatenated in xyl */
int sdvl_frames_set_corners(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const int32_t *counts, const int32_t *xyl);
int sdvl_frame_num_corners(const sdvl_frame *f);
int sdvl_frame_corner_capacity(const sdvl_frame *f); /* corners its resident list holds (sdvl_ctx_set_corner_capacity) */

/* FindShiTomasiScoreAtPoint for every corner of n frames (extra/utils.cc:61-97, called from
 * FastDetector::FilterCorners fast_detector.cc:205); out_scores[i*cap + k]; the grid logic stays on the host */
int sdvl_shi_tomasi(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, double *out_scores);

/* ORBDetector::GetDescriptor for every corner of n frames (orb_detector.cc:350-395).  Descriptors stay in HBM; out_desc
 * (may be NULL) = [n][cap][32].  The reference computes descriptors lazily (matcher.cc:266-269: the corners a search
 * compares; frame.cc:148-161: the filtered corners of a keyframe), and so does this library: the calls that read a
 * frame's descriptors (sdvl_filter_inputs, sdvl_frame_download_descriptors) run this first if it has not been, and a
 * search on a frame without descriptors computes exactly the ones it compares.  Calling it up front is never wrong —
 * the values are the same — only more work. */
int sdvl_orb_describe(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, uint8_t *out_desc);
/* host mirror of descriptors_ (GetDescriptors()): out = [n_corners][32] */
int sdvl_frame_download_descriptors(sdvl_ctx *ctx, const sdvl_frame *f, int cap, uint8_t *out);
/* Everything Frame::FilterCorners (frame.cc:133-163) needs from n frames in ONE round trip: corner lists, their
 * Shi-Tomasi scores (K3 launched here) and ORB descriptors.  xyl[n][cap][3], scores[n][cap], desc[n][cap][32]
 * (desc may be NULL), counts[n]. */
int sdvl_filter_inputs(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, int32_t *xyl, double *scores, uint8_t *desc,
                       int32_t *counts);
/* The same in two halves: _begin queues the kernels and the result copy, _end waits for exactly that and unpacks.  Between
 * the two the caller may do host work, but no other call on this context that returns data to the host. */
int sdvl_filter_inputs_begin(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, int with_desc);
int sdvl_filter_inputs_end(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, int32_t *xyl, double *scores, uint8_t *desc,
                           int32_t *counts);
/* Frame::FilterCorners (frame.cc:133-163) for n frames INCLUDING the selection of FastDetector::FilterCorners
 * (fast_detector.cc:177-218): per frame the corners outside the margin or in a locked cell are dropped, every other cell
 * keeps the corner the reference's sequential scan would keep (a later corner wins when its Shi-Tomasi score exceeds the
 * TRUNCATED score stored for the cell), cells whose score stays <= min_feature_score are dropped, and the survivors come
 * back in cell order with their ORB descriptors (the reference computes descriptors for the filtered corners only).
 * locked_cells[i][mask_words]: bit c of frame i's mask = grid cell c already holds a feature (FastDetector::LockCell on the
 * frame's features, frame.cc:139-142); grid = ceil(width / cell_size) x ceil(height / cell_size), at most 4096 cells.
 * _begin queues everything, _end waits and returns counts[n] and records out[n][cap] (cap >= the cell count is always enough). */
typedef struct sdvl_filtered_corner {
  int32_t index;        /* position in the frame's corner list (Frame::GetFilteredCorners() holds these) */
  int32_t x, y, level;  /* the corner, level coordinates */
  int32_t score;        /* the cell's stored score: (int) Shi-Tomasi score */
  int32_t pad_;
  uint8_t desc[32];
} sdvl_filtered_corner;
int sdvl_filter_corners_begin(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const uint32_t *locked_cells, int mask_words, int cell_size,
                              int margin, int min_feature_score, int with_desc);
int sdvl_filter_corners_end(sdvl_ctx *ctx, int n, int cap, int32_t *counts, sdvl_filtered_corner *out);
/* ORBDetector::Distance (extra/orb_detector.cc:398-410) batched with the arg-min of Matcher::SearchFeatures
 * (matcher.cc:254-289): query i (32 B) is compared with the candidates cand_desc[cand_offsets[i] .. cand_offsets[i+1]);
 * best_index[i] = position within that list of the FIRST smallest Hamming distance, or -1 when that distance is not below
 * `threshold` (MIN_ORB_THRESHOLD = 100, matcher.h:37) or the list is empty (strict '<' between candidates, '>= threshold' means
 * not found, as matcher.cc:280-288); best_dist[i] (may be NULL) = min(smallest distance, threshold + 1): the true minimum whenever
 * it is <= threshold (so distance == threshold comes back with index -1), threshold + 1 for an empty list or when every candidate
 * is further away.  All pointers are host memory; blocking. */
int sdvl_hamming_argmin(sdvl_ctx *ctx, int n, const uint8_t *queries, const int32_t *cand_offsets, const uint8_t *cand_desc,
                        int threshold, int32_t *best_index, int32_t *best_dist);
/* ORBDetector::GetDescriptor at arbitrary (x,y,level) points of one frame; out_angle_deg may be NULL */
int sdvl_orb_describe_points(sdvl_ctx *ctx, const sdvl_frame *f, int n, const int32_t *xyl, uint8_t *out_desc,
                             float *out_angle_deg);

/* ---- ImageAlign::ComputePose, image_align.cc:46-84 (Optimize :86-125, ComputeResiduals :127-206,
 * PrecomputePatches :208-267): n_jobs independent frame pairs, whole coarse-to-fine Gauss-Newton on device ------ */
int sdvl_image_align(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, int n_features,
                     const sdvl_align_feature *features, const sdvl_camera *cam, const sdvl_align_params *p,
                     sdvl_align_result *out);
/* The same in two halves: _begin launches and queues the result copy, _end waits for exactly that.  Between the two the
 * caller may queue work on the context that returns nothing to the host (sdvl_pyramid_build, sdvl_detect_corners,
 * sdvl_orb_describe): it runs behind the alignment while the host already has the poses. */
int sdvl_image_align_begin(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, int n_features,
                           const sdvl_align_feature *features, const sdvl_camera *cam, const sdvl_align_params *p);
int sdvl_image_align_end(sdvl_ctx *ctx, int n_jobs, sdvl_align_result *out);
/* SDVL::Relocalize (sdvl.cc:205-238) calls ImageAlign::ComputePose(keyframe, current, fast) for EVERY keyframe of the map at every
 * frame a tracker stays lost: frame1's feature records are the same from one frame to the next.  A store keeps them in HBM:
 * _create reserves `capacity` records, _write fills records [offset, offset + n) (ordered behind the context's queued work, complete
 * on return), and sdvl_image_align_begin_stored is sdvl_image_align_begin with the jobs' feat_begin / feat_end naming records of the
 * store - per frame only the job records cross the link.  Results through sdvl_image_align_end as usual. */
typedef struct sdvl_align_store sdvl_align_store;
int sdvl_align_store_create(sdvl_ctx *ctx, int capacity, sdvl_align_store **out);
int sdvl_align_store_destroy(sdvl_ctx *ctx, sdvl_align_store *store);
int sdvl_align_store_write(sdvl_ctx *ctx, sdvl_align_store *store, int offset, int n, const sdvl_align_feature *features);
int sdvl_image_align_begin_stored(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, const sdvl_align_store *store,
                                  const sdvl_camera *cam, const sdvl_align_params *p);

/* ---- Matcher::SearchPoint, matcher.cc:45-121 (WarpMatrixAffine :293-312, GetSearchLevel :314-323,
 * CreatePatch :325-357, GetCornersInRange :123-230, SearchFeatures :232-291, AlignPatch :359-445).
 * The current frames need corners (sdvl_frame_set_corners / sdvl_detect_corners).  With use_orb the descriptors of a
 * current frame are read from HBM if sdvl_orb_describe has run on it; otherwise every wave computes the descriptor of
 * each corner it finds in range (matcher.cc:266-269), about a fifth of a frame's corners in a tracking step. -- */
int sdvl_search_points(sdvl_ctx *ctx, int n, const sdvl_search_req *reqs, const sdvl_camera *cam,
                       const sdvl_search_params *p, sdvl_search_res *out);

/* The same search for callers that issue thousands of requests per call (SDVLBatch): the requests are written once,
 * in the device layout, straight into the context's pinned staging area, and name their frames by slot.
 *   sdvl_search_begin  reserves room for max_requests records and returns the array to fill;
 *   sdvl_search_slot   registers a (frame, pose) pair of this batch and returns its slot (>= 0), or a negative status;
 *                      the frame must have corners as for sdvl_search_points;
 *   sdvl_search_run    evaluates the first n records.
 * No other call on this context between begin and run. */
typedef struct sdvl_search_req_packed {
  int32_t cur, ref;      /* slots of the current frame and of feature->GetFrame() */
  int32_t level, fixed;
  double px[2];
  double bearing[3];
  double idepth, idepth_std;
  double px0[2];
  uint8_t desc[32];
} sdvl_search_req_packed;
int sdvl_search_begin(sdvl_ctx *ctx, int max_requests, sdvl_search_req_packed **reqs);
int sdvl_search_slot(sdvl_ctx *ctx, const sdvl_frame *frame, const double *pose7);
int sdvl_search_run(sdvl_ctx *ctx, int n, const sdvl_camera *cam, const sdvl_search_params *p, sdvl_search_res *out);

/* ---- the mapper's depth filter behind the search (Map::UpdateCandidates, map.cc:402-498) ------------------------------
 * One request per candidate that passed the visibility / baseline tests (map.cc:421-452).  After search_points the device
 * does, per request, what the body of the loop does with the result (map.cc:454-497):
 *   not found                -> Point::Unpromote (point.cc:109-118): n_failed + 1, b + 1, deleted when n_failed > max_failed
 *   found                    -> GetDepthFromTriangulation (extra/utils.cc:193-205), the point's world position, GetParallax
 *                               (utils.cc:207-213; >= 0.999999 skips), the two minimum-depth tests (map.cc:476-479),
 *                               Point::Update (point.cc:64-100: ComputeTau, the Gaussian x uniform posterior of inverse depth,
 *                               parallax and distance of the new estimate) and Point::HasConverged (point.cc:164-178)
 * and returns the point's new filter state.  With a tracking table set, the row of a point that a tracker follows
 * (track_row >= 0) is patched in place — position, inverse depth and its deviation, fixed, n_failed, and "deleted after the next
 * step" (Map::DeletePoint takes effect in the next frame's EmptyTrash, sdvl.cc:127) — so the table stays valid across the
 * mapper's update.  acos / sin / exp are the device library's: results agree with the host arithmetic to the last bits, not
 * bit for bit (tolerance class of the poses, 1e-4). */
struct sdvl_track_set; /* the tracking tables, below */
typedef struct sdvl_depth_state { /* the Point behind a request (point.h:136-150) */
  double rho, sigma2, a, b, z_range;
  double cos_alpha, last_distance; /* of the previous update: HasConverged reads them when Update returns early (NaN) */
  double depth_mean;   /* Frame::GetSceneDepth() of the map's current frame (map.cc:92) */
  double position[3];  /* p3d_ of a point that is fixed already (the second list entry of a point that converged in the first
                          pass of this frame, map.cc:381,389: Update and HasConverged still run on it) */
  int32_t fixed;
  int32_t n_failed;
  int32_t track_row;   /* tracker * max_points + index of the point's row in the tracking tables, -1: none */
  int32_t pad_;
} sdvl_depth_state;

typedef struct sdvl_depth_params {
  double px_error_angle;  /* Camera::GetPixelErrorAngle, camera.h:104-107 */
  double min_depth;       /* Config::MapScale() * Config::ScaleMinDist(), map.cc:476 */
  double scale_min_dist;  /* Config::ScaleMinDist(), map.cc:478 */
  int32_t max_failed;     /* Config::MaxFailed() */
  int32_t pad_;
} sdvl_depth_params;

#define SDVL_DEPTH_NOT_FOUND 0 /* Unpromote */
#define SDVL_DEPTH_SKIPPED 1   /* found, but no triangulation / no parallax / too close: the point is left alone */
#define SDVL_DEPTH_UPDATED 2   /* Point::Update ran */
#define SDVL_DEPTH_CONVERGED 3 /* ... and Point::HasConverged fixed the point at `position` */
#define SDVL_DEPTH_FIXED_STALE 4 /* Update returned early (norm_scale is NaN, point.cc:76) and HasConverged fixed the point at
                                    the position of its unchanged estimate */
#define SDVL_DEPTH_DELETED 0x100 /* | NOT_FOUND: n_failed crossed max_failed (Map::DeletePoint) */
typedef struct sdvl_depth_out {
  int32_t outcome;
  int32_t n_failed;
  double rho, sigma2, a, b;
  double cos_alpha, last_distance; /* valid when UPDATED / CONVERGED */
  double position[3];              /* Point::GetPosition() after the update (UPDATED / CONVERGED) */
} sdvl_depth_out;

/* requests as for sdvl_search_points; state[n]; set may be NULL (no table to patch); out[n], fout[n] */
int sdvl_search_points_filter(sdvl_ctx *ctx, int n, const sdvl_search_req *reqs, const sdvl_camera *cam, const sdvl_search_params *p,
                              const sdvl_depth_state *state, const sdvl_depth_params *fp, struct sdvl_track_set *set,
                              sdvl_search_res *out, sdvl_depth_out *fout);
/* the same behind sdvl_search_begin / sdvl_search_slot */
int sdvl_search_run_filter(sdvl_ctx *ctx, int n, const sdvl_camera *cam, const sdvl_search_params *p, const sdvl_depth_state *state,
                           const sdvl_depth_params *fp, struct sdvl_track_set *set, sdvl_search_res *out, sdvl_depth_out *fout);

/* Matcher::AlignPatch alone, matcher.cc:359-445: n patches against level images of frames.
 * border[n][100], patch[n][64], uv_io[n][2] (level coordinates), converged[n], its[n] (may be NULL) */
int sdvl_align_patches(sdvl_ctx *ctx, int n, const sdvl_frame *const *frames, const int32_t *levels,
                       const uint8_t *border, const uint8_t *patch, int max_its, double *uv_io,
                       uint8_t *converged, int32_t *its);

/* ---- cooperative waits ----------------------------------------------------------------------------------------------
 * Every entry point that returns results waits for the context's stream.  By default the calling thread sleeps on a
 * blocking event.  With a hook installed the wait polls instead and calls hook(user, ctx) while the stream is busy: a
 * host thread that drives several contexts (one per group of sequences) switches to another context's host work there
 * and comes back when sdvl_ctx_wait_done(ctx) says 1; when nothing else is runnable it sleeps in sdvl_ctx_wait_block. */
int sdvl_ctx_set_wait_hook(sdvl_ctx *ctx, void (*hook)(void *user, sdvl_ctx *ctx), void *user);
/* Round 5: waits of this context poll WITHOUT sleeping for the first `microseconds` (then fall back to the sleeping polls).  For a
 * context whose waits are a lone camera's 0.2-ms chains (SDVL::HandleFrame sets 500): a sleeping poll wakes ~15 us late on average. */
int sdvl_ctx_set_wait_spin(sdvl_ctx *ctx, int microseconds);
int sdvl_ctx_wait_done(sdvl_ctx *ctx);
int sdvl_ctx_wait_block(sdvl_ctx *ctx);
/* 0 while the stream is healthy (idle or busy), SDVL_ERR_HIP after a device fault: for schedulers that poll _wait_done */
int sdvl_ctx_health(sdvl_ctx *ctx);
/* Which path the work of this context took, counted on the host (a fast path must be SEEN to run: the tracked step scanned whole
 * corner lists for a round and a half without failing anything).  out4[0] = tracked jobs (sdvl_track_search) whose search reads the
 * corner bins, out4[1] = tracked jobs whose search scans the frame's whole corner list, out4[2] = undistortion maps built
 * (sdvl_undistort / sdvl_frames_upload_undistorted build one per camera, not per call), out4[3] reserved (0). */
int sdvl_ctx_counters(sdvl_ctx *ctx, int64_t *out4);

/* ---- input stage: Camera::UndistortImage = cv::undistort(in, out, K, D) (camera.cc:39-67,100-105, main.cc:133) ----
 * d[0..4] = Camera.d1..d5 of the config = (k1, k2, p1, p2, k3).  As in the reference, d[0] == 0 means "no distortion":
 * the image is copied unchanged (Camera::SetDistortions tests d0 only, camera.cc:46). */
typedef struct sdvl_distortion {
  double d[5];
} sdvl_distortion;

/* n images of one camera: src[i] (host pointers, or device pointers if src_on_device) -> dst_dev[i] (device, distinct
 * from src).  Asynchronous on the context's stream like every other producer of frame data. */
int sdvl_undistort(sdvl_ctx *ctx, int n, const void *const *src, int src_stride, int src_on_device, int width, int height,
                   const sdvl_camera *cam, const sdvl_distortion *dist, void *const *dst_dev, int dst_stride);
/* the fused form main.cc:133-135 needs: raw camera images -> undistorted level 0 of the frames (replaces
 * sdvl_frame_upload / sdvl_frame_set_image_device for a camera with distortion); follow with sdvl_pyramid_build */
int sdvl_frames_upload_undistorted(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const void *const *src, int src_stride,
                                   int src_on_device, const sdvl_camera *cam, const sdvl_distortion *dist);

/* ---- pose from matches: FeatureAlign::SelectInliers + OptimizePose (feature_align.cc:73-82,152-243,258-283,341-431) --
 * One job per frame.  obs[] are the frame's matched features in found order: ax, ay = feature bearing x/z, y/z
 * (feature_align.cc:268), p = 3D point position, inv_cov = 1 / (1 << level).  The reference draws rand() % size once
 * per RANSAC iteration and shrinks its iteration budget from log() of the supporter ratio: the caller passes
 *   rand_idx[rand_begin .. rand_begin + max_ransac_its)   the draws the reference WOULD make (a copy of the stream), and
 *   nits_table[nits_begin + s], s = 0..size               the budget after an improvement to s supporters (:199-207),
 * and commits result.n_draws draws on its real stream afterwards.  out_lists[obs_begin ..) receives the final inlier
 * indices (n_inliers of them, reference order) followed by the outlier indices (n_outliers), relative to obs_begin.
 * At most 1024 observations per job (SDVL_ERR_CAPACITY beyond). */
typedef struct sdvl_pose_obs {
  double ax, ay;
  double px, py, pz;
  double inv_cov;
} sdvl_pose_obs;

typedef struct sdvl_pose_job {
  int32_t obs_begin, obs_end;
  int32_t rand_begin, nits_begin;
  double pose[7]; /* frame pose on entry (q0 q1 q2 q3 tx ty tz) */
} sdvl_pose_job;

typedef struct sdvl_pose_params {
  int32_t max_ransac_points; /* Config::MaxRansacPoints(), <= 8 */
  int32_t max_ransac_its;    /* Config::MaxRansacIts() */
  int32_t max_optim_pose_its;/* Config::MaxOptimPoseIts() */
  int32_t pad_;
  double inlier_threshold;   /* Config::InlierErrorThreshold() / fx */
  double fx;
} sdvl_pose_params;

typedef struct sdvl_pose_result {
  double pose[7];
  int32_t n_draws;    /* rand() calls SelectInliers made */
  int32_t n_inliers, n_outliers;
  int32_t refined;    /* ConvergePose ran on a non-empty inlier set at least once (pose was set) */
} sdvl_pose_result;

int sdvl_pose_from_matches(sdvl_ctx *ctx, int n_jobs, const sdvl_pose_job *jobs, int n_obs, const sdvl_pose_obs *obs,
                           int n_rand, const int32_t *rand_idx, int n_nits, const int32_t *nits_table,
                           const sdvl_pose_params *p, sdvl_pose_result *results, int32_t *out_lists);

/* ---- FeatureAlign::Reproject + OptimizePose in one submission (feature_align.cc:59-82): sdvl_search_run, then ON THE
 * DEVICE the second half of SelectPoints (feature_align.cc:105-149: candidates cell by cell in the caller's order, a cell's
 * first found candidate is a match, at most max_matches of them) and sdvl_pose_from_matches on the matches it selects.
 * The host still needs the search results for its own bookkeeping (features, point statistics), so
 *   sdvl_search_run_chain  returns as soon as they are on the host, with selection and pose kernels still in flight;
 *   sdvl_search_chain_end  waits for the pose results: results[n_frames], n_obs[n_frames] (matches selected, = what the
 *                          host's replay of SelectPoints finds) and lists — tracker f's inlier then outlier indices start at
 *                          sum of max_matches of the trackers before it (indices relative to its own match list).
 * cand_req[k]  = request the k-th candidate was searched with (index into the batch) or -1 (never found);
 * cand_first[k]= index (into cand_req) of the first candidate of k's cell;  req_point[r] = Point::GetPosition() of request r;
 * rand_raw     = per tracker the next max_ransac_its values of its rand() stream (the device reduces them modulo the match
 *                count, feature_align.cc:180); the iteration budget table of :199-207 is built inside, with libm log.
 * No other call on this context between the two. */
typedef struct sdvl_chain_frame {
  int32_t cand_begin, cand_end; /* this tracker's candidates */
  int32_t max_matches;          /* FeatureAlign max_matches_, <= 1024 */
  int32_t rand_begin;           /* first of its max_ransac_its values in rand_raw */
  double pose[7];               /* frame pose on entry to the pose stage */
} sdvl_chain_frame;
int sdvl_search_run_chain(sdvl_ctx *ctx, int n, const sdvl_camera *cam, const sdvl_search_params *p, sdvl_search_res *out, int n_frames,
                          const sdvl_chain_frame *frames, int n_cand, const int32_t *cand_req, const int32_t *cand_first,
                          const double *req_point, int n_rand, const int32_t *rand_raw, const sdvl_pose_params *pp);
int sdvl_search_chain_end(sdvl_ctx *ctx, int n_frames, sdvl_pose_result *results, int32_t *n_obs, int32_t *lists);

/* ---- device-resident tracking state: SDVL::ProcessFrame without the host in the loop (sdvl.cc:175-203) ----------------
 * What FeatureAlign::Reproject reads every frame — the features of last_frame and the points behind them
 * (feature_align.cc:296-339: position, score; :105-125: the point's first observation, inverse depth, descriptor) — lives in
 * HBM, one table per tracker, and one submission does for n trackers what ProcessFrame does between the Frame constructor
 * and the keyframe decision:
 *   ImageAlign::ComputePose(last_frame, frame)                                        image_align.cc:46-84
 *   FeatureAlign::Reproject: ProjectPoints, the shuffled cell order, the per-cell sort by Point::Score, SearchPoint for every
 *     candidate, first hit per cell, max_matches, Promote / Unpromote / MaxFailed     feature_align.cc:59-150,285-339
 *   SelectInliers + OptimizePose + RemoveOutliers                                      feature_align.cc:73-82,152-283
 * and leaves the new frame's features (the matches) in the tracker's OTHER feature buffer; the caller decides afterwards
 * whether the frame replaces last_frame (sdvl.cc:104-123) by naming the buffer in the next job.  rand() stays on the host:
 * the caller passes the cell order it shuffled and the next max_ransac_its raw draws, and advances its stream by n_draws.
 * Per tracker and frame ~1 KB goes to the device and ~100 B + the match list come back; no Feature / Point object is touched.
 * A table is (re)built from host objects with sdvl_track_upload (keyframes: new points appear, bootstrap, relocalisation).
 * All frames named by a table or a job must have been created on the table's context. */
typedef struct sdvl_track_set sdvl_track_set;

typedef struct sdvl_track_point { /* what the tracker reads of one Point (point.h) */
  double position[3];     /* Point::GetPosition() */
  double px[2];           /* GetInitFeature()->GetPosition() */
  double bearing[3];      /* GetInitFeature()->GetVector() */
  double idepth, idepth_std; /* GetInverseDepth(), GetStd() */
  const sdvl_frame *ref;  /* GetInitFeature()->GetFrame(), registered with sdvl_frame_register */
  int32_t level, fixed;   /* GetInitFeature()->GetLevel(), IsFixed() */
  int32_t score, n_failed;/* n_successful_ (Score()), n_failed_ */
  int32_t last_frame;     /* GetLastFrame() */
  int32_t status;         /* GetStatus() as int */
  uint8_t desc[32];       /* GetInitFeature()->GetDescriptor() */
} sdvl_track_point;

typedef struct sdvl_track_feature { /* one Feature of last_frame (feature.h) */
  double px[2];       /* GetPosition() */
  double bearing[3];  /* GetVector() */
  int32_t level;
  int32_t point;      /* index into the tracker's points, -1 = no point (or ToDelete); | SDVL_TRACK_DUPLICATE when an
                         earlier feature of the list has the same point (ProjectPoints skips it, feature_align.cc:310) */
} sdvl_track_feature;
#define SDVL_TRACK_DUPLICATE 0x40000000

typedef struct sdvl_track_params {
  sdvl_align_params align;
  sdvl_search_params search;
  sdvl_pose_params pose;
  int32_t cell_size;   /* Config::CellSize(): FeatureAlign's grid */
  int32_t patch_size;  /* Config::PatchSize(): margin of ProjectPoint's IsInsideImage */
  int32_t max_failed;  /* Config::MaxFailed(): Point::Unpromote */
  int32_t pad_;
} sdvl_track_params;

typedef struct sdvl_track_job {
  int32_t tracker;        /* table */
  int32_t feat_buf;       /* which of the tracker's two feature buffers holds last_frame's features; results go to the other */
  const sdvl_frame *last; /* last_frame */
  sdvl_frame *cur;        /* the new frame: pyramid built; corners detected before sdvl_track_search */
  double T[7];            /* start of the alignment, frame2.pose * frame1.pose^-1 (image_align.cc:66) */
  double last_pose[7];    /* last_frame->GetPose() */
  int32_t frame_id;       /* frame->GetID() (Point::SetLastFrame) */
  int32_t max_matches;    /* FeatureAlign max_matches_ */
} sdvl_track_job;

typedef struct sdvl_track_result {
  double pose[7];         /* frame->GetPose() after OptimizePose */
  double align_error;     /* ImageAlign::GetError() */
  double scene_depth;     /* Frame::GetSceneDepth() of the new frame (frame.cc:70-92): median camera-frame depth of its points */
  int32_t align_meas, align_iters; /* ComputePose return value; ComputeResiduals evaluations */
  int32_t n_features;     /* features of last_frame the alignment ran on */
  int32_t n_requests;     /* SearchPoint evaluations (all candidates, speculatively) */
  int32_t matches, attempts;       /* GetMatches(), GetAttempts() */
  int32_t n_draws, n_inliers, n_outliers, refined; /* as sdvl_pose_result */
  int32_t n_points;       /* Frame::GetNumPoints() of the new frame */
  int32_t n_deleted;      /* points that crossed MaxFailed this frame (Map::DeletePoint) */
  int32_t lk_iters;       /* AlignPatch iterations over all requests */
  int32_t n_corners;      /* corners of the new frame (Frame::GetCorners().size()) */
  int32_t status;         /* 0, or SDVL_ERR_CAPACITY if a table capacity was exceeded (results invalid) */
} sdvl_track_result;

typedef struct sdvl_track_feature_out { /* a feature of the new frame as the host needs it to build a Feature */
  double px[2];
  int32_t level;
  int32_t point;   /* index into the tracker's points, -1 after RemoveOutliers */
} sdvl_track_feature_out;

typedef struct sdvl_track_point_stat { /* the mutable part of a point after the step */
  int32_t score, n_failed, last_frame;
  int32_t status;  /* Point::PointStatus | 0x100 when the point was deleted (crossed MaxFailed) */
} sdvl_track_point_stat;

int sdvl_track_create(sdvl_ctx *ctx, int n_trackers, int max_points, int max_features, int grid_cells, int max_matches,
                      int max_ransac_its, sdvl_track_set **out);
int sdvl_track_destroy(sdvl_ctx *ctx, sdvl_track_set *set);
/* (frame, pose) becomes addressable by tables: keyframes when their pose is final, the bootstrap frame */
int sdvl_frame_register(sdvl_ctx *ctx, const sdvl_frame *frame, const double *pose7);
/* the same for n frames (poses7 = [n][7]) in one submission */
int sdvl_frames_register(sdvl_ctx *ctx, int n, const sdvl_frame *const *frames, const double *poses7);
/* replaces the tables of n trackers: trackers[i] gets n_points[i] points and n_features[i] features (concatenated arrays),
 * its features go to buffer feat_buf[i].  Asynchronous. */
int sdvl_track_upload(sdvl_ctx *ctx, sdvl_track_set *set, int n, const int32_t *trackers, const int32_t *feat_buf, const int32_t *n_points,
                      const sdvl_track_point *points, const int32_t *n_features, const sdvl_track_feature *features);
/* the same rows APPENDED behind the rows the tables hold (the step that has just run left trackers[i]'s new features in buffer
 * feat_buf[i]): n_points[i] new points take the next point rows, n_features[i] new features the next feature rows of that buffer;
 * features name points by their index in the whole table.  What a keyframe's seeded points (Map::InitCandidates, map.cc:262-400,
 * with the plane-map stand-in) add to a table that is otherwise current on the device.  Asynchronous. */
int sdvl_track_append(sdvl_ctx *ctx, sdvl_track_set *set, int n, const int32_t *trackers, const int32_t *feat_buf, const int32_t *n_points,
                      const sdvl_track_point *points, const int32_t *n_features, const sdvl_track_feature *features);
/* first third of a step: alignment of n_jobs frames against their last frames, queued.  cell_rank[j][grid_cells] = position
 * of every grid cell in tracker j's shuffled cell_order_; rand_raw[j][max_ransac_its] = its next rand() values. */
int sdvl_track_align(sdvl_ctx *ctx, sdvl_track_set *set, int n_jobs, const sdvl_track_job *jobs, const uint16_t *cell_rank,
                     const int32_t *rand_raw, const sdvl_camera *cam, const sdvl_track_params *p);
/* second third: reprojection, search, match selection, pose, table update, queued behind whatever the caller queued in
 * between (sdvl_detect_corners on the new frames) */
int sdvl_track_search(sdvl_ctx *ctx, sdvl_track_set *set);
/* last third: wait, results[n_jobs].  Until the set's next sdvl_track_align, sdvl_track_features / _stats return the new
 * frame's features (result.matches of them) and the tracker's point statistics (n_points of them) */
int sdvl_track_collect(sdvl_ctx *ctx, sdvl_track_set *set, int n_jobs, sdvl_track_result *results);
const sdvl_track_feature_out *sdvl_track_features(const sdvl_track_set *set, int job);
const sdvl_track_point_stat *sdvl_track_stats(const sdvl_track_set *set, int job);

/* ---- synthetic sequence generator (no dataset ships with the repo) ------------------------------- */
struct sdvl_synth_view;
/* renders n views of the textured plane straight into HBM: dev_out + i*frame_bytes, row stride = width */
int sdvl_synth_render(sdvl_ctx *ctx, int n, const struct sdvl_synth_view *views, int width, int height,
                      void *dev_out, int64_t frame_bytes);
int sdvl_device_malloc(sdvl_ctx *ctx, int64_t bytes, void **out);
int sdvl_device_free(sdvl_ctx *ctx, void *p);
int sdvl_device_download(sdvl_ctx *ctx, const void *dev, int64_t bytes, void *host);
/* Fork / join inside one context (round 5).  Work queued between _fork_begin and _fork_end runs on a side stream that starts behind
 * the point of the main stream _fork_mark recorded and is joined into the main stream by _fork_end: everything queued afterwards waits
 * for both.  A lone camera's tracked frame uses it (host/standalone.cc): Frame::CreateCorners' FAST + selection (frame.cc:122-131,
 * reached from the Frame constructor, sdvl.cc:59) needs the pyramid only, so it runs beside ImageAlign::ComputePose (sdvl.cc:189)
 * instead of behind it.  One fork at a time; results are identical (the same kernels on the same inputs). */
int sdvl_ctx_fork_mark(sdvl_ctx *ctx);
int sdvl_ctx_fork_begin(sdvl_ctx *ctx);
int sdvl_ctx_fork_end(sdvl_ctx *ctx);
/* Page-locked host memory the GPU reads where it lies.  SDVL::HandleFrame(const cv::Mat&) (sdvl.cc:55-59) is handed a host image:
 * when that image lives in memory from sdvl_host_alloc_pinned — or in a buffer the caller registered once with
 * sdvl_host_register, e.g. the capture buffers behind the cv::Mat of main.cc:131 — sdvl_frames_upload pulls it over the link
 * with a kernel (no staging copy, no DMA queue); a pageable image goes through hipMemcpyAsync's bounce buffer instead. */
int sdvl_host_alloc_pinned(sdvl_ctx *ctx, int64_t bytes, void **out);
int sdvl_host_free_pinned(sdvl_ctx *ctx, void *p);
int sdvl_host_register(sdvl_ctx *ctx, void *p, int64_t bytes);
int sdvl_host_unregister(sdvl_ctx *ctx, void *p);

#ifdef __cplusplus
}
#endif
#endif /* SDVL_HIP_H_ */
