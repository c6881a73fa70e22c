// sdvl_search_types.h — device records of the search stage (sdvl_search.hip), shared with the device-resident tracking
// tables (sdvl_track.hip), which write them on the device instead of receiving them from the host.
#ifndef SDVL_SEARCH_TYPES_H_
#define SDVL_SEARCH_TYPES_H_

#include "sdvl_internal.h"
#include "sdvl_math.h"

constexpr int kWavesPerBlock = 4;  // requests of one workgroup; they all search the SAME current frame (block table)
constexpr int kLdsCorners = 4096;  // corners of the current frame staged in LDS (16 KB: 8 workgroups per CU); a frame with
                                   // more (up to SDVL_MAX_CORNERS) has the rest read from HBM / L2

struct SearchBlock {
  int first, count;  // requests [first, first + count) of the launch, count <= kWavesPerBlock
};

struct SearchFrame {
  const uint8_t *level[SDVL_MAX_LEVELS];
  int lw[SDVL_MAX_LEVELS], lh[SDVL_MAX_LEVELS];
  const int32_t *corners;
  const uint8_t *desc;
  const int32_t *n_ptr;  // device-resident corner count
  int levels;
  int bin_gw;            // corners binned by 32-px cell of level-0 coordinates (select_pack_kernel), bin_start == null: none
  const int32_t *bin_start;
  const uint2 *bin_entries;
  int bin_cells, pad_;
  int pad2_[2];
};

// (frame, pose) pairs are shared by many requests of a launch: they go into a table, requests carry two indices
struct SearchFramePose {
  SearchFrame f;
  double pose[7];
  double pad_;
};

struct SearchReqDev {
  int cur, ref;  // indices into the SearchFramePose table of the launch
  int level, fixed;
  double px[2], bearing[3];
  double idepth, idepth_std;
  double px0[2];
  uint32_t desc[8];
};

// what the scalar part of SearchPoint (matcher.cc:45-96) leaves for the wave part: one record per request
struct SearchPrep {
  int alive, slevel;
  double pxa[2], pxb[2];        // projected ends of the depth interval (pxb only for epipolar searches)
  double I00, I01, I10, I11;    // inverse of the affine warp (CreatePatch, matcher.cc:330)
  // wave-uniform constants of GetCornersInRange (matcher.cc:139-148, 92-94), computed once by the request's lane; the wave
  // kernel reads the record with scalar loads, so they live in SGPRs instead of 64 copies in VGPRs
  double nx, ny, normdist, xdiff, ydiff, vline, range, range2;
  // the corner bins the search region touches, looked up by the request's lane as well (round 4: in the wave kernel the box, its cell
  // range and the eight bin offsets were ~100 vector instructions every lane repeated for itself).  bin_mode 0: not prepared (no bins,
  // degenerate epipolar line, a region of more than four cell rows or kBinRegionCells cells: the wave kernel decides as before);
  // 1: the region lies outside the image; 2: cell rows 0..3 contribute entries [bin_e0[r], bin_e0[r] + bin_pre[r + 1] - bin_pre[r])
  int bin_mode, bin_pad_;
  int bin_e0[4], bin_pre[4];  // bin_pre[r] = entries of the rows before r + 1 (bin_pre[3] = all of them)
};

// search regions of up to this many 32-px cells go through the corner bins (their corners: ~3.4 per cell); larger ones scan the whole list
constexpr int kBinRegionCells = 320;

struct ChainFrameDev {
  int cand_begin, cand_end;
  int max_matches, rand_begin;
  int obs_begin, pad_;
  double pose[7];
  double pad2_;
};

// ---- rows of the device-resident tracking tables (sdvl_track.hip); the depth filter (sdvl_search.hip) patches them in place
constexpr int kDeleted = 0x100;                 // sdvl_track_point_stat::status bit
constexpr int kTrash = 0x200;                   // deleted by the mapper: becomes kDeleted at the end of the next step's commit
constexpr int kKeepBits = kDeleted | kTrash;
enum { kFound = 0, kNotFound = 1, kSeen = 2, kUnseen = 3 };  // Point::PointStatus (point.h)
constexpr int kPointMask = SDVL_TRACK_DUPLICATE - 1;

struct TrackPoint {  // device form of sdvl_track_point: the frame pointer replaced by its registry slot
  double P[3];
  double ipx[2];
  double ibearing[3];
  double idepth, idepth_std;
  int32_t ref, pad_;
  int32_t ilevel, fixed;
  int32_t score, n_failed;
  int32_t last_frame;
  int32_t status;  // Point::PointStatus | kDeleted | kTrash
  uint32_t desc[8];
};
static_assert(sizeof(TrackPoint) == sizeof(sdvl_track_point), "upload converts in place");
static_assert(sizeof(sdvl_track_point) == 144, "layout");
struct TrackFeat {  // a feature row of last_frame (sdvl_track_feature)
  double px[2];
  double bearing[3];
  int32_t level;
  int32_t point;
};
static_assert(sizeof(TrackFeat) == sizeof(sdvl_track_feature) && sizeof(TrackFeat) == 48, "layout");

// one tracked step of one tracker, written by sdvl_track_align and read by every kernel of the step (track_project, track_commit,
// and — round 4 — the image alignment itself: image_align_track_kernel builds its features from the table rows, no records in between)
struct TrackJobDev {
  SearchFrame cur;  // view of the new frame
  int tracker, feat_buf;
  int cur_id, last_id;
  int frame_id, max_matches;
  int n_feat, n_points;
  double last_pose[7];
  double pad_;
  const uint8_t *last_level[SDVL_MAX_LEVELS];  // pyramid of last_frame (frame1 of the alignment; same geometry as cur)
  double T0[7];                                // start of the alignment, frame2.pose * frame1.pose^-1 (image_align.cc:66)
  double pad2_;
};

// the set's point rows and the row count per tracker (sdvl_track.hip)
extern "C" TrackPoint *sdvl_track_points_device(sdvl_track_set *set, int *max_points, int *n_trackers);

// ---- launch helpers of sdvl_search.hip for callers whose records already live in HBM (no copies, no wait) --------------
// search_prepare + search_points over n_slots request records (a record with level < 0 is a dead slot) dealt to workgroups
// by the n_blocks entries of d_blocks (a block with count 0 is skipped); d_res receives one result per slot, h_res (may
// be null) the same records in pinned host memory
int sdvl_search_launch_device(sdvl_ctx *ctx, int n_slots, const SearchReqDev *d_reqs, const SearchFramePose *d_table,
                              const SearchBlock *d_blocks, int n_blocks, const sdvl_camera *cam, const sdvl_search_params *p,
                              SearchPrep *d_prep, sdvl_search_res *d_res, sdvl_search_res *h_res, bool prepared = false);
// second half of SelectPoints for n_frames trackers (feature_align.cc:105-149): d_cand_req may be null (candidate k of a
// tracker = request cand_begin + k); d_match_cand (may be null) receives, per selected match, its candidate index
// relative to cand_begin, at obs_begin + rank
int sdvl_select_matches_launch(sdvl_ctx *ctx, int n_frames, const ChainFrameDev *d_frames, const int32_t *d_cand_req,
                               const int32_t *d_cand_first, const sdvl_search_res *d_res, const double *d_req_point,
                               const sdvl_camera *cam, PoseJobDev *d_jobs, sdvl_pose_obs *d_obs, int32_t *d_nobs, int32_t *d_match_cand);
// sdvl_image_align.hip: the image alignment of a tracked step, features straight from the tracking tables (no records, no wait)
int sdvl_image_align_track_enqueue(sdvl_ctx *ctx, int n_jobs, const TrackJobDev *d_jobs, const TrackPoint *d_points, const TrackFeat *d_feats0,
                                   const TrackFeat *d_feats1, int np, int nfeat_cap, int max_nf, int levels, const sdvl_camera *cam,
                                   const sdvl_align_params *p, sdvl_align_result *d_results, int batch_size);
// batch_size: the capacity of the set the jobs come from (NOT the jobs of this step): it picks the one-wave or four-wave form for the
// set's life, so that a sequence's sums do not change order when a neighbour bootstraps or is lost
// the iteration-budget table of SelectInliers for all match counts up to max_size, resident in HBM (ctx->d_nits)
extern "C" int sdvl_ensure_nits_table(sdvl_ctx *ctx, int npoints_cfg, int max_its, int max_size);

#endif  // SDVL_SEARCH_TYPES_H_
