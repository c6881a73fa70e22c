// sdvl_track.hip — device-resident tracking tables: what FeatureAlign::Reproject reads every frame (the features of
// last_frame and the points behind them) lives in HBM, and the per-frame work the host used to do around the kernels —
// ProjectPoints, the grid, the per-cell sort by score, request assembly (feature_align.cc:88-150,285-339), the replay of
// SelectPoints' bookkeeping (Promote / Unpromote / MaxFailed), RemoveOutliers (:245-256) and the new frame's feature list —
// runs in three small kernels around the existing ones:
//
//   [image_align]      sdvl_image_align.hip, results stay in HBM
//   track_project      pose = T * last_pose; ProjectPoints; candidates ordered as SelectPoints visits them (cell by cell in
//                      the caller's shuffled order, inside a cell by score, stable) -> request records, cell starts, block
//                      table, the pose stage's frame record; the (frame, pose) registry entry of the new frame
//   [search_prepare, search_points, select_matches, pose_hypotheses, pose_refine]
//   track_commit       which candidates SelectPoints would have tried; Promote / Unpromote / delete; the matches become the
//                      new frame's features (other buffer); outliers lose their point; results + mirrors to pinned host memory
//
// One workgroup per tracker in the three kernels; nothing here needs a workgroup-to-workgroup hand-off.
// Compiled with -ffp-contract=off like everything else: the arithmetic is the host layer's, statement for statement.
#include <atomic>
#include <chrono>
#include <vector>

#include "sdvl_internal.h"
#include "sdvl_math.h"
#include "sdvl_search_types.h"
#include "sdvl_search_prepare.h"

namespace {

using namespace sdvl;


struct UploadRec {  // sdvl_track_upload / sdvl_track_append: where one tracker's staged rows go
  int tracker, feat_buf;
  int n_points, n_feat;
  long long point_off, feat_off;  // element offsets into the staged arrays
  int point_dst, feat_dst;        // first row written in the tracker's tables (0: the table is replaced; > 0: rows are appended)
  long long pad_;
};

struct RegisterRec {  // sdvl_frame_register
  SearchFramePose e;
  int id, pad_;
};

// bins_pending: the frame's corners (and their bins) are produced by a detection that is queued BEHIND this call and ahead of the
// kernels that will read the view (sdvl_track_align -> sdvl_detect_corners -> sdvl_track_search): the view names the frame's bin
// arrays although the host does not call them valid yet.  sdvl_track_search checks that the detection did come (track_clear_bins).
void fill_view(SearchFrame *d, const sdvl_frame *f, bool bins_pending = false) {
  memset(d, 0, sizeof(SearchFrame));
  for (int l = 0; l < f->v.levels; l++) {
    d->level[l] = f->v.level[l];
    d->lw[l] = f->v.lw[l];
    d->lh[l] = f->v.lh[l];
  }
  d->corners = f->v.corners;
  d->desc = f->desc_valid ? f->v.desc : nullptr;  // null: a search computes the descriptors it compares (search_points_kernel)
  d->n_ptr = f->v.corner_hdr;
  d->levels = f->v.levels;
  if (f->bins_valid || (bins_pending && f->bin_cells > 0)) {
    d->bin_start = f->bin_start;
    d->bin_entries = f->bin_entries;
    d->bin_gw = f->bin_gw;
    d->bin_cells = f->bin_cells;
  }
}

// the rare case behind fill_view's bins_pending: current frames whose corners did not come out of sdvl_detect_corners (set by hand)
// have no bins: their views go back to "scan the whole corner list"
__global__ __launch_bounds__(64) void track_clear_bins_kernel(TrackJobDev *__restrict__ jobs, const uint8_t *__restrict__ no_bins, int n) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j < n && no_bins[j]) {
    jobs[j].cur.bin_start = nullptr;
    jobs[j].cur.bin_entries = nullptr;
    jobs[j].cur.bin_gw = 0;
    jobs[j].cur.bin_cells = 0;
  }
}

__global__ __launch_bounds__(64) void registry_write_kernel(const RegisterRec *__restrict__ recs, int n, SearchFramePose *__restrict__ registry) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) registry[recs[i].id] = recs[i].e;
}

__global__ __launch_bounds__(256) void track_upload_kernel(const UploadRec *__restrict__ recs, const TrackPoint *__restrict__ src_points,
                                                           const TrackFeat *__restrict__ src_feats, TrackPoint *__restrict__ points,
                                                           TrackFeat *__restrict__ feats0, TrackFeat *__restrict__ feats1, int np, int nf) {
  const UploadRec r = recs[blockIdx.x];
  // rows are multiples of 16 bytes: copy them as such
  const uint4 *sp = reinterpret_cast<const uint4 *>(src_points + r.point_off);
  uint4 *dp = reinterpret_cast<uint4 *>(points + static_cast<size_t>(r.tracker) * np + r.point_dst);
  const int pw = r.n_points * static_cast<int>(sizeof(TrackPoint) / 16);
  for (int i = threadIdx.x; i < pw; i += 256) dp[i] = sp[i];
  const uint4 *sf = reinterpret_cast<const uint4 *>(src_feats + r.feat_off);
  uint4 *df = reinterpret_cast<uint4 *>((r.feat_buf ? feats1 : feats0) + static_cast<size_t>(r.tracker) * nf + r.feat_dst);
  const int fw = r.n_feat * static_cast<int>(sizeof(TrackFeat) / 16);
  for (int i = threadIdx.x; i < fw; i += 256) df[i] = sf[i];
}

constexpr int kProjThreads = 512;

// ProjectPoints + the first half of SelectPoints (feature_align.cc:88-118,285-339)
__global__ __launch_bounds__(kProjThreads) void track_project_kernel(const TrackJobDev *__restrict__ jobs, TrackPoint *__restrict__ points,
                                                                     const TrackFeat *__restrict__ feats0, const TrackFeat *__restrict__ feats1,
                                                                     int np, int nf, int stride, int mm, int max_its,
                                                                     const sdvl_align_result *__restrict__ ares,
                                                                     const uint16_t *__restrict__ cell_rank, int cells, Cam cam, int cell_size,
                                                                     int patch, SearchFramePose *__restrict__ registry,
                                                                     SearchReqDev *__restrict__ reqs, double *__restrict__ req_point,
                                                                     int32_t *__restrict__ cand_first, int32_t *__restrict__ cand_feat,
                                                                     SearchBlock *__restrict__ blocks, ChainFrameDev *__restrict__ chain,
                                                                     sdvl_search_params sprm, SearchPrep *__restrict__ prep) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  unsigned long long *s_key = reinterpret_cast<unsigned long long *>(s_dyn);  // [stride] sort key of feature i, ~0 = not a candidate
  double *s_px = reinterpret_cast<double *>(s_key + stride);                   // [stride][2] its projection
  uint16_t *s_cell = reinterpret_cast<uint16_t *>(s_px + 2 * static_cast<size_t>(stride));  // [stride] cell of the candidate at sorted position k
  __shared__ double s_pose[7];
  __shared__ int s_ncand;
  const int j = blockIdx.x, tid = threadIdx.x;
  const TrackJobDev &jb = jobs[j];
  const TrackFeat *F = (jb.feat_buf ? feats1 : feats0) + static_cast<size_t>(jb.tracker) * nf;
  TrackPoint *P = points + static_cast<size_t>(jb.tracker) * np;
  const int n_feat = jb.n_feat;
  const int base = j * stride;
  if (tid == 0) {
    // frame2.pose = T * frame1.pose (image_align.cc:79)
    const Rigid pose = se3_mul(se3_from7(ares[j].T), se3_from7(jb.last_pose));
    se3_to7(pose, s_pose);
    s_ncand = 0;
    SearchFramePose e;
    e.f = jb.cur;
    se3_to7(pose, e.pose);
    e.pad_ = 0.0;
    registry[jb.cur_id] = e;
  }
  __syncthreads();
  const Rigid pose = se3_from7(s_pose);
  const M3 R = se3_rot(pose);
  const int gw = static_cast<int>(ceil(cam.width / cell_size));
  for (int i = tid; i < n_feat; i += static_cast<int>(blockDim.x)) {
    unsigned long long key = ~0ull;
    const int praw = F[i].point;
    if (praw >= 0 && !(praw & SDVL_TRACK_DUPLICATE)) {
      TrackPoint &p = P[praw];
      if (!(p.status & kDeleted) && p.last_frame != jb.frame_id) {
        // ProjectPoint, feature_align.cc:317-339; Frame::Project, frame.cc:94-103
        const V3 rel = vadd(mvec(R, {p.P[0], p.P[1], p.P[2]}), pose.t);
        bool ok = !(rel.z < 0.0);
        V2 px = {0.0, 0.0};
        if (ok) {
          px = cam_project(cam, rel);
          ok = cam_inside(cam, static_cast<int>(px.x), static_cast<int>(px.y), patch);
        }
        if (!ok) {
          p.status = (p.status & kTrash) | kUnseen;
        } else {
          const int k = static_cast<int>(px.y / cell_size) * gw + static_cast<int>(px.x / cell_size);
          const unsigned sc = static_cast<unsigned>(p.score < 0 ? 0 : (p.score > 0xFFFFFF ? 0xFFFFFF : p.score));
          // cells in the shuffled order; inside a cell by descending score, ties in list order (std::list::sort is stable)
          key = (static_cast<unsigned long long>(k < cells ? cell_rank[static_cast<size_t>(j) * cells + k] : 0xFFFFu) << 48) |
                (static_cast<unsigned long long>(0xFFFFFFu - sc) << 24) | static_cast<unsigned>(i);
          s_px[2 * i] = px.x;
          s_px[2 * i + 1] = px.y;
          p.status = (p.status & kTrash) | kSeen;
        }
        p.last_frame = jb.frame_id;  // not relocalising (the tables are only used for ordinary tracking)
      }
    }
    s_key[i] = key;
    if (key != ~0ull) atomicAdd(&s_ncand, 1);
  }
  __syncthreads();
  const int n_cand = s_ncand;
  // position of every candidate in visiting order = number of smaller keys (keys are unique: they end in the feature index)
  for (int i = tid; i < n_feat; i += static_cast<int>(blockDim.x)) {
    const unsigned long long key = s_key[i];
    if (key == ~0ull) continue;
    int r = 0;
    for (int q = 0; q < n_feat; q++) r += s_key[q] < key ? 1 : 0;
    const TrackPoint &p = P[F[i].point];
    SearchReqDev rq;
    rq.cur = jb.cur_id;
    rq.ref = p.ref;
    rq.level = p.ilevel;
    rq.fixed = p.fixed;
    rq.px[0] = p.ipx[0]; rq.px[1] = p.ipx[1];
    rq.bearing[0] = p.ibearing[0]; rq.bearing[1] = p.ibearing[1]; rq.bearing[2] = p.ibearing[2];
    rq.idepth = p.idepth;
    rq.idepth_std = p.idepth_std;
    rq.px0[0] = s_px[2 * i]; rq.px0[1] = s_px[2 * i + 1];
#pragma unroll
    for (int w = 0; w < 8; w++) rq.desc[w] = p.desc[w];
    reqs[base + r] = rq;
    // phase 0 of SearchPoint for the request this lane has just assembled (search_prepare_kernel's work, sdvl_search_prepare.h);
    // the reference frame's pose comes out of the registry, the current frame's is the one this workgroup has just computed
    if (prep) prep[base + r] = search_prepare_one(rq, pose, se3_from7(registry[p.ref].pose), cam, sprm, &jb.cur);  // (the bins were filled earlier in this stream)
    double *rp = req_point + 3 * static_cast<size_t>(base + r);
    rp[0] = p.P[0]; rp[1] = p.P[1]; rp[2] = p.P[2];
    cand_feat[base + r] = i;
    s_cell[r] = static_cast<uint16_t>(key >> 48);
  }
  __syncthreads();
  for (int k = tid; k < stride; k += static_cast<int>(blockDim.x)) {
    if (k < n_cand) {
      int first = k;
      const uint16_t c = s_cell[k];
      while (first > 0 && s_cell[first - 1] == c) first--;
      cand_first[base + k] = base + first;
    } else {
      reqs[base + k].level = -1;  // dead slot: search_prepare / search_points skip it
      if (prep) {
        prep[base + k].alive = 0;
        prep[base + k].slevel = -1;
      }
    }
  }
  const int nblk = stride / kWavesPerBlock;
  for (int b = tid; b < nblk; b += static_cast<int>(blockDim.x)) {
    int cnt = n_cand - b * kWavesPerBlock;
    cnt = cnt < 0 ? 0 : (cnt > kWavesPerBlock ? kWavesPerBlock : cnt);
    blocks[j * nblk + b] = SearchBlock{base + b * kWavesPerBlock, cnt};
  }
  if (tid == 0) {
    ChainFrameDev c;
    c.cand_begin = base;
    c.cand_end = base + n_cand;
    c.max_matches = jb.max_matches;
    c.rand_begin = j * max_its;
    c.obs_begin = j * mm;
    c.pad_ = 0;
#pragma unroll
    for (int q = 0; q < 7; q++) c.pose[q] = s_pose[q];
    c.pad2_ = 0.0;
    chain[j] = c;
  }
}

// second half of SelectPoints as bookkeeping (feature_align.cc:105-149), RemoveOutliers (:245-256), the new feature list
__global__ __launch_bounds__(256) void track_commit_kernel(const TrackJobDev *__restrict__ jobs, TrackPoint *__restrict__ points,
                                                           const TrackFeat *__restrict__ feats0_c, const TrackFeat *__restrict__ feats1_c,
                                                           TrackFeat *__restrict__ feats0, TrackFeat *__restrict__ feats1, int np, int nf, int stride,
                                                           int mm, const ChainFrameDev *__restrict__ chain, const int32_t *__restrict__ cand_first,
                                                           const int32_t *__restrict__ cand_feat, const sdvl_search_res *__restrict__ res,
                                                           const sdvl_pose_result *__restrict__ pres, const int32_t *__restrict__ lists,
                                                           const sdvl_align_result *__restrict__ ares, Cam cam, int max_failed,
                                                           SearchFramePose *__restrict__ registry, sdvl_track_result *__restrict__ h_results,
                                                           sdvl_track_feature_out *__restrict__ h_feats, sdvl_track_point_stat *__restrict__ h_stats) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  uint16_t *s_before = reinterpret_cast<uint16_t *>(s_dyn);  // [stride + 1] matches selected among the candidates before k
  uint8_t *s_found = reinterpret_cast<uint8_t *>(s_before + stride + 1);  // [stride]
  double *s_depth = reinterpret_cast<double *>(s_dyn + (static_cast<size_t>(stride + 1) * 2 + stride + 64 + 7) / 8 * 8);  // [mm]
  __shared__ int s_wave[4];
  __shared__ int s_attempts, s_deleted, s_lk, s_npoints;
  __shared__ double s_median;
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TrackJobDev &jb = jobs[j];
  const ChainFrameDev &fr = chain[j];
  const int base = fr.cand_begin, n = fr.cand_end - fr.cand_begin;
  const TrackFeat *F = (jb.feat_buf ? feats1_c : feats0_c) + static_cast<size_t>(jb.tracker) * nf;
  TrackFeat *N = (jb.feat_buf ? feats0 : feats1) + static_cast<size_t>(jb.tracker) * nf;  // the OTHER buffer
  TrackPoint *P = points + static_cast<size_t>(jb.tracker) * np;
  if (tid == 0) { s_attempts = 0; s_deleted = 0; s_lk = 0; s_npoints = 0; }
  int lk = 0;
  for (int k = tid; k < n; k += 256) {
    const sdvl_search_res &r = res[base + k];
    s_found[k] = r.found != 0 ? 1 : 0;
    lk += r.lk_its;
  }
  __syncthreads();
  if (lk) atomicAdd(&s_lk, lk);
  // a cell's first found candidate is selected; s_before[k] = selected candidates before k
  int running = 0;
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int k = c0 + tid;
    bool sel = false;
    if (k < n && s_found[k]) {
      sel = true;
      for (int q = cand_first[base + k] - base; q < k; q++)
        if (s_found[q]) { sel = false; break; }
    }
    const unsigned long long m = __ballot(sel);
    const int below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
    __syncthreads();  // s_wave of the previous round has been read
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int b = running;
    for (int w = 0; w < wave; w++) b += s_wave[w];
    if (k < n) s_before[k] = static_cast<uint16_t>(b + below);
    running += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
  __syncthreads();
  const int matches = running < jb.max_matches ? running : jb.max_matches;
  // what the sequential loop does with candidate k: its cell is visited while fewer than max_matches matches exist; inside
  // the cell the candidates are tried until the first hit
  for (int k = tid; k < n; k += 256) {
    const int c0 = cand_first[base + k] - base;
    const bool visited = s_before[c0] < jb.max_matches;
    const bool tried = visited && s_before[k] == s_before[c0];
    if (!tried) continue;
    atomicAdd(&s_attempts, 1);
    const int fi = cand_feat[base + k];
    const int pt = F[fi].point;
    TrackPoint &p = P[pt];
    if (s_found[k]) {
      const sdvl_search_res &r = res[base + k];
      p.score += 1;      // Point::Promote
      p.n_failed = 0;
      p.status = (p.status & kKeepBits) | kFound;
      TrackFeat nf_;
      nf_.px[0] = r.px[0]; nf_.px[1] = r.px[1];
      const V3 v = cam_unproject(cam, {r.px[0], r.px[1]});  // Feature::Feature, feature.cc:28-35
      nf_.bearing[0] = v.x; nf_.bearing[1] = v.y; nf_.bearing[2] = v.z;
      nf_.level = r.level;
      nf_.point = pt;
      N[s_before[k]] = nf_;
    } else {
      p.n_failed += 1;   // Point::Unpromote; beyond MaxFailed the map deletes the point (feature_align.cc:141-142)
      int st = kNotFound;
      if (p.n_failed > max_failed && !(p.status & kDeleted)) {
        st |= kDeleted | (p.status & kTrash);
        atomicAdd(&s_deleted, 1);
      } else {
        st |= p.status & kKeepBits;
      }
      p.status = st;
    }
  }
  __syncthreads();
  // RemoveOutliers, feature_align.cc:245-256: the pose stage's outlier list indexes the matches
  const sdvl_pose_result pr = pres[j];
  const int32_t *lst = lists + fr.obs_begin;
  for (int q = tid; q < pr.n_outliers; q += 256) {
    const int r = lst[pr.n_inliers + q];
    const int pt = N[r].point;
    if (pt >= 0) {
      N[r].point = -1;
      P[pt].status = (P[pt].status & kKeepBits) | kNotFound;
    }
  }
  __syncthreads();
  int kept = 0;
  const Rigid final_pose = se3_from7(pr.pose);
  for (int r = tid; r < matches; r += 256) {
    const TrackFeat f = N[r];
    sdvl_track_feature_out o;
    o.px[0] = f.px[0]; o.px[1] = f.px[1];
    o.level = f.level;
    o.point = f.point;
    h_feats[static_cast<size_t>(j) * nf + r] = o;
    kept += f.point >= 0 ? 1 : 0;
    // Frame::GetSceneDepth, frame.cc:70-92: GetRelativePos(point position)(2) of every feature that has a point
    double z = __builtin_nan("");  // no point: compares false with everything below
    if (f.point >= 0) z = se3_apply(final_pose, {P[f.point].P[0], P[f.point].P[1], P[f.point].P[2]}).z;
    s_depth[r] = z;
  }
  if (kept) atomicAdd(&s_npoints, kept);
  if (tid == 0) s_median = 0.0;
  __syncthreads();
  {
    // GetMedianVector (extra/utils.cc:215-220): nth_element at floor(n / 2) = the value of sorted rank n / 2; a value v sits
    // there when fewer than or exactly n / 2 values are smaller and more than n / 2 are smaller or equal
    const int n_pts = s_npoints, mid = n_pts / 2;
    for (int r = tid; r < matches; r += 256) {
      const double v = s_depth[r];
      if (v != v) continue;
      int less = 0, leq = 0;
      for (int q = 0; q < matches; q++) {
        const double u = s_depth[q];
        less += u < v ? 1 : 0;
        leq += u <= v ? 1 : 0;
      }
      if (less <= mid && mid < leq) s_median = v;  // every writer holds the same value
    }
  }
  for (int q = tid; q < jb.n_points; q += 256) {
    TrackPoint &p = P[q];
    // a point the mapper deleted during the previous frame's update dies now: Map::DeletePoint only queues it, the queue is
    // emptied after the next frame has been tracked (sdvl.cc:127) — this frame still saw it
    if (p.status & kTrash) p.status = (p.status & ~kTrash) | kDeleted;
    sdvl_track_point_stat s;
    s.score = p.score; s.n_failed = p.n_failed; s.last_frame = p.last_frame; s.status = p.status;
    h_stats[static_cast<size_t>(j) * np + q] = s;
  }
  __syncthreads();
  if (tid == 0) {
    sdvl_track_result out;
#pragma unroll
    for (int q = 0; q < 7; q++) out.pose[q] = pr.pose[q];
    out.align_error = ares[j].error;
    out.scene_depth = s_median;
    out.align_meas = ares[j].n_meas;
    out.align_iters = ares[j].iters_run;
    out.n_features = jb.n_feat;
    out.n_requests = n;
    out.matches = matches;
    out.attempts = s_attempts;
    out.n_draws = pr.n_draws;
    out.n_inliers = pr.n_inliers;
    out.n_outliers = pr.n_outliers;
    out.refined = pr.refined;
    out.n_points = s_npoints;
    out.n_deleted = s_deleted;
    out.lk_iters = s_lk;
    out.n_corners = jb.cur.n_ptr[0];  // the new frame's corner count (header written by the detection's pack kernel)
    out.status = 0;
    h_results[j] = out;
#pragma unroll
    for (int q = 0; q < 7; q++) registry[jb.cur_id].pose[q] = pr.pose[q];  // the frame's final pose, for when it becomes a reference
  }
}

}  // namespace

struct sdvl_track_set {
  sdvl_ctx *ctx;
  int n, np, nf, cells, mm, max_its;
  // tables
  TrackPoint *d_points = nullptr;
  TrackFeat *d_feats[2] = {nullptr, nullptr};
  // per-step scratch, all sized for n jobs
  uint8_t *d_scratch = nullptr;
  size_t scratch_bytes = 0;
  TrackJobDev *d_jobs;
  uint16_t *d_cell_rank;
  int32_t *d_rand;
  sdvl_align_result *d_ares;
  SearchReqDev *d_reqs;
  SearchPrep *d_prep;
  sdvl_search_res *d_res;
  double *d_reqpt;
  int32_t *d_cfirst, *d_cfeat;
  SearchBlock *d_blocks;
  ChainFrameDev *d_chain;
  PoseJobDev *d_pjobs;
  sdvl_pose_obs *d_obs;
  void *d_hyp;
  sdvl_pose_result *d_pres;
  int32_t *d_lists, *d_nobs;
  // pinned host mirrors, written by track_commit
  uint8_t *h_pinned = nullptr;
  sdvl_track_result *h_results;
  sdvl_track_feature_out *h_feats;
  sdvl_track_point_stat *h_stats;
  // host view of the tables
  std::vector<int> n_points;
  std::vector<int> n_feat[2];
  // the step in flight
  std::vector<sdvl_track_job> jobs;
  int stride = 0;
  int phase = 0;  // 0 idle, 1 aligned, 2 searched
  uint32_t ticket = 0;
  // SDVL_STEP_GRAPH=1 (A/B): the search -> pose -> commit chain of a step as a HIP graph.  The chain is captured
  // every step (its scalars and launch geometry follow the step's feature counts), the instantiated graph is UPDATED in place with
  // the new capture (same topology) and launched as one submission.
  sdvl_camera cam;
  sdvl_track_params prm;
};

namespace {
template <typename T>
T *carve(uint8_t *&p, size_t count) {
  T *r = reinterpret_cast<T *>(p);
  p += (sizeof(T) * count + 255) / 256 * 256;
  return r;
}
}  // namespace

extern "C" {

int sdvl_track_create(sdvl_ctx *ctx, int n, int max_points, int max_features, int grid_cells, int max_matches, int max_ransac_its,
                      sdvl_track_set **out) {
  if (!ctx || !out) return SDVL_ERR_INVALID;
  *out = nullptr;
  SDVL_REQUIRE(ctx, n >= 1 && n <= 65536, "tracker count out of range");
  SDVL_REQUIRE(ctx, max_points >= 1 && max_points < SDVL_TRACK_DUPLICATE && max_features >= 4 && max_features <= 4096, "table capacities out of range (features <= 4096)");
  SDVL_REQUIRE(ctx, grid_cells >= 1 && grid_cells <= 65535, "grid too large");
  SDVL_REQUIRE(ctx, max_matches >= 1 && max_matches <= 1024 && max_matches <= max_features, "max_matches outside the device pose stage's range (1024)");
  SDVL_REQUIRE(ctx, max_ransac_its >= 1 && max_ransac_its <= 4096, "bad max_ransac_its");
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  sdvl_track_set *s = new sdvl_track_set();
  s->ctx = ctx;
  s->n = n;
  s->np = max_points;
  s->nf = (max_features + 3) / 4 * 4;
  s->cells = grid_cells;
  s->mm = max_matches;
  s->max_its = max_ransac_its;
  const size_t N = static_cast<size_t>(n), NF = N * s->nf;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&s->d_points), sizeof(TrackPoint) * N * s->np);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_feats[0]), sizeof(TrackFeat) * NF);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_feats[1]), sizeof(TrackFeat) * NF);
  // scratch: computed by carving a null base first
  for (int pass = 0; pass < 2 && e == hipSuccess; pass++) {
    uint8_t *p = pass ? s->d_scratch : nullptr;
    s->d_jobs = carve<TrackJobDev>(p, N);
    s->d_cell_rank = carve<uint16_t>(p, N * s->cells);
    s->d_rand = carve<int32_t>(p, N * s->max_its);
    s->d_ares = carve<sdvl_align_result>(p, N);
    s->d_reqs = carve<SearchReqDev>(p, NF);
    s->d_prep = carve<SearchPrep>(p, NF);
    s->d_res = carve<sdvl_search_res>(p, NF);
    s->d_reqpt = carve<double>(p, NF * 3);
    s->d_cfirst = carve<int32_t>(p, NF);
    s->d_cfeat = carve<int32_t>(p, NF);
    s->d_blocks = carve<SearchBlock>(p, NF / kWavesPerBlock);
    s->d_chain = carve<ChainFrameDev>(p, N);
    s->d_pjobs = carve<PoseJobDev>(p, N);
    s->d_obs = carve<sdvl_pose_obs>(p, N * s->mm);
    s->d_hyp = carve<uint8_t>(p, sdvl_pose_hyp_bytes() * N * s->max_its);
    s->d_pres = carve<sdvl_pose_result>(p, N);
    s->d_lists = carve<int32_t>(p, N * s->mm);
    s->d_nobs = carve<int32_t>(p, N);
    if (!pass) {
      s->scratch_bytes = reinterpret_cast<size_t>(p);
      e = hipMalloc(reinterpret_cast<void **>(&s->d_scratch), s->scratch_bytes);
    }
  }
  if (e == hipSuccess) {
    const size_t hb = (sizeof(sdvl_track_result) * N + 255) / 256 * 256, fb = (sizeof(sdvl_track_feature_out) * NF + 255) / 256 * 256;
    const size_t sb = sizeof(sdvl_track_point_stat) * N * s->np;
    e = hipHostMalloc(reinterpret_cast<void **>(&s->h_pinned), hb + fb + sb, hipHostMallocDefault);
    if (e == hipSuccess) {
      memset(s->h_pinned, 0, hb + fb + sb);
      s->h_results = reinterpret_cast<sdvl_track_result *>(s->h_pinned);
      s->h_feats = reinterpret_cast<sdvl_track_feature_out *>(s->h_pinned + hb);
      s->h_stats = reinterpret_cast<sdvl_track_point_stat *>(s->h_pinned + hb + fb);
    }
  }
  if (e != hipSuccess) {
    ctx->err = std::string("sdvl_track_create: ") + hipGetErrorString(e);
    sdvl_track_destroy(ctx, s);
    return SDVL_ERR_HIP;
  }
  s->n_points.assign(n, 0);
  s->n_feat[0].assign(n, 0);
  s->n_feat[1].assign(n, 0);
  *out = s;
  return SDVL_OK;
}

TrackPoint *sdvl_track_points_device(sdvl_track_set *set, int *max_points, int *n_trackers) {
  if (max_points) *max_points = set->np;
  if (n_trackers) *n_trackers = set->n;
  return set->d_points;
}

int sdvl_track_destroy(sdvl_ctx *ctx, sdvl_track_set *s) {
  if (!ctx || !s) return SDVL_ERR_INVALID;
  (void)sdvl_bind_device(ctx);
  (void)sdvl_stream_wait(ctx);
  if (s->d_points) (void)hipFree(s->d_points);
  if (s->d_feats[0]) (void)hipFree(s->d_feats[0]);
  if (s->d_feats[1]) (void)hipFree(s->d_feats[1]);
  if (s->d_scratch) (void)hipFree(s->d_scratch);
  if (s->h_pinned) (void)hipHostFree(s->h_pinned);
  delete s;
  return SDVL_OK;
}

int sdvl_frame_register(sdvl_ctx *ctx, const sdvl_frame *f, const double *pose7) { return sdvl_frames_register(ctx, 1, &f, pose7); }

// n frames in one push + one launch (a group of trackers registers ~10 reference frames per step when keyframes are created:
// one at a time that was two dispatches each on the step's critical path)
int sdvl_frames_register(sdvl_ctx *ctx, int n, const sdvl_frame *const *frames, const double *poses7) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !poses7))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  for (int i = 0; i < n; i++)
    SDVL_REQUIRE(ctx, frames[i] && frames[i]->home == ctx && frames[i]->reg_id >= 0 && frames[i]->reg_id < ctx->registry_cap,
                 "frame was not created on this context");
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_stage_alloc(ctx, sizeof(RegisterRec) * static_cast<size_t>(n), &hs, &dsx);
  if (rc) return rc;
  RegisterRec *r = static_cast<RegisterRec *>(hs);
  for (int i = 0; i < n; i++) {
    fill_view(&r[i].e.f, frames[i]);
    memcpy(r[i].e.pose, poses7 + 7 * i, sizeof(double) * 7);
    r[i].e.pad_ = 0.0;
    r[i].id = frames[i]->reg_id;
    r[i].pad_ = 0;
  }
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, sizeof(RegisterRec) * static_cast<size_t>(n)));
  SDVL_LAUNCH(ctx, "registry_write", registry_write_kernel, dim3((n + 63) / 64), dim3(64), static_cast<const RegisterRec *>(dsx), n,
              static_cast<SearchFramePose *>(ctx->d_registry));
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

// append == false: the tables of the trackers are replaced; true: the rows go behind the rows the tables hold (round 4: a keyframe of
// the plane-map configuration adds its ~60 seeded points and their features to the tables the step has just updated on the device,
// instead of the host rebuilding all ~250 rows from Feature / Point objects)
static int track_upload_rows(sdvl_ctx *ctx, sdvl_track_set *s, int n, const int32_t *trackers, const int32_t *feat_buf, const int32_t *n_points,
                             const sdvl_track_point *points, const int32_t *n_features, const sdvl_track_feature *features, bool append) {
  if (!ctx || !s || s->ctx != ctx || n < 0 || (n > 0 && (!trackers || !feat_buf || !n_points || !n_features))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, s->phase == 0, "sdvl_track_upload while a step is in flight");
  size_t tp = 0, tf = 0;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, trackers[i] >= 0 && trackers[i] < s->n && (feat_buf[i] == 0 || feat_buf[i] == 1), "bad tracker / buffer index");
    const int p0 = append ? s->n_points[trackers[i]] : 0, f0 = append ? s->n_feat[feat_buf[i]][trackers[i]] : 0;
    if (n_points[i] < 0 || p0 + n_points[i] > s->np || n_features[i] < 0 || f0 + n_features[i] > s->nf) {
      ctx->err = "table larger than the set's capacity";
      return SDVL_ERR_CAPACITY;
    }
    for (int j = 0; j < i; j++) SDVL_REQUIRE(ctx, trackers[j] != trackers[i], "a tracker is named twice in one upload");
    tp += n_points[i];
    tf += n_features[i];
  }
  SDVL_REQUIRE(ctx, (tp == 0 || points) && (tf == 0 || features), "null rows");
  const size_t rb = (sizeof(UploadRec) * n + 255) / 256 * 256, pb = (sizeof(TrackPoint) * tp + 255) / 256 * 256, fb = sizeof(TrackFeat) * tf;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_stage_alloc(ctx, rb + pb + fb, &hs, &dsx);
  if (rc) return rc;
  uint8_t *h8 = static_cast<uint8_t *>(hs), *d8 = static_cast<uint8_t *>(dsx);
  UploadRec *recs = reinterpret_cast<UploadRec *>(h8);
  TrackPoint *hp = reinterpret_cast<TrackPoint *>(h8 + rb);
  size_t po = 0, fo = 0;
  for (int i = 0; i < n; i++) {
    const int p0 = append ? s->n_points[trackers[i]] : 0, f0 = append ? s->n_feat[feat_buf[i]][trackers[i]] : 0;
    recs[i] = UploadRec{trackers[i], feat_buf[i], n_points[i], n_features[i], static_cast<long long>(po), static_cast<long long>(fo), p0, f0, 0};
    for (int k = 0; k < n_points[i]; k++) {
      const sdvl_track_point &src = points[po + k];
      SDVL_REQUIRE(ctx, src.ref && src.ref->home == ctx && src.ref->reg_id >= 0, "a point's reference frame was not created on this context");
      SDVL_REQUIRE(ctx, src.level >= 0 && src.level < src.ref->v.levels, "feature level outside the reference pyramid");
      SDVL_REQUIRE(ctx, src.idepth == src.idepth && src.idepth != 0.0, "inverse depth must be finite and non-zero");
      TrackPoint &d = hp[po + k];
      memcpy(&d, &src, sizeof(TrackPoint));  // same layout except for the frame pointer
      d.ref = src.ref->reg_id;
      d.pad_ = 0;
    }
    for (int k = 0; k < n_features[i]; k++) {
      const int pt = features[fo + k].point;
      SDVL_REQUIRE(ctx, pt < 0 || (pt & kPointMask) < p0 + n_points[i], "feature names a point outside its tracker's table");
    }
    po += n_points[i];
    fo += n_features[i];
  }
  if (fb) memcpy(h8 + rb + pb, features, fb);
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, rb + pb + fb));
  SDVL_LAUNCH(ctx, "track_upload", track_upload_kernel, dim3(n), dim3(256), reinterpret_cast<const UploadRec *>(d8),
              reinterpret_cast<const TrackPoint *>(d8 + rb), reinterpret_cast<const TrackFeat *>(d8 + rb + pb), s->d_points, s->d_feats[0],
              s->d_feats[1], s->np, s->nf);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  // every record was valid and the rows are on their way: only now do the host-side counts follow (a failed call leaves them untouched)
  for (int i = 0; i < n; i++) {
    s->n_points[trackers[i]] = recs[i].point_dst + n_points[i];
    s->n_feat[feat_buf[i]][trackers[i]] = recs[i].feat_dst + n_features[i];
  }
  return SDVL_OK;
}

int sdvl_track_upload(sdvl_ctx *ctx, sdvl_track_set *s, int n, const int32_t *trackers, const int32_t *feat_buf, const int32_t *n_points,
                      const sdvl_track_point *points, const int32_t *n_features, const sdvl_track_feature *features) {
  return track_upload_rows(ctx, s, n, trackers, feat_buf, n_points, points, n_features, features, false);
}

int sdvl_track_append(sdvl_ctx *ctx, sdvl_track_set *s, int n, const int32_t *trackers, const int32_t *feat_buf, const int32_t *n_points,
                      const sdvl_track_point *points, const int32_t *n_features, const sdvl_track_feature *features) {
  return track_upload_rows(ctx, s, n, trackers, feat_buf, n_points, points, n_features, features, true);
}

int sdvl_track_align(sdvl_ctx *ctx, sdvl_track_set *s, int n_jobs, const sdvl_track_job *jobs, const uint16_t *cell_rank, const int32_t *rand_raw,
                     const sdvl_camera *cam, const sdvl_track_params *p) {
  if (!ctx || !s || s->ctx != ctx || n_jobs <= 0 || !jobs || !cell_rank || !rand_raw || !cam || !p) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, s->phase == 0, "sdvl_track_align while a step is in flight");
  SDVL_REQUIRE(ctx, n_jobs <= s->n, "more jobs than trackers");
  SDVL_REQUIRE(ctx, p->pose.max_ransac_points >= 1 && p->pose.max_ransac_points <= 8, "max_ransac_points must be in [1,8]");
  SDVL_REQUIRE(ctx, p->pose.max_ransac_its == s->max_its && p->pose.max_optim_pose_its >= 0, "max_ransac_its differs from the set's");
  SDVL_REQUIRE(ctx, p->cell_size >= 1 && p->patch_size >= 0 && p->max_failed >= 0, "bad track parameters");
  {
    const int gw = static_cast<int>(ceil(cam->width / p->cell_size)), gh = static_cast<int>(ceil(cam->height / p->cell_size));
    SDVL_REQUIRE(ctx, gw * gh == s->cells, "the set's grid does not match camera / cell size");
  }
  int max_nf = 0;
  for (int j = 0; j < n_jobs; j++) {
    const sdvl_track_job &a = jobs[j];
    SDVL_REQUIRE(ctx, a.tracker >= 0 && a.tracker < s->n && (a.feat_buf == 0 || a.feat_buf == 1), "bad tracker / buffer index");
    SDVL_REQUIRE(ctx, a.last && a.cur && a.last->home == ctx && a.cur->home == ctx, "frames of a job must be created on the set's context");
    SDVL_REQUIRE(ctx, a.max_matches >= 0 && a.max_matches <= s->mm, "max_matches above the set's");
    for (int q = 0; q < j; q++) SDVL_REQUIRE(ctx, jobs[q].tracker != a.tracker, "a tracker appears twice in one step");
    const int nf = s->n_feat[a.feat_buf][a.tracker];
    if (nf > max_nf) max_nf = nf;
  }
  for (int k = 0; k < n_jobs * s->cells; k++) SDVL_REQUIRE(ctx, cell_rank[k] < s->cells, "cell rank out of range");
  for (int k = 0; k < n_jobs * s->max_its; k++) SDVL_REQUIRE(ctx, rand_raw[k] >= 0, "rand() values are non-negative");
  int rc = sdvl_ensure_nits_table(ctx, p->pose.max_ransac_points, p->pose.max_ransac_its, s->mm);
  if (rc) return rc;
  s->stride = (max_nf + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;
  if (s->stride < kWavesPerBlock) s->stride = kWavesPerBlock;
  s->cam = *cam;
  s->prm = *p;
  s->jobs.assign(jobs, jobs + n_jobs);
  // jobs | cell ranks | rand values: staged at the offsets the three arrays have in the set's scratch (they were carved one behind
  // the other), so ONE push lands all of them — every push is a launch on the step's critical path
  const size_t jb = static_cast<size_t>(reinterpret_cast<uint8_t *>(s->d_cell_rank) - reinterpret_cast<uint8_t *>(s->d_jobs));
  const size_t cb = static_cast<size_t>(reinterpret_cast<uint8_t *>(s->d_rand) - reinterpret_cast<uint8_t *>(s->d_cell_rank));
  const size_t rb = sizeof(int32_t) * static_cast<size_t>(n_jobs) * s->max_its;
  void *hs = nullptr, *dsx = nullptr;
  rc = sdvl_stage_alloc(ctx, jb + cb + rb, &hs, &dsx);
  if (rc) return rc;
  uint8_t *h8 = static_cast<uint8_t *>(hs), *d8 = static_cast<uint8_t *>(dsx);
  TrackJobDev *hj = reinterpret_cast<TrackJobDev *>(h8);
  for (int j = 0; j < n_jobs; j++) {
    const sdvl_track_job &a = jobs[j];
    TrackJobDev &d = hj[j];
    fill_view(&d.cur, a.cur, /*bins_pending*/ true);
    d.tracker = a.tracker;
    d.feat_buf = a.feat_buf;
    d.cur_id = a.cur->reg_id;
    d.last_id = a.last->reg_id;
    d.frame_id = a.frame_id;
    d.max_matches = a.max_matches;
    d.n_feat = s->n_feat[a.feat_buf][a.tracker];
    d.n_points = s->n_points[a.tracker];
    memcpy(d.last_pose, a.last_pose, sizeof(double) * 7);
    d.pad_ = 0.0;
    SDVL_REQUIRE(ctx, a.last->width == a.cur->width && a.last->height == a.cur->height && a.last->v.levels == a.cur->v.levels,
                 "frame pair with different geometry");
    for (int l = 0; l < SDVL_MAX_LEVELS; l++) d.last_level[l] = l < a.last->v.levels ? a.last->v.level[l] : nullptr;
    memcpy(d.T0, a.T, sizeof(double) * 7);
    d.pad2_ = 0.0;
  }
  memcpy(h8 + jb, cell_rank, sizeof(uint16_t) * static_cast<size_t>(n_jobs) * s->cells);
  memcpy(h8 + jb + cb, rand_raw, rb);
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, s->d_jobs, h8, jb + cb + rb));
  (void)d8;
  // the alignment reads the tables itself (image_align_track_*): no feature records, no launch in between
  rc = sdvl_image_align_track_enqueue(ctx, n_jobs, static_cast<const TrackJobDev *>(s->d_jobs), static_cast<const TrackPoint *>(s->d_points),
                                      static_cast<const TrackFeat *>(s->d_feats[0]), static_cast<const TrackFeat *>(s->d_feats[1]), s->np, s->nf, max_nf,
                                      jobs[0].cur->v.levels, cam, &p->align, s->d_ares, s->n);
  if (rc) return rc;
  s->phase = 1;
  return SDVL_OK;
}

int sdvl_track_search(sdvl_ctx *ctx, sdvl_track_set *s) {
  if (!ctx || !s || s->ctx != ctx) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, s->phase == 1, "sdvl_track_search without sdvl_track_align");
  const int n_jobs = static_cast<int>(s->jobs.size());
  for (const sdvl_track_job &a : s->jobs) {
    SDVL_REQUIRE(ctx, !a.cur->hdr_stale, "current frame has a new image but no corners (detect or set corners first)");
    SDVL_REQUIRE(ctx, s->prm.search.max_fast_levels <= a.cur->v.levels, "max_fast_levels exceeds the pyramid depth");
  }
  const Cam c{s->cam.width, s->cam.height, s->cam.fx, s->cam.fy, s->cam.u0, s->cam.v0};
  const int stride = s->stride;
  SearchFramePose *registry = static_cast<SearchFramePose *>(ctx->d_registry);
  {
    // sdvl_track_align named the current frames' corner bins before the detection that fills them was queued (fill_view): frames
    // whose corners were set by hand instead have none
    static const bool never = getenv("SDVL_TRACK_NO_BINS") != nullptr;  // A/B and test of this path: every view loses its bins
    bool any = false;
    for (const sdvl_track_job &a : s->jobs) any = any || (a.cur->bin_cells > 0 && (never || !a.cur->bins_valid));
    for (const sdvl_track_job &a : s->jobs) ctx->counters[(a.cur->bin_cells > 0 && !never && a.cur->bins_valid) ? 0 : 1]++;
    if (any) {
      void *hs = nullptr, *dsx = nullptr;
      const int rc_m = sdvl_stage_alloc(ctx, static_cast<size_t>(n_jobs), &hs, &dsx);
      if (rc_m) return rc_m;
      uint8_t *hm = static_cast<uint8_t *>(hs);
      for (int j = 0; j < n_jobs; j++) hm[j] = (s->jobs[j].cur->bin_cells > 0 && (never || !s->jobs[j].cur->bins_valid)) ? 1 : 0;
      SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, static_cast<size_t>(n_jobs)));
      hipLaunchKernelGGL(track_clear_bins_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, ctx->stream, s->d_jobs, static_cast<const uint8_t *>(dsx), n_jobs);
      SDVL_HIP_CHECK(ctx, hipGetLastError());
    }
  }
  // track_project's lanes run SearchPoint's scalar phase for the requests they assemble: no search_prepare launch in a tracked step.
  // (The chain as a HIP graph was measured in rounds 4 and 5 — 2.69 k against 2.96 k frames/s for a lone camera, nothing for a farm:
  //  launches are not the cost — and removed in round 6.)
  // (one-time set-up calls stay outside a capture)
  if (static_cast<size_t>(stride) * (8 + 16 + 2) + 64 > 60 * 1024) {  // track_project beyond the default dynamic LDS limit: raise it once per device
    static std::atomic<unsigned long long> attr_devices{0};
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
      SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(track_project_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              4096 * (8 + 16 + 2) + 64));
      attr_devices.fetch_or(bit, std::memory_order_release);
    }
  }
  {
    const size_t lds = static_cast<size_t>(stride) * (8 + 16 + 2) + 64;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "track_project", &ev_a, &ev_b);
    // a lane per feature of last_frame: 256 lanes cover the ~190 features of the metric configuration, configuration C's ~850 take 512.
    // Round 6: a farm's launches (sets of more than 32 trackers) take 128 — among the other streams' one-wave workgroups a workgroup
    // is placed when ALL its waves find a slot on one CU at the same moment, and four are found far less often than two: the kernel's
    // dispatch time in company is how long its workgroups wait to be placed (2.0 -> 1.5 ms per step of 16 groups, +3 % tracked frames/s;
    // 64 threads: +1 % — the rank loop gets longer)
    const int proj_threads = stride <= 256 ? (s->n > 32 ? 128 : 256) : kProjThreads;
    hipExtLaunchKernelGGL(track_project_kernel, dim3(n_jobs), dim3(proj_threads), lds, ctx->stream, ev_a, ev_b, 0,
                          static_cast<const TrackJobDev *>(s->d_jobs), s->d_points, static_cast<const TrackFeat *>(s->d_feats[0]),
                          static_cast<const TrackFeat *>(s->d_feats[1]), s->np, s->nf, stride, s->mm, s->max_its,
                          static_cast<const sdvl_align_result *>(s->d_ares), static_cast<const uint16_t *>(s->d_cell_rank), s->cells, c,
                          s->prm.cell_size, s->prm.patch_size, registry, s->d_reqs, s->d_reqpt, s->d_cfirst, s->d_cfeat, s->d_blocks, s->d_chain,
                          s->prm.search, s->d_prep);
    SDVL_HIP_CHECK(ctx, hipGetLastError());
  }
  int rc = sdvl_search_launch_device(ctx, n_jobs * stride, s->d_reqs, registry, s->d_blocks, n_jobs * (stride / kWavesPerBlock), &s->cam,
                                     &s->prm.search, s->d_prep, s->d_res, nullptr, /*prepared*/ true);
  if (rc) return rc;
  // match ranks are not needed separately: track_commit derives them again from the same flags
  rc = sdvl_select_matches_launch(ctx, n_jobs, s->d_chain, nullptr, s->d_cfirst, s->d_res, s->d_reqpt, &s->cam, s->d_pjobs, s->d_obs, s->d_nobs,
                                  nullptr);
  if (rc) return rc;
  sdvl_pose_params pp = s->prm.pose;
  pp.pad_ = 1;  // raw rand() values: the kernel reduces them modulo the match count it finds in the job
  rc = sdvl_pose_enqueue_device(ctx, n_jobs, s->d_pjobs, s->d_obs, s->d_rand, static_cast<const int32_t *>(ctx->d_nits), &pp, s->d_hyp, s->d_pres,
                                s->d_lists, s->mm, s->n);
  if (rc) return rc;
  {
    // s_before | s_found | (8-byte aligned) depths of the new frame's points, at most mm of them
    const size_t lds = (static_cast<size_t>(stride + 1) * 2 + stride + 64 + 7) / 8 * 8 + static_cast<size_t>(s->mm) * 8;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "track_commit", &ev_a, &ev_b);
    hipExtLaunchKernelGGL(track_commit_kernel, dim3(n_jobs), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const TrackJobDev *>(s->d_jobs),
                          s->d_points, static_cast<const TrackFeat *>(s->d_feats[0]), static_cast<const TrackFeat *>(s->d_feats[1]), s->d_feats[0],
                          s->d_feats[1], s->np, s->nf, stride, s->mm, static_cast<const ChainFrameDev *>(s->d_chain),
                          static_cast<const int32_t *>(s->d_cfirst), static_cast<const int32_t *>(s->d_cfeat),
                          static_cast<const sdvl_search_res *>(s->d_res), static_cast<const sdvl_pose_result *>(s->d_pres),
                          static_cast<const int32_t *>(s->d_lists), static_cast<const sdvl_align_result *>(s->d_ares), c, s->prm.max_failed, registry,
                          s->h_results, s->h_feats, s->h_stats);
    SDVL_HIP_CHECK(ctx, hipGetLastError());
  }
  SDVL_HIP_CHECK(ctx, sdvl_mark_record(ctx, SDVL_MARK_CHAIN, &s->ticket));
  s->phase = 2;
  return SDVL_OK;
}

int sdvl_track_collect(sdvl_ctx *ctx, sdvl_track_set *s, int n_jobs, sdvl_track_result *results) {
  if (!ctx || !s || s->ctx != ctx || !results) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, s->phase == 2 && n_jobs == static_cast<int>(s->jobs.size()), "sdvl_track_collect without a matching sdvl_track_search");
  SDVL_HIP_CHECK(ctx, sdvl_mark_wait(ctx, SDVL_MARK_CHAIN, s->ticket));
  s->phase = 0;
  memcpy(results, s->h_results, sizeof(sdvl_track_result) * n_jobs);
  for (int j = 0; j < n_jobs; j++) {
    const sdvl_track_job &a = s->jobs[j];
    s->n_feat[1 - a.feat_buf][a.tracker] = results[j].matches;
    if (results[j].n_corners >= 0) a.cur->v.n_corners = results[j].n_corners;  // the count rode along: no round trip to ask for it
  }
  ctx->wait_gen++;  // everything queued before the step's mark has completed (the detection's corner counts included)
  return SDVL_OK;
}

const sdvl_track_feature_out *sdvl_track_features(const sdvl_track_set *s, int job) {
  return (s && job >= 0 && job < s->n) ? s->h_feats + static_cast<size_t>(job) * s->nf : nullptr;
}

const sdvl_track_point_stat *sdvl_track_stats(const sdvl_track_set *s, int job) {
  return (s && job >= 0 && job < s->n) ? s->h_stats + static_cast<size_t>(job) * s->np : nullptr;
}

}  // extern "C"
