// sdvl_orb.hip — K3 Shi-Tomasi score and K4 ORB (orientation + 256-bit steered BRIEF), one wave64 per corner.
//   K4  orb_describe_kernel : ORBDetector::GetDescriptor / GetOrientation, extra/orb_detector.cc:350-437 — the per-corner
//        arithmetic is orb_wave_nibble() in sdvl_orb_device.h; here nibbles are merged pairwise into bytes and stored.
//   K3  shi_tomasi_kernel   : FindShiTomasiScoreAtPoint, extra/utils.cc:61-97 — lane = pixel of the 8x8 box,
//        exact int32 sums (every partial sum < 2^24 so the reference's float loop is exact too), float/double tail.
#include <algorithm>
#include <vector>

#include "sdvl_internal.h"
#include "sdvl_orb_device.h"

namespace {

struct OrbJob {
  const uint8_t *level[SDVL_MAX_LEVELS];
  int lw[SDVL_MAX_LEVELS], lh[SDVL_MAX_LEVELS];
  const int32_t *corners;  // [n][4]
  uint8_t *desc;           // [n][32] in HBM
  uint8_t *out;            // optional second copy (host-bound buffer), may be null
  float *out_angle;        // optional
  double *out_score;       // shi-tomasi output
  const int32_t *n_ptr;    // device-resident corner count (frame header); null -> use n
  int n;
  int levels;
};

__device__ __forceinline__ int wave_sum_i32(int v) { return orb_wave_sum_i32(v); }

// Blocks b and b + 8 share an XCD (observed placement, used for speed only): the `chunks` workgroups of frame f all run on XCD f % 8, so
// the windows of neighbouring corners — a 10x10 or 31x31 window touches 10 / 31 lines of 128 B, and the corners of a cell row share
// them — come out of ONE L2 instead of being fetched into all eight (round 2: shi_tomasi fetched 53 MB for 5.5 MB of windows).
// 1-D grid of 8 * ceil(n_frames / 8) * chunks blocks; returns false for the padding blocks.
__device__ __forceinline__ bool xcd_frame_block(int n_frames, int chunks, int *frame, int *chunk) {
  const int b = static_cast<int>(blockIdx.x), q = b >> 3;
  *frame = (q / chunks) * 8 + (b & 7);
  *chunk = q % chunks;
  return *frame < n_frames;
}
inline dim3 xcd_frame_grid(int n_frames, int chunks) { return dim3(static_cast<unsigned>((n_frames + 7) / 8 * 8 * chunks)); }

// `chunks` workgroups of 4 waves per job (frame), a wave per corner
__global__ __launch_bounds__(256) void orb_describe_kernel(const OrbJob *__restrict__ jobs, int n_jobs, int chunks) {
  int fj, bx;
  if (!xcd_frame_block(n_jobs, chunks, &fj, &bx)) return;
  const OrbJob &job = jobs[fj];
  const int lane = threadIdx.x & 63;
  const int n = job.n_ptr ? min(job.n_ptr[0], SDVL_MAX_CORNERS) : job.n;
  // wave-uniform grid-stride loop: the corner count may only be known on the device
  const int wpb = static_cast<int>(blockDim.x >> 6);  // waves per workgroup: one (round 3) — a wave per corner needs no company
  for (int ci = bx * wpb + (threadIdx.x >> 6); ci < n; ci += chunks * wpb) {
  const int cx = job.corners[4 * ci], cy = job.corners[4 * ci + 1], cl = job.corners[4 * ci + 2];
  if (cl < 0 || cl >= job.levels) continue;
  const int W = job.lw[cl], H = job.lh[cl];
  uint8_t *dst = job.desc + static_cast<size_t>(ci) * 32;
  // ORBDetector::IsInsideLimits (orb_detector.cc:439-445); the reference asserts it
  if (!(cx >= 19 && cx < W - 19 && cy >= 19 && cy < H - 19)) {
    if (lane < 32) {
      dst[lane] = 0;
      if (job.out) job.out[static_cast<size_t>(ci) * 32 + lane] = 0;
    }
    if (lane == 0 && job.out_angle) job.out_angle[ci] = -1.f;
    continue;
  }
  float angle_deg;
  const uint32_t nib = orb_wave_nibble(job.level[cl], static_cast<uint32_t>(cy * W + cx), W, lane, &angle_deg);
  const uint32_t hi = __shfl_down(nib, 1, 64);
  if ((lane & 1) == 0) {
    const uint8_t byte = static_cast<uint8_t>(nib | (hi << 4));
    dst[lane >> 1] = byte;
    if (job.out) job.out[static_cast<size_t>(ci) * 32 + (lane >> 1)] = byte;
  }
  if (lane == 0 && job.out_angle) job.out_angle[ci] = angle_deg;
  }
}

// Round 4: FOUR corners per wave, 16 lanes each (until round 3 a wave per corner, lane = pixel of the 8x8 box: ~130 wave instructions
// per corner, 17 M per keyframe dispatch — the path is bound by instruction issue).  Lane q of a corner's 16 owns four
// adjacent pixels of box row q / 2: two words of its row (at x - 1 and x + 1: byte k of their difference is dx of pixel k) and one
// word each of the rows above and below (dy) — four loads instead of sixteen; the three sums are exact integers in any order
// (|dx|, |dy| <= 255, 64 terms), reduced inside the 16-lane DPP row.  ~27 instructions per corner.
constexpr int kShiPerWave = 4;
__global__ __launch_bounds__(256) void shi_tomasi_kernel(const OrbJob *__restrict__ jobs, int n_jobs, int chunks) {
  int fj, bx;
  if (!xcd_frame_block(n_jobs, chunks, &fj, &bx)) return;
  const OrbJob &job = jobs[fj];
  const int lane = threadIdx.x & 63, grp = lane >> 4, q = lane & 15;
  const int n = job.n_ptr ? min(job.n_ptr[0], SDVL_MAX_CORNERS) : job.n;
  const int wpb = static_cast<int>(blockDim.x >> 6);
  for (int c0 = (bx * wpb + (threadIdx.x >> 6)) * kShiPerWave; c0 < n; c0 += chunks * wpb * kShiPerWave) {
    const int ci = c0 + grp;
    int sxx = 0, syy = 0, sxy = 0;
    bool inside = false;
    if (ci < n) {
      const int px = job.corners[4 * ci], py = job.corners[4 * ci + 1], cl = job.corners[4 * ci + 2];
      if (cl >= 0 && cl < job.levels) {
        const int W = job.lw[cl], H = job.lh[cl];
        const int x_min = px - 4, x_max = px + 4, y_min = py - 4, y_max = py + 4;
        inside = !(x_min < 1 || x_max >= W - 1 || y_min < 1 || y_max >= H - 1);
        if (inside) {
          const uint8_t *p = job.level[cl] + static_cast<size_t>(y_min + (q >> 1)) * W + (x_min + 4 * (q & 1));
          uint32_t lf, rt, up, dn;
          __builtin_memcpy(&lf, p - 1, 4);
          __builtin_memcpy(&rt, p + 1, 4);
          __builtin_memcpy(&up, p - W, 4);
          __builtin_memcpy(&dn, p + W, 4);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int dx = static_cast<int>((rt >> (8 * k)) & 0xFFu) - static_cast<int>((lf >> (8 * k)) & 0xFFu);
            const int dy = static_cast<int>((dn >> (8 * k)) & 0xFFu) - static_cast<int>((up >> (8 * k)) & 0xFFu);
            sxx += dx * dx;
            syy += dy * dy;
            sxy += dx * dy;
          }
        }
      }
    }
    // sums over the corner's 16 lanes: row_shr 1, 2, 4, 8 leave the row's total in its last lane
#define SDVL_ROW_SUM(v)                                                   \
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);       \
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);       \
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, false);       \
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, false);
    SDVL_ROW_SUM(sxx)
    SDVL_ROW_SUM(syy)
    SDVL_ROW_SUM(sxy)
#undef SDVL_ROW_SUM
    if (q == 15 && ci < n) {
      const int cl = job.corners[4 * ci + 2];
      if (cl >= 0 && cl < job.levels) {  // (corners of a level the frame does not have keep whatever the score buffer held, as before)
        double score = 0.0;
        if (inside) {
          float dXX = static_cast<float>(sxx), dYY = static_cast<float>(syy), dXY = static_cast<float>(sxy);
          dXX = static_cast<float>(dXX / (2.0 * 64));
          dYY = static_cast<float>(dYY / (2.0 * 64));
          dXY = static_cast<float>(dXY / (2.0 * 64));
          const float disc = (dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY);
          score = 0.5 * (dXX + dYY - sqrt(static_cast<double>(disc)));
        }
        job.out_score[ci] = score;
      }
    }
  }
}
inline int shi_chunks(int max_n) { return (max_n + kShiPerWave - 1) / kShiPerWave; }


// ---- Frame::FilterCorners on the device (frame.cc:133-163, FastDetector::FilterCorners fast_detector.cc:177-218) ----------
// After shi_tomasi_kernel has scored every corner: one workgroup per frame walks the corner list cell by cell in the
// reference's order (a later corner of a cell replaces the cell's choice when its score exceeds the TRUNCATED score stored
// for it: order matters, so a cell is scanned sequentially by its own thread), drops corners in locked cells or inside the
// margin, lists the survivors in cell order, and one wave per survivor computes its ORB descriptor (frame.cc:148-161 asks
// for the descriptors of the filtered corners only).  Output per frame: count + records {index, x, y, level, score, desc}.
struct FilterJob {
  const uint8_t *level[SDVL_MAX_LEVELS];
  int lw[SDVL_MAX_LEVELS], lh[SDVL_MAX_LEVELS];
  const int32_t *corners;   // [n][4]
  const int32_t *n_ptr;     // device-resident corner count
  const double *scores;     // [ccap] Shi-Tomasi scores of this frame's corners
  const uint32_t *locked;   // [mask_words] bit c = cell c holds a feature already (FastDetector::LockCell)
  sdvl_filtered_corner *out;  // [max_out] records; out_count[0] = how many
  int32_t *out_count;
  int levels, ccap;
};

constexpr int kFilterMaxCorners = SDVL_MAX_CORNERS;

__global__ __launch_bounds__(256) void filter_select_kernel(const FilterJob *__restrict__ jobs, int cell_size, int grid_w, int n_cells, int margin,
                                                            int min_score, int max_out) {
  __shared__ uint16_t s_cell[kFilterMaxCorners];  // cell of corner i, 0xFFFF = not eligible
  __shared__ int s_sel[4096];                     // per cell: chosen corner or -1
  __shared__ int s_wave[4];
  const FilterJob &job = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(min(job.n_ptr[0], job.ccap), kFilterMaxCorners);
  for (int i = tid; i < n; i += 256) {
    const int px = job.corners[4 * i], py = job.corners[4 * i + 1], level = job.corners[4 * i + 2];
    uint16_t c = 0xFFFFu;
    if (level >= 0 && level < job.levels && !(px < margin || py < margin || px >= job.lw[level] - margin || py >= job.lh[level] - margin)) {
      const int scale = 1 << level;
      const int pos = ((py * scale) / cell_size) * grid_w + (px * scale) / cell_size;
      if (pos >= 0 && pos < n_cells && !((job.locked[pos >> 5] >> (pos & 31)) & 1u)) c = static_cast<uint16_t>(pos);
    }
    s_cell[i] = c;
  }
  __syncthreads();
  for (int c = tid; c < n_cells; c += 256) {
    int best_idx = 0, best = min_score;  // cgrid_ starts as (0, MinFeatureScore), fast_detector.cc:40
    for (int i = 0; i < n; i++) {
      if (s_cell[i] != c) continue;
      const double score = job.scores[i];
      if (score > best) {
        best_idx = i;
        best = static_cast<int>(score);
      }
    }
    s_sel[c] = best > min_score ? best_idx : -1;
  }
  __syncthreads();
  // survivors in cell order (fast_detector.cc:213-217)
  int running = 0;
  for (int c0 = 0; c0 < n_cells; c0 += 256) {
    const int c = c0 + tid;
    const bool ok = c < n_cells && s_sel[c] >= 0;
    const unsigned long long m = __ballot(ok);
    const int below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int pos = running + below;
    for (int w = 0; w < wave; w++) pos += s_wave[w];
    if (ok && pos < max_out) {
      const int i = s_sel[c];
      sdvl_filtered_corner r;
      r.index = i;
      r.x = job.corners[4 * i]; r.y = job.corners[4 * i + 1]; r.level = job.corners[4 * i + 2];
      r.score = static_cast<int>(job.scores[i]);
      r.pad_ = 0;
      // descriptor filled below
      sdvl_filtered_corner *dst = job.out + pos;
      dst->index = r.index; dst->x = r.x; dst->y = r.y; dst->level = r.level; dst->score = r.score; dst->pad_ = 0;
    }
    running += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
  if (tid == 0) job.out_count[0] = running;  // the host checks it against max_out
}

// The same selection for grids of up to 2048 cells with the corners binned first: every corner appends itself to its cell's
// short list (LDS atomics, arrival order), the cell's thread puts the handful of entries back into list order and runs the
// reference's sequential comparison over them.  Work ~ corners instead of cells x corners (configuration C: 1200 cells x
// 4050 corners).  A cell with more than kBinSlots corners falls back to the scan over all corners.
// Round 6: two sizes.  The arrays are static LDS: 60 KB when they are cut for configuration C's 1200 cells and 6144 corners — a
// workgroup that has to find 60 KB free on one compute unit among the other streams' kernels (fast_cells alone fills 150 of a CU's
// 160 KB) waits: 132 us per dispatch in the farm against 15 alone.  The metric configuration's 300 cells and ~1000 corners take 16 KB.
constexpr int kBinCells = 2048, kBinSlots = 10;
constexpr int kBinCellsSmall = 512, kBinCornersSmall = 2048;

template <int kCells, int kCorners>
__global__ __launch_bounds__(256) void filter_select_binned_kernel(const FilterJob *__restrict__ jobs, int cell_size, int grid_w, int n_cells,
                                                                   int margin, int min_score, int max_out) {
  __shared__ uint16_t s_cell[kCorners];
  __shared__ uint16_t s_list[kCells * kBinSlots];
  __shared__ int s_cnt[kCells];  // corners of the cell; afterwards the chosen corner or -1
  __shared__ int s_wave[4];
  const FilterJob &job = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(min(job.n_ptr[0], job.ccap), kCorners);
  for (int c = tid; c < n_cells; c += 256) s_cnt[c] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const int px = job.corners[4 * i], py = job.corners[4 * i + 1], level = job.corners[4 * i + 2];
    uint16_t c = 0xFFFFu;
    if (level >= 0 && level < job.levels && !(px < margin || py < margin || px >= job.lw[level] - margin || py >= job.lh[level] - margin)) {
      const int scale = 1 << level;
      const int pos = ((py * scale) / cell_size) * grid_w + (px * scale) / cell_size;
      if (pos >= 0 && pos < n_cells && !((job.locked[pos >> 5] >> (pos & 31)) & 1u)) {
        c = static_cast<uint16_t>(pos);
        const int slot = atomicAdd(&s_cnt[pos], 1);
        if (slot < kBinSlots) s_list[pos * kBinSlots + slot] = static_cast<uint16_t>(i);
      }
    }
    s_cell[i] = c;
  }
  __syncthreads();
  for (int c = tid; c < n_cells; c += 256) {
    const int cnt = s_cnt[c];
    int best_idx = 0, best = min_score;  // cgrid_ starts as (0, MinFeatureScore), fast_detector.cc:40
    if (cnt <= kBinSlots) {
      int idx[kBinSlots];
#pragma unroll
      for (int q = 0; q < kBinSlots; q++) idx[q] = q < cnt ? s_list[c * kBinSlots + q] : 0x7FFFFFFF;
      // list order = ascending corner index: a fixed compare-exchange network over the (at most 10) entries
#pragma unroll
      for (int a = 0; a < kBinSlots; a++)
#pragma unroll
        for (int b = 0; b + 1 < kBinSlots - a; b++) {
          const int lo = min(idx[b], idx[b + 1]), hi = max(idx[b], idx[b + 1]);
          idx[b] = lo;
          idx[b + 1] = hi;
        }
#pragma unroll
      for (int q = 0; q < kBinSlots; q++) {
        if (q >= cnt) break;
        const double score = job.scores[idx[q]];
        if (score > best) {
          best_idx = idx[q];
          best = static_cast<int>(score);
        }
      }
    } else {
      for (int i = 0; i < n; i++) {
        if (s_cell[i] != c) continue;
        const double score = job.scores[i];
        if (score > best) {
          best_idx = i;
          best = static_cast<int>(score);
        }
      }
    }
    s_cnt[c] = best > min_score ? best_idx : -1;
  }
  __syncthreads();
  int running = 0;
  for (int c0 = 0; c0 < n_cells; c0 += 256) {
    const int c = c0 + tid;
    const bool ok = c < n_cells && s_cnt[c] >= 0;
    const unsigned long long m = __ballot(ok);
    const int below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int pos = running + below;
    for (int w = 0; w < wave; w++) pos += s_wave[w];
    if (ok && pos < max_out) {
      const int i = s_cnt[c];
      sdvl_filtered_corner *dst = job.out + pos;
      dst->index = i;
      dst->x = job.corners[4 * i]; dst->y = job.corners[4 * i + 1]; dst->level = job.corners[4 * i + 2];
      dst->score = static_cast<int>(job.scores[i]);
      dst->pad_ = 0;
    }
    running += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
  if (tid == 0) job.out_count[0] = running;
}

// ORB descriptors of the corners filter_select_kernel kept, one wave each: grid.x covers max_out / 4 workgroups, grid.y = frames
__global__ __launch_bounds__(256) void filter_describe_kernel(const FilterJob *__restrict__ jobs, int max_out, int n_jobs, int chunks) {
  int fj, bx;
  if (!xcd_frame_block(n_jobs, chunks, &fj, &bx)) return;
  const FilterJob &job = jobs[fj];
  const int lane = threadIdx.x & 63;
  const int total = min(job.out_count[0], max_out);
  const int wpb = static_cast<int>(blockDim.x >> 6);
  for (int k = bx * wpb + (threadIdx.x >> 6); k < total; k += chunks * wpb) {
    sdvl_filtered_corner *dst = job.out + k;
    const int cx = dst->x, cy = dst->y, cl = dst->level;
    const int W = job.lw[cl], H = job.lh[cl];
    uint8_t byte = 0;
    if (cx >= 19 && cx < W - 19 && cy >= 19 && cy < H - 19) {  // ORBDetector::IsInsideLimits; zeros outside, as orb_describe_kernel
      float angle_deg;
      const uint32_t nib = orb_wave_nibble(job.level[cl], static_cast<uint32_t>(cy * W + cx), W, lane, &angle_deg);
      const uint32_t hi = __shfl_down(nib, 1, 64);
      byte = static_cast<uint8_t>(nib | (hi << 4));
    }
    if ((lane & 1) == 0) dst->desc[lane >> 1] = byte;
  }
}

// header + corners (+ descriptors) of a frame -> one contiguous row, in 16-byte units (sdvl_filter_inputs)
__global__ __launch_bounds__(256) void filter_gather_kernel(const OrbJob *__restrict__ jobs, uint4 *__restrict__ dst, int row_units, int ccap,
                                                            int with_desc) {
  const OrbJob &job = jobs[blockIdx.y];
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= row_units) return;
  uint4 v;
  if (u <= ccap) {  // header (unit 0) and corner records: contiguous in the frame, starting at the header
    v = reinterpret_cast<const uint4 *>(job.n_ptr)[u];
  } else if (with_desc) {
    v = reinterpret_cast<const uint4 *>(job.desc)[u - (ccap + 1)];
  } else {
    return;
  }
  dst[static_cast<size_t>(blockIdx.y) * row_units + u] = v;
}

int fill_jobs(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, uint8_t *d_out_desc, double *d_out_score,
              OrbJob **d_jobs, int *max_n) {
  const size_t bytes = sizeof(OrbJob) * n;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_stage_alloc(ctx, bytes, &hs, &dsx);
  if (rc) return rc;
  OrbJob *hj = static_cast<OrbJob *>(hs);
  *max_n = 0;
  for (int i = 0; i < n; i++) {
    int rcf = sdvl_frame_fix_header(ctx, frames[i]);
    if (rcf) return rcf;
    const FrameView &v = frames[i]->v;
    memset(&hj[i], 0, sizeof(OrbJob));
    for (int l = 0; l < v.levels; l++) { hj[i].level[l] = v.level[l]; hj[i].lw[l] = v.lw[l]; hj[i].lh[l] = v.lh[l]; }
    hj[i].corners = v.corners;
    hj[i].desc = v.desc;
    hj[i].out = d_out_desc ? d_out_desc + static_cast<size_t>(i) * cap * 32 : nullptr;
    hj[i].out_angle = nullptr;
    hj[i].out_score = d_out_score ? d_out_score + static_cast<size_t>(i) * cap : nullptr;
    hj[i].n_ptr = v.corner_hdr;
    hj[i].n = v.n_corners;
    hj[i].levels = v.levels;
    const int bound = v.n_corners >= 0 ? v.n_corners : 1280;  // device-only count: a typical grid; the kernels grid-stride
    if (bound > *max_n) *max_n = bound;
  }
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hj, bytes));
  *d_jobs = static_cast<OrbJob *>(dsx);
  return SDVL_OK;
}

}  // namespace

// ORBDetector::Distance (extra/orb_detector.cc:398-410) over candidate lists + the arg-min of Matcher::SearchFeatures
// (matcher.cc:254-289: best starts at threshold+1, strict '<' so the first smallest wins, ">= threshold" -> not found).
// One wave per query; lanes stride the query's candidates, 32 B per candidate as two 16 B loads.
struct HammingJob {
  const uint8_t *queries;     // [n][32]
  const int32_t *offsets;     // [n+1]
  const uint8_t *cands;       // [offsets[n]][32]
  int32_t *best;              // [n][2]: index within the query's list (or -1), distance (or threshold+1)
  int n, threshold;
};

__global__ __launch_bounds__(64) void hamming_argmin_kernel(const HammingJob *jobp) {
  const HammingJob &job = *jobp;
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= job.n) return;
  const uint4 *qp = reinterpret_cast<const uint4 *>(job.queries + 32 * static_cast<size_t>(q));
  const uint4 q0 = qp[0], q1 = qp[1];
  const int begin = job.offsets[q], end = job.offsets[q + 1];
  // key = distance << 32 | position: the smallest key is the first smallest distance
  unsigned long long key = (static_cast<unsigned long long>(job.threshold + 1) << 32) | 0xffffffffull;
  for (int c = begin + lane; c < end; c += 64) {
    const uint4 *cp = reinterpret_cast<const uint4 *>(job.cands + 32 * static_cast<size_t>(c));
    const uint4 a = cp[0], b = cp[1];
    const int d = __popc(a.x ^ q0.x) + __popc(a.y ^ q0.y) + __popc(a.z ^ q0.z) + __popc(a.w ^ q0.w) + __popc(b.x ^ q1.x) +
                  __popc(b.y ^ q1.y) + __popc(b.z ^ q1.z) + __popc(b.w ^ q1.w);
    const unsigned long long k = (static_cast<unsigned long long>(d) << 32) | static_cast<unsigned>(c - begin);
    key = k < key ? k : key;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off, 64);
    key = o < key ? o : key;
  }
  if (lane == 0) {
    const int d = static_cast<int>(key >> 32);
    const bool found = d < job.threshold;
    job.best[2 * q] = found ? static_cast<int>(key & 0xffffffffull) : -1;
    job.best[2 * q + 1] = d;
  }
}

extern "C" {

int sdvl_orb_describe(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, uint8_t *out_desc) {
  if (!ctx || n < 0 || (n > 0 && !frames)) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] != nullptr, "null frame");
    if (out_desc) {
      int cnt = 0;
      int rc0 = sdvl_frame_count_host(ctx, frames[i], &cnt);
      if (rc0) return rc0;
    }
    if (out_desc && frames[i]->v.n_corners > cap) {
      ctx->err = "descriptor output capacity smaller than the corner count";
      return SDVL_ERR_CAPACITY;
    }
  }
  uint8_t *d_desc = nullptr;
  const size_t out_bytes = out_desc ? static_cast<size_t>(n) * cap * 32 : 0;
  if (out_desc) {
    int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, out_bytes, false);
    if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, out_bytes, true);
    if (rc) return rc;
    d_desc = static_cast<uint8_t *>(ctx->d_out);
  }
  OrbJob *d_jobs = nullptr;
  int max_n = 0;
  int rc = fill_jobs(ctx, n, frames, cap, d_desc, nullptr, &d_jobs, &max_n);
  if (rc) return rc;
  if (max_n > 0) SDVL_LAUNCH(ctx, "orb_describe", orb_describe_kernel, xcd_frame_grid(n, max_n), dim3(64), d_jobs, n, max_n);
    SDVL_HIP_CHECK(ctx, hipGetLastError());
  for (int i = 0; i < n; i++) frames[i]->desc_valid = 1;
  if (out_desc) {
    if (max_n > 0) {
      SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(ctx->h_out, static_cast<size_t>(cap) * 32, d_desc, static_cast<size_t>(cap) * 32,
                                           static_cast<size_t>(max_n) * 32, n, hipMemcpyDeviceToHost, ctx->stream));
      SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
      for (int i = 0; i < n; i++)
        memcpy(out_desc + static_cast<size_t>(i) * cap * 32, static_cast<uint8_t *>(ctx->h_out) + static_cast<size_t>(i) * cap * 32,
               static_cast<size_t>(frames[i]->v.n_corners) * 32);
    }
  }
  return SDVL_OK;
}

int sdvl_shi_tomasi(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, double *out_scores) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !out_scores))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] != nullptr, "null frame");
    int cnt = 0;
    int rc0 = sdvl_frame_count_host(ctx, frames[i], &cnt);
    if (rc0) return rc0;
    if (frames[i]->v.n_corners > cap) {
      ctx->err = "score output capacity smaller than the corner count";
      return SDVL_ERR_CAPACITY;
    }
  }
  const size_t out_bytes = sizeof(double) * static_cast<size_t>(n) * cap;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, out_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, out_bytes, true);
  if (rc) return rc;
  OrbJob *d_jobs = nullptr;
  int max_n = 0;
  rc = fill_jobs(ctx, n, frames, cap, nullptr, static_cast<double *>(ctx->d_out), &d_jobs, &max_n);
  if (rc) return rc;
  if (max_n == 0) return SDVL_OK;
  SDVL_LAUNCH(ctx, "shi_tomasi", shi_tomasi_kernel, xcd_frame_grid(n, shi_chunks(max_n)), dim3(64), d_jobs, n, shi_chunks(max_n));
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(ctx->h_out, sizeof(double) * cap, ctx->d_out, sizeof(double) * cap, sizeof(double) * max_n, n,
                                       hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  for (int i = 0; i < n; i++)
    memcpy(out_scores + static_cast<size_t>(i) * cap, static_cast<double *>(ctx->h_out) + static_cast<size_t>(i) * cap,
           sizeof(double) * frames[i]->v.n_corners);
  return SDVL_OK;
}

int sdvl_filter_inputs_begin(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, int with_desc) {
  if (!ctx || n < 0 || (n > 0 && !frames) || cap <= 0) return SDVL_ERR_INVALID;
  ctx->filter_pending = 0;
  if (n == 0) return SDVL_OK;
  const bool desc = with_desc != 0;
  for (int i = 0; i < n; i++) SDVL_REQUIRE(ctx, frames[i] != nullptr, "null frame");
  if (desc) {  // frames whose descriptors have not been computed yet (searches compute only what they compare) get them now
    std::vector<sdvl_frame *> missing;
    for (int i = 0; i < n; i++)
      if (!frames[i]->desc_valid) missing.push_back(frames[i]);
    if (!missing.empty()) {
      const int rc0 = sdvl_orb_describe(ctx, static_cast<int>(missing.size()), missing.data(), SDVL_MAX_CORNERS, nullptr);
      if (rc0) return rc0;
    }
  }
  // rows only as long as they have to be: the largest corner count when the host already knows every count
  int ccap = cap < SDVL_MAX_CORNERS ? cap : SDVL_MAX_CORNERS;
  {
    int known_max = 0;
    bool all_known = true;
    for (int i = 0; i < n; i++) {
      const int c = frames[i]->hdr_stale ? 0 : frames[i]->v.n_corners;
      if (c < 0) all_known = false;
      else known_max = c > known_max ? c : known_max;
    }
    if (all_known && known_max < ccap) ccap = known_max > 0 ? known_max : 1;
  }
  // device layout: scores[n][ccap] doubles | n rows of {header + corners[ccap] (+ descriptors[ccap])} gathered from the
  // frames by one kernel, so that everything returns in ONE device-to-host copy
  const size_t sc_bytes = (sizeof(double) * static_cast<size_t>(n) * ccap + 255) / 256 * 256;
  const size_t row = sizeof(int32_t) * 4 * (static_cast<size_t>(ccap) + 1) + (desc ? static_cast<size_t>(ccap) * 32 : 0);
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, sc_bytes + row * n, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, sc_bytes + row * n, true);
  if (rc) return rc;
  OrbJob *d_jobs = nullptr;
  int max_n = 0;
  rc = fill_jobs(ctx, n, frames, ccap, nullptr, static_cast<double *>(ctx->d_out), &d_jobs, &max_n);
  if (rc) return rc;
  max_n = max_n > ccap ? ccap : max_n;
  if (max_n > 0) SDVL_LAUNCH(ctx, "shi_tomasi", shi_tomasi_kernel, xcd_frame_grid(n, shi_chunks(max_n)), dim3(64), d_jobs, n, shi_chunks(max_n));
  {
    const int units = (ccap + 1) + (desc ? 2 * ccap : 0);  // 16-byte units per row
    SDVL_LAUNCH(ctx, "filter_gather", filter_gather_kernel, dim3((units + 255) / 256, n), dim3(256), static_cast<const OrbJob *>(d_jobs),
                reinterpret_cast<uint4 *>(static_cast<uint8_t *>(ctx->d_out) + sc_bytes), static_cast<int>(row / 16), ccap, desc ? 1 : 0);
  }
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  uint8_t *h = static_cast<uint8_t *>(ctx->h_out);
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(h, ctx->d_out, sc_bytes + row * n, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_mark_record(ctx, SDVL_MARK_FILTER, &ctx->filter_ticket));
  ctx->filter_pending = n;
  ctx->filter_ccap = ccap;
  ctx->filter_desc = desc ? 1 : 0;
  ctx->filter_sc_bytes = sc_bytes;
  ctx->filter_row = row;
  return SDVL_OK;
}

int sdvl_filter_inputs_end(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, int32_t *xyl, double *scores, uint8_t *desc,
                           int32_t *counts) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !xyl || !scores || !counts)) || cap <= 0) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, ctx->filter_pending == n && (desc != nullptr) == (ctx->filter_desc != 0), "sdvl_filter_inputs_end without a matching sdvl_filter_inputs_begin");
  ctx->filter_pending = 0;
  SDVL_HIP_CHECK(ctx, sdvl_mark_wait(ctx, SDVL_MARK_FILTER, ctx->filter_ticket));
  const int ccap = ctx->filter_ccap;
  const size_t sc_bytes = ctx->filter_sc_bytes, row = ctx->filter_row;
  const uint8_t *h = static_cast<const uint8_t *>(ctx->h_out);
  for (int i = 0; i < n; i++) {
    const uint8_t *src = h + sc_bytes + row * i;
    const int32_t *hdr = reinterpret_cast<const int32_t *>(src);
    int cnt = frames[i]->hdr_stale ? 0 : hdr[0];
    if (cnt > cap) {
      ctx->err = "corner output capacity smaller than the corner count";
      return SDVL_ERR_CAPACITY;
    }
    frames[i]->v.n_corners = cnt;
    counts[i] = cnt;
    for (int k = 0; k < cnt; k++) {
      xyl[(static_cast<size_t>(i) * cap + k) * 3] = hdr[4 * (k + 1)];
      xyl[(static_cast<size_t>(i) * cap + k) * 3 + 1] = hdr[4 * (k + 1) + 1];
      xyl[(static_cast<size_t>(i) * cap + k) * 3 + 2] = hdr[4 * (k + 1) + 2];
    }
    memcpy(scores + static_cast<size_t>(i) * cap, reinterpret_cast<const double *>(h) + static_cast<size_t>(i) * ccap, sizeof(double) * cnt);
    if (desc) memcpy(desc + static_cast<size_t>(i) * cap * 32, src + sizeof(int32_t) * 4 * (static_cast<size_t>(ccap) + 1), static_cast<size_t>(cnt) * 32);
  }
  return SDVL_OK;
}


// ---- sdvl_filter_corners_begin / _end: Frame::FilterCorners for n frames, selection included --------------------------------
int sdvl_filter_corners_begin(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const uint32_t *locked_cells, int mask_words, int cell_size,
                              int margin, int min_feature_score, int with_desc) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !locked_cells)) || mask_words <= 0 || cell_size <= 0) return SDVL_ERR_INVALID;
  ctx->filter_pending = 0;
  if (n == 0) return SDVL_OK;
  (void)with_desc;  // descriptors of the selected corners always ride along (32 B each)
  const int W = frames[0]->width, H = frames[0]->height;
  const int grid_w = (W + cell_size - 1) / cell_size, grid_h = (H + cell_size - 1) / cell_size, n_cells = grid_w * grid_h;
  SDVL_REQUIRE(ctx, n_cells <= 4096 && n_cells <= mask_words * 32, "filter grid too large (4096 cells) or lock mask too short");
  int ccap = 1;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] && frames[i]->width == W && frames[i]->height == H, "frames of one batch must share their size");
    const int c = frames[i]->hdr_stale ? 0 : frames[i]->v.n_corners;
    ccap = std::max(ccap, c < 0 ? SDVL_MAX_CORNERS : c);
  }
  const int max_out = n_cells;
  // device: scores[n][ccap] | counts[n] | records[n][max_out] ; host mirror of counts + records
  const size_t sc_bytes = (sizeof(double) * static_cast<size_t>(n) * ccap + 255) / 256 * 256;
  const size_t cnt_bytes = (sizeof(int32_t) * n + 255) / 256 * 256, rec_bytes = sizeof(sdvl_filtered_corner) * static_cast<size_t>(n) * max_out;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, sc_bytes + cnt_bytes + rec_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, cnt_bytes + rec_bytes, true);
  if (rc) return rc;
  OrbJob *d_jobs = nullptr;
  int max_n = 0;
  rc = fill_jobs(ctx, n, frames, ccap, nullptr, static_cast<double *>(ctx->d_out), &d_jobs, &max_n);
  if (rc) return rc;
  max_n = max_n > ccap ? ccap : max_n;
  if (max_n > 0) SDVL_LAUNCH(ctx, "shi_tomasi", shi_tomasi_kernel, xcd_frame_grid(n, shi_chunks(max_n)), dim3(64), d_jobs, n, shi_chunks(max_n));
  const size_t jb = (sizeof(FilterJob) * n + 255) / 256 * 256, mb = sizeof(uint32_t) * static_cast<size_t>(n) * mask_words;
  void *hs = nullptr, *dsx = nullptr;
  rc = sdvl_stage_alloc(ctx, jb + mb, &hs, &dsx);
  if (rc) return rc;
  FilterJob *hj = static_cast<FilterJob *>(hs);
  uint8_t *d8 = static_cast<uint8_t *>(ctx->d_out);
  for (int i = 0; i < n; i++) {
    const FrameView &v = frames[i]->v;
    memset(&hj[i], 0, sizeof(FilterJob));
    for (int l = 0; l < v.levels; l++) { hj[i].level[l] = v.level[l]; hj[i].lw[l] = v.lw[l]; hj[i].lh[l] = v.lh[l]; }
    hj[i].corners = v.corners;
    hj[i].n_ptr = v.corner_hdr;
    hj[i].scores = reinterpret_cast<const double *>(d8) + static_cast<size_t>(i) * ccap;
    hj[i].locked = reinterpret_cast<const uint32_t *>(static_cast<uint8_t *>(dsx) + jb) + static_cast<size_t>(i) * mask_words;
    hj[i].out = reinterpret_cast<sdvl_filtered_corner *>(d8 + sc_bytes + cnt_bytes) + static_cast<size_t>(i) * max_out;
    hj[i].out_count = reinterpret_cast<int32_t *>(d8 + sc_bytes) + i;
    hj[i].levels = v.levels;
    hj[i].ccap = ccap;
  }
  memcpy(static_cast<uint8_t *>(hs) + jb, locked_cells, mb);
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, jb + mb));
  if (n_cells <= kBinCellsSmall && ccap <= kBinCornersSmall)
    SDVL_LAUNCH(ctx, "filter_select", (filter_select_binned_kernel<kBinCellsSmall, kBinCornersSmall>), dim3(n), dim3(256), static_cast<const FilterJob *>(dsx),
                cell_size, grid_w, n_cells, margin, min_feature_score, max_out);
  else if (n_cells <= kBinCells)
    SDVL_LAUNCH(ctx, "filter_select", (filter_select_binned_kernel<kBinCells, kFilterMaxCorners>), dim3(n), dim3(256), static_cast<const FilterJob *>(dsx),
                cell_size, grid_w, n_cells, margin, min_feature_score, max_out);
  else
    SDVL_LAUNCH(ctx, "filter_select", filter_select_kernel, dim3(n), dim3(256), static_cast<const FilterJob *>(dsx), cell_size, grid_w, n_cells, margin,
                min_feature_score, max_out);
  SDVL_LAUNCH(ctx, "filter_describe", filter_describe_kernel, xcd_frame_grid(n, std::min(max_out, 512)), dim3(64), static_cast<const FilterJob *>(dsx),
              max_out, n, std::min(max_out, 512));
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, sdvl_pull(ctx, ctx->h_out, d8 + sc_bytes, cnt_bytes + rec_bytes));
  SDVL_HIP_CHECK(ctx, sdvl_mark_record(ctx, SDVL_MARK_FILTER, &ctx->filter_ticket));
  ctx->filter_pending = n;
  ctx->filter_ccap = max_out;
  ctx->filter_sc_bytes = cnt_bytes;
  return SDVL_OK;
}

int sdvl_filter_corners_end(sdvl_ctx *ctx, int n, int cap, int32_t *counts, sdvl_filtered_corner *out) {
  if (!ctx || n < 0 || (n > 0 && (!counts || !out)) || cap <= 0) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, ctx->filter_pending == n, "sdvl_filter_corners_end without a matching sdvl_filter_corners_begin");
  ctx->filter_pending = 0;
  SDVL_HIP_CHECK(ctx, sdvl_mark_wait(ctx, SDVL_MARK_FILTER, ctx->filter_ticket));
  const int max_out = ctx->filter_ccap;
  const uint8_t *h = static_cast<const uint8_t *>(ctx->h_out);
  const int32_t *hc = reinterpret_cast<const int32_t *>(h);
  const sdvl_filtered_corner *hr = reinterpret_cast<const sdvl_filtered_corner *>(h + ctx->filter_sc_bytes);
  for (int i = 0; i < n; i++) {
    if (hc[i] > max_out || hc[i] > cap) {
      ctx->err = "more filtered corners than the output holds";
      return SDVL_ERR_CAPACITY;
    }
    counts[i] = hc[i];
    memcpy(out + static_cast<size_t>(i) * cap, hr + static_cast<size_t>(i) * max_out, sizeof(sdvl_filtered_corner) * static_cast<size_t>(hc[i]));
  }
  return SDVL_OK;
}

int sdvl_filter_inputs(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int cap, int32_t *xyl, double *scores, uint8_t *desc,
                       int32_t *counts) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !xyl || !scores || !counts)) || cap <= 0) return SDVL_ERR_INVALID;
  const int rc = sdvl_filter_inputs_begin(ctx, n, frames, cap, desc ? 1 : 0);
  if (rc) return rc;
  return sdvl_filter_inputs_end(ctx, n, frames, cap, xyl, scores, desc, counts);
}

int sdvl_frame_download_descriptors(sdvl_ctx *ctx, const sdvl_frame *f, int cap, uint8_t *out) {
  if (!ctx || !f || !out) return SDVL_ERR_INVALID;
  {
    int cnt = 0;
    int rc0 = sdvl_frame_count_host(ctx, const_cast<sdvl_frame *>(f), &cnt);
    if (rc0) return rc0;
  }
  if (!f->desc_valid && f->v.n_corners != 0) {
    sdvl_frame *fm = const_cast<sdvl_frame *>(f);
    const int rc1 = sdvl_orb_describe(ctx, 1, &fm, SDVL_MAX_CORNERS, nullptr);
    if (rc1) return rc1;
  }
  if (f->v.n_corners > cap) {
    ctx->err = "descriptor output capacity smaller than the corner count";
    return SDVL_ERR_CAPACITY;
  }
  if (f->v.n_corners == 0) return SDVL_OK;
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(out, f->v.desc, static_cast<size_t>(f->v.n_corners) * 32, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  return SDVL_OK;
}

int sdvl_orb_describe_points(sdvl_ctx *ctx, const sdvl_frame *f, int n, const int32_t *xyl, uint8_t *out_desc,
                             float *out_angle_deg) {
  if (!ctx || !f || n < 0 || (n > 0 && (!xyl || !out_desc))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  for (int i = 0; i < n; i++) {
    const int x = xyl[3 * i], y = xyl[3 * i + 1], l = xyl[3 * i + 2];
    SDVL_REQUIRE(ctx, l >= 0 && l < f->v.levels && x >= 0 && y >= 0 && x < f->v.lw[l] && y < f->v.lh[l], "point outside its pyramid level");
  }
  // device scratch: d_stage = corners[n][4] | OrbJob ; d_out = desc[n][32] | angle[n]
  const size_t c_bytes = sizeof(int32_t) * 4 * n, d_bytes = 32 * static_cast<size_t>(n), a_bytes = sizeof(float) * n;
  const size_t job_off = (c_bytes + 255) / 256 * 256;
  const size_t a_off = (d_bytes + 255) / 256 * 256;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, a_off + a_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, a_off + a_bytes, true);
  if (!rc) rc = sdvl_stage_alloc(ctx, job_off + sizeof(OrbJob), &hs, &dsx);
  if (rc) return rc;
  int32_t *hc = static_cast<int32_t *>(hs);
  for (int i = 0; i < n; i++) { hc[4 * i] = xyl[3 * i]; hc[4 * i + 1] = xyl[3 * i + 1]; hc[4 * i + 2] = xyl[3 * i + 2]; hc[4 * i + 3] = 0; }
  OrbJob *hj = reinterpret_cast<OrbJob *>(static_cast<uint8_t *>(hs) + job_off);
  memset(hj, 0, sizeof(OrbJob));
  for (int l = 0; l < f->v.levels; l++) { hj->level[l] = f->v.level[l]; hj->lw[l] = f->v.lw[l]; hj->lh[l] = f->v.lh[l]; }
  hj->corners = static_cast<int32_t *>(dsx);
  hj->desc = static_cast<uint8_t *>(ctx->d_out);
  hj->out = nullptr;
  hj->out_angle = reinterpret_cast<float *>(static_cast<uint8_t *>(ctx->d_out) + a_off);
  hj->n_ptr = nullptr;
  hj->n = n;
  hj->levels = f->v.levels;
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, job_off + sizeof(OrbJob)));
  SDVL_LAUNCH(ctx, "orb_describe", orb_describe_kernel, xcd_frame_grid(1, n), dim3(64), reinterpret_cast<const OrbJob *>(static_cast<uint8_t *>(dsx) + job_off), 1, n);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, a_off + a_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  memcpy(out_desc, ctx->h_out, d_bytes);
  if (out_angle_deg) memcpy(out_angle_deg, static_cast<uint8_t *>(ctx->h_out) + a_off, a_bytes);
  return SDVL_OK;
}

int sdvl_hamming_argmin(sdvl_ctx *ctx, int n, const uint8_t *queries, const int32_t *cand_offsets, const uint8_t *cand_desc,
                        int threshold, int32_t *best_index, int32_t *best_dist) {
  if (!ctx || n < 0 || (n > 0 && (!queries || !cand_offsets || !best_index))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, cand_offsets[0] == 0, "cand_offsets[0] must be 0");
  for (int i = 0; i < n; i++) SDVL_REQUIRE(ctx, cand_offsets[i + 1] >= cand_offsets[i], "cand_offsets must not decrease");
  const int total = cand_offsets[n];
  SDVL_REQUIRE(ctx, total == 0 || cand_desc, "candidate descriptors missing");
  SDVL_REQUIRE(ctx, threshold >= 0 && threshold <= 256, "threshold outside [0, 256]");
  // staged: queries[n][32] | offsets[n+1] | cands[total][32] | HammingJob ; d_out = best[n][2]
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t q_bytes = 32 * static_cast<size_t>(n), o_off = up(q_bytes), c_off = o_off + up(sizeof(int32_t) * (n + 1)),
               j_off = c_off + up(32 * static_cast<size_t>(total)), r_bytes = sizeof(int32_t) * 2 * n;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, r_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, r_bytes, true);
  if (!rc) rc = sdvl_stage_alloc(ctx, j_off + sizeof(HammingJob), &hs, &dsx);
  if (rc) return rc;
  uint8_t *h = static_cast<uint8_t *>(hs), *d = static_cast<uint8_t *>(dsx);
  memcpy(h, queries, q_bytes);
  memcpy(h + o_off, cand_offsets, sizeof(int32_t) * (n + 1));
  if (total) memcpy(h + c_off, cand_desc, 32 * static_cast<size_t>(total));
  HammingJob *hj = reinterpret_cast<HammingJob *>(h + j_off);
  hj->queries = d;
  hj->offsets = reinterpret_cast<const int32_t *>(d + o_off);
  hj->cands = d + c_off;
  hj->best = static_cast<int32_t *>(ctx->d_out);
  hj->n = n;
  hj->threshold = threshold;
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, j_off + sizeof(HammingJob)));
  SDVL_LAUNCH(ctx, "hamming_argmin", hamming_argmin_kernel, dim3(n), dim3(64), reinterpret_cast<const HammingJob *>(d + j_off));
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, r_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  const int32_t *r = static_cast<const int32_t *>(ctx->h_out);
  for (int i = 0; i < n; i++) {
    best_index[i] = r[2 * i];
    if (best_dist) best_dist[i] = r[2 * i + 1];
  }
  return SDVL_OK;
}

}  // extern "C"
