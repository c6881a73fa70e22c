// sdvl_detect.hip — K1 image pyramid and K2 per-cell FAST-9/16 for batches of frames (gfx950, wave64).
//   K1  pyr_down_kernel   : Frame::CreatePyramid, frame.cc:114-120 (cv::pyrDown 8UC1, REFLECT_101, (sum+128)>>8)
//   K2  fast_cells_kernel : the cv::FAST(roi, thr, nonmax=true) calls of FastDetector::SelectPixels,
//                           fast_detector.cc:79-106 — one workgroup per (frame, level, cell); a cell's ROI
//                           (<= 32x32 px) lives in LDS; segment test with 16-bit ring masks; score by the
//                           closed form max over the 16 arcs of min over 9; 3x3 strict NMS on an LDS score tile;
//                           ordered (row-major = cv::FAST scan order) compaction by a block prefix sum.
//       compact_cells_kernel: per frame exclusive scan of the cell counts + gather into a dense list.
// Integer arithmetic only: results are bit-exact against the oracle.
#include "sdvl_internal.h"

namespace {

constexpr int kPyrTW = 64, kPyrTH = 16;              // output tile of one 256-thread workgroup
constexpr int kPyrSW = 2 * kPyrTW + 3, kPyrSH = 2 * kPyrTH + 3;  // source tile 131 x 35

struct PyrJob {
  const uint8_t *src;
  uint8_t *dst;
  int sw, sh, dw, dh;
};

__device__ __forceinline__ int reflect101_clamped(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * n - 2 - p;
  return min(max(p, 0), n - 1);
}

__global__ __launch_bounds__(256) void pyr_down_kernel(const PyrJob *__restrict__ jobs) {
  __shared__ uint8_t s_src[kPyrSH][kPyrSW + 1];
  __shared__ uint16_t s_h[kPyrSH][kPyrTW];
  const PyrJob job = jobs[blockIdx.z];
  const int tx0 = blockIdx.x * kPyrTW, ty0 = blockIdx.y * kPyrTH;
  if (tx0 >= job.dw || ty0 >= job.dh) return;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < kPyrSH * kPyrSW; idx += 256) {
    const int r = idx / kPyrSW, c = idx - r * kPyrSW;
    const int sy = reflect101_clamped(2 * ty0 - 2 + r, job.sh);
    const int sx = reflect101_clamped(2 * tx0 - 2 + c, job.sw);
    s_src[r][c] = job.src[static_cast<size_t>(sy) * job.sw + sx];
  }
  __syncthreads();
  for (int idx = tid; idx < kPyrSH * kPyrTW; idx += 256) {
    const int r = idx / kPyrTW, x = idx - r * kPyrTW;
    const uint8_t *s = &s_src[r][2 * x];
    s_h[r][x] = static_cast<uint16_t>(s[0] + s[4] + 4 * (s[1] + s[3]) + 6 * s[2]);
  }
  __syncthreads();
  for (int idx = tid; idx < kPyrTH * kPyrTW; idx += 256) {
    const int y = idx / kPyrTW, x = idx - y * kPyrTW;
    const int oy = ty0 + y, ox = tx0 + x;
    if (oy < job.dh && ox < job.dw) {
      const int v = s_h[2 * y][x] + s_h[2 * y + 4][x] + 4 * (s_h[2 * y + 1][x] + s_h[2 * y + 3][x]) + 6 * s_h[2 * y + 2][x];
      job.dst[static_cast<size_t>(oy) * job.dw + ox] = static_cast<uint8_t>((v + 128) >> 8);
    }
  }
}

// ---------------------------------------------------------------------------------------------- FAST
struct FastLevels {
  int n_levels;
  int cell_begin[SDVL_MAX_LEVELS + 1];  // first global cell index of each level
  int wcells[SDVL_MAX_LEVELS];
  int cell_size, margin, threshold;
};

struct FastJob {
  const uint8_t *level[4];
  int lw[4], lh[4];
  uint32_t *cell_kps;
  int32_t *cell_counts;
};


constexpr int kTile = 32, kTilePitch = 36;

// has the 16-bit circular mask m a run of >= 9 ones?
__device__ __forceinline__ bool ring_run9(uint32_t m) {
  const uint32_t M = m | (m << 16);
  uint32_t r = M & (M >> 1);
  r &= (r >> 2);
  r &= (r >> 4);
  r &= (M >> 8);
  return (r & 0xFFFFu) != 0;
}

// cornerScore<16> in closed form: max(t, max_arcs min9(v-p), max_arcs min9(p-v)) - 1
__device__ __forceinline__ int fast_score(const int *d, int t) {
  int best = t;
#pragma unroll
  for (int sgn = 0; sgn < 2; sgn++) {
    int e[16], m2[16], m4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = sgn ? -d[k] : d[k];
#pragma unroll
    for (int k = 0; k < 16; k++) m2[k] = min(e[k], e[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int m8 = min(m4[k], m4[(k + 4) & 15]);
      best = max(best, min(m8, e[(k + 8) & 15]));
    }
  }
  return best - 1;
}

__global__ __launch_bounds__(256) void fast_cells_kernel(const FastJob *__restrict__ jobs, FastLevels lv) {
  __shared__ uint8_t s_img[kTile][kTilePitch];
  __shared__ uint8_t s_score[kTile][kTilePitch];
  __shared__ int s_wave_tot[4];
  const FastJob job = jobs[blockIdx.y];
  const int gcell = blockIdx.x;
  int l = 0;
  while (l + 1 < lv.n_levels && gcell >= lv.cell_begin[l + 1]) l++;
  const int c = gcell - lv.cell_begin[l];
  const int ci = c / lv.wcells[l], cj = c - ci * lv.wcells[l];
  const int W = job.lw[l], H = job.lh[l];
  const int y0 = max(lv.margin, ci * lv.cell_size), y1 = min(H - lv.margin, ci * lv.cell_size + lv.cell_size);
  const int x0 = max(lv.margin, cj * lv.cell_size), x1 = min(W - lv.margin, cj * lv.cell_size + lv.cell_size);
  const int tid = threadIdx.x;
  if (y1 <= y0 || x1 <= x0) {  // cell swallowed by the margin: cv::FAST is not called (fast_detector.cc:84-92)
    if (tid == 0) job.cell_counts[gcell] = 0;
    return;
  }
  const int rw = x1 - x0, rh = y1 - y0;  // <= 32
  const uint8_t *img = job.level[l];
  const int row = tid >> 3, cg = (tid & 7) * 4;
  // stage the ROI (zero outside it)
  {
    uint32_t pack = 0;
    if (row < rh) {
      const uint8_t *src = img + static_cast<size_t>(y0 + row) * W + x0 + cg;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (cg + k < rw) pack |= static_cast<uint32_t>(src[k]) << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(&s_img[row][cg]) = pack;
  }
  __syncthreads();
  const int t = lv.threshold;
  uint32_t sc_pack = 0;
  if (row >= 3 && row < rh - 3) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int x = cg + k;
      if (x < 3 || x >= rw - 3) continue;
      const int v = s_img[row][x];
      // Bresenham circle of radius 3 (cv::FAST offsets16); folded to immediates by the full unroll
      const int ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
      const int ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
      int d[16];
      uint32_t brighter = 0, darker = 0;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int p = s_img[row + ring_dy[r]][x + ring_dx[r]];
        d[r] = v - p;
        brighter |= (p > v + t ? 1u : 0u) << r;
        darker |= (p < v - t ? 1u : 0u) << r;
      }
      if (ring_run9(brighter) || ring_run9(darker)) {
        const int s = fast_score(d, t);
        sc_pack |= static_cast<uint32_t>(s & 0xFF) << (8 * k);  // stored as uchar like OpenCV's buf[]
      }
    }
  }
  *reinterpret_cast<uint32_t *>(&s_score[row][cg]) = sc_pack;
  __syncthreads();
  // 3x3 strict non-max suppression; survivors in row-major order
  uint32_t keep = 0;
  int cnt = 0;
  if (sc_pack) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int s = (sc_pack >> (8 * k)) & 0xFF;
      if (!s) continue;
      const int x = cg + k;  // x >= 3, row >= 3 here
      const bool ok = s > s_score[row][x - 1] && s > s_score[row][x + 1] && s > s_score[row - 1][x - 1] &&
                      s > s_score[row - 1][x] && s > s_score[row - 1][x + 1] && s > s_score[row + 1][x - 1] &&
                      s > s_score[row + 1][x] && s > s_score[row + 1][x + 1];
      if (ok) { keep |= 1u << k; cnt++; }
    }
  }
  // block exclusive prefix sum of cnt (thread order == scan order)
  const int lane = tid & 63, wave = tid >> 6;
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int n = __shfl_up(incl, off, 64);
    if (lane >= off) incl += n;
  }
  if (lane == 63) s_wave_tot[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; w++) base += s_wave_tot[w];
  const int total = s_wave_tot[0] + s_wave_tot[1] + s_wave_tot[2] + s_wave_tot[3];
  int pos = base + incl - cnt;
  uint32_t *out = job.cell_kps + static_cast<size_t>(gcell) * SDVL_CELL_KP_CAP;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (keep & (1u << k)) {
      if (pos < SDVL_CELL_KP_CAP) {
        const uint32_t s = (sc_pack >> (8 * k)) & 0xFF;
        out[pos] = static_cast<uint32_t>(x0 + cg + k) | (static_cast<uint32_t>(y0 + row) << 12) | (s << 24);
      }
      pos++;
    }
  }
  if (tid == 0) job.cell_counts[gcell] = min(total, SDVL_CELL_KP_CAP);
}

// one workgroup per frame: exclusive scan of the cell counts, dense gather (cell-major, scan order inside a cell)
__global__ __launch_bounds__(256) void compact_cells_kernel(const FastJob *__restrict__ jobs, int total_cells, int cap,
                                                            uint32_t *__restrict__ out_kps, int32_t *__restrict__ out_offsets) {
  __shared__ int s_tot[4];
  __shared__ int s_carry;
  const FastJob job = jobs[blockIdx.x];
  int32_t *offs = out_offsets + static_cast<size_t>(blockIdx.x) * (total_cells + 1);
  uint32_t *dst = out_kps + static_cast<size_t>(blockIdx.x) * cap;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int c0 = 0; c0 < total_cells; c0 += 256) {
    const int c = c0 + tid;
    const int cnt = (c < total_cells) ? job.cell_counts[c] : 0;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int n = __shfl_up(incl, off, 64);
      if (lane >= off) incl += n;
    }
    if (lane == 63) s_tot[wave] = incl;
    __syncthreads();
    int base = s_carry;
    for (int w = 0; w < wave; w++) base += s_tot[w];
    const int excl = base + incl - cnt;
    if (c < total_cells) {
      offs[c] = excl;
      const uint32_t *src = job.cell_kps + static_cast<size_t>(c) * SDVL_CELL_KP_CAP;
      for (int k = 0; k < cnt; k++)
        if (excl + k < cap) dst[excl + k] = src[k];
    }
    __syncthreads();
    if (tid == 0) s_carry += s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    __syncthreads();
  }
  if (tid == 0) offs[total_cells] = s_carry;
}


// ------------------------------------------------------------------------------------ corner selection on device
// The quota / retainBest half of FastDetector::SelectPixels (fast_detector.cc:108-151) without leaving the GPU.
// cv::KeyPointsFilter::retainBest = std::nth_element + std::partition; the surviving ORDER is whatever libstdc++'s
// introselect leaves, and that order matters downstream (first-wins ties in matcher.cc:280-283 and
// fast_detector.cc:208).  The functions below restate libstdc++ (GCC 11 bits/stl_algo.h, bits/stl_heap.h)
// __introselect / __unguarded_partition_pivot / __move_median_to_first / __insertion_sort / __heap_select and the
// bidirectional std::__partition statement by statement on packed keypoints (score in the top byte), so the device
// list is identical — element for element — to what the host calls produce on the same input order.
__device__ __forceinline__ bool kp_gt(uint32_t a, uint32_t b) { return (a >> 24) > (b >> 24); }  // response greater
__device__ __forceinline__ void kp_swap(uint32_t *v, int i, int j) { const uint32_t t = v[i]; v[i] = v[j]; v[j] = t; }

__device__ void sel_move_median_to_first(uint32_t *v, int result, int a, int b, int c) {
  if (kp_gt(v[a], v[b])) {
    if (kp_gt(v[b], v[c])) kp_swap(v, result, b);
    else if (kp_gt(v[a], v[c])) kp_swap(v, result, c);
    else kp_swap(v, result, a);
  } else if (kp_gt(v[a], v[c])) kp_swap(v, result, a);
  else if (kp_gt(v[b], v[c])) kp_swap(v, result, c);
  else kp_swap(v, result, b);
}

__device__ int sel_unguarded_partition(uint32_t *v, int first, int last, int pivot) {
  while (true) {
    while (kp_gt(v[first], v[pivot])) ++first;
    --last;
    while (kp_gt(v[pivot], v[last])) --last;
    if (!(first < last)) return first;
    kp_swap(v, first, last);
    ++first;
  }
}

__device__ void sel_push_heap(uint32_t *v, int first, int hole, int top, uint32_t value) {
  int parent = (hole - 1) / 2;
  while (hole > top && kp_gt(v[first + parent], value)) {
    v[first + hole] = v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[first + hole] = value;
}

__device__ void sel_adjust_heap(uint32_t *v, int first, int hole, int len, uint32_t value) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (kp_gt(v[first + second], v[first + (second - 1)])) second--;
    v[first + hole] = v[first + second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    v[first + hole] = v[first + (second - 1)];
    hole = second - 1;
  }
  sel_push_heap(v, first, hole, top, value);
}

__device__ void sel_heap_select(uint32_t *v, int first, int middle, int last) {
  const int len = middle - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      const uint32_t value = v[first + parent];
      sel_adjust_heap(v, first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  for (int i = middle; i < last; ++i)
    if (kp_gt(v[i], v[first])) {  // __pop_heap(first, middle, i)
      const uint32_t value = v[i];
      v[i] = v[first];
      sel_adjust_heap(v, first, 0, middle - first, value);
    }
}

__device__ void sel_insertion_sort(uint32_t *v, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    const uint32_t val = v[i];
    if (kp_gt(val, v[first])) {
      for (int k = i; k > first; --k) v[k] = v[k - 1];  // move_backward(first, i, i + 1)
      v[first] = val;
    } else {  // __unguarded_linear_insert
      int lastp = i, next = i - 1;
      while (kp_gt(val, v[next])) {
        v[lastp] = v[next];
        lastp = next;
        --next;
      }
      v[lastp] = val;
    }
  }
}

// std::nth_element(v+first, v+nth, v+last, response-greater)
__device__ void sel_nth_element(uint32_t *v, int first, int nth, int last) {
  if (first == last || nth == last) return;
  int depth_limit = (31 - __clz(last - first)) * 2;  // std::__lg(n) * 2
  while (last - first > 3) {
    if (depth_limit == 0) {
      sel_heap_select(v, first, nth + 1, last);
      kp_swap(v, first, nth);
      return;
    }
    --depth_limit;
    const int mid = first + (last - first) / 2;
    sel_move_median_to_first(v, first, first + 1, mid, last - 1);
    const int cut = sel_unguarded_partition(v, first + 1, last, first);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  sel_insertion_sort(v, first, last);
}

// cv::KeyPointsFilter::retainBest(kps, n_points) on v[0..len): returns the new length
__device__ int sel_retain_best(uint32_t *v, int len, int n_points) {
  if (n_points >= 0 && len > n_points) {
    if (n_points == 0) return 0;
    sel_nth_element(v, 0, n_points, len);
    const uint32_t amb = v[n_points - 1] >> 24;
    // std::partition(v+n_points, v+len, response >= amb)   (bidirectional __partition)
    int first = n_points, last = len;
    while (true) {
      while (true) {
        if (first == last) return first;
        else if ((v[first] >> 24) >= amb) ++first;
        else break;
      }
      --last;
      while (true) {
        if (first == last) return first;
        else if (!((v[last] >> 24) >= amb)) --last;
        else break;
      }
      kp_swap(v, first, last);
      ++first;
    }
  }
  return len;
}

constexpr int kSelMaxCells = 2048;  // cells of one level
constexpr int kSelStage = 8192;     // keypoints staged in LDS per round
constexpr int kSelFts = 4096;       // concatenated selection of one level before the final retainBest

struct SelLevels {
  int n_levels;
  int cell_begin[5];
  int wcells[4], hcells[4];
  int quota[4];
  int cell_size, margin;
};

struct SelJob {
  const uint32_t *cell_kps;
  const int32_t *cell_counts;
  int32_t *level_corners;  // [4][SDVL_MAX_CORNERS][4]
  int32_t *level_counts;   // [4]
  int32_t *corner_hdr;     // {count,0,0,0} + corners
  int lw[4], lh[4];
};

// one workgroup per (level, frame)
__global__ __launch_bounds__(256) void select_corners_kernel(const SelJob *__restrict__ jobs, SelLevels lv) {
  __shared__ uint32_t s_stage[kSelStage];
  __shared__ uint32_t s_fts[kSelFts];
  __shared__ uint8_t s_nleft[kSelMaxCells], s_nsel[kSelMaxCells], s_newlen[kSelMaxCells];
  __shared__ int s_off[kSelMaxCells];  // offset of the cell inside s_stage for the current round
  __shared__ int s_round[3];           // c0, c1, total staged
  __shared__ int s_nfts, s_overflow;
  const SelJob &job = jobs[blockIdx.y];
  const int l = blockIdx.x;
  const int tid = threadIdx.x;
  const int ncells = lv.wcells[l] * lv.hcells[l];
  const int cbeg = lv.cell_begin[l];
  const int nfeatures = lv.quota[l];
  // ---- quota loop, fast_detector.cc:108-135 (thread 0; a handful of passes over <= 2048 bytes of LDS)
  for (int c = tid; c < ncells; c += 256) {
    s_nleft[c] = static_cast<uint8_t>(job.cell_counts[cbeg + c]);
    s_nsel[c] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    int nempty = 0;
    for (int i = 0; i < lv.hcells[l]; i++) {
      const int inity = max(lv.margin, i * lv.cell_size), maxy = min(job.lh[l] - lv.margin, i * lv.cell_size + lv.cell_size);
      if (maxy <= inity) continue;
      for (int j = 0; j < lv.wcells[l]; j++) {
        const int initx = max(lv.margin, j * lv.cell_size), maxx = min(job.lw[l] - lv.margin, j * lv.cell_size + lv.cell_size);
        if (maxx <= initx) continue;
        if (s_nleft[i * lv.wcells[l] + j] == 0) nempty++;  // only cells where cv::FAST ran count as empty (Appendix B)
      }
    }
    int selected = 0;
    int cells_left = ncells - nempty;
    while ((nfeatures - selected) > 0 && cells_left > 0) {
      const int rem = nfeatures - selected;
      const int npercell = (rem + cells_left - 1) / cells_left;  // ceil(double(rem) / double(cells_left)), exact for ints this small
      cells_left = 0;
      for (int c = 0; c < ncells; c++) {
        const int nl = s_nleft[c];
        if (nl > 0) {
          if (nl > npercell) {
            s_nsel[c] = static_cast<uint8_t>(s_nsel[c] + npercell);
            selected += npercell;
            s_nleft[c] = static_cast<uint8_t>(nl - npercell);
            cells_left++;
          } else {
            s_nsel[c] = static_cast<uint8_t>(s_nsel[c] + nl);
            selected += nl;
            s_nleft[c] = 0;
          }
        }
      }
    }
    s_nfts = 0;
    s_overflow = 0;
  }
  __syncthreads();
  // ---- per-cell retainBest (fast_detector.cc:138-145): cells are staged into LDS in rounds, one thread per cell
  int c0 = 0;
  while (c0 < ncells) {
    if (tid == 0) {
      int tot = 0, c = c0;
      while (c < ncells) {
        const int cnt = job.cell_counts[cbeg + c];
        if (tot + cnt > kSelStage) break;
        s_off[c] = tot;
        tot += cnt;
        c++;
      }
      s_round[0] = c0;
      s_round[1] = c;
      s_round[2] = tot;
    }
    __syncthreads();
    const int c1 = s_round[1];
    for (int c = c0 + (tid >> 2); c < c1; c += 64) {  // 4 lanes copy one cell
      const int cnt = job.cell_counts[cbeg + c];
      const uint32_t *src = job.cell_kps + static_cast<size_t>(cbeg + c) * SDVL_CELL_KP_CAP;
      for (int k = (tid & 3); k < cnt; k += 4) s_stage[s_off[c] + k] = src[k];
    }
    __syncthreads();
    for (int c = c0 + tid; c < c1; c += 256) {
      const int cnt = job.cell_counts[cbeg + c];
      s_newlen[c] = static_cast<uint8_t>(sel_retain_best(&s_stage[s_off[c]], cnt, s_nsel[c]));
    }
    __syncthreads();
    if (tid == 0) {  // append the survivors in cell order (fast_detector.cc:141-143)
      int n = s_nfts;
      for (int c = c0; c < c1; c++) {
        const int len = s_newlen[c];
        if (n + len > kSelFts) { s_overflow = 1; break; }
        for (int k = 0; k < len; k++) s_fts[n + k] = s_stage[s_off[c] + k];
        n += len;
      }
      s_nfts = n;
    }
    __syncthreads();
    c0 = c1;
    if (s_overflow) break;
  }
  // ---- final retainBest over the level (fast_detector.cc:147-148)
  if (tid == 0 && !s_overflow) {
    if (s_nfts > nfeatures) s_nfts = sel_retain_best(s_fts, s_nfts, nfeatures);
  }
  __syncthreads();
  int n = s_overflow ? -1 : min(s_nfts, SDVL_MAX_CORNERS);
  int32_t *dst = job.level_corners + static_cast<size_t>(l) * SDVL_MAX_CORNERS * 4;
  for (int k = tid; k < n; k += 256) {
    const uint32_t v = s_fts[k];
    dst[4 * k] = static_cast<int32_t>(v & 0xFFF);
    dst[4 * k + 1] = static_cast<int32_t>((v >> 12) & 0xFFF);
    dst[4 * k + 2] = l;
    dst[4 * k + 3] = 0;
  }
  if (tid == 0) job.level_counts[l] = n;
}

// one workgroup per frame: level segments -> corners_ (fast_detector.cc:171-174 concatenates levels in order)
__global__ __launch_bounds__(256) void pack_corners_kernel(const SelJob *__restrict__ jobs, int n_levels, int32_t *__restrict__ batch_counts) {
  const SelJob &job = jobs[blockIdx.x];
  int off = 0;
  bool bad = false;
  for (int l = 0; l < n_levels; l++) {
    const int cnt = job.level_counts[l];
    if (cnt < 0 || off + cnt > SDVL_MAX_CORNERS) { bad = true; break; }
    const int4 *src = reinterpret_cast<const int4 *>(job.level_corners + static_cast<size_t>(l) * SDVL_MAX_CORNERS * 4);
    int4 *dst = reinterpret_cast<int4 *>(job.corner_hdr + 4) + off;
    for (int k = threadIdx.x; k < cnt; k += 256) dst[k] = src[k];
    off += cnt;
  }
  if (threadIdx.x == 0) {
    const int total = bad ? -1 : off;  // -1: a capacity overflowed; reported by sdvl_frames_corner_counts
    job.corner_hdr[0] = total < 0 ? 0 : total;
    job.corner_hdr[1] = total;
    if (batch_counts) batch_counts[blockIdx.x] = total;
  }
}

// diagnostic: retainBest of one list by one thread (tests the libstdc++ restatement against the host calls)
__global__ void retain_best_kernel(uint32_t *v, int len, int n_points, int *out_len) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out_len = sel_retain_best(v, len, n_points);
}

}  // namespace

extern "C" {

int sdvl_pyramid_build(sdvl_ctx *ctx, int n, sdvl_frame *const *frames) {
  if (!ctx || n < 0 || (n > 0 && !frames)) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  int levels = frames[0]->v.levels;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] != nullptr, "null frame");
    SDVL_REQUIRE(ctx, frames[i]->v.levels == levels && frames[i]->width == frames[0]->width && frames[i]->height == frames[0]->height,
                 "frames of one batch must share size and pyramid depth");
  }
  if (levels < 2) return SDVL_OK;
  const size_t bytes = sizeof(PyrJob) * n * (levels - 1);
  int rc = sdvl_ensure(ctx, &ctx->h_stage, &ctx->h_stage_bytes, bytes, true);
  if (rc) return rc;
  rc = sdvl_ensure(ctx, &ctx->d_stage, &ctx->d_stage_bytes, bytes, false);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));  // staging reuse
  PyrJob *hj = static_cast<PyrJob *>(ctx->h_stage);
  for (int l = 1; l < levels; l++)
    for (int i = 0; i < n; i++) {
      const FrameView &v = frames[i]->v;
      hj[(l - 1) * n + i] = PyrJob{v.level[l - 1], v.level[l], v.lw[l - 1], v.lh[l - 1], v.lw[l], v.lh[l]};
    }
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage, hj, bytes, hipMemcpyHostToDevice, ctx->stream));
  for (int l = 1; l < levels; l++) {
    const FrameView &v = frames[0]->v;
    dim3 grid((v.lw[l] + kPyrTW - 1) / kPyrTW, (v.lh[l] + kPyrTH - 1) / kPyrTH, n);
    ScopedKernelTimer tm(ctx, "pyr_down");
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, ctx->stream, static_cast<const PyrJob *>(ctx->d_stage) + (l - 1) * n);
  }
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

int sdvl_fast_num_cells(int width, int height, const sdvl_detect_params *p, int *cells_per_level, int *total) {
  if (!p || p->cell_size <= 0 || p->max_fast_levels < 1 || p->max_fast_levels > 4) return SDVL_ERR_INVALID;
  int w = width, h = height, tot = 0;
  for (int l = 0; l < p->max_fast_levels; l++) {
    const int wc = (w + p->cell_size - 1) / p->cell_size, hc = (h + p->cell_size - 1) / p->cell_size;
    if (cells_per_level) cells_per_level[l] = wc * hc;
    tot += wc * hc;
    w /= 2;
    h /= 2;
  }
  if (total) *total = tot;
  return SDVL_OK;
}

int sdvl_fast_cells(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const sdvl_detect_params *p, int cap,
                    sdvl_keypoint *out_kps, int32_t *out_cell_offsets) {
  if (!ctx || !p || n < 0 || (n > 0 && (!frames || !out_kps || !out_cell_offsets))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, p->cell_size >= 8 && p->cell_size <= kTile, "cell_size must be in [8,32]");
  SDVL_REQUIRE(ctx, p->max_fast_levels >= 1 && p->max_fast_levels <= 4, "max_fast_levels must be in [1,4]");
  SDVL_REQUIRE(ctx, p->margin >= 0 && cap > 0, "bad margin / capacity");
  const int W = frames[0]->width, H = frames[0]->height;
  FastLevels lv;
  memset(&lv, 0, sizeof(lv));
  lv.n_levels = p->max_fast_levels;
  lv.cell_size = p->cell_size;
  lv.margin = p->margin;
  lv.threshold = p->fast_threshold < 0 ? 0 : (p->fast_threshold > 255 ? 255 : p->fast_threshold);
  int total_cells = 0;
  for (int l = 0; l < lv.n_levels; l++) {
    SDVL_REQUIRE(ctx, l < frames[0]->v.levels, "max_fast_levels exceeds the pyramid depth");
    const int w = frames[0]->v.lw[l], h = frames[0]->v.lh[l];
    lv.cell_begin[l] = total_cells;
    lv.wcells[l] = (w + p->cell_size - 1) / p->cell_size;
    total_cells += lv.wcells[l] * ((h + p->cell_size - 1) / p->cell_size);
  }
  lv.cell_begin[lv.n_levels] = total_cells;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] && frames[i]->width == W && frames[i]->height == H && frames[i]->v.levels == frames[0]->v.levels,
                 "frames of one batch must share size and pyramid depth");
    if (total_cells > frames[i]->max_cells) {
      ctx->err = "cell grid larger than the frame's per-cell list capacity";
      return SDVL_ERR_CAPACITY;
    }
  }
  const size_t job_bytes = sizeof(FastJob) * n;
  const size_t offs_bytes = sizeof(int32_t) * static_cast<size_t>(n) * (total_cells + 1);
  const size_t kps_bytes = sizeof(uint32_t) * static_cast<size_t>(n) * cap;
  int rc = sdvl_ensure(ctx, &ctx->h_stage, &ctx->h_stage_bytes, job_bytes, true);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->d_stage, &ctx->d_stage_bytes, job_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, offs_bytes + kps_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, offs_bytes + kps_bytes, true);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  FastJob *hj = static_cast<FastJob *>(ctx->h_stage);
  for (int i = 0; i < n; i++) {
    memset(&hj[i], 0, sizeof(FastJob));
    for (int l = 0; l < lv.n_levels; l++) {
      hj[i].level[l] = frames[i]->v.level[l];
      hj[i].lw[l] = frames[i]->v.lw[l];
      hj[i].lh[l] = frames[i]->v.lh[l];
    }
    hj[i].cell_kps = frames[i]->cell_kps;
    hj[i].cell_counts = frames[i]->cell_counts;
  }
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage, hj, job_bytes, hipMemcpyHostToDevice, ctx->stream));
  int32_t *d_offs = static_cast<int32_t *>(ctx->d_out);
  uint32_t *d_kps = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ctx->d_out) + offs_bytes);
  {
    ScopedKernelTimer tm(ctx, "fast_cells");
    hipLaunchKernelGGL(fast_cells_kernel, dim3(total_cells, n), dim3(256), 0, ctx->stream,
                       static_cast<const FastJob *>(ctx->d_stage), lv);
  }
  {
    ScopedKernelTimer tm(ctx, "compact_cells");
    hipLaunchKernelGGL(compact_cells_kernel, dim3(n), dim3(256), 0, ctx->stream, static_cast<const FastJob *>(ctx->d_stage),
                       total_cells, cap, d_kps, d_offs);
  }
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  int32_t *h_offs = static_cast<int32_t *>(ctx->h_out);
  uint32_t *h_kps = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ctx->h_out) + offs_bytes);
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(h_offs, d_offs, offs_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  int max_total = 0;
  for (int i = 0; i < n; i++) {
    const int tot = h_offs[static_cast<size_t>(i) * (total_cells + 1) + total_cells];
    if (tot > cap) {
      ctx->err = "FAST keypoints of a frame exceed the caller's capacity";
      return SDVL_ERR_CAPACITY;
    }
    max_total = tot > max_total ? tot : max_total;
  }
  if (max_total > 0) {
    SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(h_kps, sizeof(uint32_t) * cap, d_kps, sizeof(uint32_t) * cap,
                                         sizeof(uint32_t) * max_total, n, hipMemcpyDeviceToHost, ctx->stream));
    SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  }
  memcpy(out_cell_offsets, h_offs, offs_bytes);
  for (int i = 0; i < n; i++) {
    const int32_t *offs = h_offs + static_cast<size_t>(i) * (total_cells + 1);
    const uint32_t *src = h_kps + static_cast<size_t>(i) * cap;
    sdvl_keypoint *dst = out_kps + static_cast<size_t>(i) * cap;
    int l = 0;
    for (int c = 0; c < total_cells; c++) {
      while (l + 1 < lv.n_levels && c >= lv.cell_begin[l + 1]) l++;
      for (int k = offs[c]; k < offs[c + 1]; k++) {
        const uint32_t v = src[k];
        dst[k].x = static_cast<uint16_t>(v & 0xFFF);
        dst[k].y = static_cast<uint16_t>((v >> 12) & 0xFFF);
        dst[k].score = static_cast<uint8_t>(v >> 24);
        dst[k].level = static_cast<uint8_t>(l);
        dst[k].cell = static_cast<uint16_t>(c - lv.cell_begin[l]);
      }
    }
  }
  return SDVL_OK;
}

// FastDetector::DetectPyramid (fast_detector.cc:154-175) entirely on device: per-cell FAST, quota + retainBest in
// libstdc++ order, level concatenation.  No host synchronisation; the corner count stays in HBM (frame header).
int sdvl_detect_corners(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const sdvl_detect_params *p, int nfeatures) {
  if (!ctx || !p || n < 0 || (n > 0 && !frames)) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, p->cell_size >= 8 && p->cell_size <= kTile, "cell_size must be in [8,32]");
  SDVL_REQUIRE(ctx, p->max_fast_levels >= 1 && p->max_fast_levels <= 4, "max_fast_levels must be in [1,4]");
  SDVL_REQUIRE(ctx, p->margin >= 0 && nfeatures >= 0, "bad margin / nfeatures");
  const int W = frames[0]->width, H = frames[0]->height;
  FastLevels lv;
  SelLevels sl;
  memset(&lv, 0, sizeof(lv));
  memset(&sl, 0, sizeof(sl));
  lv.n_levels = sl.n_levels = p->max_fast_levels;
  lv.cell_size = sl.cell_size = p->cell_size;
  lv.margin = sl.margin = p->margin;
  lv.threshold = p->fast_threshold < 0 ? 0 : (p->fast_threshold > 255 ? 255 : p->fast_threshold);
  // level quotas, fast_detector.cc:161-174
  const double scale = 1.2;
  double factor = 1.0, val = 0.0;
  for (int i = 0; i < p->max_fast_levels; i++) { val += factor; factor /= scale; }
  int levelfeatures = static_cast<int>(nfeatures / val);
  int total_cells = 0;
  for (int l = 0; l < lv.n_levels; l++) {
    SDVL_REQUIRE(ctx, l < frames[0]->v.levels, "max_fast_levels exceeds the pyramid depth");
    const int w = frames[0]->v.lw[l], h = frames[0]->v.lh[l];
    lv.cell_begin[l] = sl.cell_begin[l] = total_cells;
    lv.wcells[l] = sl.wcells[l] = (w + p->cell_size - 1) / p->cell_size;
    sl.hcells[l] = (h + p->cell_size - 1) / p->cell_size;
    SDVL_REQUIRE(ctx, sl.wcells[l] * sl.hcells[l] <= kSelMaxCells, "too many cells in one level for the selection kernel");
    total_cells += sl.wcells[l] * sl.hcells[l];
    sl.quota[l] = levelfeatures;
    levelfeatures = static_cast<int>(levelfeatures / scale);
  }
  lv.cell_begin[lv.n_levels] = sl.cell_begin[sl.n_levels] = total_cells;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] && frames[i]->width == W && frames[i]->height == H && frames[i]->v.levels == frames[0]->v.levels,
                 "frames of one batch must share size and pyramid depth");
    if (total_cells > frames[i]->max_cells) {
      ctx->err = "cell grid larger than the frame's per-cell list capacity";
      return SDVL_ERR_CAPACITY;
    }
  }
  const size_t fj_bytes = (sizeof(FastJob) * n + 255) / 256 * 256, sj_bytes = sizeof(SelJob) * n;
  int rc = sdvl_ensure(ctx, &ctx->h_stage, &ctx->h_stage_bytes, fj_bytes + sj_bytes, true);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->d_stage, &ctx->d_stage_bytes, fj_bytes + sj_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->d_counts, &ctx->d_counts_bytes, sizeof(int32_t) * n, false);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  ctx->detect_frames.assign(frames, frames + n);
  FastJob *hf = static_cast<FastJob *>(ctx->h_stage);
  SelJob *hs = reinterpret_cast<SelJob *>(static_cast<uint8_t *>(ctx->h_stage) + fj_bytes);
  for (int i = 0; i < n; i++) {
    memset(&hf[i], 0, sizeof(FastJob));
    memset(&hs[i], 0, sizeof(SelJob));
    for (int l = 0; l < lv.n_levels; l++) {
      hf[i].level[l] = frames[i]->v.level[l];
      hf[i].lw[l] = hs[i].lw[l] = frames[i]->v.lw[l];
      hf[i].lh[l] = hs[i].lh[l] = frames[i]->v.lh[l];
    }
    hf[i].cell_kps = frames[i]->cell_kps;
    hf[i].cell_counts = frames[i]->cell_counts;
    hs[i].cell_kps = frames[i]->cell_kps;
    hs[i].cell_counts = frames[i]->cell_counts;
    hs[i].level_corners = frames[i]->level_corners;
    hs[i].level_counts = frames[i]->level_counts;
    hs[i].corner_hdr = frames[i]->v.corner_hdr;
    frames[i]->v.n_corners = -1;  // known on the device only
    frames[i]->desc_valid = 0;
  }
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage, ctx->h_stage, fj_bytes + sj_bytes, hipMemcpyHostToDevice, ctx->stream));
  const FastJob *df = static_cast<const FastJob *>(ctx->d_stage);
  const SelJob *ds = reinterpret_cast<const SelJob *>(static_cast<uint8_t *>(ctx->d_stage) + fj_bytes);
  {
    ScopedKernelTimer tm(ctx, "fast_cells");
    hipLaunchKernelGGL(fast_cells_kernel, dim3(total_cells, n), dim3(256), 0, ctx->stream, df, lv);
  }
  {
    ScopedKernelTimer tm(ctx, "select_corners");
    hipLaunchKernelGGL(select_corners_kernel, dim3(lv.n_levels, n), dim3(256), 0, ctx->stream, ds, sl);
  }
  {
    ScopedKernelTimer tm(ctx, "pack_corners");
    hipLaunchKernelGGL(pack_corners_kernel, dim3(n), dim3(256), 0, ctx->stream, ds, lv.n_levels, static_cast<int32_t *>(ctx->d_counts));
  }
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

// corner counts of n frames in one transfer (blocking); refreshes the host copies
int sdvl_frames_corner_counts(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, int32_t *counts) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !counts))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  bool all_known = true;
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] != nullptr, "null frame");
    if (frames[i]->v.n_corners < 0) all_known = false;
  }
  if (!all_known) {
    const size_t nd = ctx->detect_frames.size();
    int rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, sizeof(int32_t) * (4 * static_cast<size_t>(n) + nd), true);
    if (rc) return rc;
    SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
    int32_t *h = static_cast<int32_t *>(ctx->h_out);
    // fast path: the pack kernel of the last detect batch left every count in one array
    bool batch_ok = nd > 0;
    std::vector<int> where(n, -1);
    if (batch_ok) {
      for (int i = 0; i < n && batch_ok; i++) {
        if (frames[i]->v.n_corners >= 0) continue;
        if (static_cast<size_t>(i) < nd && ctx->detect_frames[i] == frames[i]) where[i] = i;
        else {
          for (size_t k = 0; k < nd; k++)
            if (ctx->detect_frames[k] == frames[i]) { where[i] = static_cast<int>(k); break; }
          if (where[i] < 0) batch_ok = false;
        }
      }
    }
    if (batch_ok) {
      SDVL_HIP_CHECK(ctx, hipMemcpyAsync(h, ctx->d_counts, sizeof(int32_t) * nd, hipMemcpyDeviceToHost, ctx->stream));
      SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
      for (int i = 0; i < n; i++)
        if (frames[i]->v.n_corners < 0) {
          const int c = h[where[i]];
          if (c < 0) {
            ctx->err = "corner selection overflowed a device capacity (SDVL_MAX_CORNERS / level staging)";
            return SDVL_ERR_CAPACITY;
          }
          frames[i]->v.n_corners = c;
        }
    } else {
      for (int i = 0; i < n; i++)
        if (frames[i]->v.n_corners < 0)
          SDVL_HIP_CHECK(ctx, hipMemcpyAsync(h + 4 * i, frames[i]->v.corner_hdr, 16, hipMemcpyDeviceToHost, ctx->stream));
      SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
      for (int i = 0; i < n; i++)
        if (frames[i]->v.n_corners < 0) {
          if (h[4 * i + 1] < 0) {
            ctx->err = "corner selection overflowed a device capacity (SDVL_MAX_CORNERS / level staging)";
            return SDVL_ERR_CAPACITY;
          }
          frames[i]->v.n_corners = h[4 * i];
        }
    }
  }
  for (int i = 0; i < n; i++) counts[i] = frames[i]->v.n_corners;
  return SDVL_OK;
}

// diagnostic: cv::KeyPointsFilter::retainBest on packed keypoints (score = top byte) by the device restatement
int sdvl_retain_best(sdvl_ctx *ctx, uint32_t *packed, int len, int n_points, int *out_len) {
  if (!ctx || !packed || !out_len || len < 0 || len > (1 << 20)) return SDVL_ERR_INVALID;
  if (len == 0) { *out_len = 0; return SDVL_OK; }
  const size_t bytes = sizeof(uint32_t) * len + 64;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, bytes, true);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  memcpy(static_cast<uint8_t *>(ctx->h_out) + 64, packed, sizeof(uint32_t) * len);
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_out, ctx->h_out, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(retain_best_kernel, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ctx->d_out) + 64), len,
                     n_points, static_cast<int *>(ctx->d_out));
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  *out_len = *static_cast<int *>(ctx->h_out);
  memcpy(packed, static_cast<uint8_t *>(ctx->h_out) + 64, sizeof(uint32_t) * len);
  return SDVL_OK;
}

}  // extern "C"
