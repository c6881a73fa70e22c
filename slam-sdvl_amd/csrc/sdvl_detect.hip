// sdvl_detect.hip — K1 image pyramid and K2 per-cell FAST-9/16 for batches of frames (gfx950, wave64).
//   K1  pyr_down_kernel   : Frame::CreatePyramid, frame.cc:114-120 (cv::pyrDown 8UC1, REFLECT_101, (sum+128)>>8)
//   K2  fast_cells_kernel : the cv::FAST(roi, thr, nonmax=true) calls of FastDetector::SelectPixels,
//                           fast_detector.cc:79-106 — one workgroup per (frame, level, cell); a cell's ROI
//                           (<= 32x32 px) lives in LDS; segment test with 16-bit ring masks; score by the
//                           closed form max over the 16 arcs of min over 9; 3x3 strict NMS on an LDS score tile;
//                           ordered (row-major = cv::FAST scan order) compaction by a block prefix sum.
//       compact_cells_kernel: per frame exclusive scan of the cell counts + gather into a dense list.
// Integer arithmetic only: results are bit-exact against the oracle.
#include <atomic>

#include "sdvl_internal.h"

namespace {

// Output tile of one workgroup: 32 x 8 outputs for ONE WAVE (the form the launches use: no s_barrier — the lanes hand the
// source tile and the horizontal sums over through LDS behind a wave fence; among the other streams' kernels the 64 x 16 tile
// of a four-wave workgroup with its three barriers ran 3.5x longer than alone, a wave that never waits for another wave does not).
constexpr int kPyrTW = 32, kPyrTH = 8;

__device__ __forceinline__ void pyr_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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

// One workgroup = one 64 x 16 output tile.  The 35 source rows are staged with aligned 32-bit loads (byte b of an LDS row
// is source column 2*tx0 - 4 + b); the columns that fall off the image are patched afterwards from their BORDER_REFLECT_101
// mirror inside the same row.  Horizontal 1-4-6-4-1 pass into 16-bit sums, then every thread finishes 4 adjacent outputs
// and stores them as one word.
template <int kPyrTW, int kPyrTH>
__global__ __launch_bounds__(kPyrTW * kPyrTH / 4) void pyr_down_kernel(const PyrJob *__restrict__ jobs, int n_jobs, int gx, int gy) {
  constexpr int kThreads = kPyrTW * kPyrTH / 4;  // every thread finishes 4 adjacent outputs
  constexpr int kPyrSH = 2 * kPyrTH + 3;         // source rows of a tile
  constexpr int kPyrSWW = kPyrTW / 2 + 2;        // source words of a tile row: bytes [2*tx0 - 4, 2*tx0 + 2*TW + 4)
  constexpr int kPyrPitch = (kPyrSWW + 1) * 4;   // LDS row pitch in bytes (odd number of words: rows start in different banks)
  const auto sync = [] {
    if (kThreads == 64) pyr_wave_sync();
    else __syncthreads();
  };
  __shared__ uint32_t s_srcw[kPyrSH * (kPyrSWW + 1)];
  __shared__ __attribute__((aligned(8))) uint16_t s_h[kPyrSH][kPyrTW];
  // Round 4: a 1-D grid dealt so that ALL tiles of a frame run on one XCD (workgroups go round-robin over the 8 XCDs: id & 7), in
  // row-major tile order.  With the plain (x, y, frame) grid the x-neighbours of a 32-px tile row landed on eight different XCDs and
  // each of their L2s fetched the same 128-B source lines, as did the vertical halo rows: 106 MB of HBM traffic per dispatch
  // against 32.6 MB algorithmic (profiles/r03/pmc_hbm_traffic.csv).
  const int id = blockIdx.x, per_frame = gx * gy;
  const int slot = id >> 3;
  const int bz = (slot / per_frame) * 8 + (id & 7);
  if (bz >= n_jobs) return;
  const int tq = slot % per_frame;
  const int by = tq / gx, bx = tq - by * gx;
  const PyrJob job = jobs[bz];
  const int tx0 = bx * kPyrTW, ty0 = by * kPyrTH;
  if (tx0 >= job.dw || ty0 >= job.dh) return;
  const int tid = threadIdx.x;
  uint8_t *s_bytes = reinterpret_cast<uint8_t *>(s_srcw);
  const int xb = 2 * tx0 - 4;  // source column of byte 0 of an LDS row
  // the part of the tile that exists: nx x ny outputs, fed by source rows 0 .. 2*ny+2 and row bytes 2 .. 2*nx+4
  const int nx = min(kPyrTW, job.dw - tx0), ny = min(kPyrTH, job.dh - ty0);
  const int rows = 2 * ny + 3, last_byte = 2 * nx + 4;
  if ((job.sw & 3) == 0 && (reinterpret_cast<uintptr_t>(job.src) & 3u) == 0) {
    const int words_row = job.sw >> 2, w0 = xb >> 2;  // w0 = -1 for the leftmost tile
    const int nwords = last_byte / 4 + 1;
    // Round 4: four words per lane and load.  The word-by-word loop below spent ~150 of the kernel's 265 instructions per wave on
    // index arithmetic (a division, two clamps, a reflection and a 64-bit address per word, 5.3 rounds); here a row is covered by
    // kRowLanes lanes with one 16-byte load each, 12 rows per round.  A load is moved inwards where it would leave the row
    // (s = clamp(first word, 0, words_row - 4)): the words keep their own LDS column, columns whose source word lies off the image
    // are not written — the border fix below fills the bytes of theirs that are ever read.
    constexpr int kRowLanes = (kPyrSWW + 3) / 4, kRowsPerRound = kThreads / kRowLanes;
    if (words_row >= 4) {
      const int lr = tid / kRowLanes, lq = tid - lr * kRowLanes;
      const int first = w0 + 4 * lq;
      const int s = min(max(first, 0), words_row - 4);
      const int col0 = s - w0;
      if (lr < kRowsPerRound && 4 * lq < nwords) {
        for (int r = lr; r < rows; r += kRowsPerRound) {
          const int sy = reflect101_clamped(2 * ty0 - 2 + r, job.sh);
          // (global memory behind a scalar base, a 32-bit word index: one 16-byte load)
          const __attribute__((address_space(1))) uint32_t *srcw = (const __attribute__((address_space(1))) uint32_t *)job.src;
          const __attribute__((address_space(1))) uint32_t *pw = srcw + static_cast<uint32_t>(sy * words_row + s);
          const uint32_t v[4] = {pw[0], pw[1], pw[2], pw[3]};
          uint32_t *dst = &s_srcw[r * (kPyrSWW + 1)];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int c = col0 + k;
            if (c >= 0 && c < nwords) dst[c] = v[k];
          }
        }
      }
    } else
    for (int idx = tid; idx < rows * kPyrSWW; idx += kThreads) {
      const int r = idx / kPyrSWW, c = idx - r * kPyrSWW;  // constant divisor; narrow tiles skip the words they do not need
      if (c >= nwords) continue;
      const int sy = reflect101_clamped(2 * ty0 - 2 + r, job.sh);
      const int wi = min(max(w0 + c, 0), words_row - 1);
      s_srcw[r * (kPyrSWW + 1) + c] = reinterpret_cast<const uint32_t *>(job.src + static_cast<size_t>(sy) * job.sw)[wi];
    }
    sync();
    if (tid < rows && (xb < 0 || xb + last_byte >= job.sw)) {  // border tiles: one thread mends one row
      uint8_t *row = s_bytes + tid * kPyrPitch;
      if (xb < 0) {  // columns -2, -1 mirror columns 2, 1
        row[2] = row[6];
        row[3] = row[5];
      }
      for (int b = max(2, job.sw - xb); b <= last_byte; b++) row[b] = row[max(reflect101_clamped(xb + b, job.sw) - xb, 0)];
    }
  } else {  // odd widths: byte by byte
    const int ncols = last_byte - 1;
    for (int idx = tid; idx < rows * ncols; idx += kThreads) {
      const int r = idx / ncols, c = idx - r * ncols;
      const int sy = reflect101_clamped(2 * ty0 - 2 + r, job.sh);
      const int sx = reflect101_clamped(2 * tx0 - 2 + c, job.sw);
      s_bytes[r * kPyrPitch + 2 + c] = job.src[static_cast<size_t>(sy) * job.sw + sx];
    }
  }
  sync();
  // horizontal 1-4-6-4-1 for two adjacent outputs per lane on packed 16-bit lanes (sums <= 255 * 16): outputs x, x+1 read source
  // bytes 2x .. 2x+6 of the row = the last two bytes of word x/2, word x/2 + 1, the first byte of word x/2 + 2
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  for (int idx = tid; idx < rows * (kPyrTW / 2); idx += kThreads) {
    const int r = idx / (kPyrTW / 2), xp = idx - r * (kPyrTW / 2);
    if (2 * xp < nx) {
      const uint32_t *w = &s_srcw[r * (kPyrSWW + 1) + xp];
      const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
      // the five taps of an output as two byte dot products (v_dot4_u32_u8), the weights laid over the bytes of the words they fall
      // on: output x = bytes 2, 3 of w0 (1, 4) and 0 .. 2 of w1 (6, 4, 1); output x + 1 = all of w1 (1, 4, 6, 4) and byte 0 of w2.
      // (Round 4; before: five byte shuffles into 16-bit lanes and six packed operations per pair.)
      const uint32_t hx = __builtin_amdgcn_udot4(w1, 0x00010406u, __builtin_amdgcn_udot4(w0, 0x04010000u, 0u, false), false);
      const uint32_t hx1 = __builtin_amdgcn_udot4(w2, 0x00000001u, __builtin_amdgcn_udot4(w1, 0x04060401u, 0u, false), false);
      const uint32_t h = hx | (hx1 << 16);
      *reinterpret_cast<uint32_t *>(&s_h[r][2 * xp]) = __builtin_bit_cast(uint32_t, h);
    }
  }
  sync();
  {
    const int y = tid / (kPyrTW / 4), x = (tid % (kPyrTW / 4)) * 4;
    const int oy = ty0 + y, ox = tx0 + x;
    if (oy < job.dh && ox < job.dw) {
      // vertical pass on the same packed lanes: sums <= 4080 * 16 = 65280, + 128 still fits 16 bits
      u16x2 a01 = {0, 0}, a23 = {0, 0};
#pragma unroll
      for (int t = 0; t < 5; t++) {
        const unsigned short wg = (t == 0 || t == 4) ? 1 : (t == 2 ? 6 : 4);
        const u16x2 wgt = {wg, wg};
        const uint2 hv = *reinterpret_cast<const uint2 *>(&s_h[2 * y + t][x]);
        a01 += wgt * __builtin_bit_cast(u16x2, hv.x);
        a23 += wgt * __builtin_bit_cast(u16x2, hv.y);
      }
      const u16x2 half = {128, 128};
      a01 = (a01 + half) >> 8;
      a23 = (a23 + half) >> 8;
      int acc[4] = {a01.x, a01.y, a23.x, a23.y};
      uint8_t *d = job.dst + static_cast<size_t>(oy) * job.dw + ox;
      if (ox + 3 < job.dw && (reinterpret_cast<uintptr_t>(job.dst) & 3u) == 0 && (job.dw & 3) == 0) {  // (ox is a multiple of 4)
        __attribute__((address_space(1))) uint32_t *dstw = (__attribute__((address_space(1))) uint32_t *)job.dst;
        dstw[static_cast<uint32_t>(oy * job.dw + ox) >> 2] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, a23), __builtin_bit_cast(uint32_t, a01), 0x06040200u);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (ox + k < job.dw) d[k] = static_cast<uint8_t>(acc[k]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- FAST
struct FastLevels {
  int n_levels;
  int cell_begin[SDVL_MAX_LEVELS + 1];  // first global cell index of each level
  int wcells[SDVL_MAX_LEVELS];
  int cell_size, margin, threshold;
  int dense_num;  // a cell takes the DENSE path when more than dense_num of 64 probed pixels pass the compass test
};

// one cell of the detection grid: pyramid level and the ROI left after the image margin (fast_detector.cc:84-92).  Four 32-bit words:
// the record of a (wave-uniform) cell index comes in with ONE scalar load — 16-bit fields came in as four vector loads.
struct __attribute__((aligned(16))) CellGeo {
  uint32_t xy;  // x0 | y0 << 16
  uint32_t wh;  // rw | rh << 16; rw == 0: the margin swallows the cell
  int32_t level;     // level | quad_inv << 8: (65536 + nq - 1) / nq for the nq = (npr + 1) / 2 pairs of pairs of a tested row
  int32_t pair_inv;  // (65536 + npr - 1) / npr for the npr = (rw - 5) / 2 pixel pairs of a tested row: q / npr == (q * pair_inv) >> 16 for q < 1024
};

struct FastJob {
  const uint8_t *level[4];
  int lw[4], lh[4];
  uint32_t *cell_kps;
  int32_t *cell_counts;
};


constexpr int kTile = 32;
constexpr int kPitchW = 12;  // LDS row pitch in 32-bit words: 4 left pad + 32 px + 12 right pad bytes = 48 B
constexpr int kPadRows = 3;  // rows of padding above and below the 32 ROI rows

// has the 16-bit circular mask m a run of >= 9 ones?
__device__ __forceinline__ bool ring_run9(uint32_t m) {
  const uint32_t M = m | (m << 16);
  uint32_t r = M & (M >> 1);
  r &= (r >> 2);
  r &= (r >> 4);
  r &= (M >> 8);
  return (r & 0xFFFFu) != 0;
}

// byte `i` (0..11) of the three consecutive words w0 w1 w2
__device__ __forceinline__ int byte_of(uint32_t w0, uint32_t w1, uint32_t w2, int i) {
  return static_cast<int>(((i < 4 ? w0 : (i < 8 ? w1 : w2)) >> (8 * (i & 3))) & 0xFFu);
}

// One thread owns 4 horizontally adjacent pixels (x = cg .. cg+3).
// Phase A — compass pre-test.  Any 9 consecutive positions of the 16-ring contain two ADJACENT compass points
// (ring positions 0, 4, 8, 12 = (0,3) (3,0) (0,-3) (-3,0)), so a FAST-9 corner needs two adjacent compass pixels that are
// both brighter than v + t or both darker than v - t.  The test reads 5 aligned LDS words per thread (row y: bytes
// cg-4 .. cg+7, rows y-3 / y+3: bytes cg .. cg+3) and rejects most pixels; only the survivors get the 16-pixel
// arithmetic of phase C, one candidate per lane.
__device__ __forceinline__ bool fast_compass_pass(int v, int p_dn, int p_rt, int p_up, int p_lf, int t) {
  const int hi = v + t, lo = v - t;
  const bool b0 = p_dn > hi, b4 = p_rt > hi, b8 = p_up > hi, b12 = p_lf > hi;
  const bool d0 = p_dn < lo, d4 = p_rt < lo, d8 = p_up < lo, d12 = p_lf < lo;
  return (b0 && b4) || (b4 && b8) || (b8 && b12) || (b12 && b0) || (d0 && d4) || (d4 && d8) || (d8 && d12) || (d12 && d0);
}

#define SDVL_MIN2(a, b) min((a), (b))
#define SDVL_MAX2(a, b) max((a), (b))

// cornerScore<16> of the pixel at byte address (row, x) of the padded LDS tile in closed form:
// best = max(t, max_arcs min9(v-p), max_arcs min9(p-v)); OpenCV's score is best - 1.  The segment test itself falls out
// of the same number: a 9-arc with every |difference| > t exists  <=>  best > t  (cv::FAST_t / cornerScore<16>).
// v is the same in all 16 differences, so min9(v-p) = v - max9(p) and min9(p-v) = min9(p) - v: the windows run on the raw
// ring bytes (no subtraction per ring pixel), best = max(t, v - min_arcs max9(p), max_arcs min9(p) - v).  Integers: exact.
__device__ __forceinline__ int fast_corner_best(const uint8_t *tile, int pitch_bytes, int t) {
  const int v = tile[0];
  // p_k over the Bresenham circle, cv::FAST offsets16:
  // (dx,dy) = (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
#define SDVL_E(K, DX, DY) const int e##K = static_cast<int>(tile[(DY) * pitch_bytes + (DX)]);
  SDVL_E(0, 0, 3) SDVL_E(1, 1, 3) SDVL_E(2, 2, 2) SDVL_E(3, 3, 1) SDVL_E(4, 3, 0) SDVL_E(5, 3, -1) SDVL_E(6, 2, -2) SDVL_E(7, 1, -3)
  SDVL_E(8, 0, -3) SDVL_E(9, -1, -3) SDVL_E(10, -2, -2) SDVL_E(11, -3, -1) SDVL_E(12, -3, 0) SDVL_E(13, -3, 1) SDVL_E(14, -2, 2) SDVL_E(15, -1, 3)
#undef SDVL_E
  // sliding windows on the ring with 3-input min / max (v_min3_i32 / v_max3_i32): 3-windows, then 9-windows = three
  // 3-windows.  (Packed 16-bit min / max are 2-input: a 3-window costs two of them for two pixels — the same count per pixel.)
#define SDVL_MIN3(a, b, c) min(min((a), (b)), (c))
#define SDVL_MAX3(a, b, c) max(max((a), (b)), (c))
#define SDVL_W3(K, A, B, C) const int n##K = SDVL_MIN3(e##A, e##B, e##C), x##K = SDVL_MAX3(e##A, e##B, e##C);
  SDVL_W3(0, 0, 1, 2) SDVL_W3(1, 1, 2, 3) SDVL_W3(2, 2, 3, 4) SDVL_W3(3, 3, 4, 5) SDVL_W3(4, 4, 5, 6) SDVL_W3(5, 5, 6, 7)
  SDVL_W3(6, 6, 7, 8) SDVL_W3(7, 7, 8, 9) SDVL_W3(8, 8, 9, 10) SDVL_W3(9, 9, 10, 11) SDVL_W3(10, 10, 11, 12) SDVL_W3(11, 11, 12, 13)
  SDVL_W3(12, 12, 13, 14) SDVL_W3(13, 13, 14, 15) SDVL_W3(14, 14, 15, 0) SDVL_W3(15, 15, 0, 1)
#undef SDVL_W3
#define SDVL_W9(K, A, B, C) const int N##K = SDVL_MIN3(n##A, n##B, n##C), X##K = SDVL_MAX3(x##A, x##B, x##C);
  SDVL_W9(0, 0, 3, 6) SDVL_W9(1, 1, 4, 7) SDVL_W9(2, 2, 5, 8) SDVL_W9(3, 3, 6, 9) SDVL_W9(4, 4, 7, 10) SDVL_W9(5, 5, 8, 11)
  SDVL_W9(6, 6, 9, 12) SDVL_W9(7, 7, 10, 13) SDVL_W9(8, 8, 11, 14) SDVL_W9(9, 9, 12, 15) SDVL_W9(10, 10, 13, 0) SDVL_W9(11, 11, 14, 1)
  SDVL_W9(12, 12, 15, 2) SDVL_W9(13, 13, 0, 3) SDVL_W9(14, 14, 1, 4) SDVL_W9(15, 15, 2, 5)
#undef SDVL_W9
  const int max_n = SDVL_MAX3(SDVL_MAX3(N0, N1, N2), SDVL_MAX3(N3, N4, N5), SDVL_MAX3(SDVL_MAX3(N6, N7, N8), SDVL_MAX3(N9, N10, N11), SDVL_MAX3(N12, N13, SDVL_MAX2(N14, N15))));
  const int min_x = SDVL_MIN3(SDVL_MIN3(X0, X1, X2), SDVL_MIN3(X3, X4, X5), SDVL_MIN3(SDVL_MIN3(X6, X7, X8), SDVL_MIN3(X9, X10, X11), SDVL_MIN3(X12, X13, SDVL_MIN2(X14, X15))));
#undef SDVL_MIN3
#undef SDVL_MAX3
  return max(max(t, v - min_x), max_n - v);
}


// ---- the score of TWO horizontally adjacent pixels per lane on packed halves -------------------------------------------------
// gfx950 has three-input packed min / max on f16 pairs (v_pk_minimum3_f16 / v_pk_maximum3_f16).  Grey levels 0..255 and their
// differences are exact in f16, so the ring windows of fast_corner_best run on pixel PAIRS: the tile is kept a second time
// as halves (kPitchHW words per row, pixel x at half 4 + x), the pair (x, x+1) with x odd reads its ring as 20 aligned words
// — position (dx, dy) of both pixels is the word at half x + dx for odd dx, and two neighbouring words funnel-shifted by 16
// bits for even dx — and every min3 / max3 serves two pixels: 32 + 32 + 16 packed operations instead of 2 x 88 scalar ones.
constexpr int kPitchHW = 20;  // LDS row pitch of the half tile in 32-bit words: 4 + 32 + 4 halves

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 pk_min3(h2 a, h2 b, h2 c) {
  h2 d;
  asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ h2 pk_max3(h2 a, h2 b, h2 c) {
  h2 d;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ h2 as_h2(uint32_t w) { return __builtin_bit_cast(h2, w); }
// the two halves that straddle the words lo | hi: (hi : lo) >> 16
__device__ __forceinline__ h2 mid_h2(uint32_t hi, uint32_t lo) { return as_h2(__builtin_amdgcn_alignbit(hi, lo, 16)); }

// the 20 words of the half tile a pair's ring lies in: rows +3 / -3 words 1, 2 (a, b), rows +2 / -2 words 0 .. 3 (c, d), rows +1 / -1
// words 0 and 3 (u, l), row 0 words 0 .. 3 (z); word K of a row = halves (x - 3 + 2 K, x - 2 + 2 K) of the pair (x, x + 1), x odd
struct PairRing {
  uint32_t a1, a2, b1, b2, c0, c1, c2, c3, d0, d1, d2, d3, z0, z1, z2, z3, u0, u3, l0, l3;
};

// (best(x), best(x+1)) with best as in fast_corner_best; t2 = (t, t)
__device__ __forceinline__ h2 fast_pair_best_ring(const PairRing &q, h2 t2) {
  // cv::FAST offsets16: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
  const h2 e0 = mid_h2(q.a2, q.a1), e1 = as_h2(q.a2), e2 = mid_h2(q.c3, q.c2), e3 = as_h2(q.u3), e4 = as_h2(q.z3), e5 = as_h2(q.l3),
           e6 = mid_h2(q.d3, q.d2), e7 = as_h2(q.b2), e8 = mid_h2(q.b2, q.b1), e9 = as_h2(q.b1), e10 = mid_h2(q.d1, q.d0), e11 = as_h2(q.l0),
           e12 = as_h2(q.z0), e13 = as_h2(q.u0), e14 = mid_h2(q.c1, q.c0), e15 = as_h2(q.a1);
  const h2 v = mid_h2(q.z2, q.z1);
#define SDVL_W3(K, A, B, C) const h2 n##K = pk_min3(e##A, e##B, e##C), x##K = pk_max3(e##A, e##B, e##C);
  SDVL_W3(0, 0, 1, 2) SDVL_W3(1, 1, 2, 3) SDVL_W3(2, 2, 3, 4) SDVL_W3(3, 3, 4, 5) SDVL_W3(4, 4, 5, 6) SDVL_W3(5, 5, 6, 7)
  SDVL_W3(6, 6, 7, 8) SDVL_W3(7, 7, 8, 9) SDVL_W3(8, 8, 9, 10) SDVL_W3(9, 9, 10, 11) SDVL_W3(10, 10, 11, 12) SDVL_W3(11, 11, 12, 13)
  SDVL_W3(12, 12, 13, 14) SDVL_W3(13, 13, 14, 15) SDVL_W3(14, 14, 15, 0) SDVL_W3(15, 15, 0, 1)
#undef SDVL_W3
#define SDVL_W9(K, A, B, C) const h2 N##K = pk_min3(n##A, n##B, n##C), X##K = pk_max3(x##A, x##B, x##C);
  SDVL_W9(0, 0, 3, 6) SDVL_W9(1, 1, 4, 7) SDVL_W9(2, 2, 5, 8) SDVL_W9(3, 3, 6, 9) SDVL_W9(4, 4, 7, 10) SDVL_W9(5, 5, 8, 11)
  SDVL_W9(6, 6, 9, 12) SDVL_W9(7, 7, 10, 13) SDVL_W9(8, 8, 11, 14) SDVL_W9(9, 9, 12, 15) SDVL_W9(10, 10, 13, 0) SDVL_W9(11, 11, 14, 1)
  SDVL_W9(12, 12, 15, 2) SDVL_W9(13, 13, 0, 3) SDVL_W9(14, 14, 1, 4) SDVL_W9(15, 15, 2, 5)
#undef SDVL_W9
  const h2 max_n = pk_max3(pk_max3(pk_max3(N0, N1, N2), pk_max3(N3, N4, N5), pk_max3(N6, N7, N8)),
                           pk_max3(pk_max3(N9, N10, N11), pk_max3(N12, N13, N14), N15), N15);
  const h2 min_x = pk_min3(pk_min3(pk_min3(X0, X1, X2), pk_min3(X3, X4, X5), pk_min3(X6, X7, X8)),
                           pk_min3(pk_min3(X9, X10, X11), pk_min3(X12, X13, X14), X15), X15);
  return pk_max3(t2, v - min_x, max_n - v);
}

// Two horizontally adjacent pairs (x .. x+3, x = 3 mod 4) at once: w as for the first pair and EVEN (8-byte aligned).  The second
// pair's ring is the first one's moved by one word, so the two share 11 of their 2 x 20 words: 29 loads, the even-odd word pairs
// as 64-bit loads (which the LDS serves at twice the rate of two 32-bit ones).
__device__ __forceinline__ void fast_quad_best(const uint32_t *w, h2 t2, h2 *best_a, h2 *best_b) {
  const auto w2 = [w](int dy, int k) { return *reinterpret_cast<const uint2 *>(&w[(dy + 3) * kPitchHW + k]); };  // k even
  const auto w1 = [w](int dy, int k) { return w[(dy + 3) * kPitchHW + k]; };
  const uint32_t a1 = w1(3, 1), b1 = w1(-3, 1);
  const uint2 a23 = w2(3, 2), b23 = w2(-3, 2);
  const uint2 c01 = w2(2, 0), c23 = w2(2, 2), d01 = w2(-2, 0), d23 = w2(-2, 2), z01 = w2(0, 0), z23 = w2(0, 2), u01 = w2(1, 0), l01 = w2(-1, 0);
  const uint32_t c4 = w1(2, 4), d4 = w1(-2, 4), z4 = w1(0, 4), u3 = w1(1, 3), u4 = w1(1, 4), l3 = w1(-1, 3), l4 = w1(-1, 4);
  const PairRing qa = {a1, a23.x, b1, b23.x, c01.x, c01.y, c23.x, c23.y, d01.x, d01.y, d23.x, d23.y, z01.x, z01.y, z23.x, z23.y, u01.x, u3, l01.x, l3};
  const PairRing qb = {a23.x, a23.y, b23.x, b23.y, c01.y, c23.x, c23.y, c4, d01.y, d23.x, d23.y, d4, z01.y, z23.x, z23.y, z4, u01.y, u4, l01.y, l4};
  *best_a = fast_pair_best_ring(qa, t2);
  *best_b = fast_pair_best_ring(qb, t2);
}

// exclusive rank of (lane, bit k) in lane-major order over the wave for the 4-bit flag sets `flags`, and the wave total
__device__ __forceinline__ int wave_rank4(uint32_t flags, int *wave_total) {
  int below = 0, total = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned long long m = __ballot((flags >> k) & 1u);
    below += __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
    total += __popcll(m);
  }
  *wave_total = total;
  return below;
}

// The same work as fast_cells_kernel with a 64-lane workgroup: no s_barrier anywhere (the lanes of one wave hand data over
// through LDS behind a wave fence), no cross-wave prefix sums (ranks come from ballots and a running base), the probe's vote
// is a ballot, and only the tile / score planes of the chosen path are written: 7.5 KB of LDS per cell instead of 13 KB and
// four wave slots.  A workgroup of four waves spent most of its life waiting — job record, geometry, tile, four barriers —
// and held four wave slots while it did; one wave per cell holds one, so four times as many cells are in flight per CU.
__device__ __forceinline__ void fc_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// slot `pos` of a cell's corner list as (uniform base + 32-bit byte offset): one shift, a store with a scalar base
typedef __attribute__((address_space(1))) uint32_t *FcGlobalWords;
__device__ __forceinline__ FcGlobalWords fc_slot(FcGlobalWords base, int pos) {
  return (FcGlobalWords)((__attribute__((address_space(1))) char *)base + (static_cast<uint32_t>(pos) << 2));
}

constexpr int kFcDenseWords = (kTile + 2 * kPadRows) * kPitchHW + (kTile + 2) * kPitchHW;          // half tile + half scores: 1872
constexpr int kFcSparseWords = (kTile + 2 * kPadRows) * kPitchW + (kTile + 2) * kPitchW + kTile * kTile / 2;  // byte tile + byte scores + list
static_assert(kFcSparseWords <= kFcDenseWords, "the sparse layout lives inside the dense one");
static_assert(((kTile + 2 * kPadRows) * kPitchHW * 4) % 16 == 0, "the half score plane is cleared with 16-byte stores");

// The dense path's scores never leave the packed halves.  best >= t always, so "corner" is best - 1 >= t, a non-corner's best - 1 is
// t - 1, and a corner's score beats every non-corner's whether the plane holds 0 or t - 1 for those: the ring phase stores best2 - 1 as
// it comes (no conversion, no compare) and keeps it in a register; the suppression folds max(t - 1, 0) into the neighbours' maximum m,
// and then "is a corner AND beats its 8 neighbours" is the SIGN of m - score per half (integers in f16: exact, m == score gives +0).
// The second pixel of a row's last pair in an odd-width ROI is not a tested pixel: it is given t - 1, a non-corner.
// A lane takes FOUR adjacent pixels, two pairs, per pass — the rings of the two pairs share loads (fast_quad_best), and the index
// arithmetic, the suppression's neighbour loads and the ordered compaction are paid once per four pixels: 3 passes of 182 lane tasks.
// (Round 6: the kernel's LDS bank conflicts — 57 % of its LDS cycles — are hidden behind its vector issue: a conflict-free layout
//  halved the LDS cycles and changed nothing, profiles/r06/fast_cells_lds_experiment/.  The four-wave, pair-per-lane and integer-score
//  forms of rounds 2-4 are gone.)
__global__ __launch_bounds__(64) void fast_cells_wave_kernel(const FastJob *__restrict__ jobs, FastLevels lv, const CellGeo *__restrict__ cells) {
  __shared__ __attribute__((aligned(16))) uint32_t s_mem[kFcDenseWords];
  const FastJob &job = jobs[blockIdx.y];
  const int total_cells = lv.cell_begin[lv.n_levels];
  // XCD placement as in fast_cells_kernel: 4 horizontally adjacent cells per XCD and run of 32
  const int gcell = static_cast<int>(blockIdx.x & ~31u) + static_cast<int>(blockIdx.x & 7u) * 4 + static_cast<int>((blockIdx.x >> 3) & 3u);
  if (gcell >= total_cells) return;
  const CellGeo geo = cells[gcell];
  const int lane = threadIdx.x;
  if ((geo.wh & 0xFFFFu) == 0u) {  // cell swallowed by the margin: cv::FAST is not called (fast_detector.cc:84-92)
    if (lane == 0) job.cell_counts[gcell] = 0;
    return;
  }
  const int l = geo.level & 0xFF;
  const int W = job.lw[l];
  const int x0 = static_cast<int>(geo.xy & 0xFFFFu), y0 = static_cast<int>(geo.xy >> 16);
  const int rw = static_cast<int>(geo.wh & 0xFFFFu), rh = static_cast<int>(geo.wh >> 16);  // <= 32
  // the level as global memory behind a scalar base, 32-bit byte offsets from it (a generic pointer costs 64-bit vector
  // arithmetic per address and flat loads, which wait on two counters)
  typedef __attribute__((address_space(1))) const uint8_t *GlobalBytes;
  const GlobalBytes img = (GlobalBytes)job.level[l];
  const uint32_t roi0 = static_cast<uint32_t>(y0) * static_cast<uint32_t>(W) + static_cast<uint32_t>(x0);  // offset of the ROI's first pixel
  const int t = lv.threshold;
  // ---- density probe: compass pre-test on an 8 x 8 sample of the tested pixels, straight from the image
  bool dense;
  {
    const int tw = rw - 6, th = rh - 6;
    bool probed = false, passed = false;
    if (tw > 0 && th > 0) {
      const int pr = 3 + ((lane >> 3) * th >> 3), px = 3 + ((lane & 7) * tw >> 3);
      const uint32_t q = roi0 + static_cast<uint32_t>(pr * W + px), w3 = static_cast<uint32_t>(3 * W);
      probed = true;
      passed = fast_compass_pass(img[q], img[q + w3], img[q + 3u], img[q - w3], img[q - 3u], t);
    }
    dense = 64 * __popcll(__ballot(passed)) > lv.dense_num * __popcll(__ballot(probed));
  }
  // ---- the tile: lane = (row, half row): 16 pixels = 4 words, so that (lane, word, byte) order is scan order
  const int row = lane >> 1, wbase = (lane & 1) * 4;
  uint32_t pk[4] = {0u, 0u, 0u, 0u};
  // Word-aligned level (every pyramid this library builds): the lane's 16 pixels are one 16-byte load from the word at or below
  // them plus — when the ROI starts off a word — the word behind, shifted into place (v_alignbyte); pixels right of the ROI come
  // along as they are (no decision reads them: every ring lies within 3 px of a tested pixel, inside the ROI).  The span of 36
  // bytes stays inside the row, or the rows below the ROI take the overrun of its last row.
  const int H = job.lh[l];
  // (W >= 40: an overrun of the span stays within the ONE row that is known to lie below)
  const bool aligned_rows = (reinterpret_cast<uintptr_t>(job.level[l]) & 3u) == 0 && (W & 3) == 0 && W >= 40 && ((x0 & ~3) + 36 <= W || y0 + rh < H);
  if (aligned_rows) {
    if (row < rh) {
      const int shift = x0 & 3;
      const uint32_t o = (roi0 & ~3u) + static_cast<uint32_t>(row * W + 4 * wbase);
      const __attribute__((address_space(1))) uint32_t *q4 = (const __attribute__((address_space(1))) uint32_t *)(img + o);
      uint32_t d[5] = {q4[0], q4[1], q4[2], q4[3], 0u};  // (one 16-byte load)
      if (shift) {
        d[4] = q4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) pk[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], static_cast<uint32_t>(shift));
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) pk[k] = d[k];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int cg = (wbase + k) * 4;
      uint32_t pack = 0;
      if (row < rh && cg < rw) {
        const uint32_t o = roi0 + static_cast<uint32_t>(row * W + cg);
#pragma unroll
        for (int b = 0; b < 4; b++)
          if (cg + b < rw) pack |= static_cast<uint32_t>(img[o + b]) << (8 * b);
      }
      pk[k] = pack;
    }
  }
  uint32_t *out = job.cell_kps + static_cast<size_t>(gcell) * SDVL_CELL_KP_CAP;
  int base = 0;  // survivors written so far, in cv::FAST's output order (row-major)
  if (dense) {
    uint32_t *s_imgh = s_mem;
    uint32_t *s_scoreh = s_mem + (kTile + 2 * kPadRows) * kPitchHW;
    for (int z = lane; z < (kTile + 2) * kPitchHW / 4; z += 64) reinterpret_cast<uint4 *>(s_scoreh)[z] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 4; k++) {  // pixels as halves: pixel x at half 4 + x
      const uint32_t pack = pk[k];
      uint2 hw;
      // the half 0x6400 | b IS 1024 + b (ulp 1 on [1024, 2048)): one byte shuffle per pixel pair instead of two conversions and
      // a pack.  Every use of a pixel is a min / max or a difference of two pixels: the offset never shows.
      hw.x = __builtin_amdgcn_perm(0x64646464u, pack, 0x04010400u);  // selector bytes 0-3: bytes of `pack`, 4-7: of the constant
      hw.y = __builtin_amdgcn_perm(0x64646464u, pack, 0x04030402u);
      *reinterpret_cast<uint2 *>(&s_imgh[(row + kPadRows) * kPitchHW + 2 + 2 * (wbase + k)]) = hw;
    }
    fc_wave_sync();
    const int tw = rw - 6;
    const int npr = tw > 0 ? (tw + 1) >> 1 : 1;
    const _Float16 th = static_cast<_Float16>(t);
    {
      const h2 t2 = h2{th, th}, one2 = h2{static_cast<_Float16>(1), static_cast<_Float16>(1)};
      const h2 below2 = t2 - one2;  // best - 1 of a pixel that is no corner
      const _Float16 tf = static_cast<_Float16>(t > 1 ? t - 1 : 0);
      const h2 floor2 = h2{tf, tf};
      const bool odd = (tw & 1) != 0;
      const int nq = (npr + 1) >> 1;                                  // pairs of pairs in a tested row, <= 7
      const int nquad = tw > 0 && rh > 6 ? nq * (rh - 6) : 0;         // <= 182
      const int qinv = geo.level >> 8;                                // q / nq == (q * qinv) >> 16
      const int nqpass = (nquad + 63) >> 6;                           // <= 3
      typedef __attribute__((address_space(3))) volatile _Float16 *LdsHalves;
      typedef __attribute__((address_space(3))) uint32_t *LdsWords;
      const FcGlobalWords gout = (FcGlobalWords)out;
      uint32_t sa[3], sb[3], wofs[3], xy[3];  // per pass: the two pairs' scores (halves), the first pair's word in the half tile, (x | y << 12) of its first pixel
#pragma unroll
      for (int ps = 0; ps < 3; ps++) {
        sa[ps] = sb[ps] = wofs[ps] = xy[ps] = 0u;
        if (ps < nqpass) {
          const int i = ps * 64 + lane;
          if (i < nquad) {
            const int qr = (i * qinv) >> 16, m = i - qr * nq;
            const int w = (qr + 3) * kPitchHW + 2 + 2 * m;  // even: tile row qr + 3 is ring row -3 of image row qr + 3; pixel x = 3 + 4 m
            h2 a, b;
            fast_quad_best(&s_imgh[w], t2, &a, &b);
            a = a - one2;
            b = b - one2;
            if (2 * m + 1 >= npr) b = below2;  // the row's last pair of pairs holds one pair when npr is odd
            if (odd) {  // the second pixel of the row's last pair (x + 1 == rw - 3) is not a tested pixel
              asm volatile("");
              if (2 * m + 1 == npr - 1) b.y = below2.y;
              if (2 * m == npr - 1) a.y = below2.y;
            }
            // halves 4 + x .. 4 + x + 3 of plane row qr + 4: the first on its own, the middle two are an aligned word, the last on its own
            _Float16 *sh = reinterpret_cast<_Float16 *>(s_scoreh) + 2 * w + 2 * kPitchHW + 3;
            const LdsHalves vsh = (LdsHalves)sh;
            vsh[0] = a.x;
            *(LdsWords)(sh + 1) = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x05040302u);  // (a.y | b.x << 16)
            vsh[3] = b.y;
            sa[ps] = __builtin_bit_cast(uint32_t, a);
            sb[ps] = __builtin_bit_cast(uint32_t, b);
            wofs[ps] = static_cast<uint32_t>(w);
            xy[ps] = static_cast<uint32_t>(x0 + 3 + 4 * m) | (static_cast<uint32_t>(y0 + 3 + qr) << 12);
          }
        }
      }
      fc_wave_sync();
#pragma unroll
      for (int ps = 0; ps < 3; ps++) {
        if (ps < nqpass) {
          const uint32_t own_a = sa[ps], own_b = sb[ps];
          uint32_t da = 0u, db = 0u;  // sign per half: corner that beats its 8 neighbours
          if (ps * 64 + lane < nquad) {
            // plane rows r, r + 1, r + 2 (r + 1 = the pixels' own), halves (x-1 | x) (x+1 | x+2) (x+3 | x+4): an odd word and an aligned pair
            const uint32_t *sw = &s_scoreh[wofs[ps] + kPitchHW + 1];
            const uint32_t u0 = sw[-kPitchHW], c0 = sw[0], d0 = sw[kPitchHW];
            const uint2 u12 = *reinterpret_cast<const uint2 *>(sw - kPitchHW + 1), c12 = *reinterpret_cast<const uint2 *>(sw + 1),
                        d12 = *reinterpret_cast<const uint2 *>(sw + kPitchHW + 1);
            const h2 up_a = pk_max3(as_h2(u0), mid_h2(u12.x, u0), as_h2(u12.x)), dn_a = pk_max3(as_h2(d0), mid_h2(d12.x, d0), as_h2(d12.x));
            const h2 up_b = pk_max3(as_h2(u12.x), mid_h2(u12.y, u12.x), as_h2(u12.y)), dn_b = pk_max3(as_h2(d12.x), mid_h2(d12.y, d12.x), as_h2(d12.y));
            const h2 m_a = pk_max3(up_a, dn_a, pk_max3(as_h2(c0), as_h2(c12.x), floor2));
            const h2 m_b = pk_max3(up_b, dn_b, pk_max3(as_h2(c12.x), as_h2(c12.y), floor2));
            da = __builtin_bit_cast(uint32_t, m_a - as_h2(own_a));
            db = __builtin_bit_cast(uint32_t, m_b - as_h2(own_b));
          }
          const uint32_t lo_a = (da >> 15) & 1u, lo_b = (db >> 15) & 1u;
          const bool ok0 = lo_a != 0u, ok1 = static_cast<int32_t>(da) < 0, ok2 = lo_b != 0u, ok3 = static_cast<int32_t>(db) < 0;
          const unsigned long long m0 = __ballot(ok0), m1 = __ballot(ok1), m2 = __ballot(ok2), m3 = __ballot(ok3);
          const auto below = [](unsigned long long m) {
            return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0)));
          };
          const int pos0 = base + below(m0) + below(m1) + below(m2) + below(m3);
          const h2 a = as_h2(own_a), b = as_h2(own_b);
          const uint32_t kp = xy[ps];
          const int after = base + __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
          // (a cell rarely fills its list: one scalar test for the pass instead of a vector compare per store)
          const auto emit = [&](auto checked) {
            int pos = pos0;
            if (ok0 && (!decltype(checked)::value || pos < SDVL_CELL_KP_CAP)) *fc_slot(gout, pos) = kp | (static_cast<uint32_t>(static_cast<uint16_t>(a.x)) << 24);
            pos += static_cast<int>(lo_a);
            if (ok1 && (!decltype(checked)::value || pos < SDVL_CELL_KP_CAP)) *fc_slot(gout, pos) = (kp + 1u) | (static_cast<uint32_t>(static_cast<uint16_t>(a.y)) << 24);
            pos += static_cast<int>(da >> 31);
            if (ok2 && (!decltype(checked)::value || pos < SDVL_CELL_KP_CAP)) *fc_slot(gout, pos) = (kp + 2u) | (static_cast<uint32_t>(static_cast<uint16_t>(b.x)) << 24);
            pos += static_cast<int>(lo_b);
            if (ok3 && (!decltype(checked)::value || pos < SDVL_CELL_KP_CAP)) *fc_slot(gout, pos) = (kp + 3u) | (static_cast<uint32_t>(static_cast<uint16_t>(b.y)) << 24);
          };
          if (after <= SDVL_CELL_KP_CAP) emit(std::false_type{});
          else emit(std::true_type{});
          base = after;
        }
      }
      if (lane == 0) job.cell_counts[gcell] = min(base, SDVL_CELL_KP_CAP);
      return;
    }
  } else {
    uint32_t *s_img = s_mem;
    uint32_t *s_score = s_mem + (kTile + 2 * kPadRows) * kPitchW;
    uint16_t *s_list = reinterpret_cast<uint16_t *>(s_mem + (kTile + 2 * kPadRows) * kPitchW + (kTile + 2) * kPitchW);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      s_img[(row + kPadRows) * kPitchW + 1 + wbase + k] = pk[k];
      s_score[(row + 1) * kPitchW + 1 + wbase + k] = 0;
    }
    fc_wave_sync();
    // ---- phase A: compass pre-test of the lane's 16 pixels (row `row`, columns 16 (lane & 1) ..)
    uint32_t cflags = 0;
    if (row >= 3 && row < rh - 3) {
      const uint32_t *qz = &s_img[(row + kPadRows) * kPitchW + wbase];  // words wbase-1 .. wbase+4 of the row (word 0 is the left pad)
      uint32_t rz[6];
#pragma unroll
      for (int k = 0; k < 6; k++) rz[k] = qz[k];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t up = s_img[(row + kPadRows - 3) * kPitchW + 1 + wbase + k], dn = s_img[(row + kPadRows + 3) * kPitchW + 1 + wbase + k];
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const int x = (wbase + k) * 4 + b;
          if (x < 3 || x >= rw - 3) continue;
          const int v = static_cast<int>((rz[k + 1] >> (8 * b)) & 0xFFu);
          if (fast_compass_pass(v, static_cast<int>((dn >> (8 * b)) & 0xFFu), byte_of(rz[k], rz[k + 1], rz[k + 2], 7 + b),
                                static_cast<int>((up >> (8 * b)) & 0xFFu), byte_of(rz[k], rz[k + 1], rz[k + 2], 1 + b), t))
            cflags |= 1u << (4 * k + b);
        }
      }
    }
    // ---- phase B: candidates listed in scan order: rank of (lane, bit) = bits of lower lanes + lower bits of this lane
    int below = 0, ncand = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const unsigned long long m = __ballot((cflags >> k) & 1u);
      below += __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
      ncand += __popcll(m);
    }
    {
      int cpos = below;
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (cflags & (1u << k)) s_list[cpos++] = static_cast<uint16_t>((row << 5) | (wbase * 4 + k));
    }
    fc_wave_sync();
    // ---- phase C: one candidate per lane and round; the corners are compacted in place at the head of the list (a corner's
    // slot never lies behind the candidate it came from), their scores go to the score plane
    uint8_t *score_bytes = reinterpret_cast<uint8_t *>(s_score);
    const uint8_t *img_bytes = reinterpret_cast<const uint8_t *>(s_img);
    int ncorner = 0;
    for (int c0 = 0; c0 < ncand; c0 += 64) {
      const int i = c0 + lane;
      int rc = 0;
      bool is = false;
      if (i < ncand) {
        rc = s_list[i];
        const int r = rc >> 5, x = rc & 31;
        const int best = fast_corner_best(img_bytes + (r + kPadRows) * (kPitchW * 4) + 4 + x, kPitchW * 4, t);
        if (best > t) {
          is = true;
          score_bytes[(r + 1) * (kPitchW * 4) + 4 + x] = static_cast<uint8_t>((best - 1) & 0xFF);  // uchar like OpenCV's score buffer
        }
      }
      const unsigned long long m = __ballot(is);
      fc_wave_sync();  // every lane has read its candidate
      if (is) s_list[ncorner + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0))] = static_cast<uint16_t>(rc);
      ncorner += __popcll(m);
    }
    fc_wave_sync();
    // ---- 3x3 strict non-max suppression of the corners and ordered output
    for (int j0 = 0; j0 < ncorner; j0 += 64) {
      const int j = j0 + lane;
      bool ok = false;
      int rc = 0, sc = 0;
      if (j < ncorner) {
        rc = s_list[j];
        const int r = rc >> 5, x = rc & 31;
        const uint8_t *q = score_bytes + (r + 1) * (kPitchW * 4) + 4 + x;
        const int pb = kPitchW * 4;
        sc = q[0];
        ok = sc > q[-1] && sc > q[1] && sc > q[-pb - 1] && sc > q[-pb] && sc > q[-pb + 1] && sc > q[pb - 1] && sc > q[pb] && sc > q[pb + 1];
      }
      const unsigned long long m = __ballot(ok);
      const int pos = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
      if (ok && pos < SDVL_CELL_KP_CAP)
        out[pos] = static_cast<uint32_t>(x0 + (rc & 31)) | (static_cast<uint32_t>(y0 + (rc >> 5)) << 12) | (static_cast<uint32_t>(sc) << 24);
      base += __popcll(m);
    }
  }
  if (lane == 0) job.cell_counts[gcell] = min(base, SDVL_CELL_KP_CAP);
}
#undef SDVL_MIN2
#undef SDVL_MAX2

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

__device__ __forceinline__ void sel_move_median_to_first(uint32_t *v, int result, int a, int b, int c) {
  if (kp_gt(v[a], v[b])) {
    if (kp_gt(v[b], v[c])) kp_swap(v, result, b);
    else if (kp_gt(v[a], v[c])) kp_swap(v, result, c);
    else kp_swap(v, result, a);
  } else if (kp_gt(v[a], v[c])) kp_swap(v, result, a);
  else if (kp_gt(v[b], v[c])) kp_swap(v, result, c);
  else kp_swap(v, result, b);
}

__device__ __forceinline__ int sel_unguarded_partition(uint32_t *v, int first, int last, int pivot) {
  while (true) {
    while (kp_gt(v[first], v[pivot])) ++first;
    --last;
    while (kp_gt(v[pivot], v[last])) --last;
    if (!(first < last)) return first;
    kp_swap(v, first, last);
    ++first;
  }
}

__device__ __forceinline__ void sel_push_heap(uint32_t *v, int first, int hole, int top, uint32_t value) {
  int parent = (hole - 1) / 2;
  while (hole > top && kp_gt(v[first + parent], value)) {
    v[first + hole] = v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[first + hole] = value;
}

__device__ __forceinline__ void sel_adjust_heap(uint32_t *v, int first, int hole, int len, uint32_t value) {
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

__device__ __forceinline__ void sel_heap_select(uint32_t *v, int first, int middle, int last) {
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

__device__ __forceinline__ void sel_insertion_sort(uint32_t *v, int first, int last) {
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
__device__ __forceinline__ void sel_nth_element(uint32_t *v, int first, int nth, int last) {
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
__device__ __forceinline__ int sel_retain_best(uint32_t *v, int len, int n_points) {
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
constexpr int kSelFts = 4096;       // concatenated selection of one level before the final retainBest

// Round 3: the selection is two launches of NARROW workgroups instead of one 1024-thread workgroup per (frame, level) that held
// 128 KB of LDS (a whole CU's worth: among the other streams' kernels it waited for a CU to drain, 104 us alone -> 460 us):
//   select_cells_kernel  one WAVE per (frame, level, 8 consecutive cells): the level's quota (one number, below), then the cells'
//                        retainBest by groups of 8 lanes; 11 KB of LDS, no workgroup barrier
//   select_pack_kernel   one 256-thread workgroup per frame: wave l concatenates level l's surviving lists and runs the level's
//                        retainBest (one wave, no barrier inside), then all four waves write corners_ and the 32-px bins
struct SelLevels {
  int n_levels;
  int cell_begin[5];
  int wcells[4], hcells[4];
  int quota[4];
  int cell_size, margin;
  int not_run[4];      // cells of the level whose ROI the margin swallows: cv::FAST is not called there (fast_detector.cc:84-94)
  int slice_begin[5];  // select_cells: first 8-cell slice of each level among the slices of one frame
  int fts_cap[4];      // select_pack: LDS entries for each level's concatenated list (longer lists go through HBM)
};

struct SelJob {
  uint32_t *cell_kps;          // detection scratch of the frame's batch slot: per-cell lists, rewritten in place by select_cells
  const int32_t *cell_counts;
  int32_t *cell_newlen;        // [total_cells] length of every cell's list after its retainBest
  uint32_t *spill;             // [n_levels][2 * kSelFts] words: a level's list + stopper lists when they outgrow the LDS share
  int32_t *corner_hdr;         // {count,0,0,0} + corners
  int lw[4], lh[4];
  int32_t *bin_start;          // [bin_cells + 1]
  uint2 *bin_entries;
  int bin_gw, bin_cells;
  int corner_cap;              // corners the frame's resident list holds
};

constexpr int kSelThreads = 1024;  // 16 waves: one lane per cell leaves <= ~20 divergent lanes per wave
constexpr int kSelWaves = kSelThreads / 64;
constexpr int kSelParMin = 96;     // ranges shorter than this are finished by one lane

// block-wide exclusive prefix sum of one int per thread; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int block_exclusive_scan(int v, int *s_wave, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int n = __shfl_up(incl, off, 64);
    if (lane >= off) incl += n;
  }
  __syncthreads();  // s_wave may still be read from a previous call
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kSelWaves; w++) {
    const int x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + incl - v;
}

// The two-pointer partitions of libstdc++ (__unguarded_partition and the bidirectional __partition) swap the k-th
// element that stops the left scan with the k-th element that stops the right scan for as long as the former lies left
// of the latter; no position takes part in two swaps.  So the final arrangement follows from two rank computations:
//   left-stoppers  L[0] < L[1] < ...  (ascending positions),  right-stoppers  R[0] > R[1] > ...  (descending positions),
//   K = #{k : L[k] < R[k]},  swap v[L[k]] <-> v[R[k]] for k < K.
// All threads call this with identical arguments.  is_left / is_right classify a packed keypoint.
template <typename FL, typename FR>
__device__ __forceinline__ void block_two_pointer_partition(uint32_t *v, int first, int last, uint16_t *Ls, uint16_t *Rs, int *s_wave, FL is_left, FR is_right,
                                            int *out_K, int *out_nL, int *out_nR) {
  const int tid = threadIdx.x;
  const int n = last - first;
  const int chunk = (n + kSelThreads - 1) / kSelThreads;
  const int my0 = min(last, first + tid * chunk), my1 = min(last, my0 + chunk);
  int cl = 0, cr = 0;
  for (int i = my0; i < my1; i++) {
    const uint32_t x = v[i];
    cl += is_left(x) ? 1 : 0;
    cr += is_right(x) ? 1 : 0;
  }
  int nL = 0, nR = 0;
  int el = block_exclusive_scan(cl, s_wave, &nL);
  int er = block_exclusive_scan(cr, s_wave, &nR);
  for (int i = my0; i < my1; i++) {
    const uint32_t x = v[i];
    if (is_left(x)) Ls[el++] = static_cast<uint16_t>(i);
    if (is_right(x)) Rs[nR - 1 - (er++)] = static_cast<uint16_t>(i);
  }
  __syncthreads();
  const int m = min(nL, nR);
  int ck = 0;
  for (int k = tid; k < m; k += kSelThreads) ck += (Ls[k] < Rs[k]) ? 1 : 0;
  int K = 0;
  block_exclusive_scan(ck, s_wave, &K);
  for (int k = tid; k < K; k += kSelThreads) {
    const int a = Ls[k], b = Rs[k];
    const uint32_t t = v[a];
    v[a] = v[b];
    v[b] = t;
  }
  __syncthreads();
  *out_K = K;
  *out_nL = nL;
  *out_nR = nR;
}

// sequential continuation of __introselect from an intermediate state (one lane)
__device__ __forceinline__ void sel_introselect_from(uint32_t *v, int first, int nth, int last, int depth_limit) {
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

// cv::KeyPointsFilter::retainBest(v[0..len), n_points) by the whole workgroup; every thread gets the new length
__device__ __forceinline__ int block_retain_best(uint32_t *v, int len, int n_points, uint16_t *Ls, uint16_t *Rs, int *s_wave) {
  if (!(n_points >= 0 && len > n_points)) return len;
  if (n_points == 0) return 0;
  const int tid = threadIdx.x;
  // ---- std::nth_element(v, v + n_points, v + len)
  {
    int first = 0, last = len;
    const int nth = n_points;
    int depth_limit = (31 - __clz(len)) * 2;
    while (last - first > 3 && last - first >= kSelParMin && depth_limit > 0) {
      --depth_limit;
      if (tid == 0) {
        const int mid = first + (last - first) / 2;
        sel_move_median_to_first(v, first, first + 1, mid, last - 1);
      }
      __syncthreads();
      const uint32_t pv = v[first] >> 24;
      int K, nL, nR;
      block_two_pointer_partition(
          v, first + 1, last, Ls, Rs, s_wave, [pv](uint32_t x) { return !((x >> 24) > pv); }, [pv](uint32_t x) { return !(pv > (x >> 24)); }, &K,
          &nL, &nR);
      // where the left scan finally stops: the next original left-stopper or the slot the last swap filled from the left
      int cut = 0x7FFFFFFF;
      if (K < nL) cut = Ls[K];
      if (K > 0) cut = min(cut, static_cast<int>(Rs[K - 1]));
      __syncthreads();  // Ls/Rs are rewritten by the next round
      if (cut <= nth) first = cut;
      else last = cut;
    }
    if (tid == 0) sel_introselect_from(v, first, nth, last, depth_limit);
    __syncthreads();
  }
  // ---- std::partition(v + n_points, v + len, response >= amb)
  const uint32_t amb = v[n_points - 1] >> 24;
  int new_len;
  if (len - n_points >= kSelParMin) {
    int K, nL, nR;
    block_two_pointer_partition(
        v, n_points, len, Ls, Rs, s_wave, [amb](uint32_t x) { return !((x >> 24) >= amb); }, [amb](uint32_t x) { return (x >> 24) >= amb; }, &K, &nL,
        &nR);
    new_len = n_points + nR;  // the elements that satisfy the predicate end up in front
    __syncthreads();
  } else {
    __shared__ int s_len;
    if (tid == 0) {
      int first = n_points, last = len;
      int res = -1;
      while (res < 0) {
        while (true) {
          if (first == last) { res = first; break; }
          else if ((v[first] >> 24) >= amb) ++first;
          else break;
        }
        if (res >= 0) break;
        --last;
        while (true) {
          if (first == last) { res = first; break; }
          else if (!((v[last] >> 24) >= amb)) --last;
          else break;
        }
        if (res >= 0) break;
        kp_swap(v, first, last);
        ++first;
      }
      s_len = res;
    }
    __syncthreads();
    new_len = s_len;
    __syncthreads();
  }
  return new_len;
}


// ---- the same retainBest by a GROUP of G consecutive lanes of one wave (G = 16: four cells per wave; G = 64: one wave for
// the per-level list).  No s_barrier: the lanes of a wave execute their LDS instructions in order, a fence keeps the
// compiler from moving them.  Groups of one wave may follow different control flow; inside a group every branch depends
// on group-uniform values only, so __ballot (active lanes) always carries the whole group.
__device__ __forceinline__ void sel_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int G>
__device__ __forceinline__ unsigned long long group_ballot(bool p, int shift) {
  const unsigned long long b = __ballot(p);
  if (G == 64) return b;
  return (b >> shift) & ((1ull << G) - 1ull);
}

constexpr int kSelGroupMin = 4;  // ranges shorter than this are finished by the group's first lane (>= 4: the median-of-3 positions must differ)
constexpr int kSelGroup = 8;     // lanes that share a cell's retainBest in select_cells_kernel

// block_two_pointer_partition for a group: left-stoppers ascending in Ls[0..nL), right-stoppers ASCENDING in Rs[0..nR) (the
// k-th from the right is Rs[nR - 1 - k]); L[k] < R[k] is monotone in k, so K is a count.  Returns K; *cut = where the left
// scan finally stops (the next original left-stopper or the slot the last swap filled from the left).
template <int G, typename IDX, typename FL, typename FR>
__device__ __forceinline__ int group_two_pointer_partition(uint32_t *v, int first, int last, IDX *Ls, IDX *Rs, int sub, int shift, FL is_left,
                                                           FR is_right, int *out_nR, int *out_cut) {
  const unsigned long long below = (1ull << sub) - 1ull;
  int nL = 0, nR = 0;
  for (int base = first; base < last; base += G) {
    const int i = base + sub;
    bool l = false, r = false;
    if (i < last) {
      const uint32_t x = v[i];
      l = is_left(x);
      r = is_right(x);
    }
    const unsigned long long bl = group_ballot<G>(l, shift), br = group_ballot<G>(r, shift);
    if (l) Ls[nL + __popcll(bl & below)] = static_cast<IDX>(i);
    if (r) Rs[nR + __popcll(br & below)] = static_cast<IDX>(i);
    nL += __popcll(bl);
    nR += __popcll(br);
  }
  sel_wave_sync();
  const int m = min(nL, nR);
  int K = 0;
  int a0 = 0, b0 = 0;  // the lane's pair of the first chunk stays in registers for the swap
  for (int base = 0; base < m; base += G) {
    const int k = base + sub;
    int a = 0, b = 0;
    if (k < m) {
      a = Ls[k];
      b = Rs[nR - 1 - k];
    }
    if (base == 0) { a0 = a; b0 = b; }
    const int c = __popcll(group_ballot<G>(k < m && a < b, shift));
    K += c;
    if (c < min(G, m - base)) break;
  }
  int cut = 0x7FFFFFFF;
  if (K < nL) cut = Ls[K];
  if (K > 0) cut = min(cut, static_cast<int>(Rs[nR - K]));
  for (int k = sub; k < K; k += G) {
    const int a = (k == sub) ? a0 : static_cast<int>(Ls[k]), b = (k == sub) ? b0 : static_cast<int>(Rs[nR - 1 - k]);
    const uint32_t t = v[a];
    v[a] = v[b];
    v[b] = t;
  }
  sel_wave_sync();
  *out_nR = nR;
  *out_cut = cut;
  return K;
}

// cv::KeyPointsFilter::retainBest(v[0..len), n_points) by a group; every lane of the group gets the new length
template <int G, typename IDX>
__device__ __forceinline__ int group_retain_best(uint32_t *v, int len, int n_points, IDX *Ls, IDX *Rs, int sub, int shift) {
  if (!(n_points >= 0 && len > n_points)) return len;
  if (n_points == 0) return 0;
  {  // ---- std::nth_element(v, v + n_points, v + len)
    int first = 0, last = len;
    const int nth = n_points;
    int depth_limit = (31 - __clz(len)) * 2;
    while (last - first > 3 && last - first >= kSelGroupMin && depth_limit > 0) {
      --depth_limit;
      // __move_median_to_first(first, first + 1, mid, last - 1): every lane reads the four values at once and decides; the
      // group's first lane stores the exchange (the range holds >= kSelGroupMin elements: the four positions differ)
      const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
      const uint32_t vf = v[first], va = v[ia], vb = v[ib], vc = v[ic];
      int pick;
      if (kp_gt(va, vb)) pick = kp_gt(vb, vc) ? ib : (kp_gt(va, vc) ? ic : ia);
      else pick = kp_gt(va, vc) ? ia : (kp_gt(vb, vc) ? ic : ib);
      const uint32_t vp = pick == ia ? va : (pick == ib ? vb : vc);
      sel_wave_sync();  // all lanes have read before the exchange is stored
      if (sub == 0) {
        v[first] = vp;
        v[pick] = vf;
      }
      sel_wave_sync();
      const uint32_t pv = vp >> 24;
      int nR, cut;
      group_two_pointer_partition<G, IDX>(
          v, first + 1, last, Ls, Rs, sub, shift, [pv](uint32_t x) { return !((x >> 24) > pv); }, [pv](uint32_t x) { return !(pv > (x >> 24)); }, &nR,
          &cut);
      if (cut <= nth) first = cut;
      else last = cut;
    }
    if (sub == 0) sel_introselect_from(v, first, nth, last, depth_limit);
    sel_wave_sync();
  }
  // ---- std::partition(v + n_points, v + len, response >= amb)
  const uint32_t amb = v[n_points - 1] >> 24;
  int nR, cut;
  group_two_pointer_partition<G, IDX>(
      v, n_points, len, Ls, Rs, sub, shift, [amb](uint32_t x) { return !((x >> 24) >= amb); }, [amb](uint32_t x) { return (x >> 24) >= amb; }, &nR, &cut);
  return n_points + nR;  // the elements that satisfy the predicate end up in front
}

constexpr int kSelCellsPerWave = 64 / kSelGroup;  // 8 cells per wave

// sum over the 64 lanes (all active), result in every lane: DPP adds inside the 16-lane rows, two row broadcasts, one readlane
__device__ __forceinline__ int sel_wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// The quota loop of FastDetector::SelectPixels (fast_detector.cc:108-135) for one level, by one wave.  Every pass hands each cell that
// still has keypoints left min(left, npercell) more, so after any pass a cell holds min(count, S) with S = the npercell values summed
// so far, and the loop's two running sums are sum(min(count, S)) and #(count > S): ONE number per level describes the outcome for
// every cell.  Returns S; cnt = the level's cell counts (bytes, LDS).
__device__ __forceinline__ int sel_level_quota(const uint8_t *cnt, int ncells, int cells_left, int nfeatures, int lane) {
  int S = 0, selected = 0;
  while ((nfeatures - selected) > 0 && cells_left > 0) {
    const int rem = nfeatures - selected;
    S += (rem + cells_left - 1) / cells_left;  // ceil(double(rem) / double(cells_left))
    int sel = 0, left = 0;
    for (int c = lane; c < ncells; c += 64) {
      const int n = cnt[c];
      sel += min(n, S);
      left += n > S ? 1 : 0;
    }
    selected = sel_wave_sum(sel);
    cells_left = sel_wave_sum(left);
  }
  return S;
}

// one wave per (frame, level, 8 consecutive cells): per-cell retainBest, fast_detector.cc:138-145.  The surviving list replaces the
// head of the cell's list in the detection scratch, its length goes to cell_newlen.
__global__ __launch_bounds__(64) void select_cells_kernel(const SelJob *__restrict__ jobs, SelLevels lv, int n_frames) {
  __shared__ uint8_t s_cnt[kSelMaxCells];
  __shared__ uint32_t s_list[kSelCellsPerWave][SDVL_CELL_KP_CAP];
  __shared__ uint8_t s_glr[kSelCellsPerWave][2 * SDVL_CELL_KP_CAP];  // stopper lists of the lane groups
  static_assert(SDVL_CELL_KP_CAP <= 255, "cell positions and counts are stored in a byte");
  // blocks b and b + 8 share an XCD (observed placement, used for speed only): all waves of a frame run on one, so the level's count
  // array and the neighbouring cells' lists come out of one L2
  const int slices = lv.slice_begin[lv.n_levels];
  const int b = static_cast<int>(blockIdx.x), q = b >> 3;
  const int fq = q / slices, s = q - fq * slices;
  const int f = fq * 8 + (b & 7);
  if (f >= n_frames) return;
  int l = 0;
  while (l + 1 < lv.n_levels && s >= lv.slice_begin[l + 1]) l++;
  const SelJob &job = jobs[f];
  const int lane = threadIdx.x;
  const int ncells = lv.wcells[l] * lv.hcells[l], cbeg = lv.cell_begin[l];
  // ---- the level's counts; cells where cv::FAST ran and found nothing (fast_detector.cc:104-105) = empty cells - cells it never ran on
  int zeros = 0;
  for (int c = lane; c < ncells; c += 64) {
    const int n = job.cell_counts[cbeg + c];
    s_cnt[c] = static_cast<uint8_t>(n);
    zeros += n == 0 ? 1 : 0;
  }
  const int nempty = sel_wave_sum(zeros) - lv.not_run[l];
  sel_wave_sync();
  const int S = sel_level_quota(s_cnt, ncells, ncells - nempty, lv.quota[l], lane);
  // ---- kSelGroup lanes per cell
  const int g = lane / kSelGroup, sub = lane % kSelGroup, shift = lane - sub;
  const int c = (s - lv.slice_begin[l]) * kSelCellsPerWave + g;
  if (c >= ncells) return;
  const int cnt = s_cnt[c];
  const int nsel = min(cnt, S);
  int nl = cnt;
  if (cnt > nsel) {
    uint32_t *list = job.cell_kps + static_cast<size_t>(cbeg + c) * SDVL_CELL_KP_CAP;
    uint32_t *v = s_list[g];
    for (int k = sub; k < cnt; k += kSelGroup) v[k] = list[k];
    sel_wave_sync();
    nl = group_retain_best<kSelGroup, uint8_t>(v, cnt, nsel, s_glr[g], s_glr[g] + cnt, sub, shift);
    sel_wave_sync();
    for (int k = sub; k < nl; k += kSelGroup) list[k] = v[k];
  }
  if (sub == 0) job.cell_newlen[cbeg + c] = nl;
}

// one workgroup per frame.  Wave l: the surviving lists of level l concatenated in cell order (fast_detector.cc:141-145), the level's
// retainBest (:147-148).  Then all waves: levels concatenated into corners_ (:151, 171-174) + the count, and the list binned by 32-px
// cell of level-0 coordinates for the searches (GetCornersInRange scans ALL corners of the frame for every point,
// matcher.cc:123-230: with the bins a search reads the handful of cells around its point).
// Dynamic LDS: [bin_cells] histogram | per level: list[fts_cap] (u32) | left-stoppers[fts_cap] (u16) | right-stoppers[fts_cap] (u16)
__global__ __launch_bounds__(256) void select_pack_kernel(const SelJob *__restrict__ jobs, SelLevels lv, int32_t *__restrict__ batch_counts,
                                                          int32_t *__restrict__ host_counts) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  __shared__ int s_n[4];
  __shared__ int s_part[256];
  const SelJob &job = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cells = job.bin_cells;
  int *s_hist = reinterpret_cast<int *>(s_dyn);
  uint32_t *lists[4];
  {
    size_t off = (static_cast<size_t>(cells > 0 ? cells : 1) * 4 + 15) / 16 * 16;
    for (int l = 0; l < 4; l++) {
      lists[l] = reinterpret_cast<uint32_t *>(s_dyn + off);
      if (l < lv.n_levels) off += static_cast<size_t>(lv.fts_cap[l]) * 8;
    }
  }
  if (wave < lv.n_levels) {
    const int l = wave;
    const int ncells = lv.wcells[l] * lv.hcells[l], cbeg = lv.cell_begin[l];
    const int per = (ncells + 63) / 64;
    const int c_lo = min(ncells, lane * per), c_hi = min(ncells, c_lo + per);
    int mine = 0;
    for (int c = c_lo; c < c_hi; c++) mine += job.cell_newlen[cbeg + c];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    const int nfts = __shfl(incl, 63, 64);
    int dst = incl - mine;
    int n = -1;  // a level of more than kSelFts candidates: reported as a capacity error, as before
    if (nfts <= kSelFts) {
      const bool fits = nfts <= lv.fts_cap[l];
      uint32_t *spill = job.spill + static_cast<size_t>(l) * 2 * kSelFts;
      // the gather: every lane copies the heads of its cells' lists
      if (fits) {
        uint32_t *v = lists[l];
        for (int c = c_lo; c < c_hi; c++) {
          const int len = job.cell_newlen[cbeg + c];
          const uint32_t *src = job.cell_kps + static_cast<size_t>(cbeg + c) * SDVL_CELL_KP_CAP;
          for (int k = 0; k < len; k++) v[dst++] = src[k];
        }
        sel_wave_sync();
        n = nfts;
        if (nfts > lv.quota[l]) {
          uint16_t *Ls = reinterpret_cast<uint16_t *>(v + lv.fts_cap[l]);
          n = group_retain_best<64, uint16_t>(v, nfts, lv.quota[l], Ls, Ls + lv.fts_cap[l], lane, 0);
        }
      } else {  // same statements on the frame's spill area in HBM (a wave's own stores and loads stay in order)
        uint32_t *v = spill;
        for (int c = c_lo; c < c_hi; c++) {
          const int len = job.cell_newlen[cbeg + c];
          const uint32_t *src = job.cell_kps + static_cast<size_t>(cbeg + c) * SDVL_CELL_KP_CAP;
          for (int k = 0; k < len; k++) v[dst++] = src[k];
        }
        sel_wave_sync();
        n = nfts;
        if (nfts > lv.quota[l]) {
          uint16_t *Ls = reinterpret_cast<uint16_t *>(v + kSelFts);
          n = group_retain_best<64, uint16_t>(v, nfts, lv.quota[l], Ls, Ls + kSelFts, lane, 0);
        }
        sel_wave_sync();
      }
      if (lane == 0) s_n[l] = fits ? n : (n | 0x40000000);  // bit 30: the list lies in the spill area
    } else if (lane == 0) {
      s_n[l] = -1;
    }
  }
  for (int c = tid; c < cells; c += 256) s_hist[c] = 0;
  __syncthreads();
  // ---- concatenation
  int lvl_off[5];
  bool bad = false;
  lvl_off[0] = 0;
  for (int l = 0; l < lv.n_levels; l++) {
    const int raw = s_n[l];
    if (raw < 0) { bad = true; lvl_off[l + 1] = lvl_off[l]; continue; }
    if (raw & 0x40000000) lists[l] = job.spill + static_cast<size_t>(l) * 2 * kSelFts;
    lvl_off[l + 1] = lvl_off[l] + (raw & 0x3FFFFFFF);
  }
  if (lvl_off[lv.n_levels] > job.corner_cap) bad = true;
  const int n = bad ? 0 : lvl_off[lv.n_levels];
  if (tid == 0) {
    const int total = bad ? -1 : n;  // -1: a capacity overflowed; reported by sdvl_frames_corner_counts
    job.corner_hdr[0] = n;
    job.corner_hdr[1] = total;
    if (batch_counts) batch_counts[blockIdx.x] = total;
    if (host_counts) host_counts[blockIdx.x] = total;
  }
  if (bad) {
    // an overflowed frame has NO corners: its bins must say so too — a tracked step's search for this frame is already queued and
    // walks the bins without looking at the corner count (stale offsets of the pooled frame's previous life otherwise)
    for (int c = tid; c <= cells; c += 256)
      if (cells > 0) job.bin_start[c] = 0;
    return;
  }
  int4 *out = reinterpret_cast<int4 *>(job.corner_hdr + 4);
  const auto cell_of = [&](uint32_t v, int l) {
    const int x = min(static_cast<int>(v & 0xFFF) << l, job.lw[0] - 1), y = min(static_cast<int>((v >> 12) & 0xFFF) << l, job.lh[0] - 1);
    return (y >> 5) * job.bin_gw + (x >> 5);
  };
  for (int l = 0; l < lv.n_levels; l++) {
    const int nl = lvl_off[l + 1] - lvl_off[l];
    const uint32_t *v = lists[l];
    for (int k = tid; k < nl; k += 256) {
      const uint32_t e = v[k];
      out[lvl_off[l] + k] = make_int4(static_cast<int>(e & 0xFFF), static_cast<int>((e >> 12) & 0xFFF), l, 0);
      if (cells > 0) atomicAdd(&s_hist[cell_of(e, l)], 1);
    }
  }
  if (cells <= 0) return;
  __syncthreads();
  // exclusive scan of the histogram: every thread owns a run of consecutive cells
  const int per = (cells + 255) / 256, c0 = min(cells, tid * per), c1 = min(cells, c0 + per);
  int sum = 0;
  for (int c = c0; c < c1; c++) sum += s_hist[c];
  s_part[tid] = sum;
  __syncthreads();
  if (tid < 64) {  // one wave scans the 256 partial sums (4 per lane)
    const int a0 = s_part[4 * tid], a1 = s_part[4 * tid + 1], a2 = s_part[4 * tid + 2], a3 = s_part[4 * tid + 3];
    int incl = a0 + a1 + a2 + a3;
    const int own = incl;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (tid >= o) incl += t;
    }
    const int ex = incl - own;
    s_part[4 * tid] = ex;
    s_part[4 * tid + 1] = ex + a0;
    s_part[4 * tid + 2] = ex + a0 + a1;
    s_part[4 * tid + 3] = ex + a0 + a1 + a2;
  }
  __syncthreads();
  {
    int run = s_part[tid];
    for (int c = c0; c < c1; c++) {
      const int v = s_hist[c];
      job.bin_start[c] = run;
      s_hist[c] = run;  // becomes the fill cursor of the cell
      run += v;
    }
    if (tid == 0) job.bin_start[cells] = n;
  }
  __syncthreads();
  for (int l = 0; l < lv.n_levels; l++) {
    const int nl = lvl_off[l + 1] - lvl_off[l];
    const uint32_t *v = lists[l];
    for (int k = tid; k < nl; k += 256) {
      const uint32_t e = v[k];
      const int slot = atomicAdd(&s_hist[cell_of(e, l)], 1);
      job.bin_entries[slot] = make_uint2((e & 0xFFFFFFu) | (static_cast<uint32_t>(l) << 24), static_cast<uint32_t>(lvl_off[l] + k));
    }
  }
}

// diagnostic: retainBest of one list by one thread (tests the libstdc++ restatement against the host calls)
__global__ __launch_bounds__(kSelThreads) void retain_best_kernel(uint32_t *v, int len, int n_points, int *out_len, int cooperative) {
  __shared__ uint32_t s_v[kSelFts];
  __shared__ uint16_t s_l[kSelFts], s_r[kSelFts];
  __shared__ int s_wave[kSelWaves];
  if (cooperative == 1 && len <= kSelFts) {  // the workgroup-cooperative form
    for (int i = threadIdx.x; i < len; i += kSelThreads) s_v[i] = v[i];
    __syncthreads();
    const int n = block_retain_best(s_v, len, n_points, s_l, s_r, s_wave);
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += kSelThreads) v[i] = s_v[i];
    if (threadIdx.x == 0) *out_len = n;
  } else if (cooperative == 2 && len <= SDVL_CELL_KP_CAP) {  // the lane-group form used per cell (run by the wave's second group: shifted ballots)
    uint8_t *gl = reinterpret_cast<uint8_t *>(s_l), *gr = reinterpret_cast<uint8_t *>(s_r);
    for (int i = threadIdx.x; i < len; i += kSelThreads) s_v[i] = v[i];
    __syncthreads();
    if (threadIdx.x >= kSelGroup && threadIdx.x < 2 * kSelGroup) {
      const int n = group_retain_best<kSelGroup, uint8_t>(s_v, len, n_points, gl, gr, threadIdx.x - kSelGroup, kSelGroup);
      if (threadIdx.x == kSelGroup) *out_len = n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += kSelThreads) v[i] = s_v[i];
  } else if (cooperative == 3 && len <= kSelFts) {            // the one-wave form used for the per-level list
    for (int i = threadIdx.x; i < len; i += kSelThreads) s_v[i] = v[i];
    __syncthreads();
    if (threadIdx.x < 64) {
      const int n = group_retain_best<64, uint16_t>(s_v, len, n_points, s_l, s_r, threadIdx.x, 0);
      if (threadIdx.x == 0) *out_len = n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += kSelThreads) v[i] = s_v[i];
  } else if (threadIdx.x == 0) {        // one lane
    *out_len = sel_retain_best(v, len, n_points);
  }
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
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_stage_alloc(ctx, bytes, &hs, &dsx);
  if (rc) return rc;
  PyrJob *hj = static_cast<PyrJob *>(hs);
  for (int l = 1; l < levels; l++)
    for (int i = 0; i < n; i++) {
      const FrameView &v = frames[i]->v;
      hj[(l - 1) * n + i] = PyrJob{v.level[l - 1], v.level[l], v.lw[l - 1], v.lh[l - 1], v.lw[l], v.lh[l]};
    }
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hj, bytes));
  for (int l = 1; l < levels; l++) {
    const FrameView &v = frames[0]->v;
    const int gx = (v.lw[l] + kPyrTW - 1) / kPyrTW, gy = (v.lh[l] + kPyrTH - 1) / kPyrTH;
    const dim3 grid = dim3(static_cast<unsigned>(gx) * gy * ((n + 7) / 8 * 8), 1, 1);
    SDVL_LAUNCH(ctx, "pyr_down", (pyr_down_kernel<kPyrTW, kPyrTH>), grid, dim3(64), static_cast<const PyrJob *>(dsx) + (l - 1) * n, n, gx, gy);
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

// Detection scratch of one batch slot: the per-cell FAST lists and counts, the lengths after the per-cell retainBest and the spill area
// of the per-level lists.  A pool of the CONTEXT sized by its largest batch, alive between fast_cells and select_pack of one
// submission only (round 3; it used to be part of every frame and stayed with a keyframe for good: 0.7 of its 1.43 MB).
struct DetectSlot {
  size_t counts_off, newlen_off, kps_off, spill_off, bytes;
};
static DetectSlot detect_slot_layout(int total_cells, int n_levels) {
  DetectSlot L;
  size_t off = 0;
  const auto take = [&off](size_t bytes) { const size_t at = off; off = (off + bytes + 255) / 256 * 256; return at; };
  L.counts_off = take(sizeof(int32_t) * (static_cast<size_t>(total_cells) + 1));
  L.newlen_off = take(sizeof(int32_t) * static_cast<size_t>(total_cells));
  L.kps_off = take(sizeof(uint32_t) * SDVL_CELL_KP_CAP * static_cast<size_t>(total_cells));
  L.spill_off = take(sizeof(uint32_t) * 2 * kSelFts * static_cast<size_t>(n_levels));
  L.bytes = off;
  return L;
}
static int detect_scratch(sdvl_ctx *ctx, int n, const DetectSlot &L, uint8_t **base) {
  const int rc = sdvl_ensure(ctx, &ctx->d_detect, &ctx->d_detect_bytes, L.bytes * static_cast<size_t>(n), false);
  if (rc) return rc;
  *base = static_cast<uint8_t *>(ctx->d_detect);
  return SDVL_OK;
}

int64_t sdvl_detect_scratch_bytes(int width, int height, const sdvl_detect_params *p) {
  int total = 0;
  if (sdvl_fast_num_cells(width, height, p, nullptr, &total) != SDVL_OK) return -1;
  return static_cast<int64_t>(detect_slot_layout(total, p->max_fast_levels).bytes);
}

// SDVL_FAST_INT_SCORES=1: the dense path's epilogue on integer scores (round 3's form; A/B measurements, tests)
// the share (of 64) of a cell's probed pixels that must pass the compass test for the cell to take the dense path
// (SDVL_FAST_DENSE_NUM: 0 = every cell dense, 64 = every cell through the candidate list; A/B and the sweep of profiles/r05)
static int fast_cells_dense_num() {
  static const int v = [] {
    const char *e = getenv("SDVL_FAST_DENSE_NUM");
    // 16 of 64: measured (profiles/r05/fast_dense_sweep.txt) — the candidate-list path only pays below ~25 % compass passes; at the
    // 29 % of the camera-like texture's level 0 (rounds 2-4: threshold 32) it cost 166 us per 256 frames against 155 now, 158 all dense
    const int n = e ? atoi(e) : 16;
    return n < 0 ? 0 : (n > 64 ? 64 : n);
  }();
  return v;
}

// the per-cell geometry table of fast_cells_kernel for this frame shape and grid, built once and kept in HBM
static int fast_cell_table(sdvl_ctx *ctx, const FastLevels &lv, const sdvl_frame *f0, const CellGeo **out) {
  const int total = lv.cell_begin[lv.n_levels];
  long long key = f0->width;
  key = key * 8191 + f0->height;
  key = key * 8191 + lv.n_levels;
  key = key * 8191 + lv.cell_size;
  key = key * 8191 + lv.margin;
  if (ctx->d_fast_table && ctx->fast_table_key == key && ctx->fast_table_cells == total) {
    *out = static_cast<const CellGeo *>(ctx->d_fast_table);
    return SDVL_OK;
  }
  std::vector<CellGeo> t(total);
  for (int l = 0; l < lv.n_levels; l++) {
    const int W = f0->v.lw[l], H = f0->v.lh[l];
    for (int c = 0; c < lv.cell_begin[l + 1] - lv.cell_begin[l]; c++) {
      const int ci = c / lv.wcells[l], cj = c - ci * lv.wcells[l];
      const int y0 = std::max(lv.margin, ci * lv.cell_size), y1 = std::min(H - lv.margin, ci * lv.cell_size + lv.cell_size);
      const int x0 = std::max(lv.margin, cj * lv.cell_size), x1 = std::min(W - lv.margin, cj * lv.cell_size + lv.cell_size);
      CellGeo &g = t[lv.cell_begin[l] + c];
      const bool empty = y1 <= y0 || x1 <= x0;
      g.xy = static_cast<uint32_t>(x0 & 0xFFFF) | (static_cast<uint32_t>(y0 & 0xFFFF) << 16);
      g.wh = empty ? 0u : (static_cast<uint32_t>(x1 - x0) | (static_cast<uint32_t>(y1 - y0) << 16));
      g.level = l;
      {
        const int tw = (empty ? 0 : x1 - x0) - 6, npr = tw > 0 ? (tw + 1) >> 1 : 1;
        g.pair_inv = (65536 + npr - 1) / npr;
        const int nq = (npr + 1) >> 1;
        g.level = l | (((65536 + nq - 1) / nq) << 8);
      }
    }
  }
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  SDVL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // a launch in flight may still read the old table
  if (ctx->d_fast_table) (void)hipFree(ctx->d_fast_table);
  ctx->d_fast_table = nullptr;
  SDVL_HIP_CHECK(ctx, hipMalloc(&ctx->d_fast_table, sizeof(CellGeo) * static_cast<size_t>(total > 0 ? total : 1)));
  if (total > 0) {
    SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_fast_table, t.data(), sizeof(CellGeo) * static_cast<size_t>(total), hipMemcpyHostToDevice, ctx->stream));
    SDVL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->fast_table_key = key;
  ctx->fast_table_cells = total;
  *out = static_cast<const CellGeo *>(ctx->d_fast_table);
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
  lv.dense_num = fast_cells_dense_num();
  int total_cells = 0;
  for (int l = 0; l < lv.n_levels; l++) {
    SDVL_REQUIRE(ctx, l < frames[0]->v.levels, "max_fast_levels exceeds the pyramid depth");
    const int w = frames[0]->v.lw[l], h = frames[0]->v.lh[l];
    lv.cell_begin[l] = total_cells;
    lv.wcells[l] = (w + p->cell_size - 1) / p->cell_size;
    total_cells += lv.wcells[l] * ((h + p->cell_size - 1) / p->cell_size);
  }
  lv.cell_begin[lv.n_levels] = total_cells;
  for (int i = 0; i < n; i++)
    SDVL_REQUIRE(ctx, frames[i] && frames[i]->width == W && frames[i]->height == H && frames[i]->v.levels == frames[0]->v.levels,
                 "frames of one batch must share size and pyramid depth");
  const DetectSlot slot = detect_slot_layout(total_cells, lv.n_levels);
  uint8_t *scratch = nullptr;
  {
    const int rc_s = detect_scratch(ctx, n, slot, &scratch);
    if (rc_s) return rc_s;
  }
  const size_t job_bytes = sizeof(FastJob) * n;
  const size_t offs_bytes = sizeof(int32_t) * static_cast<size_t>(n) * (total_cells + 1);
  const size_t kps_bytes = sizeof(uint32_t) * static_cast<size_t>(n) * cap;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, offs_bytes + kps_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, offs_bytes + kps_bytes, true);
  if (!rc) rc = sdvl_stage_alloc(ctx, job_bytes, &hs, &dsx);
  if (rc) return rc;
  FastJob *hj = static_cast<FastJob *>(hs);
  for (int i = 0; i < n; i++) {
    memset(&hj[i], 0, sizeof(FastJob));
    for (int l = 0; l < lv.n_levels; l++) {
      hj[i].level[l] = frames[i]->v.level[l];
      hj[i].lw[l] = frames[i]->v.lw[l];
      hj[i].lh[l] = frames[i]->v.lh[l];
    }
    hj[i].cell_kps = reinterpret_cast<uint32_t *>(scratch + slot.bytes * i + slot.kps_off);
    hj[i].cell_counts = reinterpret_cast<int32_t *>(scratch + slot.bytes * i + slot.counts_off);
  }
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hj, job_bytes));
  int32_t *d_offs = static_cast<int32_t *>(ctx->d_out);
  uint32_t *d_kps = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ctx->d_out) + offs_bytes);
  const CellGeo *d_cells = nullptr;
  {
    const int rc_t = fast_cell_table(ctx, lv, frames[0], &d_cells);
    if (rc_t) return rc_t;
  }
  SDVL_LAUNCH(ctx, "fast_cells", fast_cells_wave_kernel, dim3((total_cells + 31) / 32 * 32, n), dim3(64), static_cast<const FastJob *>(dsx), lv, d_cells);
  SDVL_LAUNCH(ctx, "compact_cells", compact_cells_kernel, dim3(n), dim3(256), static_cast<const FastJob *>(dsx), total_cells, cap, d_kps, d_offs);
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
  lv.dense_num = fast_cells_dense_num();
  // level quotas, fast_detector.cc:161-174
  const double scale = 1.2;
  double factor = 1.0, val = 0.0;
  for (int i = 0; i < p->max_fast_levels; i++) { val += factor; factor /= scale; }
  int levelfeatures = static_cast<int>(nfeatures / val);
  int total_cells = 0, n_slices = 0;
  size_t pack_lds = 0;
  for (int l = 0; l < lv.n_levels; l++) {
    SDVL_REQUIRE(ctx, l < frames[0]->v.levels, "max_fast_levels exceeds the pyramid depth");
    const int w = frames[0]->v.lw[l], h = frames[0]->v.lh[l];
    lv.cell_begin[l] = sl.cell_begin[l] = total_cells;
    lv.wcells[l] = sl.wcells[l] = (w + p->cell_size - 1) / p->cell_size;
    sl.hcells[l] = (h + p->cell_size - 1) / p->cell_size;
    const int ncells = sl.wcells[l] * sl.hcells[l];
    SDVL_REQUIRE(ctx, ncells <= kSelMaxCells, "too many cells in one level for the selection kernel");
    total_cells += ncells;
    sl.quota[l] = levelfeatures;
    levelfeatures = static_cast<int>(levelfeatures / scale);
    // cells whose ROI the margin swallows (cv::FAST is never called on them, fast_detector.cc:84-94)
    int not_run = 0;
    for (int ci = 0; ci < sl.hcells[l]; ci++)
      for (int cj = 0; cj < sl.wcells[l]; cj++) {
        const bool ran = std::min(h - p->margin, ci * p->cell_size + p->cell_size) > std::max(p->margin, ci * p->cell_size) &&
                         std::min(w - p->margin, cj * p->cell_size + p->cell_size) > std::max(p->margin, cj * p->cell_size);
        not_run += ran ? 0 : 1;
      }
    sl.not_run[l] = not_run;
    sl.slice_begin[l] = n_slices;
    n_slices += (ncells + kSelCellsPerWave - 1) / kSelCellsPerWave;
    // LDS entries of the level's concatenated list in select_pack: the quota loop selects < quota + cells, ties that the per-cell
    // retainBest keeps come on top (half as many again; a longer list is worked on in HBM)
    sl.fts_cap[l] = std::min(kSelFts, ((sl.quota[l] + ncells) * 3 / 2 + 63) / 64 * 64);
    pack_lds += static_cast<size_t>(sl.fts_cap[l]) * 8;
  }
  lv.cell_begin[lv.n_levels] = sl.cell_begin[sl.n_levels] = total_cells;
  sl.slice_begin[sl.n_levels] = n_slices;
  for (int i = 0; i < n; i++)
    SDVL_REQUIRE(ctx, frames[i] && frames[i]->width == W && frames[i]->height == H && frames[i]->v.levels == frames[0]->v.levels,
                 "frames of one batch must share size and pyramid depth");
  const DetectSlot slot = detect_slot_layout(total_cells, lv.n_levels);
  uint8_t *scratch = nullptr;
  {
    const int rc_s = detect_scratch(ctx, n, slot, &scratch);
    if (rc_s) return rc_s;
  }
  pack_lds += (static_cast<size_t>(std::max(1, frames[0]->bin_cells)) * 4 + 15) / 16 * 16;
  const size_t fj_bytes = (sizeof(FastJob) * n + 255) / 256 * 256, sj_bytes = sizeof(SelJob) * n;
  void *hst = nullptr, *dst = nullptr;
  int rc = sdvl_ensure(ctx, &ctx->d_counts, &ctx->d_counts_bytes, sizeof(int32_t) * n, false);
  if (!rc) rc = sdvl_stage_alloc(ctx, fj_bytes + sj_bytes, &hst, &dst);
  if (rc) return rc;
  ctx->detect_frames.assign(frames, frames + n);
  FastJob *hf = static_cast<FastJob *>(hst);
  SelJob *hs = reinterpret_cast<SelJob *>(static_cast<uint8_t *>(hst) + fj_bytes);
  for (int i = 0; i < n; i++) {
    memset(&hf[i], 0, sizeof(FastJob));
    memset(&hs[i], 0, sizeof(SelJob));
    for (int l = 0; l < lv.n_levels; l++) {
      hf[i].level[l] = frames[i]->v.level[l];
      hf[i].lw[l] = hs[i].lw[l] = frames[i]->v.lw[l];
      hf[i].lh[l] = hs[i].lh[l] = frames[i]->v.lh[l];
    }
    uint8_t *sb = scratch + slot.bytes * i;
    hf[i].cell_kps = hs[i].cell_kps = reinterpret_cast<uint32_t *>(sb + slot.kps_off);
    hf[i].cell_counts = reinterpret_cast<int32_t *>(sb + slot.counts_off);
    hs[i].cell_counts = hf[i].cell_counts;
    hs[i].cell_newlen = reinterpret_cast<int32_t *>(sb + slot.newlen_off);
    hs[i].spill = reinterpret_cast<uint32_t *>(sb + slot.spill_off);
    hs[i].corner_cap = frames[i]->corner_cap;
    hs[i].corner_hdr = frames[i]->v.corner_hdr;
    hs[i].bin_start = frames[i]->bin_start;
    hs[i].bin_entries = frames[i]->bin_entries;
    hs[i].bin_gw = frames[i]->bin_gw;
    hs[i].bin_cells = frames[i]->bin_cells;
    frames[i]->bins_valid = frames[i]->bin_cells > 0 ? 1 : 0;
    frames[i]->v.n_corners = -1;  // known on the device only
    frames[i]->hdr_stale = 0;     // the pack kernel rewrites the header
    frames[i]->desc_valid = 0;
  }
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dst, hst, fj_bytes + sj_bytes));
  const FastJob *df = static_cast<const FastJob *>(dst);
  const SelJob *ds = reinterpret_cast<const SelJob *>(static_cast<uint8_t *>(dst) + fj_bytes);
  const CellGeo *d_cells = nullptr;
  {
    const int rc_t = fast_cell_table(ctx, lv, frames[0], &d_cells);
    if (rc_t) return rc_t;
  }
  SDVL_LAUNCH(ctx, "fast_cells", fast_cells_wave_kernel, dim3((total_cells + 31) / 32 * 32, n), dim3(64), df, lv, d_cells);
  SDVL_LAUNCH(ctx, "select_cells", select_cells_kernel, dim3(static_cast<unsigned>((n + 7) / 8 * 8 * n_slices)), dim3(64), ds, sl, n);
  // the counts follow the kernels to the host without anyone waiting for them (see sdvl_frames_corner_counts): the pack kernel
  // writes them into device memory and, when results go direct, into the pinned host array as well
  const bool direct = sdvl_ensure(ctx, &ctx->h_counts, &ctx->h_counts_bytes, sizeof(int32_t) * n, true) == SDVL_OK;
  if (pack_lds > (48u << 10)) {
    // beyond the default dynamic-LDS limit.  The attribute belongs to the kernel object of a DEVICE, not to a context: it is raised
    // once per device to the most any configuration can ask for (the CU's 160 KB less the kernel's static LDS), whichever thread
    // gets there first — a per-context "already raised" mark let a second context with a smaller size lower it again
    static std::atomic<unsigned long long> attr_devices{0};
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(select_pack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(152u << 10)));
      attr_devices.fetch_or(bit, std::memory_order_release);
    }
    SDVL_REQUIRE(ctx, pack_lds <= (152u << 10), "selection lists of this configuration do not fit a compute unit's LDS");
    ctx->pack_lds_limit = pack_lds;
  }
  {
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "select_pack", &ev_a, &ev_b);
    hipExtLaunchKernelGGL(select_pack_kernel, dim3(n), dim3(256), static_cast<unsigned>(pack_lds), ctx->stream, ev_a, ev_b, 0, ds, sl,
                          static_cast<int32_t *>(ctx->d_counts), direct ? static_cast<int32_t *>(ctx->h_counts) : nullptr);
  }
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  ctx->counts_gen = ~0ull;
  if (direct) ctx->counts_gen = ctx->wait_gen;
  else if (sdvl_ensure(ctx, &ctx->h_counts, &ctx->h_counts_bytes, sizeof(int32_t) * n, true) == SDVL_OK &&
           hipMemcpyAsync(ctx->h_counts, ctx->d_counts, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess)
    ctx->counts_gen = ctx->wait_gen;
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
    if (batch_ok && ctx->counts_gen != ~0ull && ctx->wait_gen > ctx->counts_gen) {
      // the copy queued by sdvl_detect_corners has completed: some wait on this stream returned since
      const int32_t *hc = static_cast<const int32_t *>(ctx->h_counts);
      for (int i = 0; i < n; i++)
        if (frames[i]->v.n_corners < 0) {
          const int c = hc[where[i]];
          if (c < 0) {
            ctx->err = "corner selection overflowed a device capacity (SDVL_MAX_CORNERS / level staging)";
            return SDVL_ERR_CAPACITY;
          }
          frames[i]->v.n_corners = c;
        }
    } else
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
int sdvl_retain_best(sdvl_ctx *ctx, uint32_t *packed, int len, int n_points, int cooperative, int *out_len) {
  if (!ctx || !packed || !out_len || len < 0 || len > (1 << 20)) return SDVL_ERR_INVALID;
  if (len == 0) { *out_len = 0; return SDVL_OK; }
  const size_t bytes = sizeof(uint32_t) * len + 64;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, bytes, true);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  memcpy(static_cast<uint8_t *>(ctx->h_out) + 64, packed, sizeof(uint32_t) * len);
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_out, ctx->h_out, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(retain_best_kernel, dim3(1), dim3(kSelThreads), 0, ctx->stream,
                     reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ctx->d_out) + 64), len, n_points, static_cast<int *>(ctx->d_out), cooperative);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  *out_len = *static_cast<int *>(ctx->h_out);
  memcpy(packed, static_cast<uint8_t *>(ctx->h_out) + 64, sizeof(uint32_t) * len);
  return SDVL_OK;
}

}  // extern "C"
