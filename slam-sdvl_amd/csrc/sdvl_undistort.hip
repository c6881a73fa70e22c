// sdvl_undistort.hip — K0 input stage: Camera::UndistortImage = cv::undistort(in, out, K, D) (camera.cc:39-67,100-105,
// main.cc:133) as a remap from a CACHED per-camera map, optionally fused with the upload into a frame's level 0.
//
// cv::undistort builds, stripe by stripe, a fixed-point map with cv::initUndistortRectifyMap and feeds it to cv::remap
// (bilinear, BORDER_CONSTANT 0).  Nothing in that map depends on the image — Camera::SetDistortions fixes it for the life of the
// camera (camera.cc:39-67) — so it is built ONCE per (camera, distortion, size) and context, on the device, in exactly the library's
// arithmetic: the host prepares
//   xw[j] = normalised x of column j   (the library accumulates _x += ir[0] along the row: sequential rounding),
//   yw[r] = normalised y of row r      (with the principal point of the row's STRIPE, Ar(1,2) = v0 - y0),
// undistort_map_kernel evaluates the radial-tangential model per pixel in FP64 (-ffp-contract=off, the library's expression order),
// iu = cvRound(32 u), and packs integer part + 5-bit fractions into one 32-bit word per pixel (1.2 MB for 640x480: it stays in L2
// across the frames of a launch).  undistort_remap_kernel is then an integer gather: a lane takes four adjacent output pixels of
// four frames — map words decoded once, 15-bit weight table entry, 4 taps, (sum + 2^14) >> 15, one 32-bit store per frame.
// HBM-bound: one u8 neighbourhood read and one u8 write per pixel (2 W H algorithmic bytes per frame).
// Round 5's kernel recomputed the FP64 model for every pixel of every frame: 282 us per 256 frames of 640x480 (0.07 of HBM).
#include <cmath>
#include <cstring>
#include <vector>

#include "sdvl_internal.h"

namespace {

struct UndistJob {
  const uint8_t *src;
  uint8_t *dst;
};

struct UndistParams {
  double fx, fy, u0, v0, k1, k2, p1, p2, k3;
  int w, h, sstride, dstride;
};

constexpr int kMapMax = 2044;  // integer parts are stored with 11 bits each (clamped to [-2, size], + 2)

// map word: a | b << 5 | (clamp(sx, -2, W) + 2) << 10 | (clamp(sy, -2, H) + 2) << 21.  sx <= -2 or sx >= W (likewise sy) puts all four
// taps outside the image, so the clamp loses nothing.  Row pitch = W rounded up to 4 words; the padding says "outside".
__global__ __launch_bounds__(256) void undistort_map_kernel(const double *__restrict__ xw, const double *__restrict__ yw, UndistParams P, int pitch,
                                                            uint32_t *__restrict__ map) {
  const int j = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (j >= pitch) return;
  uint32_t word = 0;  // sx = sy = -2: outside
  if (j < P.w) {
    const double x = xw[j], y = yw[r];
    const double x2 = x * x, y2 = y * y;
    const double r2 = x2 + y2, _2xy = 2 * x * y;
    const double k4 = 0, k5 = 0, k6 = 0;
    const double kr = (1 + ((P.k3 * r2 + P.k2) * r2 + P.k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
    const double u = P.fx * (x * kr + P.p1 * _2xy + P.p2 * (r2 + 2 * x2)) + P.u0;
    const double v = P.fy * (y * kr + P.p1 * (r2 + 2 * y2) + P.p2 * _2xy) + P.v0;
    const int iu = __double2int_rn(u * 32), iv = __double2int_rn(v * 32);  // cvRound = round half to even
    const int sx = static_cast<short>(iu >> 5), sy = static_cast<short>(iv >> 5);
    const int a = iu & 31, b = iv & 31;
    const int sxc = min(max(sx, -2), P.w) + 2, syc = min(max(sy, -2), P.h) + 2;
    word = static_cast<uint32_t>(a) | static_cast<uint32_t>(b) << 5 | static_cast<uint32_t>(sxc) << 10 | static_cast<uint32_t>(syc) << 21;
  }
  map[static_cast<size_t>(r) * pitch + j] = word;
}

// The remap proper works on QUADS of adjacent output pixels.  The map moves by about a pixel per pixel, so the source pixels of a quad
// lie within 8 bytes of at most THREE consecutive rows (rows y0, y0 + 1 for most quads; where the map crosses a source row inside the
// quad, some of its pixels start at y0 + 1).  undistort_quad_kernel notes that once per camera in a 16-byte record:
//   off   where the quad's 8-byte window starts in the source image (first row; both clamped into the image)
//   sel   3-bit byte selectors of the left and of the right tap of each pixel inside the window, the pixels' row bits (sy_k - y0),
//         and whether the second / third row lie one stride further (they do not where the window is clamped at the image's edge)
//   a, b  the 5-bit fractions of the four pixels, and one VALID bit per tap
// cv::remap's BORDER_CONSTANT 0 reads a tap outside the image as 0: here such a tap has weight 0 and selects some byte of the
// (clamped) window.  The remap of a quad of a frame is then three unaligned 8-byte loads (the amdhsa ABI runs with unaligned access
// on), six byte permutes, four bit selects and 16 multiply-adds — no branch, border or not.
// BilinearTab_i[b * 32 + a] is 32 x {(32-b)(32-a), (32-b)a, b(32-a), ba}, so (sum + 2^14) >> 15 equals
// (S0 (32-b)(32-a) + S1 (32-b) a + S2 b (32-a) + S3 b a + 512) >> 10 — also for the table's repaired (0,0) entry {32767, 0, 0, 1}
// (saturate_cast<short>(32768) and the sum repair as OpenCV builds it): with S3 - S0 + 16384 in [16129, 16639] it yields S0, like weight
// 1024 on S0 alone, and 0 when S0 lies outside.  A quad whose live taps do not fit the window (a map that jumps: no camera's) takes the
// tap-by-tap path from the per-pixel words.
struct QuadRec {
  int off;
  uint32_t sel;  // [0:12) selL x 4, [12:24) selR x 4, [24:28) row bits, 28: row b = row a + stride, 29: row c = row b + stride; ~0: not compact
  uint32_t a;    // [0:20) a x 4, [20:32) valid bits of pixels 0..2 (bit 4k + i: tap i of pixel k; taps: 0 top left, 1 top right, 2 bottom left, 3 bottom right)
  uint32_t b;    // [0:20) b x 4, [20:24) valid bits of pixel 3
};

__global__ __launch_bounds__(256) void undistort_quad_kernel(const uint32_t *__restrict__ map, int pitch, UndistParams P, QuadRec *__restrict__ quads) {
  const int qx = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, nq = pitch / 4;
  if (qx >= nq) return;
  const uint4 words = *reinterpret_cast<const uint4 *>(map + static_cast<size_t>(r) * pitch + 4 * qx);
  const uint32_t wd[4] = {words.x, words.y, words.z, words.w};
  const int W = P.w, H = P.h, ss = P.sstride;
  int sx[4], sy[4];
  uint32_t valid = 0, qa = 0, qb = 0;
  int xlo = 1 << 20, xhi = -(1 << 20), y0 = 1 << 20, y1 = -(1 << 20);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    sx[k] = static_cast<int>((wd[k] >> 10) & 2047u) - 2;
    sy[k] = static_cast<int>(wd[k] >> 21) - 2;
    qa |= (wd[k] & 31u) << (5 * k);
    qb |= ((wd[k] >> 5) & 31u) << (5 * k);
    bool live = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int x = sx[k] + (i & 1), y = sy[k] + (i >> 1);
      if (x >= 0 && y >= 0 && x < W && y < H) {
        valid |= 1u << (4 * k + i);
        xlo = min(xlo, x);
        xhi = max(xhi, x);
        live = true;
      }
    }
    if (live) {
      y0 = min(y0, sy[k]);
      y1 = max(y1, sy[k]);
    }
  }
  QuadRec q;
  q.a = qa | (valid & 0xFFFu) << 20;
  q.b = qb | (valid >> 12) << 20;
  if (valid == 0) {  // nothing of the quad lies inside the image: all weights 0, any window
    q.off = 0;
    q.sel = 0;
  } else {
    const int xwin = min(xlo, W - 8);
    if (W < 8 || y1 - y0 > 1 || xhi - xwin > 7) {
      q.off = 0;
      q.sel = ~0u;
    } else {
      const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1), yc = min(max(y0 + 2, 0), H - 1);
      uint32_t sel = (yb != ya ? 1u << 28 : 0u) | (yc != yb ? 1u << 29 : 0u);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool live = ((valid >> (4 * k)) & 15u) != 0;
        sel |= static_cast<uint32_t>(min(max(sx[k] - xwin, 0), 7)) << (3 * k);
        sel |= static_cast<uint32_t>(min(max(sx[k] + 1 - xwin, 0), 7)) << (12 + 3 * k);
        if (live && sy[k] != y0) sel |= 1u << (24 + k);
      }
      q.off = ya * ss + xwin;
      q.sel = sel;
    }
  }
  quads[static_cast<size_t>(r) * nq + qx] = q;
}

// (S0 p0 + S1 p1 + S2 p2 + S3 p3 + 512) >> 10 for pixel k of a quad: Lt / Rt = left / right taps of its upper row as packed bytes, Lb / Rb of the lower
// (every product is below 2^18: the 24-bit multiplier with byte selects, not the quarter-rate 32-bit one)
template <int k>
__device__ __forceinline__ uint32_t quad_px(uint32_t Lt, uint32_t Rt, uint32_t Lb, uint32_t Rb, const uint32_t *p) {
  return (__umul24((Lt >> (8 * k)) & 255u, p[0]) + __umul24((Rt >> (8 * k)) & 255u, p[1]) + __umul24((Lb >> (8 * k)) & 255u, p[2]) +
          __umul24((Rb >> (8 * k)) & 255u, p[3]) + 512u) >> 10;
}

__device__ __forceinline__ uint32_t bit_select(uint32_t mask, uint32_t one, uint32_t zero) { return (mask & one) | (~mask & zero); }  // v_bfi_b32
__device__ __forceinline__ uint32_t spread3(uint32_t f) {  // four 3-bit fields -> the low bits of four bytes
  return (f & 7u) | ((f >> 3) & 7u) << 8 | ((f >> 6) & 7u) << 16 | ((f >> 9) & 7u) << 24;
}

// one workgroup = a tile of 128 x 8 output pixels (a lane: one quad) of kRemapFrames consecutive frames; the loads of all frames of the
// lane are in flight before the first result is needed.  kFull: all kRemapFrames frames exist (every workgroup but the last frame group's);
// kWide: every destination row start is 4-byte aligned and a multiple of four pixels wide (the frames' own level 0 always is)
constexpr int kRemapFrames = 4;
template <bool kFull, bool kWide>
__device__ __forceinline__ void remap_compact(const UndistJob *__restrict__ jobs, const QuadRec q, int nf, int ss, uint32_t drow, int npx) {
  // (the frame pointers are uniform and name global memory: address space 1 lets the loads take a scalar base + 32-bit lane offset
  //  instead of a 64-bit flat address per lane)
  typedef const __attribute__((address_space(1))) uint8_t *gsrc_t;
  typedef __attribute__((address_space(1))) uint8_t *gdst_t;
  typedef unsigned long long __attribute__((aligned(1))) u64_any;  // (an 8-byte window at any byte address)
  typedef const __attribute__((address_space(1))) u64_any *gwin_t;
  unsigned long long ra[kRemapFrames], rb[kRemapFrames], rc[kRemapFrames];
  gdst_t dst[kRemapFrames];
  const uint32_t off_a = static_cast<uint32_t>(q.off), off_b = off_a + ((q.sel >> 28) & 1u ? ss : 0), off_c = off_b + ((q.sel >> 29) & 1u ? ss : 0);
#pragma unroll
  for (int k = 0; k < kRemapFrames; k++) {
    const UndistJob job = jobs[kFull ? k : min(k, nf - 1)];
    const gsrc_t src = (gsrc_t)job.src;
    ra[k] = *(gwin_t)(src + off_a);
    rb[k] = *(gwin_t)(src + off_b);
    rc[k] = *(gwin_t)(src + off_c);
    dst[k] = (gdst_t)job.dst;
  }
  // the sixteen weights of the quad: p0 = (32-b)(32-a), p1 = (32-b) a, p2 = b (32-a), p3 = b a; 0 for a tap outside the image
  uint32_t p[4][4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t a = (q.a >> (5 * k)) & 31u, b = (q.b >> (5 * k)) & 31u;
    p[k][0] = (32u - b) * (32u - a);
    p[k][1] = (32u - b) * a;
    p[k][2] = b * (32u - a);
    p[k][3] = b * a;
  }
  const uint32_t valid = (q.a >> 20) | ((q.b >> 20) & 15u) << 12;
  if (valid != 0xFFFFu) {
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (!((valid >> (4 * k + i)) & 1u)) p[k][i] = 0;
  }
  const uint32_t selL = spread3(q.sel), selR = spread3(q.sel >> 12);
  const uint32_t rowbits = (q.sel >> 24) & 15u;
  const uint32_t rowbyte = (rowbits | rowbits << 7 | rowbits << 14 | rowbits << 21) & 0x01010101u;
  const uint32_t M = (rowbyte << 8) - rowbyte;  // 0xFF in the bytes of the pixels that start one row lower
#pragma unroll
  for (int k = 0; k < kRemapFrames; k++) {
    if (!kFull && k >= nf) break;
    const uint32_t alo = static_cast<uint32_t>(ra[k]), ahi = static_cast<uint32_t>(ra[k] >> 32);
    const uint32_t blo = static_cast<uint32_t>(rb[k]), bhi = static_cast<uint32_t>(rb[k] >> 32);
    const uint32_t clo = static_cast<uint32_t>(rc[k]), chi = static_cast<uint32_t>(rc[k] >> 32);
    const uint32_t La = __builtin_amdgcn_perm(ahi, alo, selL), Ra = __builtin_amdgcn_perm(ahi, alo, selR);
    const uint32_t Lb = __builtin_amdgcn_perm(bhi, blo, selL), Rb = __builtin_amdgcn_perm(bhi, blo, selR);
    const uint32_t Lc = __builtin_amdgcn_perm(chi, clo, selL), Rc = __builtin_amdgcn_perm(chi, clo, selR);
    const uint32_t Lt = bit_select(M, Lb, La), Rt = bit_select(M, Rb, Ra), Lu = bit_select(M, Lc, Lb), Ru = bit_select(M, Rc, Rb);
    const uint32_t q0 = quad_px<0>(Lt, Rt, Lu, Ru, p[0]), q1 = quad_px<1>(Lt, Rt, Lu, Ru, p[1]);
    const uint32_t q2 = quad_px<2>(Lt, Rt, Lu, Ru, p[2]), q3 = quad_px<3>(Lt, Rt, Lu, Ru, p[3]);
    gdst_t d = dst[k] + drow;
    if (kWide) {
      *(__attribute__((address_space(1))) uint32_t *)d = q0 | q1 << 8 | q2 << 16 | q3 << 24;
    } else {
      d[0] = static_cast<uint8_t>(q0);
      if (npx > 1) d[1] = static_cast<uint8_t>(q1);
      if (npx > 2) d[2] = static_cast<uint8_t>(q2);
      if (npx > 3) d[3] = static_cast<uint8_t>(q3);
    }
  }
}

__global__ __launch_bounds__(256) void undistort_remap_kernel(const UndistJob *__restrict__ jobs, const uint32_t *__restrict__ map,
                                                              const QuadRec *__restrict__ quads, int pitch, int n, int wide, UndistParams P) {
  // Workgroups are dealt to the eight XCDs round robin by their linear id: the frame group is the FASTEST index, so an XCD (its L2)
  // sees all tiles of one eighth of the frames and neighbouring tiles of a frame — which share source lines — meet in one L2.
  // (grid = frame groups x tiles across x tiles down: x is the fastest index of the dispatch order)
  const int zi = blockIdx.x;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j0 = (blockIdx.y * 32 + tx) * 4, r = blockIdx.z * 8 + ty;
  if (j0 >= P.w || r >= P.h) return;
  const int W = P.w, H = P.h, ss = P.sstride;
  const int f0 = zi * kRemapFrames, nf = min(n - f0, kRemapFrames);
  const uint32_t drow = static_cast<uint32_t>(r) * P.dstride + j0;  // (images of at most 2044 x 2044: 32 bits)
  const QuadRec q = quads[static_cast<size_t>(r) * (pitch / 4) + (j0 >> 2)];
  const int npx = min(4, W - j0);
  if (q.sel != ~0u) {
    if (nf == kRemapFrames) {
      if (wide) remap_compact<true, true>(jobs + f0, q, nf, ss, drow, npx);
      else remap_compact<true, false>(jobs + f0, q, nf, ss, drow, npx);
    } else {
      remap_compact<false, false>(jobs + f0, q, nf, ss, drow, npx);
    }
    return;
  }
  // A quad whose live taps do not fit one window: tap by tap from the per-pixel words, the same way — a tap outside the image has
  // weight 0 and an address clamped into the image, decided once per lane.
  const uint4 words = *reinterpret_cast<const uint4 *>(map + static_cast<size_t>(r) * pitch + j0);
  const uint32_t wd[4] = {words.x, words.y, words.z, words.w};
  uint32_t off[4][4], wt[4][4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t a = wd[k] & 31u, b = (wd[k] >> 5) & 31u;
    const int sx = static_cast<int>((wd[k] >> 10) & 2047u) - 2, sy = static_cast<int>(wd[k] >> 21) - 2;
    const uint32_t pw[4] = {(32u - b) * (32u - a), (32u - b) * a, b * (32u - a), b * a};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int x = sx + (i & 1), y = sy + (i >> 1);
      const bool inside = x >= 0 && y >= 0 && x < W && y < H;
      off[k][i] = static_cast<uint32_t>(min(max(y, 0), H - 1)) * ss + min(max(x, 0), W - 1);
      wt[k][i] = inside ? pw[i] : 0u;
    }
  }
  typedef const __attribute__((address_space(1))) uint8_t *gsrc_t;
  typedef __attribute__((address_space(1))) uint8_t *gdst_t;
  for (int f = 0; f < nf; f++) {
    const UndistJob job = jobs[f0 + f];
    const gsrc_t src = (gsrc_t)job.src;
    uint32_t qv[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
      qv[k] = (__umul24(src[off[k][0]], wt[k][0]) + __umul24(src[off[k][1]], wt[k][1]) + __umul24(src[off[k][2]], wt[k][2]) + __umul24(src[off[k][3]], wt[k][3]) + 512u) >> 10;
    gdst_t d = (gdst_t)job.dst + drow;
    d[0] = static_cast<uint8_t>(qv[0]);
    if (npx > 1) d[1] = static_cast<uint8_t>(qv[1]);
    if (npx > 2) d[2] = static_cast<uint8_t>(qv[2]);
    if (npx > 3) d[3] = static_cast<uint8_t>(qv[3]);
  }
}

// cv::invert of a 3x3 double matrix: closed form (the library's path for n <= 3)
bool invert3x3(const double *S, double *t) {
  double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d == 0.) return false;
  d = 1. / d;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
  return true;
}

// the per-column / per-row tables of initUndistortRectifyMap as cv::undistort calls it (stripes of 4096 / cols rows)
bool build_tables(int w, int h, const sdvl_camera *cam, double *xw, double *yw) {
  const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(w, 1)), h);
  for (int y0 = 0; y0 < h; y0 += stripe0) {
    const int stripe = std::min(stripe0, h - y0);
    const double Ar[9] = {cam->fx, 0, cam->u0, 0, cam->fy, cam->v0 - y0, 0, 0, 1};
    double ir[9];
    if (!invert3x3(Ar, ir)) return false;
    for (int i = 0; i < stripe; i++) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      // along a row _y and _w only ever receive ir[3] = ir[6] = 0, so the row's y is the value at j = 0
      yw[y0 + i] = _y * (1. / _w);
      if (y0 == 0 && i == 0)  // _x does not depend on the stripe or the row (ir[1] = 0; ir[0], ir[2] do not involve Ar(1,2))
        for (int j = 0; j < w; j++, _x += ir[0], _w += ir[6]) xw[j] = _x * (1. / _w);
    }
  }
  return true;
}

int run_undistort(sdvl_ctx *ctx, int n, const void *const *src, int src_stride, int src_on_device, int w, int h, const sdvl_camera *cam,
                  const sdvl_distortion *dist, uint8_t *const *dst, int dst_stride) {
  const size_t img_bytes = static_cast<size_t>(w) * h;
  const bool remap = dist->d[0] != 0.0;  // Camera::SetDistortions tests d0 only (camera.cc:46): otherwise out = in.clone()
  const uint8_t *dev_src[1];
  std::vector<const uint8_t *> srcs(n);
  if (src_on_device) {
    for (int i = 0; i < n; i++) srcs[i] = static_cast<const uint8_t *>(src[i]);
  } else if (remap) {  // raw images go to a device scratch area first
    int rc = sdvl_ensure(ctx, &ctx->d_work, &ctx->d_work_bytes, img_bytes * n, false);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
      uint8_t *d = static_cast<uint8_t *>(ctx->d_work) + img_bytes * i;
      SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(d, w, src[i], src_stride, w, h, hipMemcpyHostToDevice, ctx->stream));
      srcs[i] = d;
    }
    src_stride = w;
  }
  (void)dev_src;
  if (!remap) {
    for (int i = 0; i < n; i++)
      SDVL_HIP_CHECK(ctx, hipMemcpy2DAsync(dst[i], dst_stride, src[i], src_stride, w, h,
                                           src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    return SDVL_OK;
  }
  // ---- the camera's map: built once per (size, intrinsics, distortion) and context
  const int pitch = (w + 3) / 4 * 4;
  UndistParams P{cam->fx, cam->fy, cam->u0, cam->v0, dist->d[0], dist->d[1], dist->d[2], dist->d[3], dist->d[4], w, h, src_stride, dst_stride};
  double key[12] = {cam->fx, cam->fy, cam->u0, cam->v0, dist->d[0], dist->d[1], dist->d[2], dist->d[3], dist->d[4], static_cast<double>(w), static_cast<double>(h),
                    static_cast<double>(src_stride)};  // (the quad records hold source offsets)
  if (!ctx->d_undist_map || memcmp(key, ctx->undist_key, sizeof(key)) != 0) {
    const size_t px_bytes = (sizeof(uint32_t) * static_cast<size_t>(pitch) * h + 255) / 256 * 256;
    const size_t map_bytes = px_bytes + sizeof(QuadRec) * static_cast<size_t>(pitch / 4) * h;
    if (ctx->undist_map_bytes < map_bytes) {
      if (ctx->d_undist_map) {
        SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));  // remaps that read the old map may still be queued
        SDVL_HIP_CHECK(ctx, hipFree(ctx->d_undist_map));
        ctx->d_undist_map = nullptr;
        ctx->undist_map_bytes = 0;
      }
      SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
      SDVL_HIP_CHECK(ctx, hipMalloc(&ctx->d_undist_map, map_bytes));
      ctx->undist_map_bytes = map_bytes;
    }
    const size_t xb = (sizeof(double) * w + 255) / 256 * 256, yb = sizeof(double) * h;
    void *hs = nullptr, *dsx = nullptr;
    int rc = sdvl_stage_alloc(ctx, xb + yb, &hs, &dsx);
    if (rc) return rc;
    uint8_t *h8 = static_cast<uint8_t *>(hs), *d8 = static_cast<uint8_t *>(dsx);
    if (!build_tables(w, h, cam, reinterpret_cast<double *>(h8), reinterpret_cast<double *>(h8 + xb))) {
      ctx->err = "camera matrix is singular";
      return SDVL_ERR_INVALID;
    }
    SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, xb + yb));
    SDVL_LAUNCH(ctx, "undistort_map", undistort_map_kernel, dim3((pitch + 255) / 256, h), dim3(256), reinterpret_cast<const double *>(d8),
                reinterpret_cast<const double *>(d8 + xb), P, pitch, static_cast<uint32_t *>(ctx->d_undist_map));
    SDVL_LAUNCH(ctx, "undistort_map", undistort_quad_kernel, dim3((pitch / 4 + 255) / 256, h), dim3(256), static_cast<const uint32_t *>(ctx->d_undist_map), pitch, P,
                reinterpret_cast<QuadRec *>(static_cast<uint8_t *>(ctx->d_undist_map) + px_bytes));
    SDVL_HIP_CHECK(ctx, hipGetLastError());
    ctx->undist_quads = static_cast<uint8_t *>(ctx->d_undist_map) + px_bytes;
    memcpy(ctx->undist_key, key, sizeof(key));
    ctx->undist_maps_built++;
  }
  // ---- the remap
  const size_t jb = (sizeof(UndistJob) * n + 255) / 256 * 256;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_stage_alloc(ctx, jb, &hs, &dsx);
  if (rc) return rc;
  UndistJob *hj = static_cast<UndistJob *>(hs);
  int wide = ((dst_stride | w) & 3) == 0 ? 1 : 0;
  for (int i = 0; i < n; i++) {
    hj[i] = UndistJob{srcs[i], dst[i]};
    if (reinterpret_cast<uintptr_t>(dst[i]) & 3u) wide = 0;
  }
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, jb));
  SDVL_LAUNCH(ctx, "undistort", undistort_remap_kernel, dim3((n + kRemapFrames - 1) / kRemapFrames, (w + 127) / 128, (h + 7) / 8), dim3(256),
              static_cast<const UndistJob *>(dsx), static_cast<const uint32_t *>(ctx->d_undist_map), static_cast<const QuadRec *>(ctx->undist_quads), pitch, n, wide, P);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

}  // namespace

extern "C" {

int sdvl_undistort(sdvl_ctx *ctx, int n, const void *const *src, int src_stride, int src_on_device, int width, int height,
                   const sdvl_camera *cam, const sdvl_distortion *dist, void *const *dst_dev, int dst_stride) {
  if (!ctx || n < 0 || (n > 0 && (!src || !dst_dev)) || !cam || !dist) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, width >= 2 && height >= 2 && width <= kMapMax && height <= kMapMax, "image size out of range (2 .. 2044 a side)");
  SDVL_REQUIRE(ctx, src_stride >= width && dst_stride >= width, "stride smaller than width");
  for (int i = 0; i < n; i++) SDVL_REQUIRE(ctx, src[i] && dst_dev[i] && src[i] != dst_dev[i], "null image or in-place undistort");
  return run_undistort(ctx, n, src, src_stride, src_on_device, width, height, cam, dist, reinterpret_cast<uint8_t *const *>(dst_dev), dst_stride);
}

int sdvl_frames_upload_undistorted(sdvl_ctx *ctx, int n, sdvl_frame *const *frames, const void *const *src, int src_stride,
                                   int src_on_device, const sdvl_camera *cam, const sdvl_distortion *dist) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !src)) || !cam || !dist) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  std::vector<uint8_t *> dst(n);
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] && src[i], "null frame or image");
    SDVL_REQUIRE(ctx, frames[i]->width == frames[0]->width && frames[i]->height == frames[0]->height, "frames of one call share a size");
    SDVL_REQUIRE(ctx, src_stride >= frames[i]->width, "stride smaller than width");
    SDVL_REQUIRE(ctx, frames[i]->width <= kMapMax && frames[i]->height <= kMapMax, "image size out of range (2 .. 2044 a side)");
    frames[i]->v.level[0] = frames[i]->own_level0;
    dst[i] = frames[i]->own_level0;
    frames[i]->hdr_stale = 1;
    frames[i]->v.n_corners = 0;
    frames[i]->desc_valid = 0;
    frames[i]->bins_valid = 0;
  }
  return run_undistort(ctx, n, src, src_stride, src_on_device, frames[0]->width, frames[0]->height, cam, dist, dst.data(), frames[0]->width);
}

}  // extern "C"
