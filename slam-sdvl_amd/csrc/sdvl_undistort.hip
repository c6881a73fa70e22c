// sdvl_undistort.hip — K0 input stage: Camera::UndistortImage = cv::undistort(in, out, K, D) (camera.cc:39-67,100-105,
// main.cc:133) as one remap kernel, optionally fused with the upload into a frame's level 0.
//
// cv::undistort builds, stripe by stripe, a fixed-point map with cv::initUndistortRectifyMap and feeds it to cv::remap
// (bilinear, BORDER_CONSTANT 0).  Nothing in that map depends on the image, and along a row only the column index
// varies, so the host prepares two small tables in exactly the library's arithmetic —
//   xw[j] = normalised x of column j   (the library accumulates _x += ir[0] along the row: sequential rounding),
//   yw[r] = normalised y of row r      (with the principal point of the row's STRIPE, Ar(1,2) = v0 - y0),
// and the kernel does the rest per output pixel: radial-tangential model in FP64 (-ffp-contract=off, the library's
// expression order), iu = cvRound(32 u), integer part + 5-bit fractions, the 15-bit weight table entry, 4-tap gather,
// (sum + 2^14) >> 15.  HBM-bound: one u8 read neighbourhood and one u8 write per pixel (2 W H algorithmic bytes).
#include <cmath>
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

__global__ __launch_bounds__(256) void undistort_kernel(const UndistJob *__restrict__ jobs, const double *__restrict__ xw,
                                                        const double *__restrict__ yw, UndistParams P) {
  const int j = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (j >= P.w) return;
  const UndistJob job = jobs[blockIdx.z];
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
  // BilinearTab_i[b * 32 + a]: (32-b)(32-a), (32-b)a, b(32-a), ba, each times 32; the (0,0) entry is {32767, 0, 0, 1}
  // (saturate_cast<short>(32768) and the table's sum repair as OpenCV builds it)
  int w0 = (32 - b) * (32 - a) * 32, w1 = (32 - b) * a * 32, w2 = b * (32 - a) * 32, w3 = b * a * 32;
  if ((a | b) == 0) { w0 = 32767; w3 = 1; }
  const uint8_t *src = job.src;
  const int W = P.w, H = P.h, ss = P.sstride;
  int sum;
  uint8_t out;
  if (static_cast<unsigned>(sx) < static_cast<unsigned>(max(W - 1, 0)) && static_cast<unsigned>(sy) < static_cast<unsigned>(max(H - 1, 0))) {
    const uint8_t *S = src + static_cast<size_t>(sy) * ss + sx;
    sum = S[0] * w0 + S[1] * w1 + S[ss] * w2 + S[ss + 1] * w3;
    const int q = (sum + (1 << 14)) >> 15;
    out = static_cast<uint8_t>(min(max(q, 0), 255));
  } else if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) {
    out = 0;
  } else {
    const int sx1 = sx + 1, sy1 = sy + 1;
    const int v0p = (sx >= 0 && sy >= 0 && sx < W && sy < H) ? src[static_cast<size_t>(sy) * ss + sx] : 0;
    const int v1p = (sx1 >= 0 && sy >= 0 && sx1 < W && sy < H) ? src[static_cast<size_t>(sy) * ss + sx1] : 0;
    const int v2p = (sx >= 0 && sy1 >= 0 && sx < W && sy1 < H) ? src[static_cast<size_t>(sy1) * ss + sx] : 0;
    const int v3p = (sx1 >= 0 && sy1 >= 0 && sx1 < W && sy1 < H) ? src[static_cast<size_t>(sy1) * ss + sx1] : 0;
    sum = v0p * w0 + v1p * w1 + v2p * w2 + v3p * w3;
    const int q = (sum + (1 << 14)) >> 15;
    out = static_cast<uint8_t>(min(max(q, 0), 255));
  }
  job.dst[static_cast<size_t>(r) * P.dstride + j] = out;
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
  const size_t jb = (sizeof(UndistJob) * n + 255) / 256 * 256, xb = (sizeof(double) * w + 255) / 256 * 256, yb = sizeof(double) * h;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_stage_alloc(ctx, jb + xb + yb, &hs, &dsx);
  if (rc) return rc;
  uint8_t *h8 = static_cast<uint8_t *>(hs), *d8 = static_cast<uint8_t *>(dsx);
  UndistJob *hj = reinterpret_cast<UndistJob *>(h8);
  for (int i = 0; i < n; i++) hj[i] = UndistJob{srcs[i], dst[i]};
  if (!build_tables(w, h, cam, reinterpret_cast<double *>(h8 + jb), reinterpret_cast<double *>(h8 + jb + xb))) {
    ctx->err = "camera matrix is singular";
    return SDVL_ERR_INVALID;
  }
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, jb + xb + yb));
  UndistParams P{cam->fx, cam->fy, cam->u0, cam->v0, dist->d[0], dist->d[1], dist->d[2], dist->d[3], dist->d[4], w, h, src_stride, dst_stride};
  SDVL_LAUNCH(ctx, "undistort", undistort_kernel, dim3((w + 255) / 256, h, n), dim3(256), reinterpret_cast<const UndistJob *>(d8),
              reinterpret_cast<const double *>(d8 + jb), reinterpret_cast<const double *>(d8 + jb + xb), P);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

}  // namespace

extern "C" {

int sdvl_undistort(sdvl_ctx *ctx, int n, const void *const *src, int src_stride, int src_on_device, int width, int height,
                   const sdvl_camera *cam, const sdvl_distortion *dist, void *const *dst_dev, int dst_stride) {
  if (!ctx || n < 0 || (n > 0 && (!src || !dst_dev)) || !cam || !dist) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, width >= 2 && height >= 2 && width <= 4095 && height <= 4095, "image size out of range");
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
