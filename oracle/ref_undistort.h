// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of Camera::UndistortImage (camera.cc:39-67,100-105):
// cv::undistort(in, out, K, D) with K = [fx 0 u0; 0 fy v0; 0 0 1], D = (d0..d4) = (k1, k2, p1, p2, k3).
// PARITY UNPINNED: OpenCV is not in /root/reference and not installed; restated from the published 2.4/3.x sources
// (imgproc/undistort.cpp cv::undistort + cv::initUndistortRectifyMap, imgproc/imgwarp.cpp initInterTab2D + remapBilinear,
// core cv::invert 3x3 closed form).  What the restatement keeps, because the output bits depend on it:
//   * the image is processed in stripes of max(1, 4096 / cols) rows, each with the principal point shifted
//     (Ar(1,2) = v0 - y), so a row's normalised y is  i * ir[4] + ir[5]  with the STRIPE's ir[5] and local row i;
//   * _x is accumulated along a row by repeated += ir[0] (sequential rounding), _w likewise;
//   * maps are fixed point: iu = cvRound(u * 32), integer part (short) and 5+5 fraction bits;
//   * bilinear weights come from the 32x32 table of shorts scaled by 32768 (incl. saturate_cast<short>(32768) = 32767 for
//     the (0,0) entry and the table's sum-repair step as written), result = (sum + (1 << 14)) >> 15, BORDER_CONSTANT 0.
#ifndef SDVL_ORACLE_REF_UNDISTORT_H_
#define SDVL_ORACLE_REF_UNDISTORT_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace sdvlref {

constexpr int kInterBits = 5, kInterTabSize = 1 << kInterBits, kRemapCoefBits = 15, kRemapCoefScale = 1 << kRemapCoefBits;

inline int CvRoundD(double v) { return static_cast<int>(std::lrint(v)); }  // cvRound: round half to even

inline short SaturateShort(int v) { return static_cast<short>(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// initInterTab2D(INTER_LINEAR, fixpt = true): BilinearTab_i[32*32][2][2]
inline const short *BilinearTabI() {
  static std::vector<short> storage;
  if (storage.empty()) {
    const int ksize = 2;
    storage.assign(kInterTabSize * kInterTabSize * ksize * ksize + 16, 0);  // static zero-initialised in OpenCV (+ slack)
    float tab1[kInterTabSize * 2];
    const float scale = 1.f / kInterTabSize;
    for (int i = 0; i < kInterTabSize; i++) {  // initInterTab1D -> interpolateLinear
      const float x = i * scale;
      tab1[i * 2] = 1.f - x;
      tab1[i * 2 + 1] = x;
    }
    short *itab = storage.data();
    for (int i = 0; i < kInterTabSize; i++)
      for (int j = 0; j < kInterTabSize; j++, itab += ksize * ksize) {
        int isum = 0;
        for (int k1 = 0; k1 < ksize; k1++) {
          const float vy = tab1[i * ksize + k1];
          for (int k2 = 0; k2 < ksize; k2++) {
            const float v = vy * tab1[j * ksize + k2];
            itab[k1 * ksize + k2] = SaturateShort(CvRoundD(static_cast<double>(v * kRemapCoefScale)));  // saturate_cast<short>(float)
            isum += itab[k1 * ksize + k2];
          }
        }
        if (isum != kRemapCoefScale) {
          const int diff = isum - kRemapCoefScale;
          const int ksize2 = ksize / 2;
          int Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
          for (int k1 = ksize2; k1 < ksize2 + 2; k1++)
            for (int k2 = ksize2; k2 < ksize2 + 2; k2++) {
              if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2]) mk1 = k1, mk2 = k2;
              else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2]) Mk1 = k1, Mk2 = k2;
            }
          if (diff < 0) itab[Mk1 * ksize + Mk2] = static_cast<short>(itab[Mk1 * ksize + Mk2] - diff);
          else itab[mk1 * ksize + mk2] = static_cast<short>(itab[mk1 * ksize + mk2] - diff);
        }
      }
  }
  return storage.data();
}

// cv::invert of a 3x3 CV_64F matrix (DECOMP_LU takes the closed form for n <= 3)
inline bool Invert3x3(const double *S, double *t) {
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

// cv::undistort for 8UC1.  cam = fx fy u0 v0, dist = k1 k2 p1 p2 k3.  dst is w x h, row stride w.
inline void Undistort(const uint8_t *src, int w, int h, int sstep, const double *cam, const double *dist, uint8_t *dst) {
  const double fx = cam[0], fy = cam[1], u0 = cam[2], v0 = cam[3];
  const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4], k4 = 0, k5 = 0, k6 = 0;
  const int stripe_size0 = std::min(std::max(1, (1 << 12) / std::max(w, 1)), h);
  const short *wtab = BilinearTabI();
  std::vector<short> m1(static_cast<size_t>(stripe_size0) * w * 2);
  std::vector<uint16_t> m2(static_cast<size_t>(stripe_size0) * w);
  for (int y0 = 0; y0 < h; y0 += stripe_size0) {
    const int stripe = std::min(stripe_size0, h - y0);
    // initUndistortRectifyMap(A, D, I, Ar, Size(w, stripe), CV_16SC2): iR = (Ar * I).inv(DECOMP_LU)
    const double Ar[9] = {fx, 0, u0, 0, fy, v0 - y0, 0, 0, 1};
    double ir[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Invert3x3(Ar, ir);
    for (int i = 0; i < stripe; i++) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double iw = 1. / _w, x = _x * iw, y = _y * iw;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
        const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
        const int iu = CvRoundD(u * kInterTabSize), iv = CvRoundD(v * kInterTabSize);
        m1[(static_cast<size_t>(i) * w + j) * 2] = static_cast<short>(iu >> kInterBits);
        m1[(static_cast<size_t>(i) * w + j) * 2 + 1] = static_cast<short>(iv >> kInterBits);
        m2[static_cast<size_t>(i) * w + j] = static_cast<uint16_t>((iv & (kInterTabSize - 1)) * kInterTabSize + (iu & (kInterTabSize - 1)));
      }
    }
    // remap(src, dst_part, map1, map2, INTER_LINEAR, BORDER_CONSTANT = 0): remapBilinear, 8UC1, fixed-point weights
    for (int i = 0; i < stripe; i++) {
      uint8_t *D = dst + static_cast<size_t>(y0 + i) * w;
      for (int j = 0; j < w; j++) {
        const int sx = m1[(static_cast<size_t>(i) * w + j) * 2], sy = m1[(static_cast<size_t>(i) * w + j) * 2 + 1];
        const short *wt = wtab + m2[static_cast<size_t>(i) * w + j] * 4;
        int sum;
        if (static_cast<unsigned>(sx) < static_cast<unsigned>(std::max(w - 1, 0)) &&
            static_cast<unsigned>(sy) < static_cast<unsigned>(std::max(h - 1, 0))) {
          const uint8_t *S = src + static_cast<size_t>(sy) * sstep + sx;
          sum = S[0] * wt[0] + S[1] * wt[1] + S[sstep] * wt[2] + S[sstep + 1] * wt[3];
        } else if (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0) {
          D[j] = 0;
          continue;
        } else {
          const int sx0 = sx, sx1 = sx + 1, sy0 = sy, sy1 = sy + 1;
          const int v0p = (sx0 >= 0 && sy0 >= 0 && sx0 < w && sy0 < h) ? src[static_cast<size_t>(sy0) * sstep + sx0] : 0;
          const int v1p = (sx1 >= 0 && sy0 >= 0 && sx1 < w && sy0 < h) ? src[static_cast<size_t>(sy0) * sstep + sx1] : 0;
          const int v2p = (sx0 >= 0 && sy1 >= 0 && sx0 < w && sy1 < h) ? src[static_cast<size_t>(sy1) * sstep + sx0] : 0;
          const int v3p = (sx1 >= 0 && sy1 >= 0 && sx1 < w && sy1 < h) ? src[static_cast<size_t>(sy1) * sstep + sx1] : 0;
          sum = v0p * wt[0] + v1p * wt[1] + v2p * wt[2] + v3p * wt[3];
        }
        const int r = (sum + (1 << (kRemapCoefBits - 1))) >> kRemapCoefBits;  // FixedPtCast<int, uchar, 15>
        D[j] = static_cast<uint8_t>(r < 0 ? 0 : (r > 255 ? 255 : r));
      }
    }
  }
}

}  // namespace sdvlref

#endif  // SDVL_ORACLE_REF_UNDISTORT_H_
