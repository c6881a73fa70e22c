// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_math.h header).  PARITY UNPINNED.
//
// ref_orb.h — steered-BRIEF descriptor, intensity-centroid orientation, Hamming distance.
//   /root/reference/extra/orb_detector.cc:325-445   (cv::fastAtan2 / cvRound: SURVEY Appendix A.4/A.5)
// Frozen interpretations: cos/sin of the float angle evaluated in double and rounded to float;
// cvRound = round-half-to-even of the double-promoted float expression; fastAtan2 = OpenCV >= 2.4.9 polynomial.
#ifndef SDVL_ORACLE_REF_ORB_H_
#define SDVL_ORACLE_REF_ORB_H_

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

#include "ref_detect.h"

namespace sdvlref {

static const int kOrbPattern[256 * 4] = {
#include "orb_pattern_31.inc"
};

// cvRound(double): SSE2 cvtsd2si, round half to even (A.4)
inline int CvRound(double v) { return static_cast<int>(std::nearbyint(v)); }

// cv::fastAtan2(y, x) in degrees, OpenCV 2.4.9+/3.x scalar polynomial (A.5)
inline float FastAtan2(float y, float x) {
  const float atan2_p1 = 0.9997878412794807f * static_cast<float>(180 / M_PI);
  const float atan2_p3 = -0.3258083974640975f * static_cast<float>(180 / M_PI);
  const float atan2_p5 = 0.1555786518463281f * static_cast<float>(180 / M_PI);
  const float atan2_p7 = -0.04432655554792128f * static_cast<float>(180 / M_PI);
  const float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + static_cast<float>(DBL_EPSILON));
    c2 = c * c;
    a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  } else {
    c = ax / (ay + static_cast<float>(DBL_EPSILON));
    c2 = c * c;
    a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

struct OrbDetector {
  int half_patch;
  std::vector<int> umax;

  // ORBDetector::InitParameters, orb_detector.cc:325-348
  explicit OrbDetector(int orb_size = 31) {
    half_patch = orb_size / 2;
    const int vmax = static_cast<int>(std::floor(half_patch * std::sqrt(2.f) / 2 + 1));
    const int vmin = static_cast<int>(std::ceil(half_patch * std::sqrt(2.f) / 2));
    const double hp2 = half_patch * half_patch;
    umax.assign(half_patch + 1, 0);
    for (int v = 0; v <= vmax; ++v) umax[v] = CvRound(std::sqrt(hp2 - v * v));
    for (int v = half_patch, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // orb_detector.cc:439-445
  bool IsInsideLimits(const Image &src, int px, int py) const {
    const int m = half_patch + 4;
    return px >= m && px < src.cols - m && py >= m && py < src.rows - m;
  }

  // orb_detector.cc:412-437
  float GetOrientation(const Image &src, int px, int py) const {
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = src.ptr(py) + px;
    for (int u = -half_patch; u <= half_patch; ++u) m_10 += u * center[u];
    const int step = src.step;
    for (int v = 1; v <= half_patch; ++v) {
      int v_sum = 0;
      const int d = umax[v];
      for (int u = -d; u <= d; ++u) {
        const int val_plus = center[u + v * step], val_minus = center[u - v * step];
        v_sum += (val_plus - val_minus);
        m_10 += u * (val_plus + val_minus);
      }
      m_01 += v * v_sum;
    }
    return FastAtan2(static_cast<float>(m_01), static_cast<float>(m_10));
  }

  // orb_detector.cc:350-395
  void GetDescriptor(const Image &src, int px, int py, uint8_t desc[32]) const {
    const float factorPI = static_cast<float>(M_PI / 180.f);
    const int step = src.step;
    const uint8_t *center = src.ptr(py) + px;
    // GetOrientation returns double(float) in the reference; the product with a float is evaluated in
    // double and rounded to float on assignment.
    const float angle = static_cast<float>(static_cast<double>(GetOrientation(src, px, py)) * factorPI);
    const float a = static_cast<float>(std::cos(static_cast<double>(angle)));
    const float b = static_cast<float>(std::sin(static_cast<double>(angle)));
    const int *pat = kOrbPattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
      int val = 0;
      for (int k = 0; k < 8; k++) {
        const int x0 = pat[4 * k], y0 = pat[4 * k + 1], x1 = pat[4 * k + 2], y1 = pat[4 * k + 3];
        const int t0 = center[CvRound(x0 * b + y0 * a) * step + CvRound(x0 * a - y0 * b)];
        const int t1 = center[CvRound(x1 * b + y1 * a) * step + CvRound(x1 * a - y1 * b)];
        val |= (t0 < t1) << k;
      }
      desc[i] = static_cast<uint8_t>(val);
    }
  }

  // orb_detector.cc:398-410
  static int Distance(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
      uint32_t va, vb;
      std::memcpy(&va, a + 4 * i, 4);
      std::memcpy(&vb, b + 4 * i, 4);
      uint32_t v = va ^ vb;
      v = v - ((v >> 1) & 0x55555555);
      v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
      dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
  }
};

}  // namespace sdvlref

#endif  // SDVL_ORACLE_REF_ORB_H_
