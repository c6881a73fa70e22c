// sdvl_orb_device.h — the wave64 ORB descriptor (orientation + 256-bit steered BRIEF) as a device function, shared by
// orb_describe_kernel (sdvl_orb.hip: every corner of a frame) and search_points_kernel (sdvl_search.hip: only the corners
// a search actually compares, when the frame's descriptors have not been computed).
//   ORBDetector::GetDescriptor / GetOrientation, extra/orb_detector.cc:350-437.
//   Orientation: lanes stride the radius-15 disc (umax_ table of InitParameters :325-348), exact int32 moments, wave
//   butterfly; cv::fastAtan2 polynomial in float (all lanes redundantly -> no broadcast); cos/sin of the float angle in
//   double, rounded to float (DESIGN.md "frozen interpretations"); lane k evaluates tests 4k..4k+3 with cvRound =
//   v_rndne_f32.  No FMA contraction (-ffp-contract=off): sample coordinates round exactly as on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sdvl_math.h"

namespace {

__constant__ __attribute__((aligned(16))) int8_t c_orb_pattern[256 * 4] = {
#include "orb_pattern_31.inc"
};


// the same pattern as floats: the steered sample coordinates are float products of these (no int8 -> float conversion per test)
__constant__ __attribute__((aligned(16))) float c_orb_pattern_f[256 * 4] = {
#include "orb_pattern_31.inc"
};

// Intensity centroid: task t = (row v = t / 8 - 15, four-pixel segment u0 = -16 + 4 (t % 8)) of the radius-15 disc.  Per task
// two byte-coefficient words for v_dot4_u32_u8: `m` = 1 where the pixel lies inside the disc (|u| <= umax[|v|],
// orb_detector.cc:325-348), `a` = (u + 16) there — sum u p = dot(p, a) - 16 dot(p, m), sum v p = v dot(p, m): exact int32
struct OrbMomentTable {
  uint32_t a[256], m[256];
};
constexpr OrbMomentTable make_orb_moment_table() {
  OrbMomentTable t{};
  constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  for (int task = 0; task < 256; task++) {
    uint32_t a = 0, m = 0;
    if (task < 248) {
      const int v = (task >> 3) - 15, u0 = -16 + 4 * (task & 7);
      const int um = umax[v < 0 ? -v : v];
      for (int k = 0; k < 4; k++) {
        const int u = u0 + k;
        if ((u < 0 ? -u : u) <= um) {
          a |= static_cast<uint32_t>(u + 16) << (8 * k);
          m |= 1u << (8 * k);
        }
      }
    }
    t.a[task] = a;
    t.m[task] = m;
  }
  return t;
}
__constant__ OrbMomentTable c_orb_moments = make_orb_moment_table();

// sum over the 64 lanes (all active), result in every lane: data-parallel-primitive adds inside the 16-lane rows, two row
// broadcasts, one readlane — 7 VALU instructions instead of 6 x (ds_bpermute + add) with their address arithmetic
__device__ __forceinline__ int orb_wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, false);  // row_shr:4, banks 1-3
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, false);  // row_shr:8, banks 2-3: lane 15 of a row holds the row's sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __builtin_amdgcn_readlane(v, 63);
}

// cv::fastAtan2 (degrees), OpenCV >= 2.4.9 scalar polynomial
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * static_cast<float>(180 / M_PI);
  const float p3 = -0.3258083974640975f * static_cast<float>(180 / M_PI);
  const float p5 = 0.1555786518463281f * static_cast<float>(180 / M_PI);
  const float p7 = -0.04432655554792128f * static_cast<float>(180 / M_PI);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + static_cast<float>(2.2204460492503131e-16));
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + static_cast<float>(2.2204460492503131e-16));
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__device__ __forceinline__ int cv_round_f(float v) { return static_cast<int>(__builtin_rintf(v)); }

// The 4 BRIEF bits of this lane (tests 4*lane .. 4*lane+3, bit q = test 4*lane+q) for the corner at `center` (row stride
// W); the caller has checked ORBDetector::IsInsideLimits (19 px from every border).  All 64 lanes must call.
// Byte b of the descriptor = nibble of lane 2b | nibble of lane 2b+1 << 4.
// The level comes as (base pointer, byte offset of the corner): every pixel address is base + a 32-bit offset, a load with a scalar
// base register and one vector add — `center + v * W + u0` on a generic pointer cost a sign extension and a 64-bit add per load.
typedef __attribute__((address_space(1))) const uint8_t *OrbGlobalBytes;
__device__ __forceinline__ uint32_t orb_wave_nibble(const uint8_t *level, uint32_t center_off, int W, int lane, float *angle_deg_out) {
  const OrbGlobalBytes center = (OrbGlobalBytes)level;
  // intensity centroid over the disc: 31 rows x 8 four-pixel segments (u = -16 .. 15) = 248 tasks over 64 lanes,
  // one unaligned 32-bit load per task; umax_ (orb_detector.cc:325-348) lives in two immediates, 4 bits per row
  int m10 = 0, m01 = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int task = lane + 64 * r;
    if (task < 248) {
      const int v = (task >> 3) - 15, u0 = -16 + 4 * (task & 7);
      const uint32_t o = center_off + static_cast<uint32_t>(v * W + u0);
      struct __attribute__((packed)) Unaligned32 { uint32_t v; };
      const uint32_t w = ((__attribute__((address_space(1))) const Unaligned32 *)(center + o))->v;  // one unaligned 32-bit load
      const int sp = static_cast<int>(__builtin_amdgcn_udot4(w, c_orb_moments.m[task], 0u, false));   // sum of the pixels inside the disc
      const int sup = static_cast<int>(__builtin_amdgcn_udot4(w, c_orb_moments.a[task], 0u, false));  // sum of (u + 16) p
      m10 += sup - 16 * sp;
      m01 += v * sp;
    }
  }
  m10 = orb_wave_sum_i32(m10);
  m01 = orb_wave_sum_i32(m01);
  const float angle_deg = fast_atan2_deg(static_cast<float>(m01), static_cast<float>(m10));
  const float factorPI = static_cast<float>(M_PI / 180.f);
  const float angle = static_cast<float>(static_cast<double>(angle_deg) * factorPI);
  double sn_d, cs_d;
  sdvl::sincos_2pi(static_cast<double>(angle), &sn_d, &cs_d);
  const float a = static_cast<float>(cs_d);
  const float b = static_cast<float>(sn_d);
  // lane k: tests 4k .. 4k+3  (byte k/2, bits (k&1)*4 ..); its 16 pattern bytes come in one 128-bit load
  const float4 *pf = reinterpret_cast<const float4 *>(c_orb_pattern_f) + 4 * lane;
  uint32_t nib = 0;
  // (row, column) of a steered point = (x b + y a, x a - y b) = x (b, a) + y (a, -b): two packed products and a packed sum
  // (v_pk_mul_f32 / v_pk_add_f32) per point instead of four products and two sums; x a + (-(y b)) IS x a - y b, bit for bit
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 ba = {b, a}, anb = {a, -b};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float4 t = pf[q];  // x0, y0, x1, y1 of test 4 lane + q
    const f2 p0 = f2{t.x, t.x} * ba + f2{t.y, t.y} * anb;
    const f2 p1 = f2{t.z, t.z} * ba + f2{t.w, t.w} * anb;
    const int t0 = center[center_off + static_cast<uint32_t>(cv_round_f(p0.x) * W + cv_round_f(p0.y))];
    const int t1 = center[center_off + static_cast<uint32_t>(cv_round_f(p1.x) * W + cv_round_f(p1.y))];
    nib |= (t0 < t1 ? 1u : 0u) << q;
  }
  *angle_deg_out = angle_deg;
  return nib;
}

// ---- the same descriptor out of an LDS copy of the corner's neighbourhood (round 4) ----------------------------------------
// orb_wave_nibble above makes TWO dependent trips to memory per corner: the 248 moment words, then — once the angle is known —
// the 512 test pixels.  A wave that computes descriptors on demand inside a search (search_points_kernel) lives from one
// memory round trip to the next (~15 of them per request, ~40 k cycles of wave-slot time for ~1200 instructions), so here the
// whole neighbourhood any moment word or steered test can touch comes in with ONE batch of loads: rows -18 .. 18 (the pattern's
// coordinates reach 13, rotated at most sqrt(2) 13 = 18.4 -> 18), columns -20 .. 19 (the moment words start at column -16; ten
// words per row), 37 x 40 bytes.  Same bytes, same arithmetic: bit-identical descriptors.
constexpr int kOrbWinRows = 37, kOrbWinPitch = 40, kOrbWinWords = kOrbWinRows * kOrbWinPitch / 4;  // 370 words = 1480 B

// all 64 lanes: copy the window around `center` (row stride W; ORBDetector::IsInsideLimits holds: 19 px from every border, so
// every byte read lies inside the image) into win[kOrbWinWords]
__device__ __forceinline__ void orb_stage_window(const uint8_t *center, int W, int lane, uint32_t *win) {
#pragma unroll
  for (int r = 0; r < (kOrbWinWords + 63) / 64; r++) {
    const int i = lane + 64 * r;
    if (i < kOrbWinWords) {
      const int row = i / 10, cw = i - row * 10;
      uint32_t w;
      __builtin_memcpy(&w, center + (row - 18) * W + (cw * 4 - 20), 4);
      win[i] = w;
    }
  }
}

// the caller orders the wave's LDS writes before this (wave fence)
__device__ __forceinline__ uint32_t orb_wave_nibble_win(const uint32_t *win, int lane, float *angle_deg_out) {
  const uint8_t *wb = reinterpret_cast<const uint8_t *>(win) + 18 * kOrbWinPitch + 20;  // the centre pixel
  int m10 = 0, m01 = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int task = lane + 64 * r;
    if (task < 248) {
      const int v = (task >> 3) - 15, u0 = -16 + 4 * (task & 7);
      const uint32_t w = win[((v + 18) * kOrbWinPitch + 20 + u0) >> 2];  // (v + 18) * 40 + 4 + 4 (task & 7): word-aligned
      const int sp = static_cast<int>(__builtin_amdgcn_udot4(w, c_orb_moments.m[task], 0u, false));
      const int sup = static_cast<int>(__builtin_amdgcn_udot4(w, c_orb_moments.a[task], 0u, false));
      m10 += sup - 16 * sp;
      m01 += v * sp;
    }
  }
  m10 = orb_wave_sum_i32(m10);
  m01 = orb_wave_sum_i32(m01);
  const float angle_deg = fast_atan2_deg(static_cast<float>(m01), static_cast<float>(m10));
  const float factorPI = static_cast<float>(M_PI / 180.f);
  const float angle = static_cast<float>(static_cast<double>(angle_deg) * factorPI);
  double sn_d, cs_d;
  sdvl::sincos_2pi(static_cast<double>(angle), &sn_d, &cs_d);
  const float a = static_cast<float>(cs_d);
  const float b = static_cast<float>(sn_d);
  const float4 *pf = reinterpret_cast<const float4 *>(c_orb_pattern_f) + 4 * lane;
  uint32_t nib = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float4 t = pf[q];  // x0, y0, x1, y1 of test 4 lane + q
    const int t0 = wb[cv_round_f(t.x * b + t.y * a) * kOrbWinPitch + cv_round_f(t.x * a - t.y * b)];
    const int t1 = wb[cv_round_f(t.z * b + t.w * a) * kOrbWinPitch + cv_round_f(t.z * a - t.w * b)];
    nib |= (t0 < t1 ? 1u : 0u) << q;
  }
  *angle_deg_out = angle_deg;
  return nib;
}

}  // namespace
