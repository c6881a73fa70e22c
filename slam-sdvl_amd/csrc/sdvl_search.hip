// sdvl_search.hip — K7 batched Matcher::SearchPoint (matcher.cc:45-121), ONE wave64 per request.
//   phase 0  (all lanes redundantly, FP64): relative pose, depth-interval projection, margin test,
//            WarpMatrixAffine (:293-312), GetSearchLevel (:314-323), search range;
//   phase 1  CreatePatch (:325-357): the 100 samples of the 10x10 warped patch over 64 lanes (2 rounds),
//            Interpolate8U (extra/utils.cc:44-59) in float, truncated to u8 -> LDS;
//   phase 2  GetCornersInRange (:123-230) + SearchFeatures (:232-291): lanes stride the frame's corner list,
//            range test in FP64, Hamming distance on the HBM-resident ORB descriptors (or integer ZMSSD),
//            wave arg-min on the key (score << 20 | index) -> first index wins ties like the sequential loop;
//   phase 3  AlignPatch (:359-445): lane = pixel of the 8x8 patch.  Template gradients and the 3x3 H from the LDS
//            border patch (H entries are exact quarter-integers -> any summation order is exact), Eigen's cofactor
//            inverse in float, then <= max_align_its iterations: bilinear sample per lane, residual products to LDS
//            and three lanes accumulate them IN THE REFERENCE'S SEQUENTIAL ORDER so that update / convergence
//            decisions are bit-identical to the CPU path.
// Compiled with -ffp-contract=off.  sdvl_align_patches exposes phase 3 alone.
#include <unordered_map>
#include <utility>
#include <type_traits>
#include <vector>

#include <cmath>

#include "sdvl_internal.h"
#include "sdvl_math.h"
#include "sdvl_orb_device.h"
#include "sdvl_search_types.h"
#include "sdvl_search_prepare.h"

namespace {

using namespace sdvl;

struct PatchJob {  // sdvl_align_patches
  const uint8_t *img;
  int w, h;
  double u, v;
};

// Round 4 experiment, measured and switched OFF: staging the descriptor neighbourhood (37 x 40 B) and the LK neighbourhood (20 x 20 B)
// in LDS with one batch of loads each, on the theory that a search wave lives from one memory round trip to the next.  It does not:
// the second and later touches of those bytes are L1 hits, and the staging's own index arithmetic, fences and 9 more VGPRs (64 +
// 60 B of scratch instead of 55) cost more than they save — alone 155 us per 256 frames against 132, in the bench 5.3 ms of
// dispatch time per step against 3.9, 330 k tracked frames/s against 350 k.  The waves are long-lived because eight of them share
// a SIMD whose issue port is busy, not because they wait for memory: only fewer instructions help.
constexpr bool kSearchOrbWindow = false, kSearchLkWindow = false;
constexpr int kLkWin = 20, kLkWinBack = 9;  // LDS copy of the search image around the LK start: 20 x 20 bytes from (u0 - 9, v0 - 9)
struct WaveLds {
  uint8_t border[104];
  uint8_t patch[64];
  float prod[3][64];
  // round 4: one memory round trip instead of several.  orb: the 37 x 40 neighbourhood of a corner whose descriptor is computed on
  // demand (sdvl_orb_device.h); lk: the 20 x 20 neighbourhood of the LK start — every iteration's 9 x 9 window is read out of it
  // as long as the patch stays within [-5, +6] px of where it started (it moves a pixel or two; beyond that: the image itself)
  uint32_t orb[kSearchOrbWindow ? kOrbWinWords : 1];
  uint32_t lk[kSearchLkWindow ? kLkWin * kLkWin / 4 : 1];
};

__device__ __forceinline__ float interpolate8u(const uint8_t *img, int stride, float u, float v) {
  const int x = static_cast<int>(floorf(u));
  const int y = static_cast<int>(floorf(v));
  const float sx = u - x, sy = v - y;
  const float w00 = (1.0f - sx) * (1.0f - sy);
  const float w01 = (1.0f - sx) * sy;
  const float w10 = sx * (1.0f - sy);
  const float w11 = 1.0f - w00 - w01 - w10;
  // (global memory behind a scalar base, one 32-bit offset: x, y >= 0 — the caller has tested the position against the image)
  const __attribute__((address_space(1))) uint8_t *g = (const __attribute__((address_space(1))) uint8_t *)img;
  const uint32_t o = static_cast<uint32_t>(y * stride + x), st = static_cast<uint32_t>(stride);
  return w00 * g[o] + w01 * g[o + st] + w10 * g[o + 1u] + w11 * g[o + st + 1u];
}

// LDS hand-off between the lanes of ONE wave: order the compiler's view of memory, no s_barrier needed
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_sum_i32(int v) { return orb_wave_sum_i32(v); }  // DPP row adds + row broadcasts (sdvl_orb_device.h)

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned lo = static_cast<unsigned>(v), hi = static_cast<unsigned>(v >> 32);
    lo = __shfl_xor(lo, off, 64);
    hi = __shfl_xor(hi, off, 64);
    const unsigned long long o = (static_cast<unsigned long long>(hi) << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}

// minimum over the 64 lanes (all active) in every lane: DPP steps inside the 16-lane rows, two row broadcasts, one readlane
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  const auto step = [](uint32_t x, auto ctrl, auto rows) {
    const uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(x), decltype(ctrl)::value, decltype(rows)::value, 0xf, false));
    return o < x ? o : x;
  };
  v = step(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
  v = step(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
  v = step(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
  v = step(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8: lane 15 of a row holds the row's minimum
  v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3
  v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2 and 3
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

// Matcher::AlignPatch, matcher.cc:359-445.  border/patch in this wave's LDS.  Returns converged; *u,*v updated.
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <bool kWindow = false>
__device__ bool align_patch_wave(WaveLds &L, const uint8_t *img, int W, int H, int max_its, int lane, float *u_io, float *v_io,
                                 int *its_out, bool tree_sums = false) {
  const int y = lane >> 3, x = lane & 7;
  // kWindow: the neighbourhood of the start position into LDS with one batch of loads (bytes outside the image are never read
  // back: an iteration whose 9 x 9 window leaves the image ends the loop before it reads, matcher.cc:402)
  int win_x0 = 0, win_y0 = 0;
  bool win_ok = false;
  if (kWindow) {
    const float fu0 = floorf(*u_io), fv0 = floorf(*v_io);
    if (fu0 >= 4.f && fv0 >= 4.f && fu0 < static_cast<float>(W - 4) && fv0 < static_cast<float>(H - 4)) {
      win_ok = true;
      win_x0 = static_cast<int>(fu0) - kLkWinBack;
      win_y0 = static_cast<int>(fv0) - kLkWinBack;
      const long long total = static_cast<long long>(W) * H;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int i = lane + 64 * r;
        if (i < kLkWin * kLkWin / 4) {
          const int row = i / (kLkWin / 4), cw = i - row * (kLkWin / 4);
          const int gy = win_y0 + row;
          const long long o = static_cast<long long>(gy) * W + (win_x0 + 4 * cw);
          uint32_t w = 0;
          if (gy >= 0 && gy < H) {
            if (o >= 0 && o + 4 <= total) {
              __builtin_memcpy(&w, img + o, 4);
            } else {
#pragma unroll
              for (int k = 0; k < 4; k++)
                if (o + k >= 0 && o + k < total) w |= static_cast<uint32_t>(img[o + k]) << (8 * k);
            }
          }
          L.lk[i] = w;
        }
      }
    }
  }
  const uint8_t *it = &L.border[(y + 1) * 10 + 1 + x];
  const int jx = static_cast<int>(it[1]) - static_cast<int>(it[-1]);
  const int jy = static_cast<int>(it[10]) - static_cast<int>(it[-10]);
  const float dx = static_cast<float>(0.5 * jx);
  const float dy = static_cast<float>(0.5 * jy);
  // H = sum J J^T with J = (dx, dy, 1): quarter-integers, exact in float in any order
  const float h00 = 0.25f * static_cast<float>(wave_sum_i32(jx * jx));
  const float h01 = 0.25f * static_cast<float>(wave_sum_i32(jx * jy));
  const float h11 = 0.25f * static_cast<float>(wave_sum_i32(jy * jy));
  const float h02 = 0.5f * static_cast<float>(wave_sum_i32(jx));
  const float h12 = 0.5f * static_cast<float>(wave_sum_i32(jy));
  const float m[3][3] = {{h00, h01, h02}, {h01, h11, h12}, {h02, h12, 64.f}};
  // Eigen compute_inverse_size3 (cofactors of column 0, det, multiply by the reciprocal)
  const float c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  const float c10 = m[2][1] * m[0][2] - m[2][2] * m[0][1];
  const float c20 = m[0][1] * m[1][2] - m[0][2] * m[1][1];
  const float det = (c00 * m[0][0] + c10 * m[1][0]) + c20 * m[2][0];
  const float invdet = 1.0f / det;
  float inv[3][3];
  inv[0][0] = c00 * invdet;
  inv[0][1] = c10 * invdet;
  inv[0][2] = c20 * invdet;
  inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * invdet;
  inv[1][1] = (m[2][2] * m[0][0] - m[2][0] * m[0][2]) * invdet;
  inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * invdet;
  inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * invdet;
  inv[2][1] = (m[2][0] * m[0][1] - m[2][1] * m[0][0]) * invdet;
  inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * invdet;

  if (kWindow) wave_sync();  // the staged window is visible to every lane (the loads have had the set-up above to arrive)
  const float ref = static_cast<float>(L.patch[lane]);
  float mean_diff = 0.f;
  float u = *u_io, v = *v_io;
  const float min_update_squared = static_cast<float>(0.03 * 0.03);
  bool converged = false;
  int iters = 0;
  for (int iter = 0; iter < max_its; iter++) {
    const float fu = floorf(u), fv = floorf(v);
    // u_r < half || v_r < half || u_r >= cols-half || v_r >= rows-half -> break   (NaN fails the >= tests -> break)
    if (!(fu >= 4.f && fv >= 4.f && fu < static_cast<float>(W - 4) && fv < static_cast<float>(H - 4))) break;
    const int u_r = static_cast<int>(fu), v_r = static_cast<int>(fv);
    iters++;
    const float sx = u - u_r, sy = v - v_r;
    const float wTL = static_cast<float>((1.0 - sx) * (1.0 - sy));
    const float wTR = static_cast<float>(sx * (1.0 - sy));
    const float wBL = static_cast<float>((1.0 - sx) * sy);
    const float wBR = sx * sy;
    float search_pixel;
    // the 9 x 9 window of this iteration inside the staged 20 x 20? (wave-uniform: u, v are)
    if (kWindow && win_ok && u_r - 4 >= win_x0 && u_r + 4 < win_x0 + kLkWin && v_r - 4 >= win_y0 && v_r + 4 < win_y0 + kLkWin) {
      const uint8_t *ip = reinterpret_cast<const uint8_t *>(L.lk) + (v_r + y - 4 - win_y0) * kLkWin + (u_r + x - 4 - win_x0);
      search_pixel = wTL * ip[0] + wTR * ip[1] + wBL * ip[kLkWin] + wBR * ip[kLkWin + 1];
    } else {
      // (u_r, v_r >= 4: the offset is not negative; global memory behind a scalar base)
      const __attribute__((address_space(1))) uint8_t *g = (const __attribute__((address_space(1))) uint8_t *)img;
      const uint32_t o = static_cast<uint32_t>((v_r + y - 4) * W + (u_r + x - 4)), st = static_cast<uint32_t>(W);
      search_pixel = wTL * g[o] + wTR * g[o + 1u] + wBL * g[o + st] + wBR * g[o + st + 1u];
    }
    const float res = search_pixel - ref + mean_diff;
    float J0, J1, J2;
    if (tree_sums) {  // tolerance class: the same 64 terms summed as a butterfly (18 shuffles instead of a 64-step chain)
      J0 = -wave_sum_f32(res * dx);
      J1 = -wave_sum_f32(res * dy);
      J2 = -wave_sum_f32(res);
    } else {
      L.prod[0][lane] = res * dx;
      L.prod[1][lane] = res * dy;
      L.prod[2][lane] = res;
      wave_sync();
      // sequential accumulation (matcher.cc:427-429): lanes 0..2 own one component each
      float acc = 0.f;
      if (lane < 3) {
#pragma unroll 8
        for (int k = 0; k < 64; k++) acc -= L.prod[lane][k];
      }
      const int acc_bits = __builtin_bit_cast(int, acc);  // lanes 0..2 hold the three sums: one v_readlane each
      J0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(acc_bits, 0));
      J1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(acc_bits, 1));
      J2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(acc_bits, 2));
      wave_sync();
    }
    const float up0 = inv[0][0] * J0 + inv[0][1] * J1 + inv[0][2] * J2;
    const float up1 = inv[1][0] * J0 + inv[1][1] * J1 + inv[1][2] * J2;
    const float up2 = inv[2][0] * J0 + inv[2][1] * J1 + inv[2][2] * J2;
    u += up0;
    v += up1;
    mean_diff += up2;
    if (up0 * up0 + up1 * up1 < min_update_squared) {
      converged = true;
      break;
    }
  }
  *u_io = u;
  *v_io = v;
  *its_out = iters;
  return converged;
}

// Phase 0 of SearchPoint is scalar work per request (relative pose, depth-interval projection, margin test, affine warp,
// search level): one LANE per request here instead of a whole wave repeating it 64 times in search_points_kernel.
__global__ __launch_bounds__(64) void search_prepare_kernel(const SearchReqDev *__restrict__ reqs, const SearchFramePose *__restrict__ table,
                                                             int n, Cam cam, sdvl_search_params prm, SearchPrep *__restrict__ prep) {
  const int ri = blockIdx.x * 64 + threadIdx.x;  // one-wave workgroups: placed as soon as ONE wave slot is free (see search_points_kernel)
  if (ri >= n) return;
  const SearchReqDev &rq = reqs[ri];
  if (rq.level < 0) {  // a dead slot of a device-built batch: the frame indices may be anything
    prep[ri] = search_prepare_one(rq, se3_identity(), se3_identity(), cam, prm);
    return;
  }
  prep[ri] = search_prepare_one(rq, se3_from7(table[rq.cur].pose), se3_from7(table[rq.ref].pose), cam, prm, &table[rq.cur].f);
}

// kStage: frames without corner bins get their corner list staged in LDS (16 KB per workgroup).  Launches whose frames all
// carry bins (every frame that went through sdvl_detect_corners: the tracking path) use the form without the stage: 3.7 KB of
// LDS per workgroup, so its workgroups fit beside the LDS-heavy kernels of the other streams.
// kW: waves of a workgroup.  The staged form shares the frame's corner list among the kWavesPerBlock requests of a block, so its
// workgroup is the block (kW = kWavesPerBlock).  The form without the stage (round 3) runs ONE WAVE per workgroup, four workgroups per
// block: among the other streams' one-wave kernels (fast_cells, pyr_down) a workgroup that needs four free wave slots on one CU at
// the same moment is placed far less often than its share — every slot that frees is taken by a one-wave workgroup first.
#ifdef SDVL_SEARCH_STATS
// diagnostic build only (make HIPFLAGS+=-DSDVL_SEARCH_STATS): requests, binned / full scans, scan rounds, corners in range, regions (cells)
__device__ unsigned long long g_search_stats[8];
#define SDVL_STAT(i, v) do { const unsigned long long v_ = static_cast<unsigned long long>(v); if (lane == 0) atomicAdd(&g_search_stats[i], v_); } while (0)
#else
#define SDVL_STAT(i, v) do { } while (0)
#endif

// -DSDVL_SEARCH_STAMPS: diagnostic build, time per part of search_points_kernel summed over all requests (s_memtime ticks) -> sdvl_debug_search_stamps
#ifdef SDVL_SEARCH_STAMPS
__device__ unsigned long long *g_search_stamps;  // 8 per request (no atomics: they would congest what is being timed)
constexpr int kStampRequests = 1 << 18;
#define SS_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[k] += now_ - t_last_; t_last_ = now_; } while (0)
#define SS_FLUSH() do { if (lane == 0 && ri < kStampRequests && g_search_stamps) { for (int q_ = 0; q_ < 7; q_++) g_search_stamps[ri * 8 + q_] += st_[q_]; g_search_stamps[ri * 8 + 7] += 1ull; } } while (0)
#else
#define SS_STAMP(k) do { } while (0)
#define SS_FLUSH() do { } while (0)
#endif


template <bool kStage, int kW>
__global__ __launch_bounds__(64 * kW) __attribute__((amdgpu_waves_per_eu(8, 8))) void search_points_kernel(const SearchReqDev *__restrict__ reqs,
                                                                            const SearchFramePose *__restrict__ table,
                                                                            const SearchBlock *__restrict__ blocks,
                                                                            const SearchPrep *__restrict__ prep, Cam cam,
                                                                            sdvl_search_params prm, int n_blocks,
                                                                            sdvl_search_res *__restrict__ out,
                                                                            sdvl_search_res *__restrict__ out_host) {
  static_assert(kW == kWavesPerBlock || (kW == 1 && !kStage), "one wave per workgroup only without the shared corner stage");
  // round 4: descriptor neighbourhoods and the LK neighbourhood through LDS (WaveLds::orb, ::lk)
  constexpr bool kOrbWindow = kSearchOrbWindow, kLkWindow = kSearchLkWindow;
  __shared__ WaveLds s_lds[kW];
  // the current frame's corner list, packed x | y << 12 | level << 24, read from HBM once per workgroup instead of once
  // per request (GetCornersInRange scans ALL corners for every point, matcher.cc:123-230)
  __shared__ uint32_t s_corners[kStage ? kLdsCorners : 1];
  const int lane = threadIdx.x & 63;
#ifdef SDVL_SEARCH_STAMPS
  unsigned long long st_[7] = {0, 0, 0, 0, 0, 0, 0}, t_last_ = __builtin_amdgcn_s_memtime();
#endif
  // wave-uniform: request, prep and frame-table loads become scalar loads.  kW == 1: the grid holds kWavesPerBlock workgroups per block
  const int wv = kW == 1 ? static_cast<int>((blockIdx.x >> 3) % kWavesPerBlock) : __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  // Workgroups go to the 8 XCDs round-robin by linear id (gridDim.x is padded to a multiple of 8).  XCD x takes the x-th
  // eighth of the block table: blocks are ordered by current frame, so one frame's corner list, search-level image and
  // ORB windows are fetched into ONE L2 instead of all eight.
  // (kW == 1: the grid is kWavesPerBlock times as long; workgroup h runs on XCD h & 7, and within an XCD's sequence h >> 3 the
  //  kWavesPerBlock workgroups of a block follow each other)
  const int per_xcd = static_cast<int>((kW == 1 ? gridDim.x / kWavesPerBlock : gridDim.x) >> 3);
  const int bi = static_cast<int>(blockIdx.x & 7u) * per_xcd + static_cast<int>(kW == 1 ? (blockIdx.x >> 3) / kWavesPerBlock : (blockIdx.x >> 3));
  if (bi >= n_blocks) return;
  const SearchBlock blk = blocks[bi];
  if (blk.count <= 0) return;  // device-built batches reserve blocks for the most requests a tracker can have
  const SearchFramePose &tcur = table[reqs[blk.first].cur];
  const int n_corners = min(tcur.f.n_ptr[0], SDVL_MAX_CORNERS);
  const int4 *corners_g = reinterpret_cast<const int4 *>(tcur.f.corners);
  const auto pack_corner = [](const int4 c) {
    return static_cast<uint32_t>(c.x) | (static_cast<uint32_t>(c.y) << 12) | (static_cast<uint32_t>(c.z) << 24);
  };
  // A frame whose corners are binned by cell (sdvl_detect_corners) is searched through its bins: no staging, no barrier.
  // (workgroup-uniform: the workgroup's requests share the current frame)
  const bool binned = tcur.f.bin_start != nullptr;
  if (kStage && !binned) {
    for (int ci = threadIdx.x; ci < min(n_corners, kLdsCorners); ci += 64 * kW) s_corners[ci] = pack_corner(corners_g[ci]);
    __syncthreads();
  }
  const auto corner_at = [&](int ci) { return (kStage && !binned && ci < kLdsCorners) ? s_corners[ci] : pack_corner(corners_g[ci]); };
  if (wv >= blk.count) return;
  const int ri = blk.first + wv;
  WaveLds &L = s_lds[kW == 1 ? 0 : wv];
  const SearchReqDev &rq = reqs[ri];
  const SearchFramePose &tref = table[rq.ref];
  sdvl_search_res res;
  res.px[0] = rq.px0[0];
  res.px[1] = rq.px0[1];
  res.found = 0;
  res.level = -1;
  res.best_corner = -1;
  res.stage = 0;
  res.lk_its = 0;
  res.slevel = -1;

  const int level = rq.level;
  const SearchPrep pr = prep[ri];
  if (!pr.alive) {
    if (lane == 0) { out[ri] = res; if (out_host) out_host[ri] = res; }
    return;
  }
  // The epipolar constants of the request (ends of the projected depth interval, line normal, ...) are read where an epipolar
  // request uses them, through a pointer the compiler cannot trace back to prep[ri] (`epi()`: an empty asm hands it over in scalar
  // registers): loaded up front they sat in 30 scalar registers across the whole kernel, and the kernel spilled 70 scalar registers
  // into vector lanes — ~300 v_readlane / v_writelane (vector instructions: the path is bound by their issue) and as many s_nop in
  // the hot loops, 40 of them in every round of the range test.  Tracked points are `fixed` (a circle around px0): they never
  // touch these.  (Not `volatile`: volatile loads are system-coherent flat loads with a wait each.)
  const auto epi = [&]() {
    const SearchPrep *pe = prep + ri;
    asm volatile("" : "+s"(pe));
    return pe;
  };
  const int slevel = pr.slevel;
  res.slevel = slevel;
  SS_STAMP(0);
  // ---- CreatePatch, matcher.cc:325-357
  {
    const double I00 = pr.I00, I01 = pr.I01, I10 = pr.I10, I11 = pr.I11;
    const uint8_t *img = tref.f.level[level];
    const int W = tref.f.lw[level], H = tref.f.lh[level];
    // x / 2^level as ldexp(x, -level): the same double (a power of two scales exactly), one instruction instead of a division's fifteen
    const double pyrx = __builtin_ldexp(rq.px[0], -level), pyry = __builtin_ldexp(rq.px[1], -level);
    const bool bad = (I00 != I00);  // std::isnan(matrix_inv(0,0)): the reference returns leaving stale patches
    for (int s = lane; s < 100; s += 64) {
      const int y = s / 10, x = s - y * 10;
      double ppx = x - 5, ppy = y - 5;
      ppx *= (1 << slevel);
      ppy *= (1 << slevel);
      const double p0 = (I00 * ppx + I01 * ppy) + pyrx;
      const double p1 = (I10 * ppx + I11 * ppy) + pyry;
      uint8_t val = 0;
      if (!bad && !(p0 < 0 || p1 < 0 || p0 >= W - 1 || p1 >= H - 1) && p0 == p0 && p1 == p1)
        val = static_cast<uint8_t>(interpolate8u(img, W, static_cast<float>(p0), static_cast<float>(p1)));
      L.border[s] = val;
      if (y >= 1 && y < 9 && x >= 1 && x < 9) L.patch[(y - 1) * 8 + (x - 1)] = val;
    }
  }
  wave_sync();
  SS_STAMP(1);
  const double range = pr.range, range2 = pr.range2;

  // ---- GetCornersInRange + SearchFeatures
  const int threshold = prm.use_orb ? 100 : prm.patch_size * prm.patch_size * 500;
  unsigned long long best = ~0ull;
  uint32_t best_pk = 0;
  {
    const SearchFrame &cf = tcur.f;
    int sumA = 0, sumAA = 0;
    if (!prm.use_orb) {
      const int pv = L.patch[lane];
      sumA = wave_sum_i32(pv);
      sumAA = wave_sum_i32(pv * pv);
    }
    // a current frame without descriptors (sdvl_orb_describe not run): the wave computes the descriptor of every corner
    // that falls in range on the spot — same arithmetic, same values; a tracking step compares ~200 of ~1000 corners
    const bool lazy_desc = prm.use_orb && cf.desc == nullptr;
    // one round of GetCornersInRange + SearchFeatures: lane's corner (packed `pk`, list index `ci`; `have` = the lane holds one)
    const auto scan_round = [&](bool have, uint32_t pk, int ci) {
      bool inr = have;
      int cx = 0, cy = 0, cl = 0;
      if (inr) {
        cx = static_cast<int>(pk & 0xFFFu);
        cy = static_cast<int>((pk >> 12) & 0xFFFu);
        cl = static_cast<int>(pk >> 24);
        int d = cl - level;
        if (d < 0) d = -d;
        if (d > 1) inr = false;
      }
      if (inr && (cx - prm.margin < 0 || cy - prm.margin < 0)) inr = false;
      if (inr && (cy + prm.margin >= cf.lh[cl] || cx + prm.margin >= cf.lw[cl])) inr = false;
      if (inr) {
        const double posx = cx * (1 << cl), posy = cy * (1 << cl);
        if (rq.fixed) {
          const double ddx = rq.px0[0] - posx, ddy = rq.px0[1] - posy;
          if (ddx * ddx + ddy * ddy > range2) inr = false;
        } else {
          const SearchPrep *prv = epi();
          const double nx = prv->nx, ny = prv->ny, normdist = prv->normdist;
          const double dist = normdist - (posx * nx + posy * ny);
          if (fabs(dist) > range) inr = false;
          if (inr) {
            const double xdiff = prv->xdiff, ydiff = prv->ydiff, vline = prv->vline;
            const double pxa_x = prv->pxa[0], pxa_y = prv->pxa[1];
            const double uu = ((posx - pxa_x) * xdiff + (posy - pxa_y) * ydiff) / vline;
            if (uu > 1) {
              const double ddx = posx - prv->pxb[0], ddy = posy - prv->pxb[1];
              if ((ddx * ddx + ddy * ddy) > range2) inr = false;
            }
            if (inr && uu < 0) {
              const double ddx = posx - pxa_x, ddy = posy - pxa_y;
              if ((ddx * ddx + ddy * ddy) > range2) inr = false;
            }
          }
        }
      }
      int score = 0;
      SDVL_STAT(4, __popcll(__ballot(inr)));
      if (lazy_desc) {
        unsigned long long m = __ballot(inr);
        const uint32_t rq_nib = (rq.desc[lane >> 3] >> (4 * (lane & 7))) & 0xFu;
        SS_STAMP(2);
        while (m) {
          const int j = __ffsll(static_cast<long long>(m)) - 1;
          m &= m - 1;
          const uint32_t pj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(pk), j));
          const int jx = static_cast<int>(pj & 0xFFFu), jy = static_cast<int>((pj >> 12) & 0xFFFu), jl = static_cast<int>(pj >> 24);
          const int Wj = cf.lw[jl], Hj = cf.lh[jl];
          uint32_t nib = 0;  // outside ORBDetector::IsInsideLimits the descriptor is all zeros (sdvl_orb.hip)
          if (jx >= 19 && jx < Wj - 19 && jy >= 19 && jy < Hj - 19) {
            float angle_deg;
            if (kOrbWindow) {  // one batch of loads into LDS, moments and steered tests out of it (sdvl_orb_device.h)
              wave_sync();     // the previous corner's window has been read by every lane
              orb_stage_window(cf.level[jl] + static_cast<size_t>(jy) * Wj + jx, Wj, lane, L.orb);
              wave_sync();
              nib = orb_wave_nibble_win(L.orb, lane, &angle_deg);
            } else {
              nib = orb_wave_nibble(cf.level[jl], static_cast<uint32_t>(jy * Wj + jx), Wj, lane, &angle_deg);
            }
          }
          const int sc = wave_sum_i32(__popc(nib ^ rq_nib));
          if (lane == j) score = sc;
#ifdef SDVL_SEARCH_STAMPS
          st_[6]++;
#endif
        }
        SS_STAMP(3);
      }
      if (inr) {
        if (prm.use_orb) {
          if (!lazy_desc) {
            const uint32_t *dd = reinterpret_cast<const uint32_t *>(cf.desc + static_cast<size_t>(ci) * 32);
#pragma unroll
            for (int k = 0; k < 8; k++) score += __popc(dd[k] ^ rq.desc[k]);
          }
        } else {
          // CompareZMSSDScore, matcher.cc:461-476 (integer arithmetic, integer division by 64)
          const uint8_t *cp = cf.level[cl] + static_cast<size_t>(cy - 4) * cf.lw[cl] + (cx - 4);
          unsigned sumB = 0, sumBB = 0, sumAB = 0;
#pragma unroll 1
          for (int yy = 0, r = 0; yy < 8; yy++)
            for (int xx = 0; xx < 8; xx++, r++) {
              const unsigned pix = cp[yy * cf.lw[cl] + xx];
              sumB += pix;
              sumBB += pix * pix;
              sumAB += pix * L.patch[r];
            }
          const int iB = static_cast<int>(sumB), iBB = static_cast<int>(sumBB), iAB = static_cast<int>(sumAB);
          score = sumAA - 2 * iAB + iBB - (sumA * sumA - 2 * sumA * iB + iB * iB) / 64;
        }
        // score < best_score with first-index-wins == min over (score, index); scores may be negative for ZMSSD
        const unsigned long long key =
            (static_cast<unsigned long long>(static_cast<unsigned>(score + 0x40000000)) << 20) | static_cast<unsigned>(ci);
        if (key < best) {
          best = key;
          best_pk = pk;  // the corner itself travels with the key: no trip to the corner list for the winner
        }
      }
    };
    // which corners to look at: with bins, the cells the search region touches (a superset of the corners in range: the
    // exact tests above decide); otherwise, or for regions spanning many cells, the whole list
    bool scanned = false;
    // an epipolar request whose depth interval projects onto ONE pixel (zero baseline: cur pose == ref pose) has NaN line constants
    // (division by a zero-length segment, matcher.cc:139-148): every range comparison of the reference is then false and EVERY corner
    // that passes the level and margin tests counts as in range.  A box around pxa would visit a few cells only — such requests
    // scan the whole list, like the reference.
    bool line_ok = true;
    if (!rq.fixed) {
      const SearchPrep *prv = epi();
      const double vline = prv->vline, nx = prv->nx, ny = prv->ny;
      line_ok = vline > 0.0 && nx == nx && ny == ny;
    }
    // one round of the binned scan: the entries of (at most four) cell rows as ONE lane space — a search region of a few cells holds
    // ~15 corners, so a round per row ran the whole range test two or three times for a quarter of a wave each.  Row r contributes
    // entries [e0[r], e0[r] + pre[r + 1] - pre[r]) (cells of a row are consecutive); lane index i maps to the row whose prefix range holds it.
    const auto scan_rows = [&](const int (&e0)[4], const int (&pre)[5]) {
      const int total = pre[4];
      for (int base = 0; base < total; base += 64) {
        const int i = base + lane;
        const bool have = i < total;
        int r = 0;
#pragma unroll
        for (int q = 1; q < 4; q++) r += (i >= pre[q]) ? 1 : 0;
        const int e = i + (r == 0 ? e0[0] - pre[0] : r == 1 ? e0[1] - pre[1] : r == 2 ? e0[2] - pre[2] : e0[3] - pre[3]);
        uint2 ent = make_uint2(0u, 0u);
        if (have) {  // (a global load: the entry as one 64-bit scalar type — vector classes do not copy out of an address space)
          const unsigned long long e64 = ((const __attribute__((address_space(1))) unsigned long long *)cf.bin_entries)[static_cast<uint32_t>(e)];
          ent = make_uint2(static_cast<uint32_t>(e64), static_cast<uint32_t>(e64 >> 32));
        }
        SDVL_STAT(3, 1);
        scan_round(have, ent.x, static_cast<int>(ent.y));
      }
    };
    // The request's lane of the prepare step has looked the bins up already (SearchPrep::bin_mode): nothing in range, or up to four cell rows.
    int row_e0[4] = {0, 0, 0, 0}, row_pre[5] = {0, 0, 0, 0, 0};
    int ry = 0, ry_end = -1, box_cx0 = 0, box_cx1 = 0;  // cell rows ry .. ry_end still to look up here (none when the prepare step did it)
    bool rows_ready = false;
    if (binned && pr.bin_mode == 1) {
      scanned = true;
    } else if (binned && pr.bin_mode == 2) {
      scanned = true;
      rows_ready = true;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        row_e0[r] = pr.bin_e0[r];
        row_pre[r + 1] = pr.bin_pre[r];
      }
    } else if (binned && line_ok) {
      double bx0, bx1, by0, by1;
      if (rq.fixed) {
        bx0 = rq.px0[0] - range; bx1 = rq.px0[0] + range; by0 = rq.px0[1] - range; by1 = rq.px0[1] + range;
      } else {
        const SearchPrep *prv = epi();
        const double ax = prv->pxa[0], ay = prv->pxa[1], bx = prv->pxb[0], by = prv->pxb[1];
        bx0 = fmin(ax, bx) - range; bx1 = fmax(ax, bx) + range; by0 = fmin(ay, by) - range; by1 = fmax(ay, by) + range;
      }
      const int gw = cf.bin_gw, gh = cf.bin_cells / cf.bin_gw;
      // NaN or absurd coordinates fail the comparisons below and fall through to the full scan
      if (bx0 > -1.0e6 && bx1 < 1.0e6 && by0 > -1.0e6 && by1 < 1.0e6) {
        const int cx0 = max(0, static_cast<int>(floor(bx0)) >> 5), cx1 = min(gw - 1, static_cast<int>(floor(bx1)) >> 5);
        const int cy0 = max(0, static_cast<int>(floor(by0)) >> 5), cy1 = min(gh - 1, static_cast<int>(floor(by1)) >> 5);
        if (cx1 < cx0 || cy1 < cy0) {
          scanned = true;  // the region lies outside the image: no corner can be in range
        } else if ((cx1 - cx0 + 1) * (cy1 - cy0 + 1) <= kBinRegionCells) {  // (larger regions: the full scan below)
          scanned = true;
          ry = cy0; ry_end = cy1; box_cx0 = cx0; box_cx1 = cx1;
        }
      }
    }
    // four cell rows at a time (a region of the metric configuration has at most four; configuration C's span up to a dozen)
    while (rows_ready || ry <= ry_end) {
      if (!rows_ready) {
        const int gw = cf.bin_gw;
        const int nrows = min(4, ry_end - ry + 1);
        row_pre[0] = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int cyi = min(ry + r, ry_end);
          const __attribute__((address_space(1))) int *bs = (const __attribute__((address_space(1))) int *)cf.bin_start;  // global, not flat, loads
          const uint32_t brow = static_cast<uint32_t>(cyi * gw);
          const int a = bs[brow + static_cast<uint32_t>(box_cx0)], b = bs[brow + static_cast<uint32_t>(box_cx1 + 1)];
          row_e0[r] = a;
          row_pre[r + 1] = row_pre[r] + (r < nrows ? b - a : 0);
        }
        ry += 4;
      }
      scan_rows(row_e0, row_pre);
      if (rows_ready) break;
    }
    SDVL_STAT(0, 1);
    SDVL_STAT(scanned ? 1 : 2, 1);
    if (!scanned) {
      SDVL_STAT(6, binned ? 1 : 0);
      SDVL_STAT(7, line_ok ? 1 : 0);
      for (int c0 = 0; c0 < n_corners; c0 += 64) {
        SDVL_STAT(3, 1);
        const int ci = c0 + lane;
        const bool have = ci < n_corners;
        scan_round(have, have ? corner_at(ci) : 0u, ci);
      }
    }
    const unsigned long long mine = best;
    if (prm.use_orb) {
      // Hamming scores are < 4096: the low word of a key (score << 20 | corner index) orders the keys, the high word is the same in all
      // of them (0x40000, the offset that keeps ZMSSD scores positive) — a 32-bit minimum in 7 DPP steps instead of 6 exchanges of 64 bits
      const uint32_t m32 = wave_min_u32(static_cast<uint32_t>(best));
      best = m32 == ~0u ? ~0ull : ((0x40000ull << 32) | m32);
    } else {
      best = wave_min_u64(best);
    }
    if (best != ~0ull) {  // keys are unique (they end in the corner's list index): exactly one lane holds the winner
      const unsigned long long w = __ballot(mine == best);
      best_pk = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(best_pk), __ffsll(static_cast<long long>(w)) - 1));
    }
  }
  SS_STAMP(2);
  res.stage = 1;
  bool matched = false;
  int best_ci = -1;
  if (best != ~0ull) {
    const int bscore = static_cast<int>(static_cast<unsigned>(best >> 20)) - 0x40000000;
    // best_score starts at threshold+1; "best_score >= threshold -> false" (matcher.cc:247,286)
    if (bscore < threshold + 1 && !(bscore >= threshold)) {
      matched = true;
      best_ci = static_cast<int>(best & 0xFFFFF);
    }
  }
  if (!matched) {
    if (lane == 0) { out[ri] = res; if (out_host) out_host[ri] = res; }
    SS_FLUSH();
    return;
  }
  res.best_corner = best_ci;
  const uint32_t bpk = best_pk;
  const int bx = static_cast<int>(bpk & 0xFFFu), by = static_cast<int>((bpk >> 12) & 0xFFFu), bl = static_cast<int>(bpk >> 24);
  const double mpx = static_cast<double>(bx * (1 << bl)), mpy = static_cast<double>(by * (1 << bl));
  res.px[0] = mpx;
  res.px[1] = mpy;
  // ---- AlignPatch at the search level
  float u = static_cast<float>(__builtin_ldexp(mpx, -slevel)), v = static_cast<float>(__builtin_ldexp(mpy, -slevel));  // mpx / (1 << slevel), exactly
  int its = 0;
  const bool conv = kLkWindow ? align_patch_wave<true>(L, tcur.f.level[slevel], tcur.f.lw[slevel], tcur.f.lh[slevel], prm.max_align_its, lane, &u, &v,
                                                       &its, prm.lk_tree_sums != 0)
                              : align_patch_wave<false>(L, tcur.f.level[slevel], tcur.f.lw[slevel], tcur.f.lh[slevel], prm.max_align_its, lane, &u, &v,
                                                        &its, prm.lk_tree_sums != 0);
  SS_STAMP(4);
#ifdef SDVL_SEARCH_STAMPS
  st_[5] += its;
#endif
  res.lk_its = its;
  res.stage = 2;
  if (conv) {
    res.px[0] = static_cast<double>(u) * (1 << slevel);
    res.px[1] = static_cast<double>(v) * (1 << slevel);
    res.level = slevel;
    res.found = 1;
    res.stage = 3;
  }
  if (lane == 0) { out[ri] = res; if (out_host) out_host[ri] = res; }
  SS_FLUSH();
}

#ifdef SDVL_SEARCH_STAMPS
extern "C" int sdvl_debug_search_stamps(unsigned long long *out8, int reset) {
  static unsigned long long *d_buf = nullptr;
  const size_t bytes = sizeof(unsigned long long) * 8 * kStampRequests;
  if (reset) {
    if (!d_buf && hipMalloc(&d_buf, bytes) != hipSuccess) return -1;
    if (hipMemset(d_buf, 0, bytes) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_search_stamps), &d_buf, sizeof(d_buf)) == hipSuccess ? 0 : -1;
  }
  if (!d_buf) return -1;
  std::vector<unsigned long long> h(static_cast<size_t>(8) * kStampRequests);
  if (hipMemcpy(h.data(), d_buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int q = 0; q < 8; q++) out8[q] = 0;
  for (int r = 0; r < kStampRequests; r++)
    for (int q = 0; q < 8; q++) out8[q] += h[static_cast<size_t>(r) * 8 + q];
  return 0;
}
#endif

__global__ __launch_bounds__(64 * kWavesPerBlock) void align_patches_kernel(const PatchJob *__restrict__ jobs, const uint8_t *__restrict__ border,
                                                                            const uint8_t *__restrict__ patch, int n, int max_its,
                                                                            double *__restrict__ uv_out, uint8_t *__restrict__ conv_out,
                                                                            int32_t *__restrict__ its_out) {
  __shared__ WaveLds s_lds[kWavesPerBlock];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * kWavesPerBlock + wv;
  if (i >= n) return;
  WaveLds &L = s_lds[wv];
  for (int s = lane; s < 100; s += 64) L.border[s] = border[static_cast<size_t>(i) * 100 + s];
  L.patch[lane] = patch[static_cast<size_t>(i) * 64 + lane];
  wave_sync();
  const PatchJob jb = jobs[i];
  float u = static_cast<float>(jb.u), v = static_cast<float>(jb.v);
  int its = 0;
  const bool conv = align_patch_wave(L, jb.img, jb.w, jb.h, max_its, lane, &u, &v, &its);
  if (lane == 0) {
    uv_out[2 * i] = u;
    uv_out[2 * i + 1] = v;
    conv_out[i] = conv ? 1 : 0;
    its_out[i] = its;
  }
}

// ---- second half of FeatureAlign::SelectPoints (feature_align.cc:105-149) on the device, for sdvl_search_run_chain:
// candidates come cell by cell in the reference's order; a cell's first found candidate is a match; the first
// max_matches matches are kept.  One workgroup per tracker; the matches leave as pose observations (feature_align.cc:132
// creates the Feature — bearing = Camera::Unproject(px) — SelectInliers reads bearing/z, the point and 1/2^level).
constexpr int kChainMaxCand = 16384;  // candidates of one tracker the kernel can flag in LDS

__global__ __launch_bounds__(256) void select_matches_kernel(const ChainFrameDev *__restrict__ frames, const int32_t *__restrict__ cand_req,
                                                             const int32_t *__restrict__ cand_first, const sdvl_search_res *__restrict__ res,
                                                             const double *__restrict__ req_point, Cam cam, PoseJobDev *__restrict__ jobs,
                                                             sdvl_pose_obs *__restrict__ obs, int32_t *__restrict__ n_obs_out,
                                                             int32_t *__restrict__ match_cand) {
  __shared__ uint8_t s_found[kChainMaxCand];
  __shared__ int s_wave[4];
  const ChainFrameDev &fr = frames[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = fr.cand_end - fr.cand_begin;
  const int32_t *cfirst = cand_first + fr.cand_begin;
  // cand_req == nullptr: candidate k was searched with request cand_begin + k (batches built on the device)
  const auto req_of = [&](int k) { return cand_req ? cand_req[fr.cand_begin + k] : fr.cand_begin + k; };
  for (int k = tid; k < n; k += 256) {
    const int r = req_of(k);
    s_found[k] = (r >= 0 && res[r].found != 0) ? 1 : 0;
  }
  __syncthreads();
  int running = 0;
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int k = c0 + tid;
    bool sel = false;
    if (k < n && s_found[k]) {
      sel = true;
      for (int j = cfirst[k] - fr.cand_begin; j < k; j++)
        if (s_found[j]) { sel = false; break; }
    }
    const unsigned long long m = __ballot(sel);
    const int below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
    __syncthreads();  // s_wave of the previous round has been read
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int base = running;
    for (int w = 0; w < wave; w++) base += s_wave[w];
    const int rank = base + below;
    if (sel && rank < fr.max_matches) {
      const sdvl_search_res &r = res[req_of(k)];
      if (match_cand) match_cand[fr.obs_begin + rank] = k;
      const V3 v = cam_unproject(cam, {r.px[0], r.px[1]});
      sdvl_pose_obs o;
      o.ax = v.x / v.z;
      o.ay = v.y / v.z;
      const double *P = req_point + 3 * static_cast<size_t>(req_of(k));
      o.px = P[0];
      o.py = P[1];
      o.pz = P[2];
      o.inv_cov = 1.0 / (1 << r.level);
      obs[fr.obs_begin + rank] = o;
    }
    running += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
  if (tid == 0) {
    const int size = min(running, fr.max_matches);
    PoseJobDev j;
    j.obs_begin = fr.obs_begin;
    j.n_obs = size;
    j.rand_begin = fr.rand_begin;
    j.nits_begin = size * (size + 1) / 2;  // row `size` of the all-sizes budget table
    for (int q = 0; q < 7; q++) j.pose[q] = fr.pose[q];
    j.pad_ = 0.0;
    jobs[blockIdx.x] = j;
    n_obs_out[blockIdx.x] = size;
  }
}

void fill_frame(SearchFrame *d, const sdvl_frame *f) {
  memset(d, 0, sizeof(SearchFrame));
  for (int l = 0; l < f->v.levels; l++) {
    d->level[l] = f->v.level[l];
    d->lw[l] = f->v.lw[l];
    d->lh[l] = f->v.lh[l];
  }
  d->corners = f->v.corners;
  d->desc = f->desc_valid ? f->v.desc : nullptr;  // null: the search computes what it compares (search_points_kernel)
  d->n_ptr = f->v.corner_hdr;
  d->levels = f->v.levels;
  if (f->bins_valid) {
    d->bin_start = f->bin_start;
    d->bin_entries = f->bin_entries;
    d->bin_gw = f->bin_gw;
    d->bin_cells = f->bin_cells;
  }
}


// ---------------------------------------------------------------------------------------------- the mapper's depth filter
// extra/utils.cc:193-205: A = [R v_ref | v_cur], depth2 = -(A^T A)^-1 A^T t, |depth2[0]|
__device__ __forceinline__ bool depth_from_triangulation(const Rigid &pose, V3 v_ref, V3 v_cur, double *depth) {
  const M3 R = se3_rot(pose);
  const V3 a0 = mvec(R, v_ref);
  const V3 a1 = v_cur;
  const double m00 = vdot(a0, a0), m01 = vdot(a0, a1), m11 = vdot(a1, a1);
  const double det = m00 * m11 - m01 * m01;
  if (det < 0.000001) return false;
  const double invdet = 1.0 / det;
  const double i00 = m11 * invdet, i01 = -m01 * invdet;
  const double n00 = -i00, n01 = -i01;
  const double r0x = n00 * a0.x + n01 * a1.x, r0y = n00 * a0.y + n01 * a1.y, r0z = n00 * a0.z + n01 * a1.z;
  const double d0 = r0x * pose.t.x + r0y * pose.t.y + r0z * pose.t.z;
  *depth = fabs(d0);
  return true;
}

// extra/utils.cc:207-213
__device__ __forceinline__ double parallax_of(V3 src1, V3 src2, V3 p3d) {
  V3 v1 = vsub(src1, p3d), v2 = vsub(src2, p3d);
  const double n1 = vnorm(v1), n2 = vnorm(v2);
  v1 = {v1.x / n1, v1.y / n1, v1.z / n1};
  v2 = {v2.x / n2, v2.y / n2, v2.z / n2};
  return vdot(v1, v2);
}

// point.cc:189-201
__device__ __forceinline__ double compute_tau(const Rigid &pose, V3 v, double depth, double px_error_angle) {
  const double PI = 3.14159265;
  const V3 t = pose.t;
  const V3 a = {v.x * depth - t.x, v.y * depth - t.y, v.z * depth - t.z};
  const double t_norm = vnorm(t), a_norm = vnorm(a);
  const double alpha = acos((v.x * t.x + v.y * t.y + v.z * t.z) / t_norm);
  const double beta = acos((a.x * -t.x + a.y * -t.y + a.z * -t.z) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = PI - alpha - beta_plus;
  const double depth_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return depth_plus - depth;
}

// point.cc:203-217
__device__ __forceinline__ double pdf_normal(double mean, double sd, double x) {
  const double PI = 3.14159265;
  double result = 0.0;
  if (sd <= 0) return result;
  double exponent = x - mean;
  exponent *= -exponent;
  exponent /= 2 * sd * sd;
  result = exp(exponent);
  result /= sd * sqrt(2.0 * PI);
  return result;
}

// The body of Map::UpdateCandidates' loop behind SearchPoint (map.cc:454-497), one lane per request: Unpromote for a miss;
// triangulation, parallax, the minimum-depth tests, Point::Update (point.cc:64-100) and Point::HasConverged (:164-178) for a
// hit.  The point's row in the tracking tables (if it has one) receives what the tracker reads of it.
__global__ __launch_bounds__(128) void depth_filter_kernel(const SearchReqDev *__restrict__ reqs, const SearchFramePose *__restrict__ table,
                                                           const sdvl_search_res *__restrict__ res, const sdvl_depth_state *__restrict__ state,
                                                           int n, Cam cam, sdvl_depth_params fp, TrackPoint *__restrict__ rows, int n_rows,
                                                           sdvl_depth_out *__restrict__ out, sdvl_depth_out *__restrict__ out_host) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= n) return;
  const SearchReqDev &rq = reqs[i];
  const sdvl_depth_state st = state[i];
  const sdvl_search_res r = res[i];
  TrackPoint *row = (rows && st.track_row >= 0 && st.track_row < n_rows) ? rows + st.track_row : nullptr;
  sdvl_depth_out o;
  o.outcome = SDVL_DEPTH_SKIPPED;
  o.n_failed = st.n_failed;
  o.rho = st.rho; o.sigma2 = st.sigma2; o.a = st.a; o.b = st.b;
  o.cos_alpha = 0.0; o.last_distance = 0.0;
  o.position[0] = o.position[1] = o.position[2] = 0.0;
  if (!r.found) {
    // Point::Unpromote, point.cc:109-118
    o.n_failed = st.n_failed + 1;
    o.b = st.b + 1.0;
    o.outcome = SDVL_DEPTH_NOT_FOUND | (o.n_failed > fp.max_failed ? SDVL_DEPTH_DELETED : 0);
    if (row) {
      row->n_failed = o.n_failed;
      if (o.outcome & SDVL_DEPTH_DELETED) row->status |= kTrash;
    }
  } else {
    const Rigid cur = se3_from7(table[rq.cur].pose), ref = se3_from7(table[rq.ref].pose);
    const V3 fv = {rq.bearing[0], rq.bearing[1], rq.bearing[2]};
    const Rigid pose = se3_mul(cur, se3_inverse(ref));  // map.cc:459
    const V3 v3d = cam_unproject(cam, {r.px[0], r.px[1]});
    double depth = 0.0;
    bool go = depth_from_triangulation(pose, fv, v3d, &depth);
    const Rigid ref_world = se3_inverse(ref), cur_world = se3_inverse(cur);  // Frame::GetWorldPose
    if (go) {
      const V3 p3d = se3_apply(ref_world, {depth * fv.x, depth * fv.y, depth * fv.z});
      const double cos_alpha = parallax_of(ref_world.t, cur_world.t, p3d);
      if (cos_alpha >= 0.999999) go = false;
    }
    if (go && (depth < fp.min_depth || depth < st.depth_mean * fp.scale_min_dist)) go = false;
    if (go) {
      // Point::Update, point.cc:64-100
      const Rigid pose2 = se3_mul(ref, se3_inverse(cur));
      const double tau = compute_tau(pose2, fv, depth, fp.px_error_angle);
      const double tau_inverse = 0.5 * (1.0 / fmax(0.0000001, depth - tau) - 1.0 / (depth + tau));
      const double tau2 = tau_inverse * tau_inverse;
      const double x = 1. / depth;
      const double norm_scale = sqrt(st.sigma2 + tau2);
      if (norm_scale != norm_scale) {
        // std::isnan(norm_scale) -> Update returns with the point untouched (point.cc:76); HasConverged still runs
        const double std_d = sqrt(st.sigma2) / (st.rho * st.rho);
        const double l = 4 * std_d * st.cos_alpha / st.last_distance;
        if (st.fixed || l < 0.1) {
          const double sc = 1.0 / st.rho;
          const V3 pos = st.fixed ? V3{st.position[0], st.position[1], st.position[2]} : se3_apply(ref_world, {sc * fv.x, sc * fv.y, sc * fv.z});
          o.position[0] = pos.x; o.position[1] = pos.y; o.position[2] = pos.z;
          o.cos_alpha = st.cos_alpha; o.last_distance = st.last_distance;
          o.outcome = SDVL_DEPTH_FIXED_STALE;
          if (row) {
            row->P[0] = pos.x; row->P[1] = pos.y; row->P[2] = pos.z;
            row->fixed = 1;
          }
        }
      } else {
        const double s2 = 1. / (1. / st.sigma2 + 1. / tau2);
        const double m = s2 * (st.rho / st.sigma2 + x / tau2);
        double C1 = st.a / (st.a + st.b) * pdf_normal(st.rho, norm_scale, x);
        double C2 = st.b / (st.a + st.b) * 1. / st.z_range;
        const double normalization_constant = C1 + C2;
        C1 /= normalization_constant;
        C2 /= normalization_constant;
        const double f = C1 * (st.a + 1.) / (st.a + st.b + 1.) + C2 * st.a / (st.a + st.b + 1.);
        const double e = C1 * (st.a + 1.) * (st.a + 2.) / ((st.a + st.b + 1.) * (st.a + st.b + 2.)) +
                         C2 * st.a * (st.a + 1.0) / ((st.a + st.b + 1.0) * (st.a + st.b + 2.0));
        const double rho_new = C1 * m + C2 * st.rho;
        o.sigma2 = C1 * (s2 + m * m) + C2 * (st.sigma2 + st.rho * st.rho) - rho_new * rho_new;
        o.rho = rho_new;
        o.a = (e - f) / (f - e / f);
        o.b = o.a * (1.0 - f) / f;
        // Point::GetPosition (point.cc:128-142) of the candidate: the first observation's ray at the new inverse depth
        const double sc = 1.0 / o.rho;
        const V3 pos = st.fixed ? V3{st.position[0], st.position[1], st.position[2]} : se3_apply(ref_world, {sc * fv.x, sc * fv.y, sc * fv.z});
        o.cos_alpha = parallax_of(ref_world.t, cur_world.t, pos);
        o.last_distance = vnorm(vsub(cur_world.t, pos));  // Frame::DistanceTo, frame.h:133-136
        o.n_failed = 0;
        o.position[0] = pos.x; o.position[1] = pos.y; o.position[2] = pos.z;
        o.outcome = SDVL_DEPTH_UPDATED;
        // Point::HasConverged, point.cc:164-178
        const double std_d = sqrt(o.sigma2) / (o.rho * o.rho);
        const double l = 4 * std_d * o.cos_alpha / o.last_distance;
        if (st.fixed || l < 0.1) o.outcome = SDVL_DEPTH_CONVERGED;
        if (row) {
          row->P[0] = pos.x; row->P[1] = pos.y; row->P[2] = pos.z;
          row->idepth = o.rho;
          row->idepth_std = sqrt(o.sigma2);  // Point::GetStd
          row->n_failed = 0;
          if (o.outcome == SDVL_DEPTH_CONVERGED) row->fixed = 1;
        }
      }
    }
  }
  out[i] = o;
  if (out_host) out_host[i] = o;
}

}  // namespace

int sdvl_search_launch_device(sdvl_ctx *ctx, int n_slots, const SearchReqDev *d_reqs, const SearchFramePose *d_table,
                              const SearchBlock *d_blocks, int n_blocks, const sdvl_camera *cam, const sdvl_search_params *p,
                              SearchPrep *d_prep, sdvl_search_res *d_res, sdvl_search_res *h_res, bool prepared) {
  SDVL_REQUIRE(ctx, p->patch_size == 8, "only patch_size 8 is supported (one wave64 per 8x8 patch)");
  SDVL_REQUIRE(ctx, p->max_fast_levels >= 1 && p->max_fast_levels <= 4, "bad max_fast_levels");
  SDVL_REQUIRE(ctx, p->max_align_its >= 0 && p->margin >= 4, "bad max_align_its / margin");
  if (n_slots <= 0 || n_blocks <= 0) return SDVL_OK;
  Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  // `prepared`: the kernel that wrote the requests has written their prepare records too (track_project_kernel)
  if (!prepared) SDVL_LAUNCH(ctx, "search_prepare", search_prepare_kernel, dim3((n_slots + 63) / 64), dim3(64), d_reqs, d_table, n_slots, c, *p, d_prep);
  // device-built batches search frames that came out of sdvl_detect_corners: binned (a frame without bins would still be
  // searched correctly, its corner list read from HBM)
  SDVL_LAUNCH(ctx, "search_points", (search_points_kernel<false, 1>), dim3(static_cast<unsigned>((n_blocks + 7) / 8 * 8 * kWavesPerBlock)), dim3(64), d_reqs,
              d_table, d_blocks, static_cast<const SearchPrep *>(d_prep), c, *p, n_blocks, d_res, h_res);
#ifdef SDVL_SEARCH_STATS
  {
    static int launches = 0;
    if (++launches % 200 == 0) {
      unsigned long long h[8];
      if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_search_stats), sizeof(h)) == hipSuccess)
        fprintf(stderr, "search stats after %d launches: requests %llu, binned %llu, full scans %llu (frame binned %llu, line ok %llu), scan rounds %llu, corners in range %llu\n",
                launches, h[0], h[1], h[2], h[6], h[7], h[3], h[4]);
    }
  }
#endif
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

int sdvl_select_matches_launch(sdvl_ctx *ctx, int n_frames, const ChainFrameDev *d_frames, const int32_t *d_cand_req,
                               const int32_t *d_cand_first, const sdvl_search_res *d_res, const double *d_req_point,
                               const sdvl_camera *cam, PoseJobDev *d_jobs, sdvl_pose_obs *d_obs, int32_t *d_nobs, int32_t *d_match_cand) {
  if (n_frames <= 0) return SDVL_OK;
  Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  SDVL_LAUNCH(ctx, "select_matches", select_matches_kernel, dim3(n_frames), dim3(256), d_frames, d_cand_req, d_cand_first, d_res, d_req_point, c,
              d_jobs, d_obs, d_nobs, d_match_cand);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

extern "C" {

// ---- request batches ------------------------------------------------------------------------------------------------
// A batch lives in the context's pinned staging area: requests | workgroup table | frame table.  sdvl_search_begin
// reserves it, sdvl_search_slot names the (frame, pose) pairs, the caller (or sdvl_search_points) writes the requests in
// their device format, sdvl_search_run adds the tables and launches.
static_assert(sizeof(sdvl_search_req_packed) == sizeof(SearchReqDev), "public and device request records are one layout");

struct SearchBatch {
  void *hs = nullptr, *dsx = nullptr;
  int cap = 0;
  size_t in_bytes = 0, blk_cap_bytes = 0;
  std::vector<SearchFramePose> table;
  std::vector<const sdvl_frame *> frames;  // parallel to `table`
  std::unordered_map<const sdvl_frame *, int> where;
  int last = -1, last2 = -1;  // the two most recent slots: requests alternate between their current and reference frame
};
constexpr size_t kSearchTabCap = 2048;  // frame-table entries that fit the staging reserve (more go through d_work)

static SearchBatch &batch_of(sdvl_ctx *ctx) {
  static thread_local std::unordered_map<sdvl_ctx *, SearchBatch> batches;  // one builder per context
  return batches[ctx];
}

int sdvl_search_begin(sdvl_ctx *ctx, int max_requests, sdvl_search_req_packed **reqs) {
  if (!ctx || !reqs || max_requests < 0) return SDVL_ERR_INVALID;
  SearchBatch &B = batch_of(ctx);
  B.cap = max_requests;
  B.table.clear();
  B.frames.clear();
  B.where.clear();
  B.last = -1;
  B.last2 = -1;
  B.in_bytes = (sizeof(SearchReqDev) * static_cast<size_t>(max_requests) + 255) / 256 * 256;
  B.blk_cap_bytes = (sizeof(SearchBlock) * static_cast<size_t>(max_requests) + 255) / 256 * 256;
  // the batch's own pinned + device buffers, not the staging ring: the caller fills the records over time and every
  // sdvl_stream_wait in between (buffer growth inside sdvl_search_run, a staging request that wraps) restarts the ring
  if (ctx->search_busy_gen == ctx->wait_gen) SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));  // the previous batch may still be read
  const size_t need = B.in_bytes + B.blk_cap_bytes + sizeof(SearchFramePose) * kSearchTabCap;
  int rc = sdvl_ensure(ctx, &ctx->h_search, &ctx->h_search_bytes, need, true);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->d_search, &ctx->d_search_bytes, need, false);
  if (rc) return rc;
  B.hs = ctx->h_search;
  B.dsx = ctx->d_search;
  *reqs = static_cast<sdvl_search_req_packed *>(B.hs);
  return SDVL_OK;
}

int sdvl_search_slot(sdvl_ctx *ctx, const sdvl_frame *f, const double *pose) {
  if (!ctx || !f || !pose) return SDVL_ERR_INVALID;
  SearchBatch &B = batch_of(ctx);
  std::vector<SearchFramePose> &table = B.table;
  if (B.last >= 0 && B.frames[B.last] == f && memcmp(table[B.last].pose, pose, sizeof(double) * 7) == 0) return B.last;
  if (B.last2 >= 0 && B.frames[B.last2] == f && memcmp(table[B.last2].pose, pose, sizeof(double) * 7) == 0) {
    std::swap(B.last, B.last2);
    return B.last;
  }
  B.last2 = B.last;
  auto it = B.where.find(f);
  if (it != B.where.end() && memcmp(table[it->second].pose, pose, sizeof(double) * 7) == 0) { B.last = it->second; return B.last; }
  SearchFramePose e;
  fill_frame(&e.f, f);
  memcpy(e.pose, pose, sizeof(double) * 7);
  e.pad_ = 0.0;
  table.push_back(e);
  B.frames.push_back(f);
  B.where[f] = static_cast<int>(table.size()) - 1;  // the same frame under another pose: the newest entry wins the cache
  B.last = static_cast<int>(table.size()) - 1;
  return B.last;
}

// everything of sdvl_search_run up to and including the queued copy of the results into h_out[0 .. n records); the caller
// waits.  extra_d / extra_h: room the caller wants behind the search's own use of d_out / h_out (offsets returned).
static int search_enqueue(sdvl_ctx *ctx, int n, const sdvl_camera *cam, const sdvl_search_params *p, size_t extra_d, size_t extra_h,
                          size_t *extra_d_off, size_t *extra_h_off, const SearchReqDev **d_reqs_out = nullptr,
                          const SearchFramePose **d_table_out = nullptr) {
  SearchBatch &B = batch_of(ctx);
  SDVL_REQUIRE(ctx, B.hs && n <= B.cap, "sdvl_search_run without a matching sdvl_search_begin");
  SDVL_REQUIRE(ctx, p->patch_size == 8, "only patch_size 8 is supported (one wave64 per 8x8 patch)");
  SDVL_REQUIRE(ctx, p->max_fast_levels >= 1 && p->max_fast_levels <= 4, "bad max_fast_levels");
  SDVL_REQUIRE(ctx, p->max_align_its >= 0 && p->margin >= 4, "bad max_align_its / margin");
  const int n_slots = static_cast<int>(B.table.size());
  for (const SearchFramePose &e : B.table) SDVL_REQUIRE(ctx, p->max_fast_levels <= e.f.levels, "max_fast_levels exceeds the pyramid depth");
  SearchReqDev *hreq = static_cast<SearchReqDev *>(B.hs);
  for (int i = 0; i < n; i++) {
    const SearchReqDev &d = hreq[i];
    SDVL_REQUIRE(ctx, d.cur >= 0 && d.cur < n_slots && d.ref >= 0 && d.ref < n_slots, "request names a frame slot outside the batch");
    SDVL_REQUIRE(ctx, d.level >= 0 && d.level < B.table[d.ref].f.levels, "feature level outside the reference pyramid");
    SDVL_REQUIRE(ctx, d.idepth == d.idepth && d.idepth != 0.0, "inverse depth must be finite and non-zero");
  }
  const size_t in_bytes = B.in_bytes, blk_cap_bytes = B.blk_cap_bytes;
  const size_t out_bytes = sizeof(sdvl_search_res) * static_cast<size_t>(n);
  const size_t out_dev_bytes = (out_bytes + 255) / 256 * 256;
  const size_t prep_bytes = (sizeof(SearchPrep) * static_cast<size_t>(n) + 255) / 256 * 256;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, out_dev_bytes + prep_bytes + extra_d, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, out_dev_bytes + extra_h, true);
  if (rc) return rc;
  if (extra_d_off) *extra_d_off = out_dev_bytes + prep_bytes;
  if (extra_h_off) *extra_h_off = out_dev_bytes;
  void *hs = B.hs, *dsx = B.dsx;
  ctx->search_busy_gen = ctx->wait_gen;  // from here on the batch buffers are read by queued copies and kernels
  // workgroups: runs of up to kWavesPerBlock consecutive requests that search the same current frame
  SearchBlock *hblk = reinterpret_cast<SearchBlock *>(static_cast<uint8_t *>(hs) + in_bytes);
  int n_blocks = 0;
  for (int i = 0; i < n;) {
    int cnt = 1;
    while (i + cnt < n && cnt < kWavesPerBlock && hreq[i + cnt].cur == hreq[i].cur) cnt++;
    hblk[n_blocks++] = SearchBlock{i, cnt};
    const sdvl_frame *cf = B.frames[hreq[i].cur];
    SDVL_REQUIRE(ctx, !cf->hdr_stale, "current frame has a new image but no corners (detect or set corners first)");
    i += cnt;
  }
  const std::vector<SearchFramePose> &table = B.table;
  const SearchFramePose *d_table = nullptr;
  if (table.size() <= kSearchTabCap) {
    memcpy(static_cast<uint8_t *>(hs) + in_bytes + blk_cap_bytes, table.data(), sizeof(SearchFramePose) * table.size());
    SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, in_bytes + blk_cap_bytes + sizeof(SearchFramePose) * table.size()));
    d_table = reinterpret_cast<const SearchFramePose *>(static_cast<uint8_t *>(dsx) + in_bytes + blk_cap_bytes);
  } else {  // more distinct (frame, pose) pairs than the staging reserve: the table travels through the work buffer
    SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, in_bytes + blk_cap_bytes));
    rc = sdvl_ensure(ctx, &ctx->d_work, &ctx->d_work_bytes, sizeof(SearchFramePose) * table.size(), false);
    if (rc) return rc;
    SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_work, table.data(), sizeof(SearchFramePose) * table.size(), hipMemcpyHostToDevice, ctx->stream));
    SDVL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // `table` is pageable and reused by the next batch
    d_table = static_cast<const SearchFramePose *>(ctx->d_work);
  }
  if (d_reqs_out) *d_reqs_out = static_cast<const SearchReqDev *>(dsx);
  if (d_table_out) *d_table_out = d_table;
  Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  // d_out: results | per-request records of the scalar phase
  SearchPrep *d_prep = reinterpret_cast<SearchPrep *>(static_cast<uint8_t *>(ctx->d_out) + out_dev_bytes);
  SDVL_LAUNCH(ctx, "search_prepare", search_prepare_kernel, dim3((n + 63) / 64), dim3(64), static_cast<const SearchReqDev *>(dsx), d_table, n, c,
              *p, d_prep);
  bool all_binned = true;
  for (const sdvl_frame *f : B.frames) all_binned = all_binned && f->bins_valid;
  if (all_binned) {
    SDVL_LAUNCH(ctx, "search_points", (search_points_kernel<false, 1>), dim3(static_cast<unsigned>((n_blocks + 7) / 8 * 8 * kWavesPerBlock)), dim3(64),
                static_cast<const SearchReqDev *>(dsx), d_table, reinterpret_cast<const SearchBlock *>(static_cast<uint8_t *>(dsx) + in_bytes),
                static_cast<const SearchPrep *>(d_prep), c, *p, n_blocks, static_cast<sdvl_search_res *>(ctx->d_out),
                static_cast<sdvl_search_res *>(ctx->h_out));
  } else {
    SDVL_LAUNCH(ctx, "search_points", (search_points_kernel<true, kWavesPerBlock>), dim3(static_cast<unsigned>((n_blocks + 7) / 8 * 8)), dim3(64 * kWavesPerBlock),
                static_cast<const SearchReqDev *>(dsx), d_table, reinterpret_cast<const SearchBlock *>(static_cast<uint8_t *>(dsx) + in_bytes),
                static_cast<const SearchPrep *>(d_prep), c, *p, n_blocks, static_cast<sdvl_search_res *>(ctx->d_out),
                static_cast<sdvl_search_res *>(ctx->h_out));
  }
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

int sdvl_search_run(sdvl_ctx *ctx, int n, const sdvl_camera *cam, const sdvl_search_params *p, sdvl_search_res *out) {
  if (!ctx || !cam || !p || n < 0 || (n > 0 && !out)) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  const int rc = search_enqueue(ctx, n, cam, p, 0, 0, nullptr, nullptr);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  memcpy(out, ctx->h_out, sizeof(sdvl_search_res) * static_cast<size_t>(n));
  return SDVL_OK;
}

// ---- sdvl_search_run_chain / sdvl_search_chain_end
// SelectInliers' iteration budget as a function of the supporter count (feature_align.cc:199-207), for every match count
// up to max_size: row s (s + 1 entries) starts at s*(s+1)/2.  Two libm log() per entry, computed once per configuration.
int sdvl_ensure_nits_table(sdvl_ctx *ctx, int npoints_cfg, int max_its, int max_size) {
  if (ctx->d_nits && ctx->nits_points == npoints_cfg && ctx->nits_its == max_its && ctx->nits_max_size >= max_size) return SDVL_OK;
  const size_t entries = static_cast<size_t>(max_size + 1) * (max_size + 2) / 2;
  std::vector<int32_t> &t = ctx->nits_host;
  t.assign(entries, max_its);
  const double sprob = 0.99;
  for (int size = 0; size <= max_size; size++) {
    const int npoints = npoints_cfg < size ? npoints_cfg : size;
    int32_t *row = t.data() + static_cast<size_t>(size) * (size + 1) / 2;
    for (int supporters = 0; supporters <= size; supporters++) {
      int nits = max_its;
      if (size > 0) {
        const double epsilon = 1.0 - (static_cast<double>(supporters) / static_cast<double>(size));
        double tmp = 1.0 - epsilon;
        for (int k = 1; k < npoints; k++) tmp *= tmp;
        if (!(tmp < 1e-5)) {
          const int v = static_cast<int>(std::log(1.0 - sprob) / std::log(1.0 - tmp));
          nits = max_its < v ? max_its : v;
        }
      }
      row[supporters] = nits;
    }
  }
  SDVL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // nobody may still be reading the old table
  if (ctx->d_nits) SDVL_HIP_CHECK(ctx, hipFree(ctx->d_nits));
  ctx->d_nits = nullptr;
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  SDVL_HIP_CHECK(ctx, hipMalloc(&ctx->d_nits, entries * sizeof(int32_t)));
  // on the context's own stream, not the legacy stream (whose implicit synchronisation collides with a capture under way on another
  // thread of the farm: SDVL_STEP_GRAPH=1)
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_nits, t.data(), entries * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  SDVL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->nits_points = npoints_cfg;
  ctx->nits_its = max_its;
  ctx->nits_max_size = max_size;
  return SDVL_OK;
}

int sdvl_search_run_chain(sdvl_ctx *ctx, int n, const sdvl_camera *cam, const sdvl_search_params *p, sdvl_search_res *out, int n_frames,
                          const sdvl_chain_frame *frames, int n_cand, const int32_t *cand_req, const int32_t *cand_first,
                          const double *req_point, int n_rand, const int32_t *rand_raw, const sdvl_pose_params *pp) {
  if (!ctx || !cam || !p || !pp || n <= 0 || !out || n_frames <= 0 || !frames || n_cand < 0 || (n_cand > 0 && (!cand_req || !cand_first)) ||
      !req_point || n_rand < 0 || !rand_raw)
    return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, pp->max_ransac_points >= 1 && pp->max_ransac_points <= 8, "max_ransac_points must be in [1,8]");
  SDVL_REQUIRE(ctx, pp->max_ransac_its >= 1 && pp->max_ransac_its <= 4096 && pp->max_optim_pose_its >= 0, "bad iteration limits");
  ctx->chain_pending = 0;
  int obs_total = 0, max_size = 0;
  for (int f = 0; f < n_frames; f++) {
    const sdvl_chain_frame &c = frames[f];
    SDVL_REQUIRE(ctx, c.cand_begin >= 0 && c.cand_end >= c.cand_begin && c.cand_end <= n_cand, "candidate range out of bounds");
    SDVL_REQUIRE(ctx, c.cand_end - c.cand_begin <= kChainMaxCand, "too many candidates for one tracker");
    SDVL_REQUIRE(ctx, c.max_matches >= 0 && c.max_matches <= 1024, "max_matches outside the device pose stage's range (1024)");
    SDVL_REQUIRE(ctx, c.rand_begin >= 0 && c.rand_begin + pp->max_ransac_its <= n_rand, "rand range out of bounds");
    obs_total += c.max_matches;
    if (c.max_matches > max_size) max_size = c.max_matches;
  }
  for (int k = 0; k < n_cand; k++) {
    SDVL_REQUIRE(ctx, cand_req[k] >= -1 && cand_req[k] < n, "candidate names a request outside the batch");
    SDVL_REQUIRE(ctx, cand_first[k] >= 0 && cand_first[k] <= k, "cand_first must point at or before the candidate");
  }
  for (int k = 0; k < n_rand; k++) SDVL_REQUIRE(ctx, rand_raw[k] >= 0, "rand() values are non-negative");
  int rc = sdvl_ensure_nits_table(ctx, pp->max_ransac_points, pp->max_ransac_its, max_size);
  if (rc) return rc;
  // behind the search's buffers — device: jobs | obs | hypotheses | results | n_obs | lists ; host: results | n_obs | lists
  const size_t jb = (sizeof(PoseJobDev) * n_frames + 255) / 256 * 256, ob = (sizeof(sdvl_pose_obs) * static_cast<size_t>(obs_total) + 255) / 256 * 256;
  const size_t hb = (sdvl_pose_hyp_bytes() * static_cast<size_t>(n_frames) * pp->max_ransac_its + 255) / 256 * 256;
  const size_t rb = (sizeof(sdvl_pose_result) * n_frames + 255) / 256 * 256, nb = (sizeof(int32_t) * n_frames + 255) / 256 * 256;
  const size_t lb = (sizeof(int32_t) * static_cast<size_t>(obs_total) + 255) / 256 * 256;
  size_t d_off = 0, h_off = 0;
  rc = search_enqueue(ctx, n, cam, p, jb + ob + hb + rb + nb + lb, rb + nb + lb, &d_off, &h_off);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_mark_record(ctx, SDVL_MARK_CHAIN, &ctx->chain_ticket));  // the search results are on the host from here on
  // inputs of the selection + pose stage: one staged copy
  const size_t fb = (sizeof(ChainFrameDev) * n_frames + 255) / 256 * 256, cb = (sizeof(int32_t) * static_cast<size_t>(n_cand) + 255) / 256 * 256;
  const size_t pb = (sizeof(double) * 3 * static_cast<size_t>(n) + 255) / 256 * 256, rndb = (sizeof(int32_t) * static_cast<size_t>(n_rand) + 255) / 256 * 256;
  void *hs = nullptr, *dsx = nullptr;
  rc = sdvl_stage_alloc(ctx, fb + 2 * cb + pb + rndb, &hs, &dsx);
  if (rc) return rc;
  uint8_t *h8 = static_cast<uint8_t *>(hs), *d8 = static_cast<uint8_t *>(dsx);
  ChainFrameDev *hf = reinterpret_cast<ChainFrameDev *>(h8);
  int ob_run = 0;
  for (int f = 0; f < n_frames; f++) {
    hf[f].cand_begin = frames[f].cand_begin;
    hf[f].cand_end = frames[f].cand_end;
    hf[f].max_matches = frames[f].max_matches;
    hf[f].rand_begin = frames[f].rand_begin;
    hf[f].obs_begin = ob_run;
    hf[f].pad_ = 0;
    memcpy(hf[f].pose, frames[f].pose, sizeof(double) * 7);
    hf[f].pad2_ = 0.0;
    ob_run += frames[f].max_matches;
  }
  if (n_cand) {
    memcpy(h8 + fb, cand_req, sizeof(int32_t) * static_cast<size_t>(n_cand));
    memcpy(h8 + fb + cb, cand_first, sizeof(int32_t) * static_cast<size_t>(n_cand));
  }
  memcpy(h8 + fb + 2 * cb, req_point, sizeof(double) * 3 * static_cast<size_t>(n));
  if (n_rand) memcpy(h8 + fb + 2 * cb + pb, rand_raw, sizeof(int32_t) * static_cast<size_t>(n_rand));
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, fb + 2 * cb + pb + rndb));
  uint8_t *dx = static_cast<uint8_t *>(ctx->d_out) + d_off;
  PoseJobDev *d_jobs = reinterpret_cast<PoseJobDev *>(dx);
  sdvl_pose_obs *d_obs = reinterpret_cast<sdvl_pose_obs *>(dx + jb);
  void *d_hyp = dx + jb + ob;
  sdvl_pose_result *d_res = reinterpret_cast<sdvl_pose_result *>(dx + jb + ob + hb);
  int32_t *d_nobs = reinterpret_cast<int32_t *>(dx + jb + ob + hb + rb);
  int32_t *d_lists = reinterpret_cast<int32_t *>(dx + jb + ob + hb + rb + nb);
  Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  SDVL_LAUNCH(ctx, "select_matches", select_matches_kernel, dim3(n_frames), dim3(256), reinterpret_cast<const ChainFrameDev *>(d8),
              reinterpret_cast<const int32_t *>(d8 + fb), reinterpret_cast<const int32_t *>(d8 + fb + cb),
              static_cast<const sdvl_search_res *>(ctx->d_out), reinterpret_cast<const double *>(d8 + fb + 2 * cb), c, d_jobs, d_obs, d_nobs,
              static_cast<int32_t *>(nullptr));
  sdvl_pose_params prm = *pp;
  prm.pad_ = 1;  // raw rand() values: the kernel reduces them modulo the match count it finds in the job
  rc = sdvl_pose_enqueue_device(ctx, n_frames, d_jobs, d_obs, reinterpret_cast<const int32_t *>(d8 + fb + 2 * cb + pb),
                                static_cast<const int32_t *>(ctx->d_nits), &prm, d_hyp, d_res, d_lists, max_size, n_frames);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->h_out) + h_off, dx + jb + ob + hb, rb + nb + lb, hipMemcpyDeviceToHost, ctx->stream));
  ctx->chain_pending = n_frames;
  ctx->chain_host_off = h_off;
  ctx->chain_obs_total = obs_total;
  SDVL_HIP_CHECK(ctx, sdvl_mark_wait(ctx, SDVL_MARK_CHAIN, ctx->chain_ticket));
  memcpy(out, ctx->h_out, sizeof(sdvl_search_res) * static_cast<size_t>(n));
  return SDVL_OK;
}

int sdvl_search_chain_end(sdvl_ctx *ctx, int n_frames, sdvl_pose_result *results, int32_t *n_obs, int32_t *lists) {
  if (!ctx || n_frames <= 0 || !results || !n_obs || !lists) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, ctx->chain_pending == n_frames, "sdvl_search_chain_end without a matching sdvl_search_run_chain");
  ctx->chain_pending = 0;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  const size_t rb = (sizeof(sdvl_pose_result) * n_frames + 255) / 256 * 256, nb = (sizeof(int32_t) * n_frames + 255) / 256 * 256;
  const uint8_t *h = static_cast<const uint8_t *>(ctx->h_out) + ctx->chain_host_off;
  memcpy(results, h, sizeof(sdvl_pose_result) * n_frames);
  memcpy(n_obs, h + rb, sizeof(int32_t) * n_frames);
  memcpy(lists, h + rb + nb, sizeof(int32_t) * static_cast<size_t>(ctx->chain_obs_total));
  return SDVL_OK;
}


static int pack_requests(sdvl_ctx *ctx, int n, const sdvl_search_req *reqs) {
  sdvl_search_req_packed *packed = nullptr;
  int rc = sdvl_search_begin(ctx, n, &packed);
  if (rc) return rc;
  SearchReqDev *hreq = reinterpret_cast<SearchReqDev *>(packed);
  for (int i = 0; i < n; i++) {
    const sdvl_search_req &r = reqs[i];
    SDVL_REQUIRE(ctx, r.cur && r.ref, "null frame in search request");
    SearchReqDev &d = hreq[i];
    d.cur = sdvl_search_slot(ctx, r.cur, r.cur_pose);
    d.ref = sdvl_search_slot(ctx, r.ref, r.ref_pose);
    d.level = r.level; d.fixed = r.fixed;
    d.px[0] = r.px[0]; d.px[1] = r.px[1];
    d.bearing[0] = r.bearing[0]; d.bearing[1] = r.bearing[1]; d.bearing[2] = r.bearing[2];
    d.idepth = r.idepth; d.idepth_std = r.idepth_std;
    d.px0[0] = r.px0[0]; d.px0[1] = r.px0[1];
    memcpy(d.desc, r.desc, 32);
  }
  return SDVL_OK;
}

int sdvl_search_points(sdvl_ctx *ctx, int n, const sdvl_search_req *reqs, const sdvl_camera *cam,
                       const sdvl_search_params *p, sdvl_search_res *out) {
  if (!ctx || !cam || !p || n < 0 || (n > 0 && (!reqs || !out))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  const int rc = pack_requests(ctx, n, reqs);
  if (rc) return rc;
  return sdvl_search_run(ctx, n, cam, p, out);
}

// ---- search + depth filter (Map::UpdateCandidates, map.cc:402-498): one submission, one wait
int sdvl_search_run_filter(sdvl_ctx *ctx, int n, const sdvl_camera *cam, const sdvl_search_params *p, const sdvl_depth_state *state,
                           const sdvl_depth_params *fp, sdvl_track_set *set, sdvl_search_res *out, sdvl_depth_out *fout) {
  if (!ctx || !cam || !p || !fp || n < 0 || (n > 0 && (!state || !out || !fout))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  TrackPoint *rows = nullptr;
  int n_rows = 0;
  if (set) {
    int np = 0, nt = 0;
    rows = sdvl_track_points_device(set, &np, &nt);
    n_rows = np * nt;
  }
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, state[i].track_row >= -1 && (state[i].track_row < n_rows || !set), "depth filter: track_row outside the tables");
    SDVL_REQUIRE(ctx, state[i].sigma2 > 0.0 && state[i].z_range > 0.0, "depth filter: sigma2 and z_range must be positive");
  }
  const size_t sb = (sizeof(sdvl_depth_state) * static_cast<size_t>(n) + 255) / 256 * 256;
  const size_t ob = (sizeof(sdvl_depth_out) * static_cast<size_t>(n) + 255) / 256 * 256;
  size_t d_off = 0, h_off = 0;
  const SearchReqDev *d_reqs = nullptr;
  const SearchFramePose *d_table = nullptr;
  int rc = search_enqueue(ctx, n, cam, p, ob, ob, &d_off, &h_off, &d_reqs, &d_table);
  if (rc) return rc;
  void *hs = nullptr, *dsx = nullptr;
  rc = sdvl_stage_alloc(ctx, sb, &hs, &dsx);
  if (rc) return rc;
  memcpy(hs, state, sizeof(sdvl_depth_state) * static_cast<size_t>(n));
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, sb));
  Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  sdvl_depth_out *d_fout = reinterpret_cast<sdvl_depth_out *>(static_cast<uint8_t *>(ctx->d_out) + d_off);
  sdvl_depth_out *h_fout = reinterpret_cast<sdvl_depth_out *>(static_cast<uint8_t *>(ctx->h_out) + h_off);
  SDVL_LAUNCH(ctx, "depth_filter", depth_filter_kernel, dim3((n + 127) / 128), dim3(128), d_reqs, d_table,
              static_cast<const sdvl_search_res *>(ctx->d_out), static_cast<const sdvl_depth_state *>(dsx), n, c, *fp, rows, n_rows, d_fout,
              h_fout);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  memcpy(out, ctx->h_out, sizeof(sdvl_search_res) * static_cast<size_t>(n));
  memcpy(fout, h_fout, sizeof(sdvl_depth_out) * static_cast<size_t>(n));
  return SDVL_OK;
}

int sdvl_search_points_filter(sdvl_ctx *ctx, int n, const sdvl_search_req *reqs, const sdvl_camera *cam, const sdvl_search_params *p,
                              const sdvl_depth_state *state, const sdvl_depth_params *fp, sdvl_track_set *set, sdvl_search_res *out,
                              sdvl_depth_out *fout) {
  if (!ctx || !cam || !p || !fp || n < 0 || (n > 0 && (!reqs || !state || !out || !fout))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  const int rc = pack_requests(ctx, n, reqs);
  if (rc) return rc;
  return sdvl_search_run_filter(ctx, n, cam, p, state, fp, set, out, fout);
}

int sdvl_align_patches(sdvl_ctx *ctx, int n, const sdvl_frame *const *frames, const int32_t *levels, const uint8_t *border,
                       const uint8_t *patch, int max_its, double *uv_io, uint8_t *converged, int32_t *its) {
  if (!ctx || n < 0 || (n > 0 && (!frames || !levels || !border || !patch || !uv_io || !converged))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, max_its >= 0, "bad max_its");
  const size_t jb = (sizeof(PatchJob) * n + 255) / 256 * 256, bb = (static_cast<size_t>(n) * 100 + 255) / 256 * 256;
  const size_t pb = static_cast<size_t>(n) * 64;
  const size_t ob_uv = (sizeof(double) * 2 * n + 255) / 256 * 256, ob_its = (sizeof(int32_t) * n + 255) / 256 * 256;
  const size_t ob = ob_uv + ob_its + n;
  void *hsv = nullptr, *dsv = nullptr;
  int rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, ob, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, ob, true);
  if (!rc) rc = sdvl_stage_alloc(ctx, jb + bb + pb, &hsv, &dsv);
  if (rc) return rc;
  PatchJob *hj = static_cast<PatchJob *>(hsv);
  for (int i = 0; i < n; i++) {
    SDVL_REQUIRE(ctx, frames[i] && levels[i] >= 0 && levels[i] < frames[i]->v.levels, "bad frame / level");
    hj[i].img = frames[i]->v.level[levels[i]];
    hj[i].w = frames[i]->v.lw[levels[i]];
    hj[i].h = frames[i]->v.lh[levels[i]];
    hj[i].u = uv_io[2 * i];
    hj[i].v = uv_io[2 * i + 1];
  }
  uint8_t *hs = static_cast<uint8_t *>(hsv), *ds = static_cast<uint8_t *>(dsv);
  memcpy(hs + jb, border, static_cast<size_t>(n) * 100);
  memcpy(hs + jb + bb, patch, pb);
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, ds, hs, jb + bb + pb));
  uint8_t *dout = static_cast<uint8_t *>(ctx->d_out);
  SDVL_LAUNCH(ctx, "align_patches", align_patches_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(64 * kWavesPerBlock), reinterpret_cast<const PatchJob *>(ds), ds + jb, ds + jb + bb, n, max_its, reinterpret_cast<double *>(dout), dout + ob_uv + ob_its, reinterpret_cast<int32_t *>(dout + ob_uv));
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, ob, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  const uint8_t *ho = static_cast<const uint8_t *>(ctx->h_out);
  memcpy(uv_io, ho, sizeof(double) * 2 * n);
  if (its) memcpy(its, ho + ob_uv, sizeof(int32_t) * n);
  memcpy(converged, ho + ob_uv + ob_its, n);
  return SDVL_OK;
}

}  // extern "C"
