// sdvl_image_align.hip — K5/K6 sparse direct image alignment, ImageAlign::ComputePose (image_align.cc:46-84): the WHOLE coarse-to-fine
// Gauss-Newton of a frame pair on the device (levels max..min x <= max_its iterations), no host round trip per iteration.
//   image_align_pre_kernel / image_align_track_pre_kernel   PrecomputePatches (:208-267) of every level of every job as a wide launch
//                                                           (one 4-wave workgroup per job and level), with the level's factored normal matrix;
//   image_align_wave_pre_kernel / image_align_track_wave_pre_kernel   Optimize (:86-125) + ComputeResiduals (:127-206): lane = feature,
//                                                           one wave per job (four for jobs of more than 384 features and for small batches).
// Float vs double follow the reference statement by statement; the only deviation is the reduction ORDER of H, Jres (double) and chi2
// (float in the reference, accumulated in double here), hence tolerance-class parity (1e-4).
// (Rounds 1-3's forms — a 512-thread workgroup of (feature, pixel) threads with its caches in HBM, then in LDS — and round 4-5's
//  single-launch and LDS-item variants lost their A/Bs and were removed in round 6: profiles/HISTORY.md.)
#include <stdlib.h>

#include <atomic>
#include <vector>

#include "sdvl_internal.h"
#include "sdvl_math.h"
#include "sdvl_search_types.h"

namespace {

using namespace sdvl;

constexpr int kMaxF = SDVL_MAX_ALIGN_FEATURES;
constexpr int kLdsMaxF = 384;  // jobs of up to this many features: one wave per job (three rounds of 64 x 2); beyond: four waves

struct IaJob {
  const uint8_t *ref_level[SDVL_MAX_LEVELS];
  const uint8_t *cur_level[SDVL_MAX_LEVELS];
  int lw[SDVL_MAX_LEVELS], lh[SDVL_MAX_LEVELS];
  int feat_begin, n_feat;
  int out_index, pad_;  // slot of this job's result (jobs are regrouped by size before the launch)
  double T[7];
};

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------ lane = feature
// The Gauss-Newton as a chain of WAVES (round 3's workgroup of (feature, pixel) threads: 231 VGPRs x 4 waves per job, ~6 workgroup
// barriers per evaluation, thread 0 alone in the solve — 123 us per 256 jobs alone, 288 us among the other streams' kernels).  Here:
//  * lane = FEATURE; a wave walks its features in rounds of 64.  A lane reads only what it wrote itself (its feature's reference
//    items, its 3-D point): no barrier and no fence between PrecomputePatches and ComputeResiduals;
//  * the normal equations are factored per feature.  J(item) = (dx * Ja + dy * Jb) * fl with Ja, Jb the two rows of the feature's
//    2x6 Jacobian (image_align.cc:263), so  sum_items J res = fl * (Ja * A + Jb * B)  with A = sum dx res, B = sum dy res over the
//    feature's 16 pixels, and  sum_items J J^T = G^T S G  with G = fl * [Ja; Jb] and S = [Sxx Sxy; Sxy Syy] the feature's gradient
//    sums.  An evaluation costs 16 x (bilinear + 2 products) + 24 double operations per feature instead of 16 x 66; H costs 111 per
//    feature, and only when the contributing set changes.  Same mathematics, different rounding order: tolerance class (pose 1e-4;
//    every form of this kernel has summed in its own order);
//  * the 6x6 interpolated grid of the reference window is computed once and shared by the 16 pixels' value / dx / dy (the
//    reference evaluates the same expression five times per pixel: bit-identical items from 32 bilinear sums instead of 80);
//  * image windows come in as aligned dwords + v_alignbyte (10 loads per 5x5 window instead of 25 byte loads);
//  * the solve, the SE3 update and the termination tests run redundantly on every lane (uniform values): no broadcast, no
//    barrier, no idle lanes to wait for; the LDLT factor stays in LDS while H is reused;
//  * kWaves = 1 (tracking: <= 192..384 features): no s_barrier anywhere.  kWaves = 4 for the large jobs of configuration C: one
//    barrier per evaluation (per-wave partial sums through LDS, summed in wave order by everybody).
// Cites: ImageAlign::ComputePose image_align.cc:46-84, Optimize :86-125, ComputeResiduals :127-206, PrecomputePatches :208-267.
template <int N>
__device__ __forceinline__ double wave_reduce_n(double *v, int lane) {
  // halving butterfly: on return a lane holds the wave-wide sum of value index (lane >> (6 - log2 N)) & (N - 1)
  int n = N;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    if (n > 1) {
      const int h = n >> 1;
      const bool up = lane & m;
#pragma unroll
      for (int i = 0; i < h; i++) {
        const double keep = up ? v[i + h] : v[i], send = up ? v[i] : v[i + h];
        v[i] = keep + shfl_xor_f64(send, m);
      }
      n = h;
    } else {
      v[0] += shfl_xor_f64(v[0], m);
    }
  }
  return v[0];
}

__device__ __forceinline__ void ia_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// bytes [0, 8) of the row that starts at p (any alignment), from aligned dwords; `need` = how many of them the caller uses
__device__ __forceinline__ void ia_load_row8(const uint8_t *p, int need, uint32_t *lo, uint32_t *hi) {
  const uint32_t s = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)) & 3u;
  const uint32_t *q = reinterpret_cast<const uint32_t *>(p - s);
  const uint32_t d0 = q[0], d1 = q[1];
  // the third dword only where the row reaches into it: never touch memory behind the bytes the reference reads
  const uint32_t d2 = q[(s + need > 8) ? 2 : 1];
  *lo = __builtin_amdgcn_alignbyte(d1, d0, s);
  *hi = __builtin_amdgcn_alignbyte(d2, d1, s);
}
__device__ __forceinline__ float ia_byte(uint32_t w, int k) { return static_cast<float>((w >> (8 * k)) & 0xffu); }

// Rigid::Exp with the kernel's own sin / cos (sincos_2pi: fdlibm kernels, <= 1 ulp like libm's; no large-argument path — a
// Gauss-Newton step of more than a turn is folded into [0, 2 pi), the result is rejected by the chi2 test anyway)
__device__ __forceinline__ void ia_sincos(double x, double *s, double *c) {
  if (!(x <= 6.28)) x = x - 6.283185307179586 * floor(x * 0.15915494309189535);
  if (!(x >= 0.0 && x <= 6.3)) x = 0.0;  // NaN
  sincos_2pi(x, s, c);
}
__device__ __forceinline__ Rigid ia_se3_exp(const double *u) {
  const double kEps = 1e-10;
  const V3 ups = {u[0], u[1], u[2]};
  const V3 om = {u[3], u[4], u[5]};
  const double theta = vnorm(om);
  const double half_theta = 0.5 * theta;
  double imag, real, sin_half;
  ia_sincos(half_theta, &sin_half, &real);
  if (theta < kEps) {
    const double t2 = theta * theta;
    const double t4 = t2 * t2;
    imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4;
  } else {
    imag = sin_half / theta;
  }
  Rigid r;
  r.q0 = real; r.q1 = imag * om.x; r.q2 = imag * om.y; r.q3 = imag * om.z;
  M3 Om;
  Om.m[0] = 0;     Om.m[1] = -om.z; Om.m[2] = om.y;
  Om.m[3] = om.z;  Om.m[4] = 0;     Om.m[5] = -om.x;
  Om.m[6] = -om.y; Om.m[7] = om.x;  Om.m[8] = 0;
  const M3 Om2 = mmul(Om, Om);
  M3 V;
  if (theta < kEps) {
    V = quat_to_mat(r.q0, r.q1, r.q2, r.q3);
  } else {
    const double t2 = theta * theta;
    double sin_theta, cos_theta;
    ia_sincos(theta, &sin_theta, &cos_theta);
    const double ca = (1 - cos_theta) / (t2);
    const double cb = (theta - sin_theta) / (t2 * theta);
#pragma unroll
    for (int i = 0; i < 9; i++) V.m[i] = (((i % 4) == 0 ? 1.0 : 0.0) + ca * Om.m[i]) + cb * Om2.m[i];
  }
  r.t = mvec(V, ups);
  return r;
}

constexpr int kIaVis = 2, kIaOk = 4;  // feature flags in LDS: bit 0 valid, bit 1 visible (sticky, image_align.cc:233), bit 2 inside at the last evaluation

// -DSDVL_IA_STAMPS: diagnostic build, phase times of job 0 in shader-clock ticks (s_memtime) -> sdvl_debug_ia_stamps
#ifdef SDVL_IA_STAMPS
__device__ unsigned long long g_ia_stamps[8];
#define IA_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[k] += now_ - t_last_; t_last_ = now_; } while (0)
#else
#define IA_STAMP(k) do { } while (0)
#endif

// what the body needs of one job, wherever the job's record lives (IaJob of the C-ABI calls, TrackJobDev of a tracked step)
struct IaIn {
  const uint8_t *const *ref_level;  // [levels] frame 1
  const uint8_t *const *cur_level;  // [levels] frame 2
  const int *lw, *lh;
  int nf;
  const double *T0;     // [7] start of the alignment
  sdvl_align_result *out;
  // what the precompute kernel left for this job, one block per level (index 0 = max_level): items [48][pitch] floats,
  // gradient sums [3][pitch] doubles, visibility bytes [pitch], and the factored normal matrix of the level's visible set
  const float *pre_items = nullptr;
  const double *pre_S = nullptr;
  const uint8_t *pre_vis = nullptr;
  const double *pre_fac = nullptr;  // [32] per level: L[21] | tr[6] | n_visible
  int pre_pitch = 0;
};

// feature source A: the records of the C-ABI calls (sdvl_align_feature, include/sdvl_hip.h)
struct IaRecordFeats {
  const sdvl_align_feature *F;
  __device__ __forceinline__ bool load(int f, V3 *xyz) const {
    const sdvl_align_feature ft = F[f];
    *xyz = vscale({ft.fx, ft.fy, ft.fz}, ft.depth);
    return ft.valid != 0;
  }
  __device__ __forceinline__ void pos(int f, double *px, double *py) const { *px = F[f].px; *py = F[f].py; }
  // the same in two steps for the precompute kernel (see IaTableFeats::fetch)
  struct Raw { double fx, fy, fz, depth; int valid; };
  __device__ __forceinline__ void fetch(int f, Raw *r, double *px, double *py) const {
    const sdvl_align_feature ft = F[f];
    *px = ft.px;
    *py = ft.py;
    *r = Raw{ft.fx, ft.fy, ft.fz, ft.depth, ft.valid};
  }
  __device__ __forceinline__ bool valid(const Raw &r) const { return r.valid != 0; }
  __device__ __forceinline__ V3 point(const Raw &r) const { return vscale({r.fx, r.fy, r.fz}, r.depth); }
};

// feature source B: the rows of the device-resident tracking tables — what track_align_prep_kernel wrote into records until
// round 3 (image_align.cc:147-160,219-236: position, bearing, validity = has a point that is not deleted, depth = |P - C1|)
struct IaTableFeats {
  const TrackFeat *F;
  const TrackPoint *P;
  V3 first_pos;  // Frame::GetWorldPosition() of last_frame
  __device__ __forceinline__ bool load(int f, V3 *xyz) const {
    const TrackFeat ft = F[f];
    const int pt = ft.point < 0 ? -1 : (ft.point & kPointMask);
    // status and position of the point's row in one round trip (row 0 when there is no point: read and ignored)
    const TrackPoint &row = P[pt < 0 ? 0 : pt];
    const int status = row.status;
    const double p0 = row.P[0], p1 = row.P[1], p2 = row.P[2];
    const bool valid = pt >= 0 && !(status & kDeleted);
    double depth = 0.0;
    if (valid) {
      const double dx = p0 - first_pos.x, dy = p1 - first_pos.y, dz = p2 - first_pos.z;
      depth = sqrt(dx * dx + dy * dy + dz * dz);
    }
    *xyz = vscale({ft.bearing[0], ft.bearing[1], ft.bearing[2]}, depth);
    return valid;
  }
  __device__ __forceinline__ void pos(int f, double *px, double *py) const { *px = F[f].px[0]; *py = F[f].px[1]; }
  // Round 6, for the precompute kernel: the row, and behind it the point's status AND position ASKED FOR (row 0 when the feature has no
  // point: read and ignored) but not looked at — the caller issues the reference window's loads (which need the row's pixel only) before
  // valid() / point() wait for the point's row.  Row -> status -> position -> pixel -> window rows one by one were eight dependent
  // memory round trips per feature in a kernel that does little else; this way there are two.
  struct Raw { double b0, b1, b2, p0, p1, p2; int pt, status; };
  __device__ __forceinline__ void fetch(int f, Raw *r, double *px, double *py) const {
    const TrackFeat ft = F[f];
    *px = ft.px[0];
    *py = ft.px[1];
    const int pt = ft.point < 0 ? -1 : (ft.point & kPointMask);
    const TrackPoint &row = P[pt < 0 ? 0 : pt];
    *r = Raw{ft.bearing[0], ft.bearing[1], ft.bearing[2], row.P[0], row.P[1], row.P[2], pt, row.status};
    asm volatile("" ::: "memory");  // the point's loads are issued here, not sunk to where valid() / point() look at them
  }
  __device__ __forceinline__ bool valid(const Raw &r) const { return r.pt >= 0 && !(r.status & kDeleted); }
  __device__ __forceinline__ V3 point(const Raw &r) const {  // (of a valid feature)
    const double dx = r.p0 - first_pos.x, dy = r.p1 - first_pos.y, dz = r.p2 - first_pos.z;
    return vscale({r.b0, r.b1, r.b2}, sqrt(dx * dx + dy * dy + dz * dz));
  }
};

// PrecomputePatches of ONE feature at one level (image_align.cc:208-267): border test, the 16 reference items {patch, dx, dy} and
// the gradient sums Sxx, Sxy, Syy.  Returns whether the feature is visible at the level; writes nothing otherwise.
template <class Valid>
__device__ __forceinline__ bool ia_precompute_feature(const uint8_t *ref_img, int W, int H, float scale, double fpx, double fpy, const Valid valid, int f,
                                                      int pitch, float2 *it_pd, float *it_dy, double *sums3) {
  const float u_ref = static_cast<float>(fpx * scale);
  const float v_ref = static_cast<float>(fpy * scale);
  const int ui = static_cast<int>(floorf(u_ref)), vi = static_cast<int>(floorf(v_ref));
  const int border = 3;
  if (ui - border < 0 || vi - border < 0 || ui + border >= W || vi + border >= H) return false;  // (`valid`: below, behind the window's loads)
  const float su = u_ref - ui, sv = v_ref - vi;
  const float w_tl = static_cast<float>((1.0 - su) * (1.0 - sv));
  const float w_tr = static_cast<float>(su * (1.0 - sv));
  const float w_bl = static_cast<float>((1.0 - su) * sv);
  const float w_br = su * sv;
  // g[y][x] = the bilinear sum whose top-left pixel is (ui - 3 + x, vi - 3 + y): patch value of pixel (px, py) = g[py+1][px+1],
  // dx = 0.5 (g[py+1][px+2] - g[py+1][px]), dy = 0.5 (g[py+2][px+1] - g[py][px+1]) — the reference's own five expressions
  const uint8_t *wp = ref_img + static_cast<size_t>(vi - 3) * W + (ui - 3);
  float g[3][6];  // rolling: rows y-2, y-1, y of the grid
  float ra[7], rb[7];
  // all seven rows of the window are asked for before the first is used (round 6: row by row, every row's wait also waited for the
  // item stores queued in front of it — a memory round trip per row on a chain that is nothing but round trips)
  uint32_t wlo[7], whi[7];
#pragma unroll
  for (int y = 0; y < 7; y++) ia_load_row8(wp + static_cast<size_t>(y) * W, 7, &wlo[y], &whi[y]);
  // a feature without a live point is not visible either (image_align.cc:219-221); the test sits here so that the window's loads do not
  // wait for the point's row, on which `valid` depends (the window of such a feature is read and dropped)
  asm volatile("" ::: "memory");  // (the loads above stay above: the compiler otherwise tests `valid` first and the window waits for the point's row)
  if (!valid()) return false;
  {
#pragma unroll
    for (int k = 0; k < 4; k++) ra[k] = ia_byte(wlo[0], k);
#pragma unroll
    for (int k = 4; k < 7; k++) ra[k] = ia_byte(whi[0], k - 4);
  }
  double sxx = 0.0, sxy = 0.0, syy = 0.0;
#pragma unroll
  for (int y = 0; y < 6; y++) {
    const uint32_t lo = wlo[y + 1], hi = whi[y + 1];
#pragma unroll
    for (int k = 0; k < 4; k++) rb[k] = ia_byte(lo, k);
#pragma unroll
    for (int k = 4; k < 7; k++) rb[k] = ia_byte(hi, k - 4);
#pragma unroll
    for (int x = 0; x < 6; x++) g[y % 3][x] = w_tl * ra[x] + w_tr * ra[x + 1] + w_bl * rb[x] + w_br * rb[x + 1];
#pragma unroll
    for (int k = 0; k < 7; k++) ra[k] = rb[k];
    if (y >= 2) {  // grid rows y-2, y-1, y are there: patch row py = y - 2
      const int py = y - 2;
#pragma unroll
      for (int px = 0; px < 4; px++) {
        const float patch = g[(y - 1) % 3][px + 1];
        const float dx = 0.5f * (g[(y - 1) % 3][px + 2] - g[(y - 1) % 3][px]);
        const float dy = 0.5f * (g[y % 3][px + 1] - g[(y - 2) % 3][px + 1]);
        it_pd[(py * 4 + px) * pitch + f] = make_float2(patch, dx);
        it_dy[(py * 4 + px) * pitch + f] = dy;
        const double ddx = dx, ddy = dy;
        sxx += ddx * ddx;
        sxy += ddx * ddy;
        syy += ddy * ddy;
      }
    }
  }
  sums3[0] = sxx;
  sums3[1] = sxy;
  sums3[2] = syy;
  return true;
}

// the 21 entries of G^T S G of one feature (see the head of this section), added to h16 / h8
__device__ __forceinline__ void ia_feature_h(double X, double Y, double z_inv, double fl, double sxx, double sxy, double syy, double *h16, double *h8) {
  const double z_inv_2 = z_inv * z_inv;
  double ga[6], gb[6];
  {
    const double j2 = X * z_inv_2, j3 = Y * j2, j8 = Y * z_inv_2;
    ga[0] = -z_inv * fl; ga[1] = 0.0 * fl; ga[2] = j2 * fl; ga[3] = j3 * fl; ga[4] = -(1.0 + X * j2) * fl; ga[5] = (Y * z_inv) * fl;
    gb[0] = 0.0 * fl; gb[1] = -z_inv * fl; gb[2] = j8 * fl; gb[3] = (1.0 + Y * j8) * fl; gb[4] = -j3 * fl; gb[5] = (-X * z_inv) * fl;
  }
  double ma[6], mb[6];  // M = S G
#pragma unroll
  for (int c = 0; c < 6; c++) {
    ma[c] = sxx * ga[c] + sxy * gb[c];
    mb[c] = sxy * ga[c] + syy * gb[c];
  }
  int k = 0;
#pragma unroll
  for (int rr = 0; rr < 6; rr++)
#pragma unroll
    for (int c = rr; c < 6; c++) {
      const double v = ga[rr] * ma[c] + gb[rr] * mb[c];
      if (k < 16) h16[k] += v; else h8[k - 16] += v;
      k++;
    }
}

template <int kWaves, class Feats>
__device__ __forceinline__ void ia_wave_body(const IaIn job, const Feats F, const Cam cam, const sdvl_align_params prm, const int max_f, uint8_t *s_dyn) {
  // carve (max_f is a multiple of 64 * kWaves): x[4][max_f] doubles (point in frame 1: x, y, z, 1/z) | flags[max_f] bytes.  The
  // reference items {patch, dx}[16][pitch], dy[16][pitch] and the gradient sums of a level come from the precompute kernel's
  // output in the work buffer (L2-resident: 36 KB per job and level, every load coalesced over the wave's 64 features)
  double *s_x = reinterpret_cast<double *>(s_dyn);
  uint8_t *s_flag = reinterpret_cast<uint8_t *>(s_x + static_cast<size_t>(4) * max_f);
  __shared__ double s_red[2][kWaves][8];
  __shared__ int s_chg[2][kWaves];
  __shared__ double s_redH[kWaves][24];
  __shared__ double s_L[kWaves][21];  // the factor of H while H is reused: row-major lower triangle incl. D on the diagonal
  __shared__ int s_tr[kWaves][6];
  // the optimiser's pose and its roll-back copy (image_align.cc:95,112) — uniform values, kept out of the vector registers
  __shared__ double s_T[kWaves][7], s_Tbk[kWaves][7];

  const int nf = job.nf;
  const int tid = threadIdx.x, lane = tid & 63, wave = kWaves > 1 ? (tid >> 6) : 0;
  const int rounds = (nf + 64 * kWaves - 1) / (64 * kWaves);
  const int pitch = job.pre_pitch;
  const float2 *it_pd = nullptr;
  const float *it_dy = nullptr;
  const double *S3 = nullptr;  // gradient sums of the level

  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 7; q++) s_T[wave][q] = job.T0[q];
  }
#ifdef SDVL_IA_STAMPS
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last_ = __builtin_amdgcn_s_memtime();
#endif
  double chi2 = 1e10, error = 1e10;
  bool stop = false;
  int n_meas = 0, iters_run = 0;
  int its0 = 0, its1 = 0, its2 = 0, its3 = 0, its4 = 0, its5 = 0, its6 = 0, its7 = 0;

  for (int r = 0; r < rounds; r++) {
    const int f = (r * kWaves + wave) * 64 + lane;
    if (f < nf) {
      V3 xyz;
      const bool valid = F.load(f, &xyz);
      s_x[f] = xyz.x;
      s_x[max_f + f] = xyz.y;
      s_x[2 * max_f + f] = xyz.z;
      s_x[3 * max_f + f] = 1. / xyz.z;  // z_inv of Jacobian3DToPlane, extra/utils.cc:103
      s_flag[f] = valid ? 1 : 0;
    }
  }
  int eval = 0;  // parity of the partial-sum buffers
  IA_STAMP(0);

  for (int level = prm.max_level; level >= prm.min_level; level--) {
    const int W = job.lw[level], H = job.lh[level];
    const uint8_t *cur_img = job.cur_level[level];
    const float scale = 1.0f / (1 << level);
    const double fl = cam.fx / (1 << level);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 7; q++) s_Tbk[wave][q] = s_T[wave][q];
    }
    // ---- PrecomputePatches(level), image_align.cc:208-267: done by the precompute kernel for every level at once (the patches do
    // not depend on the pose): this level's block
    {
      const int li = prm.max_level - level;
      it_pd = reinterpret_cast<const float2 *>(job.pre_items + static_cast<size_t>(li) * 48 * pitch);
      it_dy = job.pre_items + static_cast<size_t>(li) * 48 * pitch + static_cast<size_t>(32) * pitch;
      S3 = job.pre_S + static_cast<size_t>(li) * 3 * pitch;
      const uint8_t *vis = job.pre_vis + static_cast<size_t>(li) * pitch;
      for (int r = 0; r < rounds; r++) {
        const int f = (r * kWaves + wave) * 64 + lane;
        if (f >= nf) continue;
        // visible at this level: assumed inside the current image too (kIaOk) — that is the set the precomputed H belongs to
        s_flag[f] = static_cast<uint8_t>((s_flag[f] & 1) | (vis[f] ? (kIaVis | kIaOk) : 0));
      }
      const double *fac = job.pre_fac + static_cast<size_t>(li) * 32;
      if (lane < 21) s_L[wave][lane] = fac[lane];
      if (lane >= 32 && lane < 38) s_tr[wave][lane - 32] = static_cast<int>(fac[21 + lane - 32]);
    }

    IA_STAMP(1);
    for (int it = 0; it < prm.max_its; it++) {
      const int b = eval & 1;
      eval++;
      ia_wave_fence();
      M3 R;
      V3 Tt;
      {
        const Rigid T = se3_from7(s_T[wave]);
        R = se3_rot(T);
        Tt = T.t;
      }
      // ---- ComputeResiduals, image_align.cc:127-206
      double acc[8];
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = 0.0;
      bool chg = false;
      for (int r = 0; r < rounds; r++) {
        const int f = (r * kWaves + wave) * 64 + lane;
        if (f >= nf) continue;
        const int flag = s_flag[f];
        if (!(flag & kIaVis)) continue;
        // the feature's 48 reference items do not depend on the pose: their loads go out first and are in flight under the projection
        // and the window's loads (round 6: the compiler had split them into three batches around the window's rows — three memory
        // round trips per round of a wave that has nothing else to do meanwhile)
        float2 pdv[16];
        float dyv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
          pdv[k] = it_pd[k * pitch + f];
          dyv[k] = it_dy[k * pitch + f];
        }
        const double X = s_x[f], Y = s_x[max_f + f], Z = s_x[2 * max_f + f];
        const V3 xc = {R.m[0] * X + R.m[1] * Y + R.m[2] * Z + Tt.x, R.m[3] * X + R.m[4] * Y + R.m[5] * Z + Tt.y,
                       R.m[6] * X + R.m[7] * Y + R.m[8] * Z + Tt.z};
        const V2 pr = cam_project(cam, xc);
        const float u_cur = static_cast<float>(pr.x * scale);
        const float v_cur = static_cast<float>(pr.y * scale);
        const float fu = floorf(u_cur), fv = floorf(v_cur);
        // (int)floorf of NaN / huge values is undefined on the CPU; treat anything outside the image as a miss
        const bool ok = fu >= 3.f && fv >= 3.f && fu < static_cast<float>(W - 3) && fv < static_cast<float>(H - 3);
        if (ok != ((flag & kIaOk) != 0)) {
          chg = true;
          s_flag[f] = static_cast<uint8_t>(ok ? (flag | kIaOk) : (flag & ~kIaOk));
        }
        if (!ok) continue;
        const int ui = static_cast<int>(fu), vi = static_cast<int>(fv);
        const float su = u_cur - ui, sv = v_cur - vi;
        const float w0 = static_cast<float>((1.0 - su) * (1.0 - sv));
        const float w1 = static_cast<float>(su * (1.0 - sv));
        const float w2 = static_cast<float>((1.0 - su) * sv);
        const float w3 = su * sv;
        const uint8_t *wp = cur_img + static_cast<size_t>(vi - 2) * W + (ui - 2);
        float ra[5], rb[5];
        uint32_t wlo[5], whi[5];
#pragma unroll
        for (int y = 0; y < 5; y++) ia_load_row8(wp + static_cast<size_t>(y) * W, 5, &wlo[y], &whi[y]);
        {
#pragma unroll
          for (int k = 0; k < 4; k++) ra[k] = ia_byte(wlo[0], k);
          ra[4] = ia_byte(whi[0], 0);
        }
        double A = 0.0, B = 0.0, c2 = 0.0;
#pragma unroll
        for (int y = 0; y < 4; y++) {
          const uint32_t lo = wlo[y + 1], hi = whi[y + 1];
#pragma unroll
          for (int k = 0; k < 4; k++) rb[k] = ia_byte(lo, k);
          rb[4] = ia_byte(hi, 0);
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const float intensity = w0 * ra[x] + w1 * ra[x + 1] + w2 * rb[x] + w3 * rb[x + 1];
            const float2 pd = pdv[y * 4 + x];
            const float dy = dyv[y * 4 + x];
            const float res = intensity - pd.x;
            const double dres = res;
            A += static_cast<double>(pd.y) * dres;
            B += static_cast<double>(dy) * dres;
            c2 += static_cast<double>(res * res);
          }
#pragma unroll
          for (int k = 0; k < 5; k++) ra[k] = rb[k];
        }
        double fj[12];
        {  // Jacobian3DToPlane with the stored 1/z, extra/utils.cc:99-118
          const double z_inv = s_x[3 * max_f + f], z_inv_2 = z_inv * z_inv;
          fj[0] = -z_inv; fj[1] = 0.0; fj[2] = X * z_inv_2; fj[3] = Y * fj[2]; fj[4] = -(1.0 + X * fj[2]); fj[5] = Y * z_inv;
          fj[6] = 0.0; fj[7] = -z_inv; fj[8] = Y * z_inv_2; fj[9] = 1.0 + Y * fj[8]; fj[10] = -fj[3]; fj[11] = -X * z_inv;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) acc[c] -= (A * fj[c] + B * fj[6 + c]) * fl;
        acc[6] += c2;
        acc[7] += 16.0;
      }
      IA_STAMP(2);
      {
        const bool any_chg = __any(chg ? 1 : 0) != 0;
        const double tot = wave_reduce_n<8>(acc, lane);
        if ((lane & 7) == 0) s_red[b][wave][(lane >> 3) & 7] = tot;
        if (lane == 0) s_chg[b][wave] = any_chg ? 1 : 0;
      }
      if (kWaves > 1) __syncthreads(); else ia_wave_fence();
      double Jres[6], sum_c2 = 0.0, sum_n = 0.0;
      bool changed = false;
      {
        double s8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) s8[i] = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
#pragma unroll
          for (int i = 0; i < 8; i++) s8[i] += s_red[b][w][i];
          changed = changed || s_chg[b][w] != 0;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) Jres[c] = s8[c];
        sum_c2 = s8[6];
        sum_n = s8[7];
      }
      const bool rebuild_h = changed;  // the level starts with the precomputed factor of its visible set
      IA_STAMP(3);
      if (rebuild_h) {
        // H = sum over the contributing features of G^T S G (see the head of this section); it changes only with that set
        double h16[16], h8[8];
#pragma unroll
        for (int i = 0; i < 16; i++) h16[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; i++) h8[i] = 0.0;
        for (int r = 0; r < rounds; r++) {
          const int f = (r * kWaves + wave) * 64 + lane;
          if (f >= nf) continue;
          if (!(s_flag[f] & kIaOk) || !(s_flag[f] & kIaVis)) continue;
          ia_feature_h(s_x[f], s_x[max_f + f], s_x[3 * max_f + f], fl, S3[f], S3[pitch + f], S3[2 * pitch + f], h16, h8);
        }
        const double t16 = wave_reduce_n<16>(h16, lane);
        const double t8 = wave_reduce_n<8>(h8, lane);
        if ((lane & 3) == 0) s_redH[wave][(lane >> 2) & 15] = t16;
        if ((lane & 7) == 0) s_redH[wave][16 + ((lane >> 3) & 7)] = t8;
        if (kWaves > 1) __syncthreads(); else ia_wave_fence();
        double Hm[36];
        {
          int k = 0;
#pragma unroll
          for (int rr = 0; rr < 6; rr++)
#pragma unroll
            for (int c = rr; c < 6; c++) {
              double s = 0.0;
#pragma unroll
              for (int w = 0; w < kWaves; w++) s += s_redH[w][k];
              Hm[6 * rr + c] = s;
              Hm[6 * c + rr] = s;
              k++;
            }
        }
        double La[36];
        int tr[6];
        ldlt_factor6_reg<true>(Hm, La, tr);
        if (lane == 0) {
          int k = 0;
#pragma unroll
          for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) s_L[wave][k++] = La[6 * i + j];
#pragma unroll
          for (int q = 0; q < 6; q++) s_tr[wave][q] = tr[q];
        }
        ia_wave_fence();
      }
      IA_STAMP(4);
      // ---- Optimize body, image_align.cc:93-124 (every lane, uniform values)
      double xs[6];
      {
        double La[36];
        int tr[6];
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j < 6; j++) La[6 * i + j] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j <= i; j++) La[6 * i + j] = s_L[wave][k++];
#pragma unroll
        for (int q = 0; q < 6; q++) tr[q] = s_tr[wave][q];
        ldlt_apply6_reg(La, tr, Jres, xs);
      }
      n_meas = static_cast<int>(sum_n);
      iters_run++;
      const double new_chi2 = static_cast<double>(static_cast<float>(sum_c2) / static_cast<float>(n_meas));
      if (n_meas == 0) stop = true;
      if (xs[0] != xs[0]) stop = true;
      bool brk = false;
      if ((it > 0 && new_chi2 > chi2) || stop) {
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 7; q++) s_T[wave][q] = s_Tbk[wave][q];
        }
        brk = true;
      } else {
        double mx[6];
#pragma unroll
        for (int r = 0; r < 6; r++) mx[r] = -xs[r];
        const Rigid T0 = se3_from7(s_T[wave]);
        const Rigid T1 = se3_mul(T0, ia_se3_exp(mx));
        ia_wave_fence();  // every lane has read T before lane 0 replaces it
        if (lane == 0) {
          se3_to7(T0, s_Tbk[wave]);
          se3_to7(T1, s_T[wave]);
        }
        chi2 = new_chi2;
        switch (level) {
          case 0: its0++; break; case 1: its1++; break; case 2: its2++; break; case 3: its3++; break;
          case 4: its4++; break; case 5: its5++; break; case 6: its6++; break; default: its7++; break;
        }
        error = abs_max6(xs);
        if (error <= 1e-10) brk = true;
      }
      IA_STAMP(5);
      if (brk) break;
    }
    // image_align.cc:73-76
    if (prm.fast && error > 0.01) {
      error = 1e10;
      break;
    }
  }
  ia_wave_fence();
  if (tid == 0) {
    sdvl_align_result r;
#pragma unroll
    for (int q = 0; q < 7; q++) r.T[q] = s_T[0][q];
    r.error = error;
    r.chi2 = chi2;
    r.n_meas = n_meas / 16;
    r.its[0] = its0; r.its[1] = its1; r.its[2] = its2; r.its[3] = its3; r.its[4] = its4; r.its[5] = its5; r.its[6] = its6; r.its[7] = its7;
    r.stop = stop ? 1 : 0;
    r.iters_run = iters_run;
    r.pad_ = 0;
    *job.out = r;
#ifdef SDVL_IA_STAMPS
    if (blockIdx.x == 0) {
      st_[6] = iters_run;
      for (int q = 0; q < 7; q++) g_ia_stamps[q] = st_[q];
    }
    atomicMax(&g_ia_stamps[7], ((st_[0] + st_[1] + st_[2] + st_[3] + st_[4] + st_[5]) << 8) | static_cast<unsigned>(iters_run));
#endif
  }
}

// ---- PrecomputePatches as a launch of its own (VERDICT r03 #1b): the reference patches, their gradients and the normal matrix of
// a level do not depend on the pose, so one workgroup per (job, level) prepares ALL levels of all jobs at once — 768 workgroups of 4
// waves for a tracked step instead of 3 x (3 rounds of lane work + a 21-value reduction + an LDLT factorisation) inside every job's
// one-wave Gauss-Newton chain (80 of the chain's 235 k clock ticks).  Per (job, level): items [48][pitch] floats, gradient sums
// [3][pitch], visibility [pitch] bytes, and fac[32] = the LDLT factor of H over the level's visible features (L lower triangle 21 |
// transpositions 6 | number of visible features): the Gauss-Newton kernel starts every level with it and rebuilds H only when a
// feature leaves the image.
struct IaPreOut {
  float *items;
  double *S;
  uint8_t *vis;
  double *fac;
};
__device__ __forceinline__ IaPreOut ia_pre_block(void *base, int n_jobs, int n_lv, int pitch, int job, int li) {
  // [items: n_jobs * n_lv * 48 * pitch floats][S: n_jobs * n_lv * 3 * pitch doubles][fac: n_jobs * n_lv * 32 doubles][vis bytes]
  const size_t blk = static_cast<size_t>(job) * n_lv + li, nb = static_cast<size_t>(n_jobs) * n_lv;
  uint8_t *b = static_cast<uint8_t *>(base);
  IaPreOut o;
  o.items = reinterpret_cast<float *>(b) + blk * 48 * pitch;
  o.S = reinterpret_cast<double *>(b + nb * 48 * pitch * sizeof(float)) + blk * 3 * pitch;
  o.fac = reinterpret_cast<double *>(b + nb * 48 * pitch * sizeof(float) + nb * 3 * pitch * sizeof(double)) + blk * 32;
  o.vis = b + nb * 48 * pitch * sizeof(float) + nb * 3 * pitch * sizeof(double) + nb * 32 * sizeof(double) + blk * pitch;
  return o;
}
inline size_t ia_pre_bytes(int n_jobs, int n_lv, int pitch) {
  const size_t nb = static_cast<size_t>(n_jobs) * n_lv;
  return (nb * (48 * pitch * sizeof(float) + 3 * pitch * sizeof(double) + 32 * sizeof(double) + pitch) + 255) / 256 * 256;
}

template <int kPreWaves, class Feats>
__device__ __forceinline__ void ia_pre_body(const uint8_t *ref_img, int W, int H, int level, int nf, const Feats F, double fx, int pitch, IaPreOut o) {
  __shared__ double s_redH[kPreWaves][24];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the level's image pointer and size come out of the job record by a run-time index: have them in scalar registers NOW — loaded where
  // they are first used, in front of the window's loads, they made those loads wait for everything issued before them
  asm volatile("" : "+s"(ref_img), "+s"(W), "+s"(H));
  const float scale = 1.0f / (1 << level);
  const double fl = fx / (1 << level);
  float2 *it_pd = reinterpret_cast<float2 *>(o.items);
  float *it_dy = o.items + static_cast<size_t>(32) * pitch;
  double h16[16], h8[8];
#pragma unroll
  for (int i = 0; i < 16; i++) h16[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 8; i++) h8[i] = 0.0;
  for (int f = tid; f < pitch; f += 64 * kPreWaves) {
    bool vis = false;
    double sums[3] = {0.0, 0.0, 0.0};
    if (f < nf) {
      typename Feats::Raw raw;
      double fpx, fpy;
      F.fetch(f, &raw, &fpx, &fpy);
      vis = ia_precompute_feature(ref_img, W, H, scale, fpx, fpy, [&]() { return F.valid(raw); }, f, pitch, it_pd, it_dy, sums);
      if (vis) {
        const V3 xyz = F.point(raw);
        ia_feature_h(xyz.x, xyz.y, 1. / xyz.z, fl, sums[0], sums[1], sums[2], h16, h8);
      }
    }
    o.S[f] = sums[0];
    o.S[pitch + f] = sums[1];
    o.S[2 * pitch + f] = sums[2];
    o.vis[f] = vis ? 1 : 0;
  }
  const double t16 = wave_reduce_n<16>(h16, lane);
  const double t8 = wave_reduce_n<8>(h8, lane);
  if ((lane & 3) == 0) s_redH[wave][(lane >> 2) & 15] = t16;
  if ((lane & 7) == 0) s_redH[wave][16 + ((lane >> 3) & 7)] = t8;
  if (kPreWaves > 1) __syncthreads(); else ia_wave_fence();
  if (wave == 0) {
    double Hm[36];
    int k = 0;
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
#pragma unroll
      for (int c = rr; c < 6; c++) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kPreWaves; w++) sum += s_redH[w][k];
        Hm[6 * rr + c] = sum;
        Hm[6 * c + rr] = sum;
        k++;
      }
    double La[36];
    int tr[6];
    ldlt_factor6_reg<true>(Hm, La, tr);
    if (lane == 0) {
      int q = 0;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) o.fac[q++] = La[6 * i + j];
#pragma unroll
      for (int i = 0; i < 6; i++) o.fac[21 + i] = static_cast<double>(tr[i]);
    }
  }
}

__global__ __launch_bounds__(256) void image_align_pre_kernel(const IaJob *__restrict__ jobs, const sdvl_align_feature *__restrict__ feats_all, Cam cam,
                                                              sdvl_align_params prm, int n_jobs, int pitch, void *pre) {
  const int n_lv = prm.max_level - prm.min_level + 1;
  const int job = static_cast<int>(blockIdx.x) / n_lv, li = static_cast<int>(blockIdx.x) - job * n_lv, level = prm.max_level - li;
  const IaJob &jb = jobs[job];
  ia_pre_body<4>(jb.ref_level[level], jb.lw[level], jb.lh[level], level, jb.n_feat, IaRecordFeats{feats_all + jb.feat_begin}, cam.fx, pitch,
              ia_pre_block(pre, n_jobs, n_lv, pitch, job, li));
}

template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(1, 8))) void image_align_wave_pre_kernel(
    const IaJob *__restrict__ jobs, const sdvl_align_feature *__restrict__ feats_all, Cam cam, sdvl_align_params prm, int n_jobs, int max_f, void *pre,
    sdvl_align_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  const IaJob &jb = jobs[blockIdx.x];
  const int n_lv = prm.max_level - prm.min_level + 1;
  const IaPreOut o = ia_pre_block(pre, n_jobs, n_lv, max_f, static_cast<int>(blockIdx.x), 0);
  IaIn in{jb.ref_level, jb.cur_level, jb.lw, jb.lh, jb.n_feat, jb.T, out + jb.out_index};
  in.pre_items = o.items; in.pre_S = o.S; in.pre_vis = o.vis; in.pre_fac = o.fac; in.pre_pitch = max_f;
  ia_wave_body<kWaves>(in, IaRecordFeats{feats_all + jb.feat_begin}, cam, prm, max_f, s_dyn);
}

// The same two kernels for a tracked step (sdvl_track.hip): job j = tracker record j of the step, its features are the rows of
// last_frame's feature buffer in the tracking tables — no IaJob records, no feature records, no launch in between
__device__ __forceinline__ IaTableFeats ia_table_feats(const TrackJobDev &jb, const TrackPoint *points, const TrackFeat *feats0, const TrackFeat *feats1,
                                                       int np, int nfeat_cap) {
  return IaTableFeats{(jb.feat_buf ? feats1 : feats0) + static_cast<size_t>(jb.tracker) * nfeat_cap, points + static_cast<size_t>(jb.tracker) * np,
                      se3_inverse(se3_from7(jb.last_pose)).t};
}

template <int kPreWaves>
__global__ __launch_bounds__(64 * kPreWaves) void image_align_track_pre_kernel(const TrackJobDev *__restrict__ jobs, const TrackPoint *__restrict__ points,
                                                                    const TrackFeat *__restrict__ feats0, const TrackFeat *__restrict__ feats1, int np,
                                                                    int nfeat_cap, Cam cam, sdvl_align_params prm, int n_jobs, int pitch, void *pre) {
  const int n_lv = prm.max_level - prm.min_level + 1;
  const int job = static_cast<int>(blockIdx.x) / n_lv, li = static_cast<int>(blockIdx.x) - job * n_lv, level = prm.max_level - li;
  const TrackJobDev &jb = jobs[job];
  ia_pre_body<kPreWaves>(jb.last_level[level], jb.cur.lw[level], jb.cur.lh[level], level, jb.n_feat, ia_table_feats(jb, points, feats0, feats1, np, nfeat_cap), cam.fx,
              pitch, ia_pre_block(pre, n_jobs, n_lv, pitch, job, li));
}

template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(1, 8))) void image_align_track_wave_pre_kernel(
    const TrackJobDev *__restrict__ jobs, const TrackPoint *__restrict__ points, const TrackFeat *__restrict__ feats0, const TrackFeat *__restrict__ feats1,
    int np, int nfeat_cap, Cam cam, sdvl_align_params prm, int n_jobs, int max_f, void *pre, sdvl_align_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  const TrackJobDev &jb = jobs[blockIdx.x];
  const int n_lv = prm.max_level - prm.min_level + 1;
  const IaPreOut o = ia_pre_block(pre, n_jobs, n_lv, max_f, static_cast<int>(blockIdx.x), 0);
  IaIn in{jb.last_level, jb.cur.level, jb.cur.lw, jb.cur.lh, jb.n_feat, jb.T0, out + blockIdx.x};
  in.pre_items = o.items; in.pre_S = o.S; in.pre_vis = o.vis; in.pre_fac = o.fac; in.pre_pitch = max_f;
  ia_wave_body<kWaves>(in, ia_table_feats(jb, points, feats0, feats1, np, nfeat_cap), cam, prm, max_f, s_dyn);
}

// dynamic LDS of the Gauss-Newton kernels: x[4][max_f] doubles + one flag byte per feature
size_t ia_wave_lds_bytes(int max_f) { return static_cast<size_t>(max_f) * (4 * sizeof(double) + 1) + 64; }

}  // namespace

// the four-wave kernels may ask for more dynamic LDS than the default limit: the attribute belongs to the kernel object of ONE
// device — set once per device, whichever thread gets there first
static int ia_allow_dynamic_lds(sdvl_ctx *ctx) {
  static std::atomic<unsigned long long> attr_devices{0};
  const unsigned long long bit = 1ull << (ctx->device & 63);
  if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
    SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
    SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_track_wave_pre_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(ia_wave_lds_bytes(kMaxF))));
    SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_wave_pre_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(ia_wave_lds_bytes(kMaxF))));
    attr_devices.fetch_or(bit, std::memory_order_release);
  }
  return SDVL_OK;
}

// Queues the alignment of n_jobs frame pairs (no wait).  The feature records come from the host (`features`, staged and
// copied here) or already sit in HBM (`d_features`: an sdvl_align_store).  Results go to `d_results` when given (device
// memory, nothing returns to the host), otherwise to the context's result buffers for sdvl_image_align_end.
// Jobs of up to kLdsMaxF features run one wave per job; the (few) larger ones of the same call — a fresh keyframe of configuration C
// with its ~850 features — four waves per job in a second launch pair.
int sdvl_image_align_enqueue(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, int n_features, const sdvl_align_feature *features,
                             const sdvl_align_feature *d_features, const sdvl_camera *cam, const sdvl_align_params *p,
                             sdvl_align_result *d_results) {
  SDVL_REQUIRE(ctx, p->patch_size == 4, "only align_patch_size 4 is supported");
  SDVL_REQUIRE(ctx, p->min_level >= 0 && p->max_level >= p->min_level && p->max_level < SDVL_MAX_LEVELS, "bad align levels");
  SDVL_REQUIRE(ctx, p->max_its >= 0, "bad max_its");
  std::vector<int> order(n_jobs);
  int n_small = 0, max_nf_small = 0, max_nf_big = 0;
  for (int j = 0; j < n_jobs; j++) {
    const sdvl_align_job &a = jobs[j];
    SDVL_REQUIRE(ctx, a.ref && a.cur, "null frame in alignment job");
    SDVL_REQUIRE(ctx, a.ref->width == a.cur->width && a.ref->height == a.cur->height && a.ref->v.levels == a.cur->v.levels,
                 "frame pair with different geometry");
    SDVL_REQUIRE(ctx, p->max_level < a.ref->v.levels, "max_align_level exceeds the pyramid depth");
    SDVL_REQUIRE(ctx, a.feat_begin >= 0 && a.feat_end >= a.feat_begin && a.feat_end <= n_features, "feature range out of bounds");
    if (a.feat_end - a.feat_begin > kMaxF) {
      ctx->err = "too many features in one alignment job (SDVL_MAX_ALIGN_FEATURES)";
      return SDVL_ERR_CAPACITY;
    }
  }
  {  // one-wave jobs first, the others behind them; results are written to each job's own slot, so the order is free
    int lo = 0, hi = n_jobs;
    for (int j = 0; j < n_jobs; j++) {
      const int nf = jobs[j].feat_end - jobs[j].feat_begin;
      if (nf <= kLdsMaxF) {
        order[lo++] = j;
        if (nf > max_nf_small) max_nf_small = nf;
      } else {
        order[--hi] = j;
        if (nf > max_nf_big) max_nf_big = nf;
      }
    }
    n_small = lo;
  }
  const int n_big = n_jobs - n_small;
  const int n_lv = p->max_level - p->min_level + 1;
  const int max_f_small = max_nf_small == 0 ? 64 : (max_nf_small + 63) / 64 * 64;
  const int max_f_big = (max_nf_big + 255) / 256 * 256;
  const size_t pre_small_bytes = n_small > 0 ? ia_pre_bytes(n_small, n_lv, max_f_small) : 0;
  const size_t work = pre_small_bytes + (n_big > 0 ? ia_pre_bytes(n_big, n_lv, max_f_big) : 0);
  const size_t job_bytes = (sizeof(IaJob) * n_jobs + 255) / 256 * 256;
  const size_t feat_bytes = d_features ? 0 : sizeof(sdvl_align_feature) * static_cast<size_t>(n_features);
  const size_t res_bytes = sizeof(sdvl_align_result) * n_jobs;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_ensure(ctx, &ctx->d_work, &ctx->d_work_bytes, work + 256, false);
  if (!rc && !d_results) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, res_bytes, true);
  if (!rc) rc = sdvl_stage_alloc(ctx, job_bytes + feat_bytes, &hs, &dsx);
  if (rc) return rc;
  IaJob *hj = static_cast<IaJob *>(hs);
  for (int q = 0; q < n_jobs; q++) {
    const sdvl_align_job &a = jobs[order[q]];
    IaJob &d = hj[q];
    memset(&d, 0, sizeof(IaJob));
    for (int l = 0; l < a.ref->v.levels; l++) {
      d.ref_level[l] = a.ref->v.level[l];
      d.cur_level[l] = a.cur->v.level[l];
      d.lw[l] = a.ref->v.lw[l];
      d.lh[l] = a.ref->v.lh[l];
    }
    d.feat_begin = a.feat_begin;
    d.n_feat = a.feat_end - a.feat_begin;
    d.out_index = order[q];
    for (int k = 0; k < 7; k++) d.T[k] = a.T[k];
  }
  if (feat_bytes) memcpy(static_cast<uint8_t *>(hs) + job_bytes, features, feat_bytes);
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, job_bytes + feat_bytes));
  const sdvl_align_feature *feats_dev = d_features ? d_features : reinterpret_cast<const sdvl_align_feature *>(static_cast<uint8_t *>(dsx) + job_bytes);
  Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  // results straight into the context's pinned host buffer (posted PCIe writes, visible once the kernel has completed) unless the caller keeps them on the device
  sdvl_align_result *dst = d_results ? d_results : static_cast<sdvl_align_result *>(ctx->h_out);
  if (n_small > 0) {
    SDVL_LAUNCH(ctx, "image_align_pre", image_align_pre_kernel, dim3(static_cast<unsigned>(n_small) * n_lv), dim3(256), static_cast<const IaJob *>(dsx), feats_dev,
                c, *p, n_small, max_f_small, ctx->d_work);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "image_align", &ev_a, &ev_b);
    hipExtLaunchKernelGGL((image_align_wave_pre_kernel<1>), dim3(n_small), dim3(64), ia_wave_lds_bytes(max_f_small), ctx->stream, ev_a, ev_b, 0,
                          static_cast<const IaJob *>(dsx), feats_dev, c, *p, n_small, max_f_small, ctx->d_work, dst);
  }
  if (n_big > 0) {
    const int rc_a = ia_allow_dynamic_lds(ctx);
    if (rc_a) return rc_a;
    void *pre_big = static_cast<uint8_t *>(ctx->d_work) + pre_small_bytes;
    SDVL_LAUNCH(ctx, "image_align_pre", image_align_pre_kernel, dim3(static_cast<unsigned>(n_big) * n_lv), dim3(256), static_cast<const IaJob *>(dsx) + n_small,
                feats_dev, c, *p, n_big, max_f_big, pre_big);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "image_align_big", &ev_a, &ev_b);
    hipExtLaunchKernelGGL((image_align_wave_pre_kernel<4>), dim3(n_big), dim3(256), ia_wave_lds_bytes(max_f_big), ctx->stream, ev_a, ev_b, 0,
                          static_cast<const IaJob *>(dsx) + n_small, feats_dev, c, *p, n_big, max_f_big, pre_big, dst);
  }
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

// The alignment of a tracked step (sdvl_track_align): the jobs are the step's TrackJobDev records, already in HBM; features come
// out of the tracking tables inside the kernels.  max_nf = the largest feature count among the jobs.
int sdvl_image_align_track_enqueue(sdvl_ctx *ctx, int n_jobs, const TrackJobDev *d_jobs, const TrackPoint *d_points, const TrackFeat *d_feats0,
                                   const TrackFeat *d_feats1, int np, int nfeat_cap, int max_nf, int levels, const sdvl_camera *cam,
                                   const sdvl_align_params *p, sdvl_align_result *d_results, int batch_size) {
  SDVL_REQUIRE(ctx, p->patch_size == 4, "only align_patch_size 4 is supported");
  SDVL_REQUIRE(ctx, p->min_level >= 0 && p->max_level >= p->min_level && p->max_level < SDVL_MAX_LEVELS, "bad align levels");
  SDVL_REQUIRE(ctx, p->max_level < levels, "max_align_level exceeds the pyramid depth");
  SDVL_REQUIRE(ctx, p->max_its >= 0, "bad max_its");
  if (max_nf > kMaxF) {
    ctx->err = "too many features in one alignment job (SDVL_MAX_ALIGN_FEATURES)";
    return SDVL_ERR_CAPACITY;
  }
  // configuration C's ~850 features per job: four waves share them.  So do the jobs of a SMALL batch (a lone camera's HandleFrame):
  // one feature per lane instead of three rounds per lane shortens every Gauss-Newton evaluation of a chain that has the chip to
  // itself; a farm's launches of hundreds of jobs keep one wave per job (fewer instructions in total).  batch_size = the set's
  // capacity, not this step's job count: the form — and with it the order of a sequence's sums — is fixed for the set's life.
  const int kw = (max_nf > kLdsMaxF || batch_size <= 32) ? 4 : 1;
  const int max_f = max_nf <= 0 ? 64 * kw : (max_nf + 64 * kw - 1) / (64 * kw) * (64 * kw);
  const int n_lv = p->max_level - p->min_level + 1;
  const int rc = sdvl_ensure(ctx, &ctx->d_work, &ctx->d_work_bytes, ia_pre_bytes(n_jobs, n_lv, max_f) + 256, false);
  if (rc) return rc;
  {
    const int rc_a = ia_allow_dynamic_lds(ctx);
    if (rc_a) return rc_a;
  }
  const Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  // Round 6: beside the one-wave Gauss-Newton form (a farm's launches) the precompute runs ONE wave per (job, level) too, its lanes
  // taking the level's features in rounds of 64: a four-wave workgroup is placed when four slots of one CU are free at the same moment,
  // which among the other streams' one-wave workgroups happens far less often than its share — 2.4 -> 1.3 ms of dispatch time per step
  // of 16 groups for the same instructions (+1.4 % tracked frames/s).  Small sets (a lone camera) keep four waves: nobody else wants the CU.
  if (kw == 1)
    SDVL_LAUNCH(ctx, "image_align_pre", image_align_track_pre_kernel<1>, dim3(static_cast<unsigned>(n_jobs) * n_lv), dim3(64), d_jobs, d_points, d_feats0, d_feats1,
                np, nfeat_cap, c, *p, n_jobs, max_f, ctx->d_work);
  else
  SDVL_LAUNCH(ctx, "image_align_pre", image_align_track_pre_kernel<4>, dim3(static_cast<unsigned>(n_jobs) * n_lv), dim3(256), d_jobs, d_points, d_feats0, d_feats1,
              np, nfeat_cap, c, *p, n_jobs, max_f, ctx->d_work);
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  sdvl_timer_events(ctx, max_nf > kLdsMaxF ? "image_align_big" : "image_align", &ev_a, &ev_b);
  const size_t lds = ia_wave_lds_bytes(max_f);
  if (kw == 4)
    hipExtLaunchKernelGGL((image_align_track_wave_pre_kernel<4>), dim3(n_jobs), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, d_jobs, d_points, d_feats0, d_feats1,
                          np, nfeat_cap, c, *p, n_jobs, max_f, ctx->d_work, d_results);
  else
    hipExtLaunchKernelGGL((image_align_track_wave_pre_kernel<1>), dim3(n_jobs), dim3(64), lds, ctx->stream, ev_a, ev_b, 0, d_jobs, d_points, d_feats0, d_feats1,
                          np, nfeat_cap, c, *p, n_jobs, max_f, ctx->d_work, d_results);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

// launch + result copy, no wait: the caller may queue more work on the context (kernels that neither read results nor use
// the context's result buffers: sdvl_pyramid_build, sdvl_detect_corners, sdvl_orb_describe) before sdvl_image_align_end
extern "C" int sdvl_image_align_begin(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, int n_features,
                                      const sdvl_align_feature *features, const sdvl_camera *cam, const sdvl_align_params *p) {
  if (!ctx || !cam || !p || n_jobs < 0 || (n_jobs > 0 && !jobs) || n_features < 0 || (n_features > 0 && !features)) return SDVL_ERR_INVALID;
  ctx->align_pending = 0;
  if (n_jobs == 0) return SDVL_OK;
  const int rc = sdvl_image_align_enqueue(ctx, n_jobs, jobs, n_features, features, nullptr, cam, p, nullptr);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_mark_record(ctx, SDVL_MARK_ALIGN, &ctx->align_ticket));
  ctx->align_pending = n_jobs;
  return SDVL_OK;
}

// ---- feature records that stay in HBM (Relocalize, sdvl.cc:205-238: every new frame is aligned against the SAME keyframes)
struct sdvl_align_store {
  sdvl_align_feature *d;
  int cap;
};

extern "C" int sdvl_align_store_create(sdvl_ctx *ctx, int capacity, sdvl_align_store **out) {
  if (!ctx || !out || capacity <= 0) return SDVL_ERR_INVALID;
  SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
  void *d = nullptr;
  SDVL_HIP_CHECK(ctx, hipMalloc(&d, sizeof(sdvl_align_feature) * static_cast<size_t>(capacity)));
  *out = new sdvl_align_store{static_cast<sdvl_align_feature *>(d), capacity};
  return SDVL_OK;
}

extern "C" int sdvl_align_store_destroy(sdvl_ctx *ctx, sdvl_align_store *store) {
  if (!ctx) return SDVL_ERR_INVALID;
  if (!store) return SDVL_OK;
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));  // alignments that read it may still be queued
  SDVL_HIP_CHECK(ctx, hipFree(store->d));
  delete store;
  return SDVL_OK;
}

// records [offset, offset + n) of the store <- features; ordered behind the context's queued work, complete on return
extern "C" int sdvl_align_store_write(sdvl_ctx *ctx, sdvl_align_store *store, int offset, int n, const sdvl_align_feature *features) {
  if (!ctx || !store || offset < 0 || n < 0 || (n > 0 && !features)) return SDVL_ERR_INVALID;
  SDVL_REQUIRE(ctx, offset + n <= store->cap, "sdvl_align_store_write beyond the store's capacity");
  if (n == 0) return SDVL_OK;
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(store->d + offset, features, sizeof(sdvl_align_feature) * static_cast<size_t>(n), hipMemcpyHostToDevice, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  return SDVL_OK;
}

// sdvl_image_align_begin with the jobs' feat_begin / feat_end naming records of `store`: nothing but the job records crosses the link
extern "C" int sdvl_image_align_begin_stored(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, const sdvl_align_store *store, const sdvl_camera *cam,
                                             const sdvl_align_params *p) {
  if (!ctx || !cam || !p || !store || n_jobs < 0 || (n_jobs > 0 && !jobs)) return SDVL_ERR_INVALID;
  ctx->align_pending = 0;
  if (n_jobs == 0) return SDVL_OK;
  const int rc = sdvl_image_align_enqueue(ctx, n_jobs, jobs, store->cap, nullptr, store->d, cam, p, nullptr);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, sdvl_mark_record(ctx, SDVL_MARK_ALIGN, &ctx->align_ticket));
  ctx->align_pending = n_jobs;
  return SDVL_OK;
}

// waits for the results of the matching sdvl_image_align_begin only — not for work queued after it
extern "C" int sdvl_image_align_end(sdvl_ctx *ctx, int n_jobs, sdvl_align_result *out) {
  if (!ctx || n_jobs < 0 || (n_jobs > 0 && !out)) return SDVL_ERR_INVALID;
  if (n_jobs == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, ctx->align_pending == n_jobs, "sdvl_image_align_end without a matching sdvl_image_align_begin");
  SDVL_HIP_CHECK(ctx, sdvl_mark_wait(ctx, SDVL_MARK_ALIGN, ctx->align_ticket));
  memcpy(out, ctx->h_out, sizeof(sdvl_align_result) * n_jobs);
  ctx->align_pending = 0;
  return SDVL_OK;
}

extern "C" int sdvl_image_align(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, int n_features,
                                const sdvl_align_feature *features, const sdvl_camera *cam, const sdvl_align_params *p,
                                sdvl_align_result *out) {
  if (n_jobs > 0 && !out) return SDVL_ERR_INVALID;
  const int rc = sdvl_image_align_begin(ctx, n_jobs, jobs, n_features, features, cam, p);
  if (rc) return rc;
  return sdvl_image_align_end(ctx, n_jobs, out);
}

#ifdef SDVL_IA_STAMPS
extern "C" int sdvl_debug_ia_stamps(unsigned long long *out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ia_stamps), 64) == hipSuccess ? 0 : -1;
}
#endif
