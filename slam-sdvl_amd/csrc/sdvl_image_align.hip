// sdvl_image_align.hip — K5/K6 sparse direct image alignment, ImageAlign::ComputePose (image_align.cc:46-84).
// One 512-thread workgroup per frame pair runs the WHOLE coarse-to-fine Gauss-Newton on device (levels
// max..min x <= max_its iterations) — no host round trip per iteration:
//   PrecomputePatches (:208-267)  work item = (feature, pixel of the 4x4 patch): bilinear reference patch (float)
//                                 and the 6-vector Jacobian (double) into an L2-resident cache in HBM;
//   ComputeResiduals (:127-206)   phase A, thread = feature: project with the current Rigid, bilinear weights -> LDS;
//                                 phase B, thread = (feature, pixel): residual, upper triangle of J J^T (21), J res (6),
//                                 chi2, count in FP64 registers; halving butterfly over the wave (32 DP shuffles for 32
//                                 values instead of 6 per value), fixed-order sum over the 8 waves in LDS;
//   Optimize (:86-125)            thread 0: pivoted LDLT 6x6, chi2 / NaN / stop_ tests, roll-back, T <- T * Exp(-x).
// Float vs double follow the reference statement by statement; the only deviation is the reduction ORDER of H, Jres
// (double) and chi2 (float in the reference, accumulated in double here), hence tolerance-class parity (1e-4).
#include <stdlib.h>

#include <atomic>
#include <vector>

#include "sdvl_internal.h"
#include "sdvl_math.h"
#include "sdvl_search_types.h"

namespace {

using namespace sdvl;

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxF = SDVL_MAX_ALIGN_FEATURES;

struct IaJob {
  const uint8_t *ref_level[SDVL_MAX_LEVELS];
  const uint8_t *cur_level[SDVL_MAX_LEVELS];
  int lw[SDVL_MAX_LEVELS], lh[SDVL_MAX_LEVELS];
  int feat_begin, n_feat;
  int out_index, pad_;  // slot of this job's result (jobs are regrouped by size before the launch)
  double T[7];
  float *patch_cache;  // [n_feat*16]
  double *jac_cache;   // [n_feat*16][6]
};

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return __hiloint2double(hi, lo);
}

// Halving butterfly: on return the even lane 2j (and its odd partner) holds the wave-wide sum of value index
// idx(2j) = lane bits (5,4,3,2,1) read as a 5-bit number.  32 values in, 32 DP shuffles.
// The same for 8 values (10 DP shuffles): on return every lane holds the wave-wide sum of value index (lane >> 3) & 7.
__device__ __forceinline__ double wave_reduce8(const double *v8, int lane) {
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = v8[i];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const bool up = lane & 32;
    const double keep = up ? v[i + 4] : v[i], send = up ? v[i] : v[i + 4];
    v[i] = keep + shfl_xor_f64(send, 32);
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const bool up = lane & 16;
    const double keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2];
    v[i] = keep + shfl_xor_f64(send, 16);
  }
  {
    const bool up = lane & 8;
    const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    v[0] = keep + shfl_xor_f64(send, 8);
  }
  v[0] += shfl_xor_f64(v[0], 4);
  v[0] += shfl_xor_f64(v[0], 2);
  return v[0] + shfl_xor_f64(v[0], 1);
}

__device__ __forceinline__ double wave_reduce32(double *v, int lane) {
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const bool up = lane & 32;
    const double keep = up ? v[i + 16] : v[i], send = up ? v[i] : v[i + 16];
    v[i] = keep + shfl_xor_f64(send, 32);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const bool up = lane & 16;
    const double keep = up ? v[i + 8] : v[i], send = up ? v[i] : v[i + 8];
    v[i] = keep + shfl_xor_f64(send, 16);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const bool up = lane & 8;
    const double keep = up ? v[i + 4] : v[i], send = up ? v[i] : v[i + 4];
    v[i] = keep + shfl_xor_f64(send, 8);
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const bool up = lane & 4;
    const double keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2];
    v[i] = keep + shfl_xor_f64(send, 4);
  }
  {
    const bool up = lane & 2;
    const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    v[0] = keep + shfl_xor_f64(send, 2);
  }
  return v[0] + shfl_xor_f64(v[0], 1);
}

__global__ __launch_bounds__(kThreads) void image_align_kernel(const IaJob *__restrict__ jobs,
                                                               const sdvl_align_feature *__restrict__ feats_all, Cam cam,
                                                               sdvl_align_params prm, sdvl_align_result *__restrict__ out) {
  __shared__ int s_ui[kMaxF], s_vi[kMaxF];
  __shared__ float s_w[4][kMaxF];
  __shared__ uint8_t s_ok[kMaxF], s_vis[kMaxF];
  __shared__ double s_red[kWaves][32];
  __shared__ double s_sum[32];
  __shared__ double s_T[7], s_R[9];
  __shared__ int s_break, s_abort;

  const IaJob &job = jobs[blockIdx.x];
  const int nf = job.n_feat;
  const sdvl_align_feature *F = feats_all + job.feat_begin;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_items = nf * 16;

  // thread-0 optimiser state (image_align.cc:33-39)
  Rigid T = se3_identity(), T_bk = se3_identity();
  double chi2 = 1e10, error = 1e10;
  bool stop = false;
  int n_meas = 0, iters_run = 0;
  int its[SDVL_MAX_LEVELS];
#pragma unroll
  for (int i = 0; i < SDVL_MAX_LEVELS; i++) its[i] = 0;

  for (int f = tid; f < nf; f += kThreads) s_vis[f] = 0;
  if (tid == 0) {
    T = se3_from7(job.T);
    se3_to7(T, s_T);
    const M3 R = se3_rot(T);
    for (int i = 0; i < 9; i++) s_R[i] = R.m[i];
    s_abort = 0;
  }
  __syncthreads();

  for (int level = prm.max_level; level >= prm.min_level; level--) {
    const int W = job.lw[level], H = job.lh[level];
    const uint8_t *ref_img = job.ref_level[level];
    const uint8_t *cur_img = job.cur_level[level];
    const float scale = 1.0f / (1 << level);
    // jacobian_cache_.setZero(), image_align.cc:69
    for (int i = tid; i < n_items * 6; i += kThreads) job.jac_cache[i] = 0.0;
    if (tid == 0) T_bk = T;
    __syncthreads();

    for (int it = 0; it < prm.max_its; it++) {
      if (it == 0) {
        // ---- PrecomputePatches(level), image_align.cc:208-267
        const double fl = cam.fx / (1 << level);
        for (int idx = tid; idx < n_items; idx += kThreads) {
          const int f = idx >> 4, p = idx & 15;
          const sdvl_align_feature ft = F[f];
          const float u_ref = static_cast<float>(ft.px * scale);
          const float v_ref = static_cast<float>(ft.py * scale);
          const int ui = static_cast<int>(floorf(u_ref)), vi = static_cast<int>(floorf(v_ref));
          const int border = 3;
          if (!ft.valid || ui - border < 0 || vi - border < 0 || ui + border >= W || vi + border >= H) continue;
          if (p == 0) s_vis[f] = 1;
          const V3 xyz = vscale({ft.fx, ft.fy, ft.fz}, ft.depth);
          double fj[12];
          jacobian_3d_to_plane(xyz, fj);
          const float su = u_ref - ui, sv = v_ref - vi;
          const float w_tl = static_cast<float>((1.0 - su) * (1.0 - sv));
          const float w_tr = static_cast<float>(su * (1.0 - sv));
          const float w_bl = static_cast<float>((1.0 - su) * sv);
          const float w_br = su * sv;
          const int y = p >> 2, x = p & 3;
          const uint8_t *ip = ref_img + static_cast<size_t>(vi + y - 2) * W + (ui + x - 2);
          const int st = W;
          job.patch_cache[idx] = w_tl * ip[0] + w_tr * ip[1] + w_bl * ip[st] + w_br * ip[st + 1];
          const float dx = 0.5f * ((w_tl * ip[1] + w_tr * ip[2] + w_bl * ip[st + 1] + w_br * ip[st + 2]) -
                                   (w_tl * ip[-1] + w_tr * ip[0] + w_bl * ip[st - 1] + w_br * ip[st]));
          const float dy = 0.5f * ((w_tl * ip[st] + w_tr * ip[1 + st] + w_bl * ip[st * 2] + w_br * ip[st * 2 + 1]) -
                                   (w_tl * ip[-st] + w_tr * ip[1 - st] + w_bl * ip[0] + w_br * ip[1]));
          double *J = job.jac_cache + static_cast<size_t>(idx) * 6;
#pragma unroll
          for (int c = 0; c < 6; c++) J[c] = (dx * fj[c] + dy * fj[6 + c]) * fl;
        }
        __syncthreads();
      }
      // ---- ComputeResiduals phase A: per-feature projection, image_align.cc:147-181
      for (int f = tid; f < nf; f += kThreads) {
        uint8_t ok = 0;
        if (s_vis[f]) {
          const sdvl_align_feature ft = F[f];
          const V3 xr = vscale({ft.fx, ft.fy, ft.fz}, ft.depth);
          const V3 xc = {s_R[0] * xr.x + s_R[1] * xr.y + s_R[2] * xr.z + s_T[4], s_R[3] * xr.x + s_R[4] * xr.y + s_R[5] * xr.z + s_T[5],
                         s_R[6] * xr.x + s_R[7] * xr.y + s_R[8] * xr.z + s_T[6]};
          const V2 pr = cam_project(cam, xc);
          const float u_cur = static_cast<float>(pr.x * scale);
          const float v_cur = static_cast<float>(pr.y * scale);
          const float fu = floorf(u_cur), fv = floorf(v_cur);
          // (int)floorf of NaN / huge values is undefined on the CPU; treat anything outside the image as a miss
          if (fu >= 3.f && fv >= 3.f && fu < static_cast<float>(W - 3) && fv < static_cast<float>(H - 3)) {
            const int ui = static_cast<int>(fu), vi = static_cast<int>(fv);
            const float su = u_cur - ui, sv = v_cur - vi;
            s_ui[f] = ui;
            s_vi[f] = vi;
            s_w[0][f] = static_cast<float>((1.0 - su) * (1.0 - sv));
            s_w[1][f] = static_cast<float>(su * (1.0 - sv));
            s_w[2][f] = static_cast<float>((1.0 - su) * sv);
            s_w[3][f] = su * sv;
            ok = 1;
          }
        }
        s_ok[f] = ok;
      }
      __syncthreads();
      // ---- phase B: residuals + normal equations, image_align.cc:182-203
      double acc[32];
#pragma unroll
      for (int i = 0; i < 32; i++) acc[i] = 0.0;
      for (int idx = tid; idx < n_items; idx += kThreads) {
        const int f = idx >> 4;
        if (!s_ok[f]) continue;
        const int p = idx & 15, y = p >> 2, x = p & 3;
        const uint8_t *ip = cur_img + static_cast<size_t>(s_vi[f] + y - 2) * W + (s_ui[f] + x - 2);
        const float intensity = s_w[0][f] * ip[0] + s_w[1][f] * ip[1] + s_w[2][f] * ip[W] + s_w[3][f] * ip[W + 1];
        const float res = intensity - job.patch_cache[idx];
        const double *Jp = job.jac_cache + static_cast<size_t>(idx) * 6;
        double J[6];
#pragma unroll
        for (int c = 0; c < 6; c++) J[c] = Jp[c];
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = r; c < 6; c++) acc[k++] += J[r] * J[c];
#pragma unroll
        for (int r = 0; r < 6; r++) acc[21 + r] -= J[r] * res;
        acc[27] += static_cast<double>(res * res);
        acc[28] += 1.0;
      }
      const double tot = wave_reduce32(acc, lane);
      if ((lane & 1) == 0) {
        const int vidx = (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) |
                         ((lane >> 1) & 1);
        s_red[wave][vidx] = tot;
      }
      __syncthreads();
      if (tid < 32) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) s += s_red[w][tid];
        s_sum[tid] = s;
      }
      __syncthreads();
      // ---- Optimize body, image_align.cc:93-124
      if (tid == 0) {
        double Hm[36], Jres[6], x[6];
        int k = 0;
        for (int r = 0; r < 6; r++)
          for (int c = r; c < 6; c++) {
            Hm[6 * r + c] = s_sum[k];
            Hm[6 * c + r] = s_sum[k];
            k++;
          }
        for (int r = 0; r < 6; r++) Jres[r] = s_sum[21 + r];
        n_meas = static_cast<int>(s_sum[28]);
        iters_run++;
        const double new_chi2 = static_cast<double>(static_cast<float>(s_sum[27]) / static_cast<float>(n_meas));
        if (n_meas == 0) stop = true;
        ldlt_solve6_reg<true>(Hm, Jres, x);
        if (x[0] != x[0]) stop = true;
        int brk = 0;
        if ((it > 0 && new_chi2 > chi2) || stop) {
          T = T_bk;
          brk = 1;
        } else {
          T_bk = T;
          double mx[6];
          for (int r = 0; r < 6; r++) mx[r] = -x[r];
          T = se3_mul(T, se3_exp(mx));
          chi2 = new_chi2;
          its[level]++;
          error = abs_max6(x);
          if (error <= 1e-10) brk = 1;
        }
        se3_to7(T, s_T);
        const M3 R = se3_rot(T);
        for (int i = 0; i < 9; i++) s_R[i] = R.m[i];
        s_break = brk;
      }
      __syncthreads();
      if (s_break) break;
    }
    // image_align.cc:73-76
    if (tid == 0 && prm.fast && error > 0.01) {
      error = 1e10;
      s_abort = 1;
    }
    __syncthreads();
    if (s_abort) break;
  }
  if (tid == 0) {
    sdvl_align_result r;
    se3_to7(T, r.T);
    r.error = error;
    r.chi2 = chi2;
    r.n_meas = n_meas / 16;
    for (int i = 0; i < SDVL_MAX_LEVELS; i++) r.its[i] = its[i];
    r.stop = stop ? 1 : 0;
    r.iters_run = iters_run;
    r.pad_ = 0;
    out[job.out_index] = r;
  }
}


// ------------------------------------------------------------------------------------------------ LDS-resident path
// Same algorithm for jobs with <= kLdsMaxF features (the tracking case: <= max_matches + one seed per grid cell).
//  * reference patch value and template gradient (dx, dy) of every (feature, pixel) item live in LDS, the 2x6 frame
//    Jacobian of every feature too; the 6-vector J of an item is rebuilt in registers exactly as the reference computes it
//    ((dx*J0 + dy*J1) * fx/2^level, image_align.cc:263) — no Jacobian cache in HBM, no zero-fill per level;
//  * J does not depend on the iteration (inverse compositional), so H = sum J J^T changes only when the SET of features
//    that project inside the image changes; phase A detects that, and H is re-accumulated only then.  Reusing the stored
//    H yields the value a full recomputation would (same items, same fixed reduction order);
//  * thread 0 solves with a scratch-free, fully unrolled pivoted LDLT (same pivots and operation order as ldlt_solve6).
constexpr int kLdsMaxF = 384;

struct IaItem { float patch, dx, dy; };

// kSpill: the two big per-feature caches (the 2x6 Jacobians and the 16 reference items of every feature, 288 B per feature)
// live in the job's slice of the context's work buffer (L2-resident: a job touches nothing else) instead of LDS, so that
// jobs of up to SDVL_MAX_ALIGN_FEATURES features run the same kernel — configuration C aligns ~850 features per frame.
// Same statements, same order: results are those of the LDS-resident form.
template <bool kSpill, int kT>
__global__ __launch_bounds__(kT) void image_align_lds_kernel(const IaJob *__restrict__ jobs,
                                                                   const sdvl_align_feature *__restrict__ feats_all, Cam cam,
                                                                   sdvl_align_params prm, int max_f, sdvl_align_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  // carve: fjac[max_f*12] doubles | items[max_f*16] (both only when not spilled) | ui, vi ints | w[4] floats | fpos | flag bytes
  double *s_fj = kSpill ? jobs[blockIdx.x].jac_cache : reinterpret_cast<double *>(s_dyn);  // [max_f][12] raw 2x6 Jacobians
  IaItem *s_item = kSpill ? reinterpret_cast<IaItem *>(jobs[blockIdx.x].patch_cache)
                          : reinterpret_cast<IaItem *>(reinterpret_cast<double *>(s_dyn) + static_cast<size_t>(max_f) * 12);  // [max_f*16]
  int *s_ui = kSpill ? reinterpret_cast<int *>(s_dyn) : reinterpret_cast<int *>(s_item + static_cast<size_t>(max_f) * 16);
  int *s_vi = s_ui + max_f;
  float *s_w = reinterpret_cast<float *>(s_vi + max_f);                   // [4][max_f]
  double *s_fpos = reinterpret_cast<double *>(s_w + static_cast<size_t>(4) * max_f);  // [max_f][5]: px, py, and f * depth (3)
  uint8_t *s_ok = reinterpret_cast<uint8_t *>(s_fpos + static_cast<size_t>(5) * max_f);
  uint8_t *s_okprev = s_ok + max_f;
  uint8_t *s_vis = s_okprev + max_f;
  uint8_t *s_valid = s_vis + max_f;
  __shared__ double s_red[kT / 64][32];
  __shared__ double s_sum[32];
  __shared__ double s_H[21], s_L[36];
  __shared__ int s_tr[6];
  __shared__ double s_T[7], s_R[9];
  __shared__ int s_break, s_abort, s_changed;

  const IaJob &job = jobs[blockIdx.x];
  const int nf = job.n_feat;
  const sdvl_align_feature *F = feats_all + job.feat_begin;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_items = nf * 16;

  Rigid T = se3_identity(), T_bk = se3_identity();
  double chi2 = 1e10, error = 1e10;
  bool stop = false;
  int n_meas = 0, iters_run = 0;
  int its0 = 0, its1 = 0, its2 = 0, its3 = 0, its4 = 0, its5 = 0, its6 = 0, its7 = 0;

  // the feature records are read once: every level's PrecomputePatches and every iteration's projection use the LDS copy
  for (int f = tid; f < nf; f += kT) {
    const sdvl_align_feature ft = F[f];
    const V3 xyz = vscale({ft.fx, ft.fy, ft.fz}, ft.depth);
    s_fpos[f * 5 + 0] = ft.px;
    s_fpos[f * 5 + 1] = ft.py;
    s_fpos[f * 5 + 2] = xyz.x;
    s_fpos[f * 5 + 3] = xyz.y;
    s_fpos[f * 5 + 4] = xyz.z;
    s_valid[f] = ft.valid ? 1 : 0;
    s_vis[f] = 0;
  }
  if (tid == 0) {
    T = se3_from7(job.T);
    se3_to7(T, s_T);
    const M3 R = se3_rot(T);
    for (int i = 0; i < 9; i++) s_R[i] = R.m[i];
    s_abort = 0;
  }
  __syncthreads();

  for (int level = prm.max_level; level >= prm.min_level; level--) {
    const int W = job.lw[level], H = job.lh[level];
    const uint8_t *ref_img = job.ref_level[level];
    const uint8_t *cur_img = job.cur_level[level];
    const float scale = 1.0f / (1 << level);
    const double fl = cam.fx / (1 << level);
    // jacobian_cache_.setZero() (image_align.cc:69): items of features that fail this level's border test keep a zero J
    for (int i = tid; i < n_items; i += kT) { s_item[i].dx = 0.f; s_item[i].dy = 0.f; }
    for (int f = tid; f < nf; f += kT) s_okprev[f] = 2;  // forces H to be accumulated in the first iteration
    if (tid == 0) T_bk = T;
    __syncthreads();

    for (int it = 0; it < prm.max_its; it++) {
      if (it == 0) {
        // ---- PrecomputePatches(level), image_align.cc:208-267
        for (int idx = tid; idx < n_items; idx += kT) {
          const int f = idx >> 4, p = idx & 15;
          const float u_ref = static_cast<float>(s_fpos[f * 5 + 0] * scale);
          const float v_ref = static_cast<float>(s_fpos[f * 5 + 1] * scale);
          const int ui = static_cast<int>(floorf(u_ref)), vi = static_cast<int>(floorf(v_ref));
          const int border = 3;
          if (!s_valid[f] || ui - border < 0 || vi - border < 0 || ui + border >= W || vi + border >= H) continue;
          if (p == 0) {
            s_vis[f] = 1;
            const V3 xyz = {s_fpos[f * 5 + 2], s_fpos[f * 5 + 3], s_fpos[f * 5 + 4]};
            double fj[12];
            jacobian_3d_to_plane(xyz, fj);
#pragma unroll
            for (int c = 0; c < 12; c++) s_fj[f * 12 + c] = fj[c];
          }
          const float su = u_ref - ui, sv = v_ref - vi;
          const float w_tl = static_cast<float>((1.0 - su) * (1.0 - sv));
          const float w_tr = static_cast<float>(su * (1.0 - sv));
          const float w_bl = static_cast<float>((1.0 - su) * sv);
          const float w_br = su * sv;
          const int y = p >> 2, x = p & 3;
          const uint8_t *ip = ref_img + static_cast<size_t>(vi + y - 2) * W + (ui + x - 2);
          const int st = W;
          IaItem item;
          item.patch = w_tl * ip[0] + w_tr * ip[1] + w_bl * ip[st] + w_br * ip[st + 1];
          item.dx = 0.5f * ((w_tl * ip[1] + w_tr * ip[2] + w_bl * ip[st + 1] + w_br * ip[st + 2]) -
                            (w_tl * ip[-1] + w_tr * ip[0] + w_bl * ip[st - 1] + w_br * ip[st]));
          item.dy = 0.5f * ((w_tl * ip[st] + w_tr * ip[1 + st] + w_bl * ip[st * 2] + w_br * ip[st * 2 + 1]) -
                            (w_tl * ip[-st] + w_tr * ip[1 - st] + w_bl * ip[0] + w_br * ip[1]));
          s_item[idx] = item;
        }
        __syncthreads();
      }
      // ---- ComputeResiduals phase A: per-feature projection, image_align.cc:147-181
      if (tid == 0) s_changed = 0;
      __syncthreads();
      for (int f = tid; f < nf; f += kT) {
        uint8_t ok = 0;
        if (s_vis[f]) {
          const V3 xr = {s_fpos[f * 5 + 2], s_fpos[f * 5 + 3], s_fpos[f * 5 + 4]};
          const V3 xc = {s_R[0] * xr.x + s_R[1] * xr.y + s_R[2] * xr.z + s_T[4], s_R[3] * xr.x + s_R[4] * xr.y + s_R[5] * xr.z + s_T[5],
                         s_R[6] * xr.x + s_R[7] * xr.y + s_R[8] * xr.z + s_T[6]};
          const V2 pr = cam_project(cam, xc);
          const float u_cur = static_cast<float>(pr.x * scale);
          const float v_cur = static_cast<float>(pr.y * scale);
          const float fu = floorf(u_cur), fv = floorf(v_cur);
          if (fu >= 3.f && fv >= 3.f && fu < static_cast<float>(W - 3) && fv < static_cast<float>(H - 3)) {
            const int ui = static_cast<int>(fu), vi = static_cast<int>(fv);
            const float su = u_cur - ui, sv = v_cur - vi;
            s_ui[f] = ui;
            s_vi[f] = vi;
            s_w[f] = static_cast<float>((1.0 - su) * (1.0 - sv));
            s_w[max_f + f] = static_cast<float>(su * (1.0 - sv));
            s_w[2 * max_f + f] = static_cast<float>((1.0 - su) * sv);
            s_w[3 * max_f + f] = su * sv;
            ok = 1;
          }
        }
        if (ok != s_okprev[f]) s_changed = 1;
        s_ok[f] = ok;
        s_okprev[f] = ok;
      }
      __syncthreads();
      const bool rebuild_h = s_changed != 0;
      // ---- phase B: residuals + normal equations, image_align.cc:182-203
      double acc[32];
#pragma unroll
      for (int i = 0; i < 32; i++) acc[i] = 0.0;
      for (int idx = tid; idx < n_items; idx += kT) {
        const int f = idx >> 4;
        if (!s_ok[f]) continue;
        const int p = idx & 15, y = p >> 2, x = p & 3;
        const uint8_t *ip = cur_img + static_cast<size_t>(s_vi[f] + y - 2) * W + (s_ui[f] + x - 2);
        const float intensity = s_w[f] * ip[0] + s_w[max_f + f] * ip[1] + s_w[2 * max_f + f] * ip[W] + s_w[3 * max_f + f] * ip[W + 1];
        const IaItem item = s_item[idx];
        const float res = intensity - item.patch;
        const double *fj = &s_fj[f * 12];
        double J[6];
#pragma unroll
        for (int c = 0; c < 6; c++) J[c] = (item.dx * fj[c] + item.dy * fj[6 + c]) * fl;
        if (rebuild_h) {
          int k = 0;
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = r; c < 6; c++) acc[k++] += J[r] * J[c];
        }
#pragma unroll
        for (int r = 0; r < 6; r++) acc[21 + r] -= J[r] * res;
        acc[27] += static_cast<double>(res * res);
        acc[28] += 1.0;
      }
      if (rebuild_h) {
        const double tot = wave_reduce32(acc, lane);
        if ((lane & 1) == 0) {
          const int vidx = (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) |
                           ((lane >> 1) & 1);
          s_red[wave][vidx] = tot;
        }
      } else {  // H is reused: only Jres, chi2 and the count (values 21..28) are new
        const double tot = wave_reduce8(acc + 21, lane);
        if ((lane & 7) == 0) s_red[wave][21 + ((lane >> 3) & 7)] = tot;
      }
      __syncthreads();
      if (tid < 32) {
        double sacc = 0.0;
#pragma unroll
        for (int w = 0; w < kT / 64; w++) sacc += s_red[w][tid];
        if (tid < 21) {
          if (rebuild_h) s_H[tid] = sacc;  // keep for the iterations in which the contributing set stays the same
          else sacc = s_H[tid];
        }
        s_sum[tid] = sacc;
      }
      __syncthreads();
      // ---- Optimize body, image_align.cc:93-124
      if (tid == 0) {
        double Hm[36], Jres[6], xs[6];
        {
          int k = 0;
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = r; c < 6; c++) {
              Hm[6 * r + c] = s_sum[k];
              Hm[6 * c + r] = s_sum[k];
              k++;
            }
        }
#pragma unroll
        for (int r = 0; r < 6; r++) Jres[r] = s_sum[21 + r];
        n_meas = static_cast<int>(s_sum[28]);
        iters_run++;
        const double new_chi2 = static_cast<double>(static_cast<float>(s_sum[27]) / static_cast<float>(n_meas));
        if (n_meas == 0) stop = true;
        {  // H is the same for as long as the contributing set stays the same: so is its factorisation (kept in LDS)
          double La[36];
          int tr[6];
          if (rebuild_h) {
            ldlt_factor6_reg<true>(Hm, La, tr);
#pragma unroll
            for (int q = 0; q < 36; q++) s_L[q] = La[q];
#pragma unroll
            for (int q = 0; q < 6; q++) s_tr[q] = tr[q];
          } else {
#pragma unroll
            for (int q = 0; q < 36; q++) La[q] = s_L[q];
#pragma unroll
            for (int q = 0; q < 6; q++) tr[q] = s_tr[q];
          }
          ldlt_apply6_reg(La, tr, Jres, xs);
        }
        if (xs[0] != xs[0]) stop = true;
        int brk = 0;
        if ((it > 0 && new_chi2 > chi2) || stop) {
          T = T_bk;
          brk = 1;
        } else {
          T_bk = T;
          double mx[6];
#pragma unroll
          for (int r = 0; r < 6; r++) mx[r] = -xs[r];
          T = se3_mul(T, se3_exp(mx));
          chi2 = new_chi2;
          switch (level) {
            case 0: its0++; break; case 1: its1++; break; case 2: its2++; break; case 3: its3++; break;
            case 4: its4++; break; case 5: its5++; break; case 6: its6++; break; default: its7++; break;
          }
          error = abs_max6(xs);
          if (error <= 1e-10) brk = 1;
        }
        se3_to7(T, s_T);
        const M3 R = se3_rot(T);
#pragma unroll
        for (int i = 0; i < 9; i++) s_R[i] = R.m[i];
        s_break = brk;
      }
      __syncthreads();
      if (s_break) break;
    }
    if (tid == 0 && prm.fast && error > 0.01) {
      error = 1e10;
      s_abort = 1;
    }
    __syncthreads();
    if (s_abort) break;
  }
  if (tid == 0) {
    sdvl_align_result r;
    se3_to7(T, r.T);
    r.error = error;
    r.chi2 = chi2;
    r.n_meas = n_meas / 16;
    r.its[0] = its0; r.its[1] = its1; r.its[2] = its2; r.its[3] = its3; r.its[4] = its4; r.its[5] = its5; r.its[6] = its6; r.its[7] = its7;
    r.stop = stop ? 1 : 0;
    r.iters_run = iters_run;
    r.pad_ = 0;
    out[job.out_index] = r;
  }
}

// ------------------------------------------------------------------------------------------------ round 4: lane = feature
// The same Gauss-Newton as a chain of WAVES instead of a workgroup of (feature, pixel) threads.  What the round-3 profile said
// about the kernel above: 231 VGPRs x 4 waves per job, ~6 workgroup barriers per evaluation, thread 0 alone in the solve — alone
// 123 us per 256 jobs, 288 us among the other streams' kernels (every barrier waits for the slowest of four waves that all compete
// with the neighbours' waves for issue slots).  Here:
//  * lane = FEATURE; a wave walks its features in rounds of 64.  A lane reads only what it wrote itself (its feature's reference
//    items, its 3-D point): no barrier and no fence between PrecomputePatches and ComputeResiduals;
//  * the normal equations are factored per feature.  J(item) = (dx * Ja + dy * Jb) * fl with Ja, Jb the two rows of the feature's
//    2x6 Jacobian (image_align.cc:263), so  sum_items J res = fl * (Ja * A + Jb * B)  with A = sum dx res, B = sum dy res over the
//    feature's 16 pixels, and  sum_items J J^T = G^T S G  with G = fl * [Ja; Jb] and S = [Sxx Sxy; Sxy Syy] the feature's gradient
//    sums.  An evaluation costs 16 x (bilinear + 2 products) + 24 double operations per feature instead of 16 x 66; H costs 111 per
//    feature, and only when the contributing set changes.  Same mathematics, different rounding order: tolerance class (pose 1e-4;
//    the kernel above already summed in its own order);
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
  float *items;         // kGlobalItems: the job's slice of the work buffer
  sdvl_align_result *out;
  // kPre: what image_align_pre_kernel left for this job, one block per level (index 0 = max_level): items [48][pitch] floats,
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
    const bool valid = pt >= 0 && !(P[pt].status & kDeleted);
    double depth = 0.0;
    if (valid) {
      const double dx = P[pt].P[0] - first_pos.x, dy = P[pt].P[1] - first_pos.y, dz = P[pt].P[2] - first_pos.z;
      depth = sqrt(dx * dx + dy * dy + dz * dz);
    }
    *xyz = vscale({ft.bearing[0], ft.bearing[1], ft.bearing[2]}, depth);
    return valid;
  }
  __device__ __forceinline__ void pos(int f, double *px, double *py) const { *px = F[f].px[0]; *py = F[f].px[1]; }
};

// PrecomputePatches of ONE feature at one level (image_align.cc:208-267): border test, the 16 reference items {patch, dx, dy} and
// the gradient sums Sxx, Sxy, Syy.  Returns whether the feature is visible at the level; writes nothing otherwise.
__device__ __forceinline__ bool ia_precompute_feature(const uint8_t *ref_img, int W, int H, float scale, double fpx, double fpy, bool valid, int f,
                                                      int pitch, float2 *it_pd, float *it_dy, double *sums3) {
  const float u_ref = static_cast<float>(fpx * scale);
  const float v_ref = static_cast<float>(fpy * scale);
  const int ui = static_cast<int>(floorf(u_ref)), vi = static_cast<int>(floorf(v_ref));
  const int border = 3;
  if (!valid || ui - border < 0 || vi - border < 0 || ui + border >= W || vi + border >= H) return false;
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
  {
    uint32_t lo, hi;
    ia_load_row8(wp, 7, &lo, &hi);
#pragma unroll
    for (int k = 0; k < 4; k++) ra[k] = ia_byte(lo, k);
#pragma unroll
    for (int k = 4; k < 7; k++) ra[k] = ia_byte(hi, k - 4);
  }
  double sxx = 0.0, sxy = 0.0, syy = 0.0;
#pragma unroll
  for (int y = 0; y < 6; y++) {
    uint32_t lo, hi;
    ia_load_row8(wp + static_cast<size_t>(y + 1) * W, 7, &lo, &hi);
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

template <int kWaves, bool kGlobalItems, bool kPre, class Feats>
__device__ __forceinline__ void ia_wave_body(const IaIn job, const Feats F, const Cam cam, const sdvl_align_params prm, const int max_f, uint8_t *s_dyn) {
  static_assert(!kPre || kGlobalItems, "precomputed items live in the work buffer");
  // carve (max_f is a multiple of 64 * kWaves): x[4][max_f] doubles (point in frame 1: x, y, z, 1/z) | S[3][max_f] doubles (gradient
  // sums of the level) | items: pd[16][max_f] float2 {patch, dx}, dy[16][max_f] floats (LDS, or the job's slice of the work buffer)
  double *s_x = reinterpret_cast<double *>(s_dyn);
  double *s_S = s_x + static_cast<size_t>(4) * max_f;
  float2 *s_pd_l = reinterpret_cast<float2 *>(s_S + static_cast<size_t>(3) * max_f);
  float *s_dy_l = reinterpret_cast<float *>(s_pd_l + static_cast<size_t>(16) * max_f);
  uint8_t *s_flag = kGlobalItems ? reinterpret_cast<uint8_t *>(s_pd_l) : reinterpret_cast<uint8_t *>(s_dy_l + static_cast<size_t>(16) * max_f);
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
  // items of a job too large for LDS live in its slice of the work buffer, same layout with the job's own pitch
  const int pitch = kPre ? job.pre_pitch : (kGlobalItems ? (nf + 63) / 64 * 64 : max_f);
  float2 *it_pd = kGlobalItems ? reinterpret_cast<float2 *>(job.items) : s_pd_l;
  float *it_dy = kGlobalItems ? job.items + static_cast<size_t>(32) * pitch : s_dy_l;
  const double *S3 = s_S;  // gradient sums of the level: LDS, or (kPre) the level's block of the precompute kernel's output
  int S_pitch = max_f;

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
    const uint8_t *ref_img = job.ref_level[level];
    const uint8_t *cur_img = job.cur_level[level];
    const float scale = 1.0f / (1 << level);
    const double fl = cam.fx / (1 << level);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 7; q++) s_Tbk[wave][q] = s_T[wave][q];
    }
    // ---- PrecomputePatches(level), image_align.cc:208-267
    if (kPre) {
      // done by image_align_pre_kernel for every level at once (the patches do not depend on the pose): this level's block
      const int li = prm.max_level - level;
      it_pd = reinterpret_cast<float2 *>(const_cast<float *>(job.pre_items) + static_cast<size_t>(li) * 48 * pitch);
      it_dy = const_cast<float *>(job.pre_items) + static_cast<size_t>(li) * 48 * pitch + static_cast<size_t>(32) * pitch;
      S3 = job.pre_S + static_cast<size_t>(li) * 3 * pitch;
      S_pitch = pitch;
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
    } else {
      for (int r = 0; r < rounds; r++) {
        const int f = (r * kWaves + wave) * 64 + lane;
        if (f >= nf) continue;
        const int flag = s_flag[f];
        double fpx, fpy;
        F.pos(f, &fpx, &fpy);
        // (a feature seen at a coarser level cannot fail a finer level's border test: u doubles exactly, W_l = W_{l-1} / 2 rounds
        //  down — so "visible" is simply "passes this level's test", image_align.cc:229-233)
        double sums[3] = {0.0, 0.0, 0.0};
        const bool vis = ia_precompute_feature(ref_img, W, H, scale, fpx, fpy, (flag & 1) != 0, f, pitch, it_pd, it_dy, sums);
        s_S[f] = sums[0];
        s_S[max_f + f] = sums[1];
        s_S[2 * max_f + f] = sums[2];
        s_flag[f] = static_cast<uint8_t>((flag & ~kIaVis) | (vis ? kIaVis : 0));
      }
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
        {
          uint32_t lo, hi;
          ia_load_row8(wp, 5, &lo, &hi);
#pragma unroll
          for (int k = 0; k < 4; k++) ra[k] = ia_byte(lo, k);
          ra[4] = ia_byte(hi, 0);
        }
        double A = 0.0, B = 0.0, c2 = 0.0;
#pragma unroll
        for (int y = 0; y < 4; y++) {
          uint32_t lo, hi;
          ia_load_row8(wp + static_cast<size_t>(y + 1) * W, 5, &lo, &hi);
#pragma unroll
          for (int k = 0; k < 4; k++) rb[k] = ia_byte(lo, k);
          rb[4] = ia_byte(hi, 0);
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const float intensity = w0 * ra[x] + w1 * ra[x + 1] + w2 * rb[x] + w3 * rb[x + 1];
            const float2 pd = it_pd[(y * 4 + x) * pitch + f];
            const float dy = it_dy[(y * 4 + x) * pitch + f];
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
      const bool rebuild_h = kPre ? changed : (it == 0 || changed);  // kPre: the level starts with the precomputed factor of its visible set
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
          ia_feature_h(s_x[f], s_x[max_f + f], s_x[3 * max_f + f], fl, S3[f], S3[S_pitch + f], S3[2 * S_pitch + f], h16, h8);
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

template <int kWaves, bool kGlobalItems>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 8))) void image_align_wave_kernel(const IaJob *__restrict__ jobs,
                                                                       const sdvl_align_feature *__restrict__ feats_all, Cam cam,
                                                                       sdvl_align_params prm, int max_f, sdvl_align_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  const IaJob &jb = jobs[blockIdx.x];
  const IaIn in{jb.ref_level, jb.cur_level, jb.lw, jb.lh, jb.n_feat, jb.T, jb.patch_cache, out + jb.out_index};
  ia_wave_body<kWaves, kGlobalItems, false>(in, IaRecordFeats{feats_all + jb.feat_begin}, cam, prm, max_f, s_dyn);
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

template <class Feats>
__device__ __forceinline__ void ia_pre_body(const uint8_t *ref_img, int W, int H, int level, int nf, const Feats F, double fx, int pitch, IaPreOut o) {
  __shared__ double s_redH[4][24];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float scale = 1.0f / (1 << level);
  const double fl = fx / (1 << level);
  float2 *it_pd = reinterpret_cast<float2 *>(o.items);
  float *it_dy = o.items + static_cast<size_t>(32) * pitch;
  double h16[16], h8[8];
#pragma unroll
  for (int i = 0; i < 16; i++) h16[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 8; i++) h8[i] = 0.0;
  for (int f = tid; f < pitch; f += 256) {
    bool vis = false;
    double sums[3] = {0.0, 0.0, 0.0};
    if (f < nf) {
      V3 xyz;
      const bool valid = F.load(f, &xyz);
      double fpx, fpy;
      F.pos(f, &fpx, &fpy);
      vis = ia_precompute_feature(ref_img, W, H, scale, fpx, fpy, valid, f, pitch, it_pd, it_dy, sums);
      if (vis) {
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
  __syncthreads();
  if (wave == 0) {
    double Hm[36];
    int k = 0;
#pragma unroll
    for (int rr = 0; rr < 6; rr++)
#pragma unroll
      for (int c = rr; c < 6; c++) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < 4; w++) sum += s_redH[w][k];
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
  ia_pre_body(jb.ref_level[level], jb.lw[level], jb.lh[level], level, jb.n_feat, IaRecordFeats{feats_all + jb.feat_begin}, cam.fx, pitch,
              ia_pre_block(pre, n_jobs, n_lv, pitch, job, li));
}

template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 8))) void image_align_wave_pre_kernel(
    const IaJob *__restrict__ jobs, const sdvl_align_feature *__restrict__ feats_all, Cam cam, sdvl_align_params prm, int n_jobs, int max_f, void *pre,
    sdvl_align_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  const IaJob &jb = jobs[blockIdx.x];
  const int n_lv = prm.max_level - prm.min_level + 1;
  const IaPreOut o = ia_pre_block(pre, n_jobs, n_lv, max_f, static_cast<int>(blockIdx.x), 0);
  IaIn in{jb.ref_level, jb.cur_level, jb.lw, jb.lh, jb.n_feat, jb.T, nullptr, out + jb.out_index};
  in.pre_items = o.items; in.pre_S = o.S; in.pre_vis = o.vis; in.pre_fac = o.fac; in.pre_pitch = max_f;
  ia_wave_body<kWaves, true, true>(in, IaRecordFeats{feats_all + jb.feat_begin}, cam, prm, max_f, s_dyn);
}

// The alignment of a tracked step straight from the tracking tables (sdvl_track.hip): job j = tracker record j of the step; its
// features are the rows of last_frame's feature buffer, its items live in slice j of the work buffer (`item_pitch` features each).
// No IaJob records, no feature records, no launch in between (track_align_prep + one stage_push per group-step until round 3).
// kGlobalItems = false (round 5, SDVL_IA_TRACK_FUSED=1): the reference items of the level being optimised live in LDS (48 floats per feature)
// and PrecomputePatches runs inside the chain — no image_align_pre launch, nothing handed through L2 / HBM between two kernels
template <int kWaves, bool kGlobalItems = true>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 8))) void image_align_track_kernel(
    const TrackJobDev *__restrict__ jobs, const TrackPoint *__restrict__ points, const TrackFeat *__restrict__ feats0, const TrackFeat *__restrict__ feats1,
    int np, int nfeat_cap, Cam cam, sdvl_align_params prm, int max_f, float *__restrict__ items, sdvl_align_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  const TrackJobDev &jb = jobs[blockIdx.x];
  const IaIn in{jb.last_level, jb.cur.level, jb.cur.lw, jb.cur.lh, jb.n_feat, jb.T0,
                kGlobalItems ? items + static_cast<size_t>(blockIdx.x) * 48 * static_cast<size_t>(max_f) : nullptr, out + blockIdx.x};
  const IaTableFeats F{(jb.feat_buf ? feats1 : feats0) + static_cast<size_t>(jb.tracker) * nfeat_cap, points + static_cast<size_t>(jb.tracker) * np,
                       se3_inverse(se3_from7(jb.last_pose)).t};
  ia_wave_body<kWaves, kGlobalItems, false>(in, F, cam, prm, max_f, s_dyn);
}

// the same two kernels for a tracked step: features out of the tracking tables
__device__ __forceinline__ IaTableFeats ia_table_feats(const TrackJobDev &jb, const TrackPoint *points, const TrackFeat *feats0, const TrackFeat *feats1,
                                                       int np, int nfeat_cap) {
  return IaTableFeats{(jb.feat_buf ? feats1 : feats0) + static_cast<size_t>(jb.tracker) * nfeat_cap, points + static_cast<size_t>(jb.tracker) * np,
                      se3_inverse(se3_from7(jb.last_pose)).t};
}

__global__ __launch_bounds__(256) void image_align_track_pre_kernel(const TrackJobDev *__restrict__ jobs, const TrackPoint *__restrict__ points,
                                                                    const TrackFeat *__restrict__ feats0, const TrackFeat *__restrict__ feats1, int np,
                                                                    int nfeat_cap, Cam cam, sdvl_align_params prm, int n_jobs, int pitch, void *pre) {
  const int n_lv = prm.max_level - prm.min_level + 1;
  const int job = static_cast<int>(blockIdx.x) / n_lv, li = static_cast<int>(blockIdx.x) - job * n_lv, level = prm.max_level - li;
  const TrackJobDev &jb = jobs[job];
  ia_pre_body(jb.last_level[level], jb.cur.lw[level], jb.cur.lh[level], level, jb.n_feat, ia_table_feats(jb, points, feats0, feats1, np, nfeat_cap), cam.fx,
              pitch, ia_pre_block(pre, n_jobs, n_lv, pitch, job, li));
}

template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(2, 8))) void image_align_track_wave_pre_kernel(
    const TrackJobDev *__restrict__ jobs, const TrackPoint *__restrict__ points, const TrackFeat *__restrict__ feats0, const TrackFeat *__restrict__ feats1,
    int np, int nfeat_cap, Cam cam, sdvl_align_params prm, int n_jobs, int max_f, void *pre, sdvl_align_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  const TrackJobDev &jb = jobs[blockIdx.x];
  const int n_lv = prm.max_level - prm.min_level + 1;
  const IaPreOut o = ia_pre_block(pre, n_jobs, n_lv, max_f, static_cast<int>(blockIdx.x), 0);
  IaIn in{jb.last_level, jb.cur.level, jb.cur.lw, jb.cur.lh, jb.n_feat, jb.T0, nullptr, out + blockIdx.x};
  in.pre_items = o.items; in.pre_S = o.S; in.pre_vis = o.vis; in.pre_fac = o.fac; in.pre_pitch = max_f;
  ia_wave_body<kWaves, true, true>(in, ia_table_feats(jb, points, feats0, feats1, np, nfeat_cap), cam, prm, max_f, s_dyn);
}

size_t ia_wave_lds_bytes(int max_f, bool global_items) {
  return static_cast<size_t>(max_f) * (7 * sizeof(double) + (global_items ? 0 : 48 * sizeof(float)) + 1) + 64;
}
size_t ia_wave_work_bytes(int nf) { return (static_cast<size_t>((nf + 63) / 64 * 64) * 48 * sizeof(float) + 255) / 256 * 256; }

size_t ia_lds_bytes(int max_f) {
  return static_cast<size_t>(max_f) * (12 * sizeof(double) + 16 * sizeof(IaItem) + 2 * sizeof(int) + 4 * sizeof(float) + 5 * sizeof(double) + 4) + 64;
}
size_t ia_spill_lds_bytes(int max_f) { return static_cast<size_t>(max_f) * (2 * sizeof(int) + 4 * sizeof(float) + 5 * sizeof(double) + 4) + 64; }
size_t ia_spill_work_bytes(int nf) { return (static_cast<size_t>(nf) * (12 * sizeof(double) + 16 * sizeof(IaItem)) + 255) / 256 * 256; }

}  // namespace

// Queues the alignment of n_jobs frame pairs (no wait).  The feature records come from the host (`features`, staged and
// copied here) or already sit in HBM (`d_features`, written by sdvl_track.hip).  Results go to `d_results` when given (device
// memory, nothing returns to the host), otherwise to the context's result buffers for sdvl_image_align_end.
int sdvl_image_align_enqueue(sdvl_ctx *ctx, int n_jobs, const sdvl_align_job *jobs, int n_features, const sdvl_align_feature *features,
                             const sdvl_align_feature *d_features, const sdvl_camera *cam, const sdvl_align_params *p,
                             sdvl_align_result *d_results) {
  SDVL_REQUIRE(ctx, p->patch_size == 4, "only align_patch_size 4 is supported");
  SDVL_REQUIRE(ctx, p->min_level >= 0 && p->max_level >= p->min_level && p->max_level < SDVL_MAX_LEVELS, "bad align levels");
  SDVL_REQUIRE(ctx, p->max_its >= 0, "bad max_its");
  // Jobs whose features fit the LDS-resident kernel run there; the (few) larger ones of the same call go to the global-memory
  // kernel in a second launch.  One oversized job used to send the whole batch to the slow kernel: with 256 trackers per
  // launch there is almost always a fresh keyframe with more than 384 features among them.
  // round 4: the lane-per-feature kernels (image_align_wave_kernel) are the default; SDVL_IA_WAVE=0 brings the round-3 workgroup
  // kernels back (A/B), SDVL_IA_WAVES=2|4 gives the LDS-sized jobs more than one wave
  static const bool wave_form = !(getenv("SDVL_IA_WAVE") && atoi(getenv("SDVL_IA_WAVE")) == 0);
  // The tracking-sized jobs keep their reference items in the work buffer too (L2-resident: 36 KB per job, every load coalesced
  // over the wave's 64 features), not in LDS: alone the kernel takes the same 160-165 us per 256 jobs either way, but a workgroup
  // that asks for 48 KB of LDS waits for a CU whose LDS the neighbours' small workgroups (fast_cells: 27 x 5.8 KB per CU) keep
  // refilling — among the other streams' kernels 240 us per dispatch with 12 KB of LDS, 290 with 48 (the round-3 kernel: 74 KB,
  // 290-320 us).  SDVL_IA_GLOBAL_ITEMS=0: items in LDS (A/B).
  static const bool small_global = !(getenv("SDVL_IA_GLOBAL_ITEMS") && atoi(getenv("SDVL_IA_GLOBAL_ITEMS")) == 0);
  const bool force_generic = getenv("SDVL_IMAGE_ALIGN_GENERIC") != nullptr;
  const bool legacy = force_generic || getenv("SDVL_IMAGE_ALIGN_LEGACY_BIG") != nullptr;  // the round-1 global-memory kernel for the big jobs
  // jobs up to this many features keep their caches in LDS (SDVL_IA_LDS_MAX_F: experiments with the LDS / L2 trade-off)
  static const int lds_max_f = [] {
    const char *e = getenv("SDVL_IA_LDS_MAX_F");
    const int v = e ? atoi(e) : kLdsMaxF;
    return v < 0 ? 0 : (v > kLdsMaxF ? kLdsMaxF : v);
  }();
  std::vector<int> order(n_jobs);
  int n_lds = 0, max_nf_lds = 0, max_nf_big = 0;
  size_t work = 0;
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
  {  // LDS-sized jobs first, the others behind them; results are written to each job's own slot, so the order is free
    int lo = 0, hi = n_jobs;
    for (int j = 0; j < n_jobs; j++) {
      const int nf = jobs[j].feat_end - jobs[j].feat_begin;
      if (!force_generic && nf <= lds_max_f) {
        order[lo++] = j;
        if (nf > max_nf_lds) max_nf_lds = nf;
        if (wave_form && small_global) work += ia_wave_work_bytes(nf);
      } else {
        order[--hi] = j;
        work += legacy ? (static_cast<size_t>(nf) * 16 * (sizeof(float) + 6 * sizeof(double)) + 255) / 256 * 256
                       : (wave_form ? ia_wave_work_bytes(nf) : ia_spill_work_bytes(nf));
        if (nf > max_nf_big) max_nf_big = nf;
      }
    }
    n_lds = lo;
  }
  const int n_gen = n_jobs - n_lds;
  // PrecomputePatches as a wide launch in front of the Gauss-Newton chains (image_align_pre_kernel); SDVL_IA_PRE=0: inside the chains
  static const bool pre_env = !(getenv("SDVL_IA_PRE") && atoi(getenv("SDVL_IA_PRE")) == 0);
  const bool pre = pre_env && wave_form && small_global && !legacy;
  const int n_lv = p->max_level - p->min_level + 1;
  static const int ia_waves_env = getenv("SDVL_IA_WAVES") ? atoi(getenv("SDVL_IA_WAVES")) : 1;
  const int kw_lds = ia_waves_env == 4 ? 4 : (ia_waves_env == 2 ? 2 : 1);
  const int max_f_lds = (max_nf_lds + 64 * kw_lds - 1) / (64 * kw_lds) * (64 * kw_lds) + (max_nf_lds == 0 ? 64 * kw_lds : 0);
  const int max_f_big = (max_nf_big + 255) / 256 * 256;
  const size_t pre_lds_bytes = pre && n_lds > 0 ? ia_pre_bytes(n_lds, n_lv, max_f_lds) : 0;
  if (pre) work = pre_lds_bytes + (n_gen > 0 ? ia_pre_bytes(n_gen, n_lv, max_f_big) : 0);
  const size_t job_bytes = (sizeof(IaJob) * n_jobs + 255) / 256 * 256;
  const size_t feat_bytes = d_features ? 0 : sizeof(sdvl_align_feature) * static_cast<size_t>(n_features);
  const size_t res_bytes = sizeof(sdvl_align_result) * n_jobs;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_ensure(ctx, &ctx->d_work, &ctx->d_work_bytes, work + 256, false);
  if (!rc && !d_results) rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, res_bytes, false);
  if (!rc && !d_results) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, res_bytes, true);
  if (!rc) rc = sdvl_stage_alloc(ctx, job_bytes + feat_bytes, &hs, &dsx);
  if (rc) return rc;
  IaJob *hj = static_cast<IaJob *>(hs);
  uint8_t *wbase = static_cast<uint8_t *>(ctx->d_work);
  size_t woff = 0;
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
    if (q >= n_lds && legacy) {
      const size_t items = static_cast<size_t>(d.n_feat) * 16;
      d.jac_cache = reinterpret_cast<double *>(wbase + woff);
      d.patch_cache = reinterpret_cast<float *>(wbase + woff + items * 6 * sizeof(double));
      woff += items * (sizeof(float) + 6 * sizeof(double));
      woff = (woff + 255) / 256 * 256;
    } else if (wave_form && (q >= n_lds || small_global)) {  // the items of the job: {patch, dx}[16][pitch], dy[16][pitch], pitch = n_feat rounded up to 64
      d.jac_cache = nullptr;
      d.patch_cache = reinterpret_cast<float *>(wbase + woff);
      woff += ia_wave_work_bytes(d.n_feat);
    } else if (q >= n_lds) {  // Jacobians [n_feat][12] doubles, then the items [n_feat][16] {patch, dx, dy}
      d.jac_cache = reinterpret_cast<double *>(wbase + woff);
      d.patch_cache = reinterpret_cast<float *>(wbase + woff + static_cast<size_t>(d.n_feat) * 12 * sizeof(double));
      woff += ia_spill_work_bytes(d.n_feat);
    }
  }
  if (feat_bytes) memcpy(static_cast<uint8_t *>(hs) + job_bytes, features, feat_bytes);
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, job_bytes + feat_bytes));
  const sdvl_align_feature *feats_dev = d_features ? d_features : reinterpret_cast<const sdvl_align_feature *>(static_cast<uint8_t *>(dsx) + job_bytes);
  Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  const bool direct = sdvl_direct_results();
  sdvl_align_result *dst = d_results ? d_results : static_cast<sdvl_align_result *>(direct ? ctx->h_out : ctx->d_out);
  {
    // the attribute belongs to the kernel object of ONE device: set it once per device, whichever thread gets there first
    static std::atomic<unsigned long long> attr_devices{0};
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
      SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_lds_kernel<false, 512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_lds_bytes(kLdsMaxF + 16))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_lds_kernel<false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_lds_bytes(kLdsMaxF + 16))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_lds_kernel<false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_lds_bytes(kLdsMaxF + 16))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_lds_kernel<true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_spill_lds_bytes(kMaxF + 16))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_wave_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kLdsMaxF, false))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_wave_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kLdsMaxF, false))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_wave_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kLdsMaxF + 128, false))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_wave_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kMaxF, true))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_wave_pre_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kMaxF, true))));
      attr_devices.fetch_or(bit, std::memory_order_release);
    }
  }
  if (n_lds > 0 && wave_form) {
    const int kw = kw_lds;
    const int max_f = max_f_lds;
    const size_t lds = ia_wave_lds_bytes(max_f, small_global);
    if (pre)
      SDVL_LAUNCH(ctx, "image_align_pre", image_align_pre_kernel, dim3(static_cast<unsigned>(n_lds) * n_lv), dim3(256), static_cast<const IaJob *>(dsx), feats_dev,
                  c, *p, n_lds, max_f, ctx->d_work);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "image_align", &ev_a, &ev_b);
    if (pre && kw == 4)
      hipExtLaunchKernelGGL((image_align_wave_pre_kernel<4>), dim3(n_lds), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx), feats_dev, c,
                            *p, n_lds, max_f, ctx->d_work, dst);
    else if (pre && kw == 2)
      hipExtLaunchKernelGGL((image_align_wave_pre_kernel<2>), dim3(n_lds), dim3(128), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx), feats_dev, c,
                            *p, n_lds, max_f, ctx->d_work, dst);
    else if (pre)
      hipExtLaunchKernelGGL((image_align_wave_pre_kernel<1>), dim3(n_lds), dim3(64), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx), feats_dev, c,
                            *p, n_lds, max_f, ctx->d_work, dst);
    else if (small_global && kw == 4)
      hipExtLaunchKernelGGL((image_align_wave_kernel<4, true>), dim3(n_lds), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
    else if (small_global && kw == 2)
      hipExtLaunchKernelGGL((image_align_wave_kernel<2, true>), dim3(n_lds), dim3(128), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
    else if (small_global)
      hipExtLaunchKernelGGL((image_align_wave_kernel<1, true>), dim3(n_lds), dim3(64), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
    else if (kw == 4)
      hipExtLaunchKernelGGL((image_align_wave_kernel<4, false>), dim3(n_lds), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
    else if (kw == 2)
      hipExtLaunchKernelGGL((image_align_wave_kernel<2, false>), dim3(n_lds), dim3(128), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
    else
      hipExtLaunchKernelGGL((image_align_wave_kernel<1, false>), dim3(n_lds), dim3(64), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
  } else if (n_lds > 0) {
    const int max_f = (max_nf_lds + 7) / 8 * 8 + 8;
    const size_t lds = ia_lds_bytes(max_f);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "image_align", &ev_a, &ev_b);
    // Round 3: 256 threads per job instead of 512.  The kernel holds ~235 VGPRs per lane: eight waves of it are a whole CU's register
    // file, so a launch of 256 jobs kept all 256 CUs to itself for its ~100 us (one slow chain of dependent phases per job) and
    // nothing of the other streams' work ran beside it.  Four waves take half of every SIMD's registers: the job takes ~15 % longer
    // alone, the other half of the CU keeps working — 289 k -> 298 k tracked frames/s, 307 k together with the same change in
    // pose_hypotheses (one box, alternating runs).  128 threads: 296 k.  SDVL_IA_THREADS=512 / 128 select the other forms (A/B).
    static const int ia_threads = getenv("SDVL_IA_THREADS") ? atoi(getenv("SDVL_IA_THREADS")) : 256;
    if (ia_threads == 128)
      hipExtLaunchKernelGGL((image_align_lds_kernel<false, 128>), dim3(n_lds), dim3(128), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
    else if (ia_threads == 256)
      hipExtLaunchKernelGGL((image_align_lds_kernel<false, 256>), dim3(n_lds), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
    else
      hipExtLaunchKernelGGL((image_align_lds_kernel<false, 512>), dim3(n_lds), dim3(512), lds, ctx->stream, ev_a, ev_b, 0, static_cast<const IaJob *>(dsx),
                            feats_dev, c, *p, max_f, dst);
  }
  if (n_gen > 0 && pre) {
    void *pre_big = static_cast<uint8_t *>(ctx->d_work) + pre_lds_bytes;
    SDVL_LAUNCH(ctx, "image_align_pre", image_align_pre_kernel, dim3(static_cast<unsigned>(n_gen) * n_lv), dim3(256), static_cast<const IaJob *>(dsx) + n_lds,
                feats_dev, c, *p, n_gen, max_f_big, pre_big);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "image_align_big", &ev_a, &ev_b);
    hipExtLaunchKernelGGL((image_align_wave_pre_kernel<4>), dim3(n_gen), dim3(256), ia_wave_lds_bytes(max_f_big, true), ctx->stream, ev_a, ev_b, 0,
                          static_cast<const IaJob *>(dsx) + n_lds, feats_dev, c, *p, n_gen, max_f_big, pre_big, dst);
  } else if (n_gen > 0 && wave_form && !legacy) {
    const int max_f = (max_nf_big + 255) / 256 * 256;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "image_align_big", &ev_a, &ev_b);
    hipExtLaunchKernelGGL((image_align_wave_kernel<4, true>), dim3(n_gen), dim3(256), ia_wave_lds_bytes(max_f, true), ctx->stream, ev_a, ev_b, 0,
                          static_cast<const IaJob *>(dsx) + n_lds, feats_dev, c, *p, max_f, dst);
  } else if (n_gen > 0 && legacy) {
    SDVL_LAUNCH(ctx, "image_align_big", image_align_kernel, dim3(n_gen), dim3(kThreads), static_cast<const IaJob *>(dsx) + n_lds, feats_dev, c, *p, dst);
  } else if (n_gen > 0) {
    const int max_f = (max_nf_big + 7) / 8 * 8 + 8;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    sdvl_timer_events(ctx, "image_align_big", &ev_a, &ev_b);
    hipExtLaunchKernelGGL((image_align_lds_kernel<true, 512>), dim3(n_gen), dim3(512), ia_spill_lds_bytes(max_f), ctx->stream, ev_a, ev_b, 0,
                          static_cast<const IaJob *>(dsx) + n_lds, feats_dev, c, *p, max_f, dst);
  }
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  if (!d_results && !direct) SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
  return SDVL_OK;
}

// The alignment of a tracked step (sdvl_track_align): the jobs are the step's TrackJobDev records, already in HBM; features come
// out of the tracking tables inside the kernel.  max_nf = the largest feature count among the jobs.
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
  // configuration C's ~850 features per job: four waves share them.  Round 5: so do the jobs of a SMALL batch (a lone camera's
  // HandleFrame): one feature per lane instead of three rounds per lane shortens every Gauss-Newton evaluation of a chain that has the
  // chip to itself; a farm's launches of hundreds of jobs keep one wave per job (fewer instructions in total).  SDVL_IA_SMALL_WAVES=1|4
  static const int small_kw = getenv("SDVL_IA_SMALL_WAVES") ? atoi(getenv("SDVL_IA_SMALL_WAVES")) : 4;
  const int kw = (max_nf > kLdsMaxF || (batch_size <= 32 && small_kw == 4)) ? 4 : 1;
  const int max_f = max_nf <= 0 ? 64 * kw : (max_nf + 64 * kw - 1) / (64 * kw) * (64 * kw);
  // PrecomputePatches of all levels as a wide launch in front of the Gauss-Newton chains (SDVL_IA_PRE=0: inside the chain, A/B)
  // SDVL_IA_TRACK_FUSED=1 (round 5, measured): no precompute launch and the items of the level at hand in LDS — jobs of up to
  // kLdsMaxF features only (48 floats per feature: 47 KB of LDS for S-A's 192)
  static const bool fused_env = getenv("SDVL_IA_TRACK_FUSED") && atoi(getenv("SDVL_IA_TRACK_FUSED")) != 0;
  const bool fused = fused_env && max_nf <= kLdsMaxF;
  static const bool pre_env = !(getenv("SDVL_IA_PRE") && atoi(getenv("SDVL_IA_PRE")) == 0);
  const bool pre = pre_env && !fused;
  const int n_lv = p->max_level - p->min_level + 1;
  const size_t work = pre ? ia_pre_bytes(n_jobs, n_lv, max_f) : static_cast<size_t>(n_jobs) * 48 * sizeof(float) * max_f;
  const int rc = sdvl_ensure(ctx, &ctx->d_work, &ctx->d_work_bytes, work + 256, false);
  if (rc) return rc;
  {
    static std::atomic<unsigned long long> attr_devices{0};
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
      SDVL_HIP_CHECK(ctx, sdvl_bind_device(ctx));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_track_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kMaxF, true))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_track_wave_pre_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kMaxF, true))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_track_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kLdsMaxF, false))));
      SDVL_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(image_align_track_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(ia_wave_lds_bytes(kLdsMaxF + 128, false))));
      attr_devices.fetch_or(bit, std::memory_order_release);
    }
  }
  const Cam c{cam->width, cam->height, cam->fx, cam->fy, cam->u0, cam->v0};
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  sdvl_timer_events(ctx, max_nf > kLdsMaxF ? "image_align_big" : "image_align", &ev_a, &ev_b);
  const size_t lds = ia_wave_lds_bytes(max_f, !fused);
  if (fused) {
    if (kw == 4)
      hipExtLaunchKernelGGL((image_align_track_kernel<4, false>), dim3(n_jobs), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, d_jobs, d_points, d_feats0, d_feats1, np,
                            nfeat_cap, c, *p, max_f, static_cast<float *>(nullptr), d_results);
    else
      hipExtLaunchKernelGGL((image_align_track_kernel<1, false>), dim3(n_jobs), dim3(64), lds, ctx->stream, ev_a, ev_b, 0, d_jobs, d_points, d_feats0, d_feats1, np,
                            nfeat_cap, c, *p, max_f, static_cast<float *>(nullptr), d_results);
  } else if (pre) {
    SDVL_LAUNCH(ctx, "image_align_pre", image_align_track_pre_kernel, dim3(static_cast<unsigned>(n_jobs) * n_lv), dim3(256), d_jobs, d_points, d_feats0, d_feats1,
                np, nfeat_cap, c, *p, n_jobs, max_f, ctx->d_work);
    if (kw == 4)
      hipExtLaunchKernelGGL((image_align_track_wave_pre_kernel<4>), dim3(n_jobs), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, d_jobs, d_points, d_feats0, d_feats1,
                            np, nfeat_cap, c, *p, n_jobs, max_f, ctx->d_work, d_results);
    else
      hipExtLaunchKernelGGL((image_align_track_wave_pre_kernel<1>), dim3(n_jobs), dim3(64), lds, ctx->stream, ev_a, ev_b, 0, d_jobs, d_points, d_feats0, d_feats1,
                            np, nfeat_cap, c, *p, n_jobs, max_f, ctx->d_work, d_results);
  } else if (kw == 4)
    hipExtLaunchKernelGGL((image_align_track_kernel<4>), dim3(n_jobs), dim3(256), lds, ctx->stream, ev_a, ev_b, 0, d_jobs, d_points, d_feats0, d_feats1, np,
                          nfeat_cap, c, *p, max_f, static_cast<float *>(ctx->d_work), d_results);
  else
    hipExtLaunchKernelGGL((image_align_track_kernel<1>), dim3(n_jobs), dim3(64), lds, ctx->stream, ev_a, ev_b, 0, d_jobs, d_points, d_feats0, d_feats1, np,
                          nfeat_cap, c, *p, max_f, static_cast<float *>(ctx->d_work), d_results);
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
