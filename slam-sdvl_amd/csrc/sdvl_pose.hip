// sdvl_pose.hip — K8 pose from matches: FeatureAlign::SelectInliers (RANSAC, feature_align.cc:152-216) and
// FeatureAlign::OptimizePose / RescueOutliers (feature_align.cc:73-82,218-243) around ConvergePose (:341-421) and
// CheckReprojectionError (:258-283), for a batch of frames.  SURVEY §8(f) "next" row 1.
//
//   pose_hypotheses_kernel : one workgroup per (frame, 128 RANSAC draws).  The reference evaluates draws one after the other
//       and adapts the iteration budget as it goes; a draw's result does not depend on earlier draws, so all
//       max_ransac_its draws are evaluated at once and the sequential bookkeeping is replayed afterwards.  Phase 1: one LANE
//       per draw converges the 5-point pose in registers; phase 2: the (draw, match) pairs of the supporter counts are
//       dealt to all 512 lanes (integer sums: order-free).
//   pose_refine_kernel     : one WAVE per frame.  Replays the RANSAC loop over the draw results (iteration budget from a
//       host-computed table, so no device log()), then inliers / Tukey-weighted Gauss-Newton / rescue / second pass.
//       Per-observation work (projection, Jacobian, weight, 28 normal-equation terms) is lane-parallel; lanes 0..27 each
//       add one term IN OBSERVATION ORDER — the sums carry the rounding of the reference's sequential loop; inlier and
//       outlier lists keep the reference's order through ballot-prefix appends; the median is found by rank counting.
// FP64 throughout, -ffp-contract=off, same expression order as host/sdvl_host.cc (which is the reference's).
// Only sin/cos inside SE3::Exp differ from the host libm by <= 1 ulp (tolerance class, like image alignment).
#include "sdvl_internal.h"
#include <algorithm>

#include "sdvl_math.h"

namespace {

using namespace sdvl;

constexpr int kMaxObs = 1024;  // observations (matched features) per frame handled on device (config C: max_matches 1000)
constexpr double kMADNorm = 1.4826;
constexpr double kTukeyC = 4.6851 * 4.6851;

struct HypResult {
  double se3[7];
  int ok, supporters;
};

__device__ __forceinline__ double tukey(double x) {  // feature_align.cc:423-431
  const double x_square = x * x;
  if (x_square <= kTukeyC) {
    const double tmp = 1.0 - x_square / kTukeyC;
    return tmp * tmp;
  }
  return 0.0;
}

// scaled reprojection error of one observation under (R, t): feature_align.cc:270-274
__device__ __forceinline__ void reproj_error(const sdvl_pose_obs &o, const M3 &R, const V3 &t, double *ex, double *ey, V3 *pos_out) {
  const V3 pos = vadd(mvec(R, {o.px, o.py, o.pz}), t);
  double x = o.ax - pos.x / pos.z, y = o.ay - pos.y / pos.z;
  x *= o.inv_cov;
  y *= o.inv_cov;
  *ex = x;
  *ey = y;
  *pos_out = pos;
}

// ---------------------------------------------------------------------------------------------- hypotheses (lane each)
// ConvergePose over `npts` observations idx[0..npts) (npts <= 8), all in registers / small local arrays
// kCache (round 6): the draw's observations are copied ONCE into the lane's column of an LDS block [npts][6][64] and read from there —
// from global memory they were a dependent load per point and iteration (the loops over a run-time point count are not unrolled):
// up to 5 x 10 memory round trips on a chain that has one wave per SIMD to hide them behind.
template <bool kCache>
__device__ bool converge_pose_small(const sdvl_pose_obs *obs, const int *idx, int npts, const Rigid &frame_pose, double fx, int max_its,
                                    Rigid *se3, double *cache = nullptr, int lane = 0) {
  if (kCache) {
    for (int q = 0; q < npts; q++) {
      const sdvl_pose_obs o = obs[idx[q]];
      double *c = cache + (q * 6) * 64 + lane;
      c[0] = o.px; c[64] = o.py; c[128] = o.pz; c[192] = o.ax; c[256] = o.ay; c[320] = o.inv_cov;
    }
  }
  const auto ob = [&](int q) {
    if (kCache) {
      const double *c = cache + (q * 6) * 64 + lane;
      sdvl_pose_obs o;
      o.px = c[0]; o.py = c[64]; o.pz = c[128]; o.ax = c[192]; o.ay = c[256]; o.inv_cov = c[320];
      return o;
    }
    return obs[idx[q]];
  };
  Rigid last = frame_pose;
  *se3 = last;
  double chi2 = 0.0;
  double errs[8];
  {
    const M3 R = se3_rot(*se3);
    for (int q = 0; q < npts; q++) {
      double ex, ey;
      V3 pos;
      reproj_error(ob(q), R, se3->t, &ex, &ey, &pos);
      errs[q] = sqrt(ex * ex + ey * ey);
    }
  }
  if (npts == 0) return false;
  // GetMedianVector: element floor(n/2) of the sorted order (extra/utils.cc:215-220) — insertion sort of <= 8 values
  for (int i = 1; i < npts; i++) {
    const double v = errs[i];
    int j = i - 1;
    while (j >= 0 && errs[j] > v) { errs[j + 1] = errs[j]; j--; }
    errs[j + 1] = v;
  }
  double scale = kMADNorm * errs[npts / 2];
  for (int i = 0; i < max_its; i++) {
    double A[36], b[6];
#pragma unroll
    for (int r = 0; r < 6; r++) b[r] = 0.0;
#pragma unroll
    for (int r = 0; r < 36; r++) A[r] = 0.0;
    double new_chi2 = 0.0;
    if (i == 5) scale = 0.85 / fx;
    const M3 R = se3_rot(*se3);
    for (int q = 0; q < npts; q++) {
      double ex, ey;
      V3 pos;
      const sdvl_pose_obs o = ob(q);
      reproj_error(o, R, se3->t, &ex, &ey, &pos);
      double J[12];
      jacobian_3d_to_plane(pos, J);
      const double ic = o.inv_cov;
#pragma unroll
      for (int c = 0; c < 12; c++) J[c] *= ic;
      const double weight = tukey(sqrt(ex * ex + ey * ey) / scale);
      // A is symmetric term by term (products commute), so only its upper triangle is accumulated; mirrored before the solve
#pragma unroll
      for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int c = r; c < 6; c++) A[6 * r + c] += (J[r] * J[c] + J[6 + r] * J[6 + c]) * weight;
        b[r] -= (J[r] * ex + J[6 + r] * ey) * weight;
      }
      new_chi2 += (ex * ex + ey * ey) * weight;
    }
#pragma unroll
    for (int r = 1; r < 6; r++)
#pragma unroll
      for (int c = 0; c < r; c++) A[6 * r + c] = A[6 * c + r];
    double dT[6];
    ldlt_solve6_reg(A, b, dT);
    if ((i > 0 && new_chi2 > chi2) || dT[0] != dT[0]) {
      *se3 = last;
      break;
    }
    const Rigid T_new = se3_mul(se3_exp(dT), *se3);
    last = *se3;
    *se3 = T_new;
    chi2 = new_chi2;
    if (abs_max6(dT) <= 1e-10) break;
  }
  return true;
}

// Round 3: two launches.  pose_hypotheses_kernel converges the draws, one LANE per draw (a dependent FP64 chain of ~105 us; at ~180
// VGPRs per lane its waves are few and narrow: 64-thread workgroups, two per frame for 100 draws), and
// pose_supporters_kernel counts every draw's supporters over all matches with one wave per (draw, 256 matches) — 100 x 1000 pairs
// per frame in configuration C took 250 of the 370 us when the two waves of the first kernel walked them alone.
constexpr int kHypDraws = 64;     // draws of one workgroup = lanes of its wave
constexpr int kSupChunk = 256;    // matches one supporter wave tests against its draw (4 per lane)

__global__ __launch_bounds__(kHypDraws) void pose_hypotheses_kernel(const PoseJobDev *__restrict__ jobs, const sdvl_pose_obs *__restrict__ obs_all,
                                                                    const int32_t *__restrict__ rand_idx, sdvl_pose_params prm,
                                                                    HypResult *__restrict__ hyp) {
  const PoseJobDev &job = jobs[blockIdx.y];
  const int h = blockIdx.x * kHypDraws + threadIdx.x;
  if (h >= prm.max_ransac_its) return;
  const int size = job.n_obs;
  const sdvl_pose_obs *obs = obs_all + job.obs_begin;
  // ConvergePose of this draw (feature_align.cc:176-197)
  HypResult r;
  r.ok = 0;
  r.supporters = 0;
  for (int k = 0; k < 7; k++) r.se3[k] = job.pose[k];
  if (size > 0) {
    const int npoints = min(prm.max_ransac_points, size);
    int index = rand_idx[job.rand_begin + h];  // rand() % size, drawn on the host (feature_align.cc:180) ...
    if (prm.pad_ & 1) index %= size;           // ... or the raw rand() value when the host could not know `size` yet
    int sel[8];
    for (int i = 0; i < npoints; i++) sel[i] = (index + i) % size;
    Rigid se3;
    extern __shared__ double s_hyp_cache[];  // [max_ransac_points][6][64]
    if (converge_pose_small<true>(obs, sel, npoints, se3_from7(job.pose), prm.fx, prm.max_optim_pose_its, &se3, s_hyp_cache, threadIdx.x)) {
      r.ok = 1;
      se3_to7(se3, r.se3);
    }
  }
  HypResult &dst = hyp[static_cast<size_t>(blockIdx.y) * prm.max_ransac_its + h];
  for (int k = 0; k < 7; k++) dst.se3[k] = r.se3[k];
  dst.ok = r.ok;
  dst.supporters = 0;  // pose_supporters_kernel adds to it
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Round 6: ONE WAVE per draw, for the sets whose launches leave the chip empty (a lone camera: sets of <= 4 trackers; configuration C's
// groups of 16 — 1600 such waves per launch, four groups at a time — lose 10 % on it and keep the lane form).  The lane
// form above makes a draw a chain of ~27 k dependent vector instructions (five points one after the other, the pivoted LDLT with every
// swap predicated, and a wave runs until its SLOWEST draw has converged); here lanes 0..npts-1 take one point each, lanes 0..27 add the
// points' normal-equation terms IN POINT ORDER (the rounding of the sequential loop, feature_align.cc:370-400), the solve and the SE3
// update run uniformly (scalar pivot branches), the wave leaves as soon as ITS draw has converged — and counts the draw's supporters
// over all matches itself (feature_align.cc:190, 245-283: an integer sum), so pose_supporters_kernel is not launched behind this form.
// Same operations on the same operands in the same order as converge_pose_small: the draws' results are bit-identical.
struct HypWaveLds {
  double terms[8][29];
  double sums[28];
  double errs[8];
};

__device__ __forceinline__ void hyp_terms(const sdvl_pose_obs &o, const M3 &R, const V3 &t, double scale, double *row) {
  double ex, ey;
  V3 pos;
  reproj_error(o, R, t, &ex, &ey, &pos);
  double J[12];
  jacobian_3d_to_plane(pos, J);
#pragma unroll
  for (int c = 0; c < 12; c++) J[c] *= o.inv_cov;
  const double weight = tukey(sqrt(ex * ex + ey * ey) / scale);
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = r; c < 6; c++) row[k++] = (J[r] * J[c] + J[6 + r] * J[6 + c]) * weight;
#pragma unroll
  for (int r = 0; r < 6; r++) row[21 + r] = (J[r] * ex + J[6 + r] * ey) * weight;
  row[27] = (ex * ex + ey * ey) * weight;
}

__global__ __launch_bounds__(64) void pose_hypotheses_wave_kernel(const PoseJobDev *__restrict__ jobs, const sdvl_pose_obs *__restrict__ obs_all,
                                                                  const int32_t *__restrict__ rand_idx, sdvl_pose_params prm,
                                                                  HypResult *__restrict__ hyp) {
  __shared__ HypWaveLds L;
  const PoseJobDev &job = jobs[blockIdx.y];
  const int h = blockIdx.x, lane = threadIdx.x;
  const int size = job.n_obs;
  const sdvl_pose_obs *obs = obs_all + job.obs_begin;
  HypResult &dst = hyp[static_cast<size_t>(blockIdx.y) * prm.max_ransac_its + h];
  Rigid se3 = se3_from7(job.pose);
  int ok = 0, supporters = 0;
  if (size > 0) {
    const int npts = min(prm.max_ransac_points, size);
    int index = rand_idx[job.rand_begin + h];  // rand() % size, drawn on the host (feature_align.cc:180) ...
    if (prm.pad_ & 1) index %= size;           // ... or the raw rand() value when the host could not know `size` yet
    const bool mine = lane < npts;
    const sdvl_pose_obs o = obs[(index + (mine ? lane : 0)) % size];
    Rigid last = se3;
    double chi2 = 0.0;
    {
      const M3 R = se3_rot(se3);
      if (mine) {
        double ex, ey;
        V3 pos;
        reproj_error(o, R, se3.t, &ex, &ey, &pos);
        L.errs[lane] = sqrt(ex * ex + ey * ey);
      }
    }
    wave_lds_sync();
    // GetMedianVector: element floor(n/2) of the sorted order (extra/utils.cc:215-220) — the insertion sort of converge_pose_small
    // with every index a compile-time constant (the moves predicated), uniform across the wave
    double errs[8];
#pragma unroll
    for (int q = 0; q < 8; q++) errs[q] = q < npts ? L.errs[q] : 0.0;
#pragma unroll
    for (int i = 1; i < 8; i++) {
      if (i < npts) {
        const double v = errs[i];
        bool moving = true;
#pragma unroll
        for (int j = i - 1; j >= 0; j--) {
          const bool shift = moving && errs[j] > v;
          errs[j + 1] = shift ? errs[j] : (moving ? v : errs[j + 1]);
          moving = shift;
        }
        if (moving) errs[0] = v;
      }
    }
    double med = errs[0];
#pragma unroll
    for (int q = 1; q < 8; q++)
      if (q == npts / 2) med = errs[q];
    double scale = kMADNorm * med;
    for (int i = 0; i < prm.max_optim_pose_its; i++) {
      if (i == 5) scale = 0.85 / prm.fx;
      const M3 R = se3_rot(se3);
      if (mine) hyp_terms(o, R, se3.t, scale, L.terms[lane]);
      wave_lds_sync();
      if (lane < 28) {
        const bool neg = lane >= 21 && lane < 27;  // b accumulates with -= in the reference, and x - t == x + (-t) exactly
        double acc = 0.0;
        for (int q = 0; q < npts; q++) acc += neg ? -L.terms[q][lane] : L.terms[q][lane];
        L.sums[lane] = acc;
      }
      wave_lds_sync();
      double A[36], b[6], dT[6];
      {
        int t = 0;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = r; c < 6; c++) {
            A[6 * r + c] = L.sums[t];
            A[6 * c + r] = L.sums[t];
            t++;
          }
      }
#pragma unroll
      for (int r = 0; r < 6; r++) b[r] = L.sums[21 + r];
      const double new_chi2 = L.sums[27];
      wave_lds_sync();
      ldlt_solve6_reg<true>(A, b, dT);
      if ((i > 0 && new_chi2 > chi2) || dT[0] != dT[0]) {
        se3 = last;
        break;
      }
      const Rigid T_new = se3_mul(se3_exp(dT), se3);
      last = se3;
      se3 = T_new;
      chi2 = new_chi2;
      if (abs_max6(dT) <= 1e-10) break;
    }
    ok = 1;
    // CheckReprojectionError of this draw over every match
    const M3 R = se3_rot(se3);
    for (int q0 = 0; q0 < size; q0 += 64) {
      const int q = q0 + lane;
      bool in = false;
      if (q < size) {
        double ex, ey;
        V3 pos;
        reproj_error(obs[q], R, se3.t, &ex, &ey, &pos);
        in = sqrt(ex * ex + ey * ey) <= prm.inlier_threshold;
      }
      supporters += __popcll(__ballot(in));
    }
  }
  if (lane == 0) {
    se3_to7(se3, dst.se3);
    dst.ok = ok;
    dst.supporters = supporters;
  }
}

// CheckReprojectionError of every draw over every match (feature_align.cc:190, 245-283).  blockIdx.x = draw, blockIdx.y = chunk of
// kSupChunk matches, blockIdx.z = frame; a supporter count is an integer sum, so its order is free: one add per wave.
__global__ __launch_bounds__(64) void pose_supporters_kernel(const PoseJobDev *__restrict__ jobs, const sdvl_pose_obs *__restrict__ obs_all,
                                                             sdvl_pose_params prm, HypResult *__restrict__ hyp) {
  const PoseJobDev &job = jobs[blockIdx.z];
  const int size = job.n_obs;
  const int q0 = static_cast<int>(blockIdx.y) * kSupChunk;
  if (q0 >= size) return;
  HypResult &H = hyp[static_cast<size_t>(blockIdx.z) * prm.max_ransac_its + blockIdx.x];
  if (!H.ok) return;
  const Rigid se3 = se3_from7(H.se3);
  const M3 R = se3_rot(se3);
  const sdvl_pose_obs *obs = obs_all + job.obs_begin;
  const int lane = threadIdx.x;
  int mine = 0;
#pragma unroll
  for (int u = 0; u < kSupChunk / 64; u++) {
    const int q = q0 + u * 64 + lane;
    if (q < size) {
      double ex, ey;
      V3 pos;
      reproj_error(obs[q], R, se3.t, &ex, &ey, &pos);
      mine += sqrt(ex * ex + ey * ey) <= prm.inlier_threshold ? 1 : 0;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if (lane == 0 && mine > 0) atomicAdd(&H.supporters, mine);
}

// ---------------------------------------------------------------------------------------------- refinement (wave each)
// Round 3: pose_refine_kernel runs on kRefWaves waves.  Wave 0 executes the function exactly as before (the replay of the RANSAC
// budget, the ordered inlier / outlier lists, the median, the solves); the other waves only help where the time goes — the
// normal-equation terms of ConvergePose, 28 per observation: a round covers 64 * kRefWaves observations (wave w computes
// observations 64 w .. 64 w + 63 of the round), then lanes 0..27 of wave 0 add the round's terms IN OBSERVATION ORDER (the
// reference's rounding).  The helpers wait at a workgroup barrier for wave 0's next command (RefineCtl): run a round / leave.
struct RefineCtl {
  double R[9], t[3], scale;
  const uint16_t *list;
  int n, q0, cmd;  // cmd: 1 = compute the terms of round q0, 0 = leave
  int gen;         // counts the ConvergePose calls: a lane's cached observation belongs to one call's list
};
// the observation a lane computed terms for last time: with a list that fits one round (<= 64 lanes x waves: the metric configuration's
// ~165 inliers on three waves) a lane has the SAME observation in every iteration of a ConvergePose call — one global load per call
// instead of one dependent load per iteration (round 6)
struct RefineObsCache {
  int q = -1, gen = -1;
  sdvl_pose_obs o;
};
template <int kRefWaves>
struct RefineLds {
  double terms[64 * kRefWaves][29];  // 28 normal-equation terms per observation of a round (+1 pad: no 2-way bank conflict pattern)
  RefineCtl ctl;
  double sums[28];
  // the one-wave form is launched for jobs of at most 256 observations only (sdvl_pose_enqueue_device): a quarter of the lists — 19 KB
  // of LDS per workgroup instead of 29.5, which is what a workgroup waits for among the other streams' kernels
  static constexpr int kObs = kRefWaves == 1 ? 256 : kMaxObs;
  double errs[kObs];
  uint16_t inl[kObs], outl[kObs], tmp[kObs];
  // one-wave form: the observations of the list that ConvergePose iterates over, copied once per call ([field][position in the list]) —
  // from global memory they were a dependent load per round and iteration (round 6)
  double obs_c[kRefWaves == 1 ? 6 : 1][kRefWaves == 1 ? kObs : 1];
};

// CheckReprojectionError over list[0..n) in order; appends to inl (at *n_in) and outl (at *n_out)
__device__ void check_list(const sdvl_pose_obs *obs, const uint16_t *list, int n, const Rigid &se3, double thr, uint16_t *inl, int *n_in,
                           uint16_t *outl, int *n_out, int lane) {
  const M3 R = se3_rot(se3);
  int ni = *n_in, no = *n_out;
  for (int q0 = 0; q0 < n; q0 += 64) {
    const int q = q0 + lane;
    bool is_in = false, is_out = false;
    int id = 0;
    if (q < n) {
      id = list[q];
      double ex, ey;
      V3 pos;
      reproj_error(obs[id], R, se3.t, &ex, &ey, &pos);
      is_in = sqrt(ex * ex + ey * ey) <= thr;
      is_out = !is_in;
    }
    const unsigned long long bi = __ballot(is_in), bo = __ballot(is_out);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (is_in) inl[ni + __popcll(bi & below)] = static_cast<uint16_t>(id);
    if (is_out) outl[no + __popcll(bo & below)] = static_cast<uint16_t>(id);
    ni += __popcll(bi);
    no += __popcll(bo);
  }
  *n_in = ni;
  *n_out = no;
}

// k-th smallest (0-based) of the n non-negative doubles vals[0..n), n <= 64 * kSlots, by one wave; uniform result.
// Non-negative doubles order like their bit patterns: the answer is built bit by bit from the top, counting with ballots
// how many values lie below the candidate prefix — 63 uniform steps, no sorting.
template <int kSlots>
__device__ __forceinline__ double wave_kth_smallest(const double *vals, int n, int k, int lane) {
  unsigned long long v[kSlots];
#pragma unroll
  for (int u = 0; u < kSlots; u++) {
    const int i = u * 64 + lane;
    v[u] = i < n ? static_cast<unsigned long long>(__double_as_longlong(vals[i])) : ~0ull;
  }
  unsigned long long prefix = 0;
  for (int bit = 62; bit >= 0; bit--) {
    const unsigned long long cand = prefix | (1ull << bit);
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < kSlots; u++) cnt += __popcll(__ballot(v[u] < cand));
    if (cnt <= k) prefix = cand;
  }
  return __longlong_as_double(static_cast<long long>(prefix));
}

// the 28 terms of observation list[q] under (R, t) -> L.terms[row] (feature_align.cc:370-400)
template <int kRefWaves>
__device__ __forceinline__ sdvl_pose_obs refine_obs(const RefineLds<kRefWaves> &L, const sdvl_pose_obs *obs, const uint16_t *list, int q) {
  if constexpr (kRefWaves == 1) {
    sdvl_pose_obs o;
    o.px = L.obs_c[0][q]; o.py = L.obs_c[1][q]; o.pz = L.obs_c[2][q]; o.ax = L.obs_c[3][q]; o.ay = L.obs_c[4][q]; o.inv_cov = L.obs_c[5][q];
    return o;
  } else {
    return obs[list[q]];
  }
}

template <int kRefWaves>
__device__ __forceinline__ void refine_terms(RefineLds<kRefWaves> &L, const sdvl_pose_obs *obs, const uint16_t *list, int q, int row, const M3 &R, const V3 &t,
                                             double scale, RefineObsCache &cache, int gen) {
  if (kRefWaves > 1 && (cache.q != q || cache.gen != gen)) {
    cache.o = obs[list[q]];
    cache.q = q;
    cache.gen = gen;
  }
  const sdvl_pose_obs o = kRefWaves > 1 ? cache.o : refine_obs(L, obs, list, q);
  double ex, ey;
  V3 pos;
  reproj_error(o, R, t, &ex, &ey, &pos);
  double J[12];
  jacobian_3d_to_plane(pos, J);
#pragma unroll
  for (int c = 0; c < 12; c++) J[c] *= o.inv_cov;
  const double weight = tukey(sqrt(ex * ex + ey * ey) / scale);
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = r; c < 6; c++) L.terms[row][k++] = (J[r] * J[c] + J[6 + r] * J[6 + c]) * weight;
#pragma unroll
  for (int r = 0; r < 6; r++) L.terms[row][21 + r] = -((J[r] * ex + J[6 + r] * ey) * weight);  // b accumulates with -= in the reference: x - t == x + (-t) exactly
  L.terms[row][27] = (ex * ex + ey * ey) * weight;
}

// what a helper wave (1 .. kRefWaves - 1) does for the whole kernel: wait for wave 0's command, compute its 64 observations of the round
template <int kRefWaves>
__device__ void refine_helper_loop(RefineLds<kRefWaves> &L, const sdvl_pose_obs *obs, int wave, int lane) {
  RefineObsCache cache;
  for (;;) {
    __syncthreads();  // A: the command is published
    if (L.ctl.cmd == 0) return;
    const int q = L.ctl.q0 + 64 * wave + lane;
    if (q < L.ctl.n) {
      M3 R;
#pragma unroll
      for (int k = 0; k < 9; k++) R.m[k] = L.ctl.R[k];
      const V3 t = {L.ctl.t[0], L.ctl.t[1], L.ctl.t[2]};
      refine_terms(L, obs, L.ctl.list, q, 64 * wave + lane, R, t, L.ctl.scale, cache, L.ctl.gen);
    }
    __syncthreads();  // B: the round's terms are complete
  }
}

// ConvergePose over list[0..n) (n <= kMaxObs) by wave 0 with the helpers; returns false when the list is empty
template <int kRefWaves>
__device__ bool converge_pose_wave(RefineLds<kRefWaves> &L, const sdvl_pose_obs *obs, const uint16_t *list, int n, const Rigid &frame_pose, double fx,
                                   int max_its, Rigid *se3, int lane, int gen) {
  Rigid last = frame_pose;
  *se3 = last;
  double chi2 = 0.0;
  if (n == 0) return false;
  RefineObsCache cache;
  if constexpr (kRefWaves == 1) {
    for (int q = lane; q < n; q += 64) {
      const sdvl_pose_obs o = obs[list[q]];
      L.obs_c[0][q] = o.px; L.obs_c[1][q] = o.py; L.obs_c[2][q] = o.pz; L.obs_c[3][q] = o.ax; L.obs_c[4][q] = o.ay; L.obs_c[5][q] = o.inv_cov;
    }
  }
  {
    const M3 R = se3_rot(*se3);
    for (int q = lane; q < n; q += 64) {
      double ex, ey;
      V3 pos;
      reproj_error(refine_obs(L, obs, list, q), R, se3->t, &ex, &ey, &pos);
      L.errs[q] = sqrt(ex * ex + ey * ey);
    }
  }
  wave_lds_sync();
  // median = element floor(n/2) of the sorted order (what nth_element leaves there)
  const double median = n <= 256 ? wave_kth_smallest<4>(L.errs, n, n / 2, lane) : wave_kth_smallest<RefineLds<kRefWaves>::kObs / 64>(L.errs, n, n / 2, lane);
  double scale = kMADNorm * median;
  for (int i = 0; i < max_its; i++) {
    if (i == 5) scale = 0.85 / fx;
    const M3 R = se3_rot(*se3);
    if (lane < 28) L.sums[lane] = 0.0;
    double acc = 0.0;  // lanes 0..27: running sum of term `lane` in observation order
    for (int q0 = 0; q0 < n; q0 += 64 * kRefWaves) {
      if (lane == 0) {  // the round's command for the helper waves
#pragma unroll
        for (int k = 0; k < 9; k++) L.ctl.R[k] = R.m[k];
        L.ctl.t[0] = se3->t.x; L.ctl.t[1] = se3->t.y; L.ctl.t[2] = se3->t.z;
        L.ctl.scale = scale;
        L.ctl.list = list;
        L.ctl.n = n;
        L.ctl.q0 = q0;
        L.ctl.gen = gen;
        L.ctl.cmd = 1;
      }
      __syncthreads();  // A
      const int q = q0 + lane;
      if (q < n) refine_terms(L, obs, list, q, lane, R, se3->t, scale, cache, gen);
      __syncthreads();  // B
      if (lane < 28) {
        // in observation order (the b terms were stored negated: refine_terms).  The adds are a dependent chain; the LDS reads of the
        // next eight terms are in flight while the current eight are added (round 6: the loop used to wait for every group of eight —
        // with configuration C's ~850 matches that wait was half of the kernel)
        const int m = min(64 * kRefWaves, n - q0);
        double cur[8], nxt[8];
        int j = 0;
        if (m >= 8) {
#pragma unroll
          for (int u = 0; u < 8; u++) cur[u] = L.terms[u][lane];
        }
        for (; j + 16 <= m; j += 8) {
#pragma unroll
          for (int u = 0; u < 8; u++) nxt[u] = L.terms[j + 8 + u][lane];
#pragma unroll
          for (int u = 0; u < 8; u++) acc += cur[u];
#pragma unroll
          for (int u = 0; u < 8; u++) cur[u] = nxt[u];
        }
        if (j + 8 <= m) {
#pragma unroll
          for (int u = 0; u < 8; u++) acc += cur[u];
          j += 8;
        }
        for (; j < m; j++) acc += L.terms[j][lane];
      }
      wave_lds_sync();
    }
    if (lane < 28) L.sums[lane] = acc;
    wave_lds_sync();
    double A[36], b[6], dT[6];
    {
      int t = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++) {
          A[6 * r + c] = L.sums[t];
          A[6 * c + r] = L.sums[t];  // the reference adds the same terms in the same order to both halves
          t++;
        }
    }
#pragma unroll
    for (int r = 0; r < 6; r++) b[r] = L.sums[21 + r];
    const double new_chi2 = L.sums[27];
    wave_lds_sync();
    ldlt_solve6_reg<true>(A, b, dT);
    if ((i > 0 && new_chi2 > chi2) || dT[0] != dT[0]) {
      *se3 = last;
      break;
    }
    const Rigid T_new = se3_mul(se3_exp(dT), *se3);
    last = *se3;
    *se3 = T_new;
    chi2 = new_chi2;
    if (abs_max6(dT) <= 1e-10) break;
  }
  return true;
}

template <int kRefWaves>
__global__ __launch_bounds__(64 * kRefWaves) void pose_refine_kernel(const PoseJobDev *__restrict__ jobs, const sdvl_pose_obs *__restrict__ obs_all,
                                                         const int32_t *__restrict__ nits_table, const HypResult *__restrict__ hyp,
                                                         sdvl_pose_params prm, sdvl_pose_result *__restrict__ results,
                                                         int32_t *__restrict__ out_lists, int lazy_supporters, const int32_t *__restrict__ rand_idx,
                                                         int hyp_ready) {
  __shared__ RefineLds<kRefWaves> L;
  const PoseJobDev &job = jobs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int size = job.n_obs;
  const sdvl_pose_obs *obs = obs_all + job.obs_begin;
  if (wave > 0) {  // helper waves: ConvergePose's terms only (size == 0: wave 0 returns without a command)
    if (size > 0) refine_helper_loop(L, obs, wave, lane);
    return;
  }
  sdvl_pose_result res;
  for (int k = 0; k < 7; k++) res.pose[k] = job.pose[k];
  res.n_draws = 0;
  res.n_inliers = 0;
  res.n_outliers = 0;
  res.refined = 0;
  if (size == 0) {  // SelectInliers returns at once; OptimizePose finds no features (feature_align.cc:165-166,220-221)
    if (lane == 0) results[blockIdx.x] = res;
    return;
  }
  // ---- RANSAC bookkeeping replayed over the draw results (feature_align.cc:172-212).  The loop is inherently serial
  // (the budget shrinks as supporters improve), so it must not touch memory: lane h holds the supporter count of draws h
  // and h + 64 (-1 = ConvergePose failed), the loop reads them with readlane; uniform across the lanes.
  const HypResult *hy = hyp + static_cast<size_t>(blockIdx.x) * prm.max_ransac_its;
  const int32_t *nits_of = nits_table + job.nits_begin;  // iteration budget after an improvement to s supporters
  int best_it = -1;
  int nits = prm.max_ransac_its, best_sup = 0, it = 0;
  double best7[7] = {1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (lazy_supporters) {
    // Round 6: the supporters of a draw are counted HERE, when the replay reaches the draw — the reference stops drawing once the budget is
    // met (typically after 5-10 of the 100 draws when nine matches in ten are inliers), and so does this loop: pose_supporters_kernel
    // tested every draw against every match (24 k of a tracked frame's 694 k instructions) for counts nobody read.  Lane h holds draw
    // base + h; the loop takes a draw's pose out of it with readlane and counts over the matches in rounds of 64 (an integer sum).
    for (int base = 0; base < prm.max_ransac_its && it < nits; base += 64) {
      const int h = base + lane;
      int ok_h = 0;
      double p7[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (h < prm.max_ransac_its) {
        if (h < hyp_ready) {
          ok_h = hy[h].ok;
#pragma unroll
          for (int k = 0; k < 7; k++) p7[k] = hy[h].se3[k];
        } else {
          // a draw beyond those pose_hypotheses_kernel converged (it is launched for the first 64 only: a budget that 64 draws do not
          // meet is the exception — fewer than ~6 matches in 10 are inliers): its lane converges it here, exactly as that kernel would
          const int npoints = min(prm.max_ransac_points, size);
          int index = rand_idx[job.rand_begin + h];
          if (prm.pad_ & 1) index %= size;
          int sel[8];
          for (int i = 0; i < npoints; i++) sel[i] = (index + i) % size;
          Rigid se3;
          if (converge_pose_small<false>(obs, sel, npoints, se3_from7(job.pose), prm.fx, prm.max_optim_pose_its, &se3)) {
            ok_h = 1;
            se3_to7(se3, p7);
          }
        }
      }
      const int end = min(base + 64, prm.max_ransac_its);
      while (it < nits && it < end) {
        const int src = it - base;
        int s_it = -1;
        double u7[7] = {1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (__builtin_amdgcn_readlane(ok_h, src)) {
#pragma unroll
          for (int k = 0; k < 7; k++)
            u7[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(p7[k]), src), __builtin_amdgcn_readlane(__double2loint(p7[k]), src));
          const Rigid se3 = se3_from7(u7);
          const M3 R = se3_rot(se3);
          s_it = 0;
          for (int q0 = 0; q0 < size; q0 += 64) {
            const int q = q0 + lane;
            bool in = false;
            if (q < size) {
              double ex, ey;
              V3 pos;
              reproj_error(obs[q], R, se3.t, &ex, &ey, &pos);
              in = sqrt(ex * ex + ey * ey) <= prm.inlier_threshold;
            }
            s_it += __popcll(__ballot(in));
          }
        }
        if (s_it > best_sup) {
          best_sup = s_it;
          best_it = it;
          nits = nits_of[best_sup];
#pragma unroll
          for (int k = 0; k < 7; k++) best7[k] = u7[k];  // (a draw converged on demand exists in registers only)
        }
        it++;
      }
    }
  } else
  for (int base = 0; base < prm.max_ransac_its && it < nits; base += 64) {
    const int h = base + lane;
    int sup = -1;
    if (h < prm.max_ransac_its && hy[h].ok) sup = hy[h].supporters;
    const int end = min(base + 64, prm.max_ransac_its);
    while (it < nits && it < end) {
      const int s_it = __builtin_amdgcn_readlane(sup, it - base);
      if (s_it > best_sup) {
        best_sup = s_it;
        best_it = it;
        nits = nits_of[best_sup];
      }
      it++;
    }
  }
  Rigid best = se3_identity();  // SE3 best_se3 default-constructed
  if (best_it >= 0) best = lazy_supporters ? se3_from7(best7) : se3_from7(hy[best_it].se3);
  res.n_draws = it;
  // ---- final CheckReprojectionError with the best hypothesis -> inliers / outliers (feature_align.cc:215)
  for (int q = lane; q < size; q += 64) L.tmp[q] = static_cast<uint16_t>(q);
  wave_lds_sync();
  int n_in = 0, n_out = 0;
  check_list(obs, L.tmp, size, best, prm.inlier_threshold, L.inl, &n_in, L.outl, &n_out, lane);
  wave_lds_sync();
  // ---- OptimizePose(frame): feature_align.cc:73-82
  Rigid pose = se3_from7(job.pose);
  for (int pass = 0; pass < 2; pass++) {
    // OptimizePose(frame, &inliers, &outliers), :218-230
    Rigid se3;
    if (converge_pose_wave(L, obs, L.inl, n_in, pose, prm.fx, prm.max_optim_pose_its, &se3, lane, pass)) {
      pose = se3;
      res.refined = 1;
      for (int q = lane; q < n_in; q += 64) L.tmp[q] = L.inl[q];
      wave_lds_sync();
      const int n_c = n_in;
      n_in = 0;
      check_list(obs, L.tmp, n_c, pose, prm.inlier_threshold, L.inl, &n_in, L.outl, &n_out, lane);
      wave_lds_sync();
    }
    if (pass == 1) break;
    // RescueOutliers, :232-243
    const int init_in = n_in;
    for (int q = lane; q < n_out; q += 64) L.tmp[q] = L.outl[q];
    wave_lds_sync();
    const int n_c = n_out;
    n_out = 0;
    check_list(obs, L.tmp, n_c, pose, 2 * prm.inlier_threshold, L.inl, &n_in, L.outl, &n_out, lane);
    wave_lds_sync();
    if (!(n_in > init_in)) break;
  }
  se3_to7(pose, res.pose);
  res.n_inliers = n_in;
  res.n_outliers = n_out;
  int32_t *lists = out_lists + job.obs_begin;
  for (int q = lane; q < n_in; q += 64) lists[q] = L.inl[q];
  for (int q = lane; q < n_out; q += 64) lists[n_in + q] = L.outl[q];
  if (lane == 0) {
    results[blockIdx.x] = res;
    L.ctl.cmd = 0;  // the helpers leave
  }
  __syncthreads();  // A
}

}  // namespace

size_t sdvl_pose_hyp_bytes() { return sizeof(HypResult); }

int sdvl_pose_enqueue_device(sdvl_ctx *ctx, int n_jobs, const PoseJobDev *d_jobs, const sdvl_pose_obs *d_obs, const int32_t *d_rand,
                             const int32_t *d_nits, const sdvl_pose_params *p, void *d_hyp, sdvl_pose_result *d_res, int32_t *d_lists, int max_obs,
                             int batch_size) {
  if (max_obs < 1) max_obs = 1;
  if (max_obs > kMaxObs) max_obs = kMaxObs;
  const bool all_supporters = getenv("SDVL_POSE_ALL_SUPPORTERS") != nullptr;  // (read per call: the test flips it inside one process)
  int hyp_ready = p->max_ransac_its;
  // one wave per draw pays while every wave finds a SIMD of its own: a lone camera (100 waves), not configuration C's groups of 16
  // (1600 waves per launch, four groups at a time: 46.3 k tracked frames/s with the lane form against 41.3 k, two alternating pairs)
  const bool wave_form = batch_size <= 4;
  if (wave_form) {
    // a small set: one wave per draw, the supporters counted by the same wave (see pose_hypotheses_wave_kernel)
    SDVL_LAUNCH(ctx, "pose_hypotheses", pose_hypotheses_wave_kernel, dim3(p->max_ransac_its, n_jobs), dim3(64), d_jobs, d_obs, d_rand, *p,
                static_cast<HypResult *>(d_hyp));
  } else {
    hyp_ready = all_supporters ? p->max_ransac_its : std::min(p->max_ransac_its, kHypDraws);  // the first wave of draws; the rest on demand (pose_refine)
    {
      hipEvent_t ev_a = nullptr, ev_b = nullptr;
      sdvl_timer_events(ctx, "pose_hypotheses", &ev_a, &ev_b);
      const size_t cache_bytes = static_cast<size_t>(p->max_ransac_points) * 6 * 64 * sizeof(double);  // <= 24.5 KB (max_ransac_points <= 8)
      hipExtLaunchKernelGGL(pose_hypotheses_kernel, dim3((hyp_ready + kHypDraws - 1) / kHypDraws, n_jobs), dim3(kHypDraws), cache_bytes, ctx->stream, ev_a, ev_b,
                            0, d_jobs, d_obs, d_rand, *p, static_cast<HypResult *>(d_hyp));
    }
    // the supporters: counted by pose_refine as its replay of the RANSAC loop reaches a draw; SDVL_POSE_ALL_SUPPORTERS=1 (A/B and test):
    // every draw against every match in a launch of its own, as in rounds 3-5
    if (all_supporters)
      SDVL_LAUNCH(ctx, "pose_supporters", pose_supporters_kernel, dim3(p->max_ransac_its, (max_obs + kSupChunk - 1) / kSupChunk, n_jobs), dim3(64), d_jobs, d_obs, *p,
                  static_cast<HypResult *>(d_hyp));
  }
  const int lazy = (!wave_form && !all_supporters) ? 1 : 0;
  // a frame of the metric configuration has <= 200 observations: one wave (three waves with 59 KB of LDS wait longer for a CU among the
  // other streams' kernels than they save: 2.4 -> 3.9 ms of dispatch time per step); configuration C's ~850: wave 0 + two helpers
  // Round 5: a small batch (a lone camera) takes the helper waves too — nobody else wants the CU
  if (max_obs > 256 || batch_size <= 32)
    SDVL_LAUNCH(ctx, "pose_refine", pose_refine_kernel<3>, dim3(n_jobs), dim3(192), d_jobs, d_obs, d_nits, static_cast<const HypResult *>(d_hyp), *p, d_res,
                d_lists, lazy, d_rand, hyp_ready);
  else
  SDVL_LAUNCH(ctx, "pose_refine", pose_refine_kernel<1>, dim3(n_jobs), dim3(64), d_jobs, d_obs, d_nits, static_cast<const HypResult *>(d_hyp), *p, d_res,
              d_lists, lazy, d_rand, hyp_ready);
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  return SDVL_OK;
}

extern "C" int sdvl_pose_from_matches(sdvl_ctx *ctx, int n_jobs, const sdvl_pose_job *jobs, int n_obs, const sdvl_pose_obs *obs, int n_rand,
                                      const int32_t *rand_idx, int n_nits, const int32_t *nits_table, const sdvl_pose_params *p,
                                      sdvl_pose_result *results, int32_t *out_lists) {
  if (!ctx || !p || n_jobs < 0 || (n_jobs > 0 && (!jobs || !results)) || n_obs < 0 || (n_obs > 0 && (!obs || !out_lists))) return SDVL_ERR_INVALID;
  if (n_jobs == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, p->max_ransac_points >= 1 && p->max_ransac_points <= 8, "max_ransac_points must be in [1,8]");
  SDVL_REQUIRE(ctx, p->max_ransac_its >= 1 && p->max_ransac_its <= 4096 && p->max_optim_pose_its >= 0, "bad iteration limits");
  SDVL_REQUIRE(ctx, rand_idx && nits_table, "rand / nits tables missing");
  for (int j = 0; j < n_jobs; j++) {
    const sdvl_pose_job &a = jobs[j];
    SDVL_REQUIRE(ctx, a.obs_begin >= 0 && a.obs_end >= a.obs_begin && a.obs_end <= n_obs, "observation range out of bounds");
    if (a.obs_end - a.obs_begin > kMaxObs) {
      ctx->err = "too many matches in one pose job for the device path (1024)";
      return SDVL_ERR_CAPACITY;
    }
    const int size = a.obs_end - a.obs_begin;
    SDVL_REQUIRE(ctx, a.rand_begin >= 0 && a.rand_begin + p->max_ransac_its <= n_rand, "rand range out of bounds");
    SDVL_REQUIRE(ctx, a.nits_begin >= 0 && a.nits_begin + size + 1 <= n_nits, "nits table range out of bounds");
    for (int h = 0; h < p->max_ransac_its && size > 0; h++)
      SDVL_REQUIRE(ctx, rand_idx[a.rand_begin + h] >= 0 && rand_idx[a.rand_begin + h] < size, "rand index outside the match list");
  }
  const size_t jb = (sizeof(PoseJobDev) * n_jobs + 255) / 256 * 256, ob = (sizeof(sdvl_pose_obs) * static_cast<size_t>(n_obs) + 255) / 256 * 256;
  const size_t rb = (sizeof(int32_t) * static_cast<size_t>(n_rand) + 255) / 256 * 256, nb = (sizeof(int32_t) * static_cast<size_t>(n_nits) + 255) / 256 * 256;
  const size_t hyp_bytes = sizeof(HypResult) * static_cast<size_t>(n_jobs) * p->max_ransac_its;
  const size_t res_bytes = (sizeof(sdvl_pose_result) * n_jobs + 255) / 256 * 256, list_bytes = sizeof(int32_t) * static_cast<size_t>(n_obs);
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_ensure(ctx, &ctx->d_work, &ctx->d_work_bytes, hyp_bytes + 256, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->d_out, &ctx->d_out_bytes, res_bytes + list_bytes, false);
  if (!rc) rc = sdvl_ensure(ctx, &ctx->h_out, &ctx->h_out_bytes, res_bytes + list_bytes, true);
  if (!rc) rc = sdvl_stage_alloc(ctx, jb + ob + rb + nb, &hs, &dsx);
  if (rc) return rc;
  uint8_t *h8 = static_cast<uint8_t *>(hs), *d8 = static_cast<uint8_t *>(dsx);
  PoseJobDev *hj = reinterpret_cast<PoseJobDev *>(h8);
  for (int j = 0; j < n_jobs; j++) {
    hj[j].obs_begin = jobs[j].obs_begin;
    hj[j].n_obs = jobs[j].obs_end - jobs[j].obs_begin;
    hj[j].rand_begin = jobs[j].rand_begin;
    hj[j].nits_begin = jobs[j].nits_begin;
    memcpy(hj[j].pose, jobs[j].pose, sizeof(double) * 7);
    hj[j].pad_ = 0.0;
  }
  if (n_obs) memcpy(h8 + jb, obs, sizeof(sdvl_pose_obs) * static_cast<size_t>(n_obs));
  memcpy(h8 + jb + ob, rand_idx, sizeof(int32_t) * static_cast<size_t>(n_rand));
  memcpy(h8 + jb + ob + rb, nits_table, sizeof(int32_t) * static_cast<size_t>(n_nits));
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, jb + ob + rb + nb));
  const PoseJobDev *dj = reinterpret_cast<const PoseJobDev *>(d8);
  const sdvl_pose_obs *dobs = reinterpret_cast<const sdvl_pose_obs *>(d8 + jb);
  const int32_t *drand = reinterpret_cast<const int32_t *>(d8 + jb + ob), *dnits = reinterpret_cast<const int32_t *>(d8 + jb + ob + rb);
  HypResult *dhyp = static_cast<HypResult *>(ctx->d_work);
  sdvl_pose_result *dres = static_cast<sdvl_pose_result *>(ctx->d_out);
  int32_t *dlists = reinterpret_cast<int32_t *>(static_cast<uint8_t *>(ctx->d_out) + res_bytes);
  sdvl_pose_params prm = *p;
  prm.pad_ = 0;  // rand_idx holds indices already reduced modulo the match count
  int max_obs = 0;
  for (int j = 0; j < n_jobs; j++) max_obs = std::max(max_obs, jobs[j].obs_end - jobs[j].obs_begin);
  rc = sdvl_pose_enqueue_device(ctx, n_jobs, dj, dobs, drand, dnits, &prm, dhyp, dres, dlists, max_obs, n_jobs);
  if (rc) return rc;
  SDVL_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, res_bytes + list_bytes, hipMemcpyDeviceToHost, ctx->stream));
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  memcpy(results, ctx->h_out, sizeof(sdvl_pose_result) * n_jobs);
  if (n_obs) memcpy(out_lists, static_cast<uint8_t *>(ctx->h_out) + res_bytes, list_bytes);
  return SDVL_OK;
}
