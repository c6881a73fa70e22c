// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_math.h header).  PARITY UNPINNED.
//
// ref_align.h — sparse direct image alignment and the per-point matcher.
//   ImageAlign: /root/reference/image_align.cc:33-269
//   Matcher:    /root/reference/matcher.cc:34-478, extra/utils.cc:44-59 (Interpolate8U)
// Float vs double kept exactly as written in the reference; compile with -ffp-contract=off.
#ifndef SDVL_ORACLE_REF_ALIGN_H_
#define SDVL_ORACLE_REF_ALIGN_H_

#include <cmath>
#include <cstdint>
#include <vector>

#include "ref_detect.h"
#include "ref_math.h"
#include "ref_orb.h"

namespace sdvlref {

// What ImageAlign reads of one feature of frame1 (image_align.cc:147-160, 219-236)
struct AlignFeature {
  double px, py;   // Feature::GetPosition (level-0 pixels)
  Vec3 f;          // Feature::GetVector (unit bearing)
  double depth;    // (point->GetPosition() - frame1->GetWorldPosition()).norm()
  uint8_t valid;   // feature->GetPoint() && !ToDelete()
};

struct ImageAlign {
  // image_align.cc:33-39
  bool stop = false;
  double chi2 = 1e10;
  double error = 1e10;
  size_t n_meas = 0;
  std::vector<float> patch_cache;
  std::vector<double> jac_cache;  // 6 x (N*area), column-major
  std::vector<uint8_t> visible;
  double H[6][6];
  double Jres[6];
  int its_per_level[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  const std::vector<Image> *pyr1 = nullptr, *pyr2 = nullptr;
  const std::vector<AlignFeature> *feats = nullptr;
  Camera cam;
  Params prm;

  // image_align.cc:46-84.  T_io = frame2.pose * frame1.pose^-1 on entry, refined on exit.
  int ComputePose(const std::vector<Image> &p1, const std::vector<Image> &p2, const std::vector<AlignFeature> &features,
                  const Camera &camera, const Params &params, SE3 *T_io, bool fast) {
    pyr1 = &p1; pyr2 = &p2; feats = &features; cam = camera; prm = params;
    const int size = static_cast<int>(features.size());
    if (size == 0) return 0;
    const int area = prm.align_patch_size * prm.align_patch_size;
    patch_cache.assign(static_cast<size_t>(size) * area, 0.f);
    jac_cache.assign(static_cast<size_t>(6) * size * area, 0.0);
    visible.assign(size, 0);
    SE3 cur = *T_io;
    for (int level = prm.max_align_level; level >= prm.min_align_level; level--) {
      std::fill(jac_cache.begin(), jac_cache.end(), 0.0);
      Optimize(&cur, level);
      if (fast && error > 0.01) {
        error = 1e10;
        break;
      }
    }
    *T_io = cur;
    return static_cast<int>(n_meas / area);
  }

  // image_align.cc:86-125
  void Optimize(SE3 *se3, int level) {
    double x[6];
    SE3 se3_bk = *se3;
    for (int i = 0; i < prm.max_img_align_its; i++) {
      for (int r = 0; r < 6; r++) { Jres[r] = 0.0; for (int c = 0; c < 6; c++) H[r][c] = 0.0; }
      n_meas = 0;
      const double new_chi2 = ComputeResiduals(*se3, level, true, i == 0);
      if (n_meas == 0) stop = true;
      LdltSolve6(H, Jres, x);
      if (std::isnan(x[0])) stop = true;
      if ((i > 0 && new_chi2 > chi2) || stop) {
        *se3 = se3_bk;
        break;
      }
      se3_bk = *se3;
      double mx[6];
      for (int r = 0; r < 6; r++) mx[r] = -x[r];
      *se3 = (*se3) * SE3Exp(mx);
      chi2 = new_chi2;
      its_per_level[level]++;
      error = AbsMax6(x);
      if (error <= 1e-10) break;
    }
  }

  // image_align.cc:127-206
  double ComputeResiduals(const SE3 &se3, int level, bool linearize, bool patches) {
    const int psize = prm.align_patch_size;
    const int half_patch = psize / 2;
    const int area = psize * psize;
    const Image &last_img = (*pyr2)[level];
    if (patches) PrecomputePatches(level);
    const int stride = last_img.cols;
    const int border = half_patch + 1;
    const float scale = 1.0f / (1 << level);
    float chi2f = 0.0;
    const Mat3 R = se3.Rotation();
    for (size_t counter = 0; counter < feats->size(); counter++) {
      const AlignFeature &ft = (*feats)[counter];
      if (!visible[counter]) continue;
      if (!ft.valid) continue;
      const Vec3 xyz_ref = ft.f * ft.depth;
      const Vec3 xyz_cur = MatVec(R, xyz_ref) + se3.t;
      const Vec2 proj = cam.Project(xyz_cur);
      const double uvx = proj.x * scale, uvy = proj.y * scale;
      const float u_cur = static_cast<float>(uvx);
      const float v_cur = static_cast<float>(uvy);
      const int u_last_i = static_cast<int>(floorf(u_cur));
      const int v_last_i = static_cast<int>(floorf(v_cur));
      if (u_last_i < 0 || v_last_i < 0 || u_last_i - border < 0 || v_last_i - border < 0 ||
          u_last_i + border >= last_img.cols || v_last_i + border >= last_img.rows)
        continue;
      const float subpix_u_cur = u_cur - u_last_i;
      const float subpix_v_cur = v_cur - v_last_i;
      const float w_tl = static_cast<float>((1.0 - subpix_u_cur) * (1.0 - subpix_v_cur));
      const float w_tr = static_cast<float>(subpix_u_cur * (1.0 - subpix_v_cur));
      const float w_bl = static_cast<float>((1.0 - subpix_u_cur) * subpix_v_cur);
      const float w_br = subpix_u_cur * subpix_v_cur;
      const float *pc = &patch_cache[static_cast<size_t>(area) * counter];
      size_t pixel_counter = 0;
      for (int y = 0; y < psize; y++) {
        const uint8_t *ip = last_img.data + static_cast<size_t>(v_last_i + y - half_patch) * stride + (u_last_i - half_patch);
        for (int x = 0; x < psize; x++, pixel_counter++, ip++, pc++) {
          const float intensity_cur = w_tl * ip[0] + w_tr * ip[1] + w_bl * ip[stride] + w_br * ip[stride + 1];
          const float res = intensity_cur - (*pc);
          const float weight = 1.0;
          chi2f += res * res * weight;
          n_meas++;
          if (linearize) {
            const double *J = &jac_cache[(counter * area + pixel_counter) * 6];
            for (int r = 0; r < 6; r++) {
              for (int c = 0; c < 6; c++) H[r][c] += J[r] * J[c] * weight;
              Jres[r] -= J[r] * res * weight;
            }
          }
        }
      }
    }
    return chi2f / n_meas;
  }

  // image_align.cc:208-267
  void PrecomputePatches(int level) {
    const int psize = prm.align_patch_size;
    const int half_patch = psize / 2;
    const int area = psize * psize;
    const int border = half_patch + 1;
    const Image &first_img = (*pyr1)[level];
    const int stride = first_img.cols;
    const float scale = 1.0f / (1 << level);
    const double focal_length = cam.fx;
    double frame_jac[2][6];
    for (size_t counter = 0; counter < feats->size(); counter++) {
      const AlignFeature &ft = (*feats)[counter];
      const float u_ref = static_cast<float>(ft.px * scale);
      const float v_ref = static_cast<float>(ft.py * scale);
      const int u_first_i = static_cast<int>(floorf(u_ref));
      const int v_first_i = static_cast<int>(floorf(v_ref));
      if (!ft.valid || u_first_i - border < 0 || v_first_i - border < 0 || u_first_i + border >= first_img.cols ||
          v_first_i + border >= first_img.rows)
        continue;
      visible[counter] = 1;
      const Vec3 xyz_ref = ft.f * ft.depth;
      Jacobian3DToPlane(xyz_ref, frame_jac);
      const float subpix_u_ref = u_ref - u_first_i;
      const float subpix_v_ref = v_ref - v_first_i;
      const float w_tl = static_cast<float>((1.0 - subpix_u_ref) * (1.0 - subpix_v_ref));
      const float w_tr = static_cast<float>(subpix_u_ref * (1.0 - subpix_v_ref));
      const float w_bl = static_cast<float>((1.0 - subpix_u_ref) * subpix_v_ref);
      const float w_br = subpix_u_ref * subpix_v_ref;
      size_t pixel_counter = 0;
      float *cache_ptr = &patch_cache[static_cast<size_t>(area) * counter];
      for (int y = 0; y < psize; y++) {
        const uint8_t *ip = first_img.data + static_cast<size_t>(v_first_i + y - half_patch) * stride + (u_first_i - half_patch);
        for (int x = 0; x < psize; x++, ip++, cache_ptr++, pixel_counter++) {
          *cache_ptr = w_tl * ip[0] + w_tr * ip[1] + w_bl * ip[stride] + w_br * ip[stride + 1];
          const float dx = 0.5f * ((w_tl * ip[1] + w_tr * ip[2] + w_bl * ip[stride + 1] + w_br * ip[stride + 2]) -
                                   (w_tl * ip[-1] + w_tr * ip[0] + w_bl * ip[stride - 1] + w_br * ip[stride]));
          const float dy = 0.5f * ((w_tl * ip[stride] + w_tr * ip[1 + stride] + w_bl * ip[stride * 2] + w_br * ip[stride * 2 + 1]) -
                                   (w_tl * ip[-stride] + w_tr * ip[1 - stride] + w_bl * ip[0] + w_br * ip[1]));
          double *J = &jac_cache[(counter * area + pixel_counter) * 6];
          const double fl = focal_length / (1 << level);
          for (int c = 0; c < 6; c++) J[c] = (dx * frame_jac[0][c] + dy * frame_jac[1][c]) * fl;
        }
      }
    }
  }
};

// extra/utils.cc:44-59
inline float Interpolate8U(const Image &mat, float u, float v) {
  const int x = static_cast<int>(std::floor(u));
  const int y = static_cast<int>(std::floor(v));
  const float subpix_x = u - x;
  const float subpix_y = v - y;
  const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  const float w01 = (1.0f - subpix_x) * subpix_y;
  const float w10 = subpix_x * (1.0f - subpix_y);
  const float w11 = 1.0f - w00 - w01 - w10;
  const int stride = mat.step;
  const uint8_t *ptr = mat.data + static_cast<size_t>(y) * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}

// What Matcher::SearchPoint reads of the reference feature and its frame (matcher.cc:45-121)
struct SearchRef {
  const std::vector<Image> *ref_pyr;
  SE3 ref_pose;        // feature->GetFrame()->GetPose()
  double px, py;       // feature->GetPosition()
  Vec3 f;              // feature->GetVector()
  int level;           // feature->GetLevel()
  uint8_t desc[32];    // feature->GetDescriptor()
};

// What it reads/writes of the current frame
struct SearchCur {
  const std::vector<Image> *pyr;
  SE3 pose;
  const std::vector<Corner> *corners;
  std::vector<std::vector<uint8_t>> *descriptors;  // lazily filled, matcher.cc:266-269
};

struct Matcher {
  static const int kMaxSsdPerPixel = 500;  // matcher.h:36
  static const int kMinOrbThreshold = 100; // matcher.h:37
  int patch_size;
  std::vector<uint8_t> patch, border_patch;
  OrbDetector detector;
  Params prm;
  Camera cam;

  Matcher(int size, const Params &p, const Camera &c) : patch_size(size), detector(p.orb_size), prm(p), cam(c) {
    patch.assign(static_cast<size_t>(size) * size, 0);
    border_patch.assign(static_cast<size_t>(size + 2) * (size + 2), 0);
  }

  // Frame::Project, frame.cc:94-103
  bool FrameProject(const SE3 &pose, const Vec3 &p3d, Vec2 *p2d) const {
    const Vec3 rel = pose * p3d;
    if (rel.z < 0.0) return false;
    *p2d = cam.Project(rel);
    return true;
  }

  // matcher.cc:45-121
  bool SearchPoint(SearchCur *cur, const SearchRef &ref, double idepth, double idepth_std, bool fixed, Vec2 *px, int *flevel) {
    Mat2 affine;
    double range, zmin, zmax;
    int slevel;
    const int level = ref.level;
    Vec2 pxa{0, 0}, pxb{0, 0};
    const SE3 pose = cur->pose * ref.ref_pose.Inverse();
    const SE3 ref_world = ref.ref_pose.Inverse();
    if (fixed) {
      zmin = 1.0 / (idepth + 2.0 * idepth_std);
      const Vec3 p3d_min = ref_world * (zmin * ref.f);
      if (!FrameProject(cur->pose, p3d_min, &pxa)) return false;
    } else {
      zmin = 1.0 / (idepth + 2.0 * idepth_std);
      zmax = 1.0 / (std::max(idepth - 2.0 * idepth_std, 0.00000001));
      const Vec3 p3d_min = ref_world * (zmin * ref.f);
      const Vec3 p3d_max = ref_world * (zmax * ref.f);
      if (!FrameProject(cur->pose, p3d_min, &pxa)) return false;
      if (!FrameProject(cur->pose, p3d_max, &pxb)) return false;
    }
    // feature->GetLevelPosition().cast<int>()
    const int lx = static_cast<int>(ref.px / (1 << level)), ly = static_cast<int>(ref.py / (1 << level));
    if (!cam.IsInsideImage(lx, ly, patch_size / 2 + 2, level)) return false;

    const Image &img = (*ref.ref_pyr)[level];
    WarpMatrixAffine(Vec2{ref.px, ref.py}, ref.f, 1.0 / idepth, pose, level, &affine);
    slevel = GetSearchLevel(affine);
    CreatePatch(affine, img, Vec2{ref.px, ref.py}, level, slevel);

    range = prm.search_size;
    for (int i = 1; i <= slevel; i++) range *= 1.2;

    std::vector<int> indices;
    if (fixed) GetCornersInRangeCircle(*cur, *px, level, range, &indices);
    else GetCornersInRangeLine(*cur, pxa, pxb, level, range, &indices);

    if (!SearchFeatures(cur, indices, px, ref.desc)) return false;

    Vec2 px_scaled{px->x / (1 << slevel), px->y / (1 << slevel)};
    if (AlignPatch((*cur->pyr)[slevel], border_patch.data(), patch.data(), &px_scaled)) {
      px->x = px_scaled.x * (1 << slevel);
      px->y = px_scaled.y * (1 << slevel);
      *flevel = slevel;
      return true;
    }
    return false;
  }

  int Margin() const { return prm.use_orb ? 4 + prm.orb_size / 2 : 1 + patch_size / 2; }

  // matcher.cc:123-192 — abs() frozen to the double overload (SURVEY §7 hard part 7)
  void GetCornersInRangeLine(const SearchCur &cur, const Vec2 &pxa, const Vec2 &pxb, int level, double range,
                             std::vector<int> *indices) const {
    const double range2 = range * range;
    const int margin = Margin();
    double ex = pxa.x - pxb.x, ey = pxa.y - pxb.y;
    const double en = std::sqrt(ex * ex + ey * ey);
    ex /= en; ey /= en;
    const double nx = ey, ny = -ex;
    const double normdist = pxa.x * nx + pxa.y * ny;
    const double xdiff = pxb.x - pxa.x;
    const double ydiff = pxb.y - pxa.y;
    const double vline = (xdiff) * (xdiff) + (ydiff) * (ydiff);
    int index = 0;
    for (auto it = cur.corners->begin(); it != cur.corners->end(); it++, index++) {
      const int clevel = it->level;
      if (std::abs(clevel - level) > 1) continue;
      if (it->x - margin < 0 || it->y - margin < 0) continue;
      if (it->y + margin >= (*cur.pyr)[clevel].rows || it->x + margin >= (*cur.pyr)[clevel].cols) continue;
      const double posx = it->x * (1 << clevel), posy = it->y * (1 << clevel);
      const double dist = normdist - (posx * nx + posy * ny);
      if (std::fabs(dist) > range) continue;
      const double u = ((posx - pxa.x) * xdiff + (posy - pxa.y) * ydiff) / vline;
      if (u > 1) {
        const double dx = posx - pxb.x, dy = posy - pxb.y;
        if ((dx * dx + dy * dy) > range2) continue;
      }
      if (u < 0) {
        const double dx = posx - pxa.x, dy = posy - pxa.y;
        if ((dx * dx + dy * dy) > range2) continue;
      }
      indices->push_back(index);
    }
  }

  // matcher.cc:194-230
  void GetCornersInRangeCircle(const SearchCur &cur, const Vec2 &cpos, int level, double range, std::vector<int> *indices) const {
    const double range2 = range * range;
    const int margin = Margin();
    int index = 0;
    for (auto it = cur.corners->begin(); it != cur.corners->end(); it++, index++) {
      const int clevel = it->level;
      if (std::abs(clevel - level) > 1) continue;
      if (it->x - margin < 0 || it->y - margin < 0) continue;
      if (it->y + margin >= (*cur.pyr)[clevel].rows || it->x + margin >= (*cur.pyr)[clevel].cols) continue;
      const double posx = it->x * (1 << clevel), posy = it->y * (1 << clevel);
      const double dx = cpos.x - posx, dy = cpos.y - posy;
      if (dx * dx + dy * dy > range2) continue;
      indices->push_back(index);
    }
  }

  // matcher.cc:232-291
  bool SearchFeatures(SearchCur *cur, const std::vector<int> &indices, Vec2 *px, const uint8_t *desc) {
    int sumA = 0, sumAA = 0, sumB, sumBB, sumAB;
    Vec2 best_px{0, 0};
    int threshold, best_score, score;
    if (prm.use_orb) threshold = kMinOrbThreshold;
    else threshold = patch_size * patch_size * kMaxSsdPerPixel;
    best_score = threshold + 1;
    if (!prm.use_orb) GetZMSSDScore(patch.data(), &sumA, &sumAA);
    for (auto it = indices.begin(); it != indices.end(); it++) {
      const int index = *it;
      const Corner &corner = (*cur->corners)[index];
      const int level = corner.level;
      const Image &cimg = (*cur->pyr)[level];
      if (prm.use_orb) {
        std::vector<uint8_t> &d = (*cur->descriptors)[index];
        if (d.empty()) {
          d.resize(32);
          detector.GetDescriptor(cimg, corner.x, corner.y, d.data());
        }
        score = OrbDetector::Distance(desc, d.data());
      } else {
        const uint8_t *cur_patch = cimg.data + static_cast<size_t>(corner.y - patch_size / 2) * cimg.cols + (corner.x - patch_size / 2);
        score = static_cast<int>(CompareZMSSDScore(patch.data(), cur_patch, sumA, sumAA, cimg.cols, &sumB, &sumBB, &sumAB));
      }
      if (score < best_score) {
        best_score = score;
        best_px = Vec2{static_cast<double>(corner.x * (1 << level)), static_cast<double>(corner.y * (1 << level))};
      }
    }
    if (best_score >= threshold) return false;
    *px = best_px;
    return true;
  }

  // matcher.cc:293-312
  void WarpMatrixAffine(const Vec2 &px, const Vec3 &v, double depth, const SE3 &pose, int level, Mat2 *res) const {
    const int half_size = 5;
    const Vec3 p3d = v * depth;
    Vec3 xyz_du = cam.Unproject(Vec2{px.x + static_cast<double>(half_size) * (1 << level), px.y + 0.0 * (1 << level)});
    Vec3 xyz_dv = cam.Unproject(Vec2{px.x + 0.0 * (1 << level), px.y + static_cast<double>(half_size) * (1 << level)});
    const double su = p3d.z / xyz_du.z;
    xyz_du = xyz_du * su;
    const double sv = p3d.z / xyz_dv.z;
    xyz_dv = xyz_dv * sv;
    const Vec2 px_cur = cam.Project(pose * p3d);
    const Vec2 px_du = cam.Project(pose * xyz_du);
    const Vec2 px_dv = cam.Project(pose * xyz_dv);
    res->a = (px_du.x - px_cur.x) / half_size;
    res->c = (px_du.y - px_cur.y) / half_size;
    res->b = (px_dv.x - px_cur.x) / half_size;
    res->d = (px_dv.y - px_cur.y) / half_size;
  }

  // matcher.cc:314-323
  int GetSearchLevel(const Mat2 &m) const {
    int search_level = 0;
    double det = m.a * m.d - m.b * m.c;
    const int max = prm.max_fast_levels - 1;
    while (det > 3.0 && search_level < max) {
      search_level += 1;
      det *= 0.25;
    }
    return search_level;
  }

  // matcher.cc:325-357 ; Eigen Matrix2d::inverse = adjugate * (1/det)
  void CreatePatch(const Mat2 &m, const Image &img, const Vec2 &px, int level, int search_level) {
    const int bpatch_size = patch_size + 2;
    const int half_size = bpatch_size / 2;
    const double det = m.a * m.d - m.b * m.c;
    const double invdet = 1.0 / det;
    const Mat2 inv{m.d * invdet, -m.b * invdet, -m.c * invdet, m.a * invdet};
    if (std::isnan(inv.a)) return;
    uint8_t *patch_ptr = patch.data();
    uint8_t *bpatch_ptr = border_patch.data();
    const double pyrx = px.x / (1 << level), pyry = px.y / (1 << level);
    for (int y = 0; y < bpatch_size; y++) {
      for (int x = 0; x < bpatch_size; x++, bpatch_ptr++) {
        double ppx = x - half_size, ppy = y - half_size;
        ppx *= (1 << search_level);
        ppy *= (1 << search_level);
        const double p0 = (inv.a * ppx + inv.b * ppy) + pyrx;
        const double p1 = (inv.c * ppx + inv.d * ppy) + pyry;
        if (p0 < 0 || p1 < 0 || p0 >= img.cols - 1 || p1 >= img.rows - 1)
          *bpatch_ptr = 0;
        else
          *bpatch_ptr = static_cast<uint8_t>(Interpolate8U(img, static_cast<float>(p0), static_cast<float>(p1)));
        if (y >= 1 && y < bpatch_size - 1 && x >= 1 && x < bpatch_size - 1) {
          *patch_ptr = *bpatch_ptr;
          patch_ptr++;
        }
      }
    }
  }

  // matcher.cc:359-445 ; Eigen Matrix3f::inverse = cofactor transpose * (1/det) (Appendix C)
  bool AlignPatch(const Image &img, const uint8_t *bpatch, const uint8_t *ptch, Vec2 *px) const {
    const int half_size = patch_size / 2;
    const int patch_area = patch_size * patch_size;
    bool converged = false;
    std::vector<float> patch_dx(patch_area), patch_dy(patch_area);
    float H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const int ref_step = patch_size + 2;
    float *it_dx = patch_dx.data();
    float *it_dy = patch_dy.data();
    for (int y = 0; y < patch_size; y++) {
      const uint8_t *it = bpatch + (y + 1) * ref_step + 1;
      for (int x = 0; x < patch_size; x++, it++, it_dx++, it_dy++) {
        float J[3];
        J[0] = static_cast<float>(0.5 * (it[1] - it[-1]));
        J[1] = static_cast<float>(0.5 * (it[ref_step] - it[-ref_step]));
        J[2] = 1;
        *it_dx = J[0];
        *it_dy = J[1];
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) H[r][c] += J[r] * J[c];
      }
    }
    float Hinv[3][3];
    Inverse3f(H, Hinv);
    float mean_diff = 0;
    float u = static_cast<float>(px->x);
    float v = static_cast<float>(px->y);
    const float min_update_squared = static_cast<float>(0.03 * 0.03);
    const int cur_step = img.step;
    for (int iter = 0; iter < prm.max_align_its; iter++) {
      const int u_r = static_cast<int>(std::floor(u));
      const int v_r = static_cast<int>(std::floor(v));
      if (u_r < half_size || v_r < half_size || u_r >= img.cols - half_size || v_r >= img.rows - half_size) break;
      if (std::isnan(u) || std::isnan(v)) return false;
      const float subpix_x = u - u_r;
      const float subpix_y = v - v_r;
      const float wTL = static_cast<float>((1.0 - subpix_x) * (1.0 - subpix_y));
      const float wTR = static_cast<float>(subpix_x * (1.0 - subpix_y));
      const float wBL = static_cast<float>((1.0 - subpix_x) * subpix_y);
      const float wBR = subpix_x * subpix_y;
      const uint8_t *it_ref = ptch;
      const float *it_ref_dx = patch_dx.data();
      const float *it_ref_dy = patch_dy.data();
      float Jres[3] = {0, 0, 0};
      for (int y = 0; y < patch_size; y++) {
        const uint8_t *it = img.data + static_cast<size_t>(v_r + y - half_size) * cur_step + u_r - half_size;
        for (int x = 0; x < patch_size; x++, it++, it_ref++, it_ref_dx++, it_ref_dy++) {
          const float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
          const float res = search_pixel - *it_ref + mean_diff;
          Jres[0] -= res * (*it_ref_dx);
          Jres[1] -= res * (*it_ref_dy);
          Jres[2] -= res;
        }
      }
      float update[3];
      for (int r = 0; r < 3; r++) update[r] = Hinv[r][0] * Jres[0] + Hinv[r][1] * Jres[1] + Hinv[r][2] * Jres[2];
      u += update[0];
      v += update[1];
      mean_diff += update[2];
      if (update[0] * update[0] + update[1] * update[1] < min_update_squared) {
        converged = true;
        break;
      }
    }
    px->x = u;
    px->y = v;
    return converged;
  }

  static void Inverse3f(const float m[3][3], float inv[3][3]) {
    // Eigen compute_inverse_size3: cofactor<i,j> = m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1), i1=(i+1)%3, i2=(i+2)%3 ...
    const float c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
    const float c10 = m[2][1] * m[0][2] - m[2][2] * m[0][1];
    const float c20 = m[0][1] * m[1][2] - m[0][2] * m[1][1];
    const float det = (c00 * m[0][0] + c10 * m[1][0]) + c20 * m[2][0];
    const float invdet = 1.0f / det;
    inv[0][0] = c00 * invdet;
    inv[0][1] = c10 * invdet;
    inv[0][2] = c20 * invdet;
    inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * invdet;
    inv[1][1] = (m[2][2] * m[0][0] - m[2][0] * m[0][2]) * invdet;
    inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * invdet;
    inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * invdet;
    inv[2][1] = (m[2][0] * m[0][1] - m[2][1] * m[0][0]) * invdet;
    inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * invdet;
  }

  // matcher.cc:447-476
  void GetZMSSDScore(const uint8_t *p, int *sumA, int *sumAA) const {
    uint32_t a = 0, aa = 0;
    const int parea = patch_size * patch_size;
    for (int r = 0; r < parea; r++) {
      const uint8_t n = p[r];
      a += n;
      aa += n * n;
    }
    *sumA = a;
    *sumAA = aa;
  }
  double CompareZMSSDScore(const uint8_t *ref_patch, const uint8_t *p, int sumA, int sumAA, int cols, int *sumB, int *sumBB,
                           int *sumAB) const {
    uint32_t b = 0, bb = 0, ab = 0;
    for (int y = 0, r = 0; y < patch_size; y++) {
      const uint8_t *patch_ptr = p + y * cols;
      for (int x = 0; x < patch_size; x++, r++) {
        const uint8_t pixel = patch_ptr[x];
        b += pixel;
        bb += pixel * pixel;
        ab += pixel * ref_patch[r];
      }
    }
    *sumB = b;
    *sumBB = bb;
    *sumAB = ab;
    return sumAA - 2 * (*sumAB) + (*sumBB) - (sumA * sumA - 2 * sumA * (*sumB) + (*sumB) * (*sumB)) / (patch_size * patch_size);
  }
};

}  // namespace sdvlref

#endif  // SDVL_ORACLE_REF_ALIGN_H_
