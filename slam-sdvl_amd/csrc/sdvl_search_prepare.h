// sdvl_search_prepare.h — phase 0 of Matcher::SearchPoint (matcher.cc:45-96 with WarpMatrixAffine :293-312 and GetSearchLevel
// :314-323): scalar work per request — relative pose, depth-interval projection, margin test, affine warp, search level, the
// constants of GetCornersInRange (:139-148).  One LANE per request; search_points_kernel (one WAVE per request) reads the record
// with scalar loads.  Shared by search_prepare_kernel (sdvl_search.hip: requests that came from the host) and, round 4,
// track_project_kernel (sdvl_track.hip), whose lanes have the request they have just assembled in registers: a tracked step no
// longer launches search_prepare (10 us alone, 25-58 us among the other streams' kernels, per group-step).
#ifndef SDVL_SEARCH_PREPARE_H_
#define SDVL_SEARCH_PREPARE_H_

#include "sdvl_search_types.h"

// cf: the current frame's view when its corner bins are final at this point of the stream (null: the wave kernel looks them up)
__device__ __forceinline__ SearchPrep search_prepare_one(const SearchReqDev &rq, const sdvl::Rigid &cur_pose, const sdvl::Rigid &ref_pose,
                                                         const sdvl::Cam &cam, const sdvl_search_params &prm, const SearchFrame *cf = nullptr) {
  using namespace sdvl;
  SearchPrep out;
  out.bin_mode = 0;
  out.bin_pad_ = 0;
  for (int r = 0; r < 4; r++) out.bin_e0[r] = out.bin_pre[r] = 0;
  out.alive = 0;
  out.slevel = -1;
  out.pxa[0] = out.pxa[1] = out.pxb[0] = out.pxb[1] = 0.0;
  out.I00 = out.I01 = out.I10 = out.I11 = 0.0;
  out.nx = out.ny = out.normdist = out.xdiff = out.ydiff = out.vline = out.range = out.range2 = 0.0;
  const int level = rq.level;
  if (level < 0) return out;  // a dead slot of a device-built batch (sdvl_track.hip): no request here
  const Rigid ref_world = se3_inverse(ref_pose);
  const Rigid pose = se3_mul(cur_pose, ref_world);
  const V3 fvec = {rq.bearing[0], rq.bearing[1], rq.bearing[2]};
  const double idepth = rq.idepth, istd = rq.idepth_std;
  bool alive = true;
  V2 pxa = {0, 0}, pxb = {0, 0};
  {
    const double zmin = 1.0 / (idepth + 2.0 * istd);
    const V3 rel = se3_apply(cur_pose, se3_apply(ref_world, vscale_l(zmin, fvec)));
    if (rel.z < 0.0) alive = false;
    else pxa = cam_project(cam, rel);
    if (alive && !rq.fixed) {
      const double zmax = 1.0 / (fmax(idepth - 2.0 * istd, 0.00000001));
      const V3 rel2 = se3_apply(cur_pose, se3_apply(ref_world, vscale_l(zmax, fvec)));
      if (rel2.z < 0.0) alive = false;
      else pxb = cam_project(cam, rel2);
    }
  }
  if (alive) {
    const int lx = static_cast<int>(rq.px[0] / (1 << level)), ly = static_cast<int>(rq.px[1] / (1 << level));
    if (!cam_inside_level(cam, lx, ly, prm.patch_size / 2 + 2, level)) alive = false;
  }
  if (!alive) return out;
  // ---- WarpMatrixAffine, matcher.cc:293-312
  double A00, A01, A10, A11;
  {
    const int half_size = 5;
    const double depth = 1.0 / idepth;
    const V3 p3d = vscale(fvec, depth);
    V3 xyz_du = cam_unproject(cam, {rq.px[0] + static_cast<double>(half_size) * (1 << level), rq.px[1] + 0.0 * (1 << level)});
    V3 xyz_dv = cam_unproject(cam, {rq.px[0] + 0.0 * (1 << level), rq.px[1] + static_cast<double>(half_size) * (1 << level)});
    const double su = p3d.z / xyz_du.z;
    xyz_du = vscale(xyz_du, su);
    const double sv = p3d.z / xyz_dv.z;
    xyz_dv = vscale(xyz_dv, sv);
    const V2 px_cur = cam_project(cam, se3_apply(pose, p3d));
    const V2 px_du = cam_project(cam, se3_apply(pose, xyz_du));
    const V2 px_dv = cam_project(cam, se3_apply(pose, xyz_dv));
    A00 = (px_du.x - px_cur.x) / half_size;
    A10 = (px_du.y - px_cur.y) / half_size;
    A01 = (px_dv.x - px_cur.x) / half_size;
    A11 = (px_dv.y - px_cur.y) / half_size;
  }
  // ---- GetSearchLevel, matcher.cc:314-323
  int slevel = 0;
  {
    double det = A00 * A11 - A01 * A10;
    const int mx = prm.max_fast_levels - 1;
    while (det > 3.0 && slevel < mx) {
      slevel += 1;
      det *= 0.25;
    }
  }
  {
    const double det = A00 * A11 - A01 * A10;
    const double invdet = 1.0 / det;
    out.I00 = A11 * invdet; out.I01 = -A01 * invdet; out.I10 = -A10 * invdet; out.I11 = A00 * invdet;
  }
  out.alive = 1;
  out.slevel = slevel;
  out.pxa[0] = pxa.x; out.pxa[1] = pxa.y; out.pxb[0] = pxb.x; out.pxb[1] = pxb.y;
  {
    double range = prm.search_size;
    for (int i = 1; i <= slevel; i++) range *= 1.2;
    out.range = range;
    out.range2 = range * range;
    if (!rq.fixed) {  // epipolar line constants (matcher.cc:139-148); a fixed search tests a circle around px0 only
      double ex = pxa.x - pxb.x, ey = pxa.y - pxb.y;
      const double en = sqrt(ex * ex + ey * ey);
      ex /= en;
      ey /= en;
      out.nx = ey;
      out.ny = -ex;
      out.normdist = pxa.x * out.nx + pxa.y * out.ny;
      out.xdiff = pxb.x - pxa.x;
      out.ydiff = pxb.y - pxa.y;
      out.vline = (out.xdiff) * (out.xdiff) + (out.ydiff) * (out.ydiff);
    }
  }
  // ---- the cells of the corner bins the search region touches (the statements of search_points_kernel's own look-up, same doubles)
  if (cf && cf->bin_start != nullptr) {
    const bool line_ok = rq.fixed || (out.vline > 0.0 && out.nx == out.nx && out.ny == out.ny);
    if (line_ok) {
      const double range = out.range;
      double bx0, bx1, by0, by1;
      if (rq.fixed) {
        bx0 = rq.px0[0] - range; bx1 = rq.px0[0] + range; by0 = rq.px0[1] - range; by1 = rq.px0[1] + range;
      } else {
        bx0 = fmin(pxa.x, pxb.x) - range; bx1 = fmax(pxa.x, pxb.x) + range; by0 = fmin(pxa.y, pxb.y) - range; by1 = fmax(pxa.y, pxb.y) + range;
      }
      if (bx0 > -1.0e6 && bx1 < 1.0e6 && by0 > -1.0e6 && by1 < 1.0e6) {
        const int gw = cf->bin_gw, gh = cf->bin_cells / cf->bin_gw;
        const int cx0 = max(0, static_cast<int>(floor(bx0)) >> 5), cx1 = min(gw - 1, static_cast<int>(floor(bx1)) >> 5);
        const int cy0 = max(0, static_cast<int>(floor(by0)) >> 5), cy1 = min(gh - 1, static_cast<int>(floor(by1)) >> 5);
        if (cx1 < cx0 || cy1 < cy0) {
          out.bin_mode = 1;
        } else if (cy1 - cy0 < 4 && (cx1 - cx0 + 1) * (cy1 - cy0 + 1) <= kBinRegionCells) {
          out.bin_mode = 2;
          int acc = 0;
          for (int r = 0; r < 4; r++) {
            const int cyi = min(cy0 + r, cy1);
            const int a = cf->bin_start[cyi * gw + cx0], b = cf->bin_start[cyi * gw + cx1 + 1];
            out.bin_e0[r] = a;
            acc += (cy0 + r <= cy1) ? b - a : 0;
            out.bin_pre[r] = acc;
          }
        }
      }
    }
  }
  return out;
}

#endif  // SDVL_SEARCH_PREPARE_H_
