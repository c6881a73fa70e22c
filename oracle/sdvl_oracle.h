/* ORACLE — TEST INFRASTRUCTURE ONLY.  C-ABI of the CPU restatement (libsdvl_oracle.so).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * PARITY UNPINNED (see oracle/ref_math.h).  All images are 8-bit, row-major; poses are 7 doubles
 * (qw,qx,qy,qz,tx,ty,tz) of the world->camera transform the reference stores in Frame::pose_. */
#ifndef SDVL_ORACLE_H_
#define SDVL_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors sdvlref::Params (oracle/ref_detect.h) */
typedef struct sdvl_ref_params {
  int pyramid_levels, cell_size, max_fast_levels, fast_threshold, num_features, use_orb, orb_size, patch_size,
      max_align_its, search_size, align_patch_size, max_align_level, min_align_level, max_img_align_its,
      min_feature_score, max_matches, min_matches, max_failed, max_optim_pose_its, max_ransac_points,
      max_ransac_its, min_keyframe_its;
  double inlier_error_threshold, lost_ratio;
} sdvl_ref_params;

typedef struct sdvl_ref_frame_stats {
  int state, quality, matches, attempts, inliers, outliers, n_corners, align_meas, keyframe, relocalized;
  double pose[7];
} sdvl_ref_frame_stats;

void sdvl_ref_default_params(sdvl_ref_params *p);

/* frame.cc:114-120 */
int sdvl_ref_pyr_down(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dst_stride);
/* cv::FAST on a whole image / ROI; out_xys = [cap][3] (x, y, score); returns count (may exceed cap) */
int sdvl_ref_fast(const uint8_t *img, int w, int h, int stride, int thr, int nonmax, int cap, int32_t *out_xys);
/* fast_detector.cc:79-106: per-cell lists, scan order, image coords; cell_offsets[ncells+1]; ran[ncells] */
int sdvl_ref_fast_cells(const uint8_t *img, int w, int h, int stride, const sdvl_ref_params *p, int cap,
                        int32_t *out_xys, int32_t *cell_offsets, uint8_t *ran);
/* fast_detector.cc:154-175 on a pyramid built from img; corners = [cap][3] (x,y,level) */
int sdvl_ref_detect_pyramid(const uint8_t *img, int w, int h, int stride, const sdvl_ref_params *p, int nfeatures,
                            int cap, int32_t *corners);
/* cv::KeyPointsFilter::retainBest on packed keypoints (x | y<<12 | response<<24), in place; returns the new length */
int sdvl_ref_retain_best(uint32_t *packed, int len, int n_points);
/* extra/utils.cc:61-97 */
double sdvl_ref_shi_tomasi(const uint8_t *img, int w, int h, int stride, int x, int y);
/* frame.cc:133-163 / fast_detector.cc:177-218; locked = [n_locked][2] level-0 positions of existing features */
int sdvl_ref_filter_corners(const uint8_t *img, int w, int h, int stride, const sdvl_ref_params *p, int n,
                            const int32_t *corners, int n_locked, const double *locked, int cap, int32_t *indices);
/* orb_detector.cc:350-437 */
void sdvl_ref_orb_describe(const uint8_t *img, int w, int h, int stride, int n, const int32_t *xy, uint8_t *desc,
                           float *angles_deg);
int sdvl_ref_orb_distance(const uint8_t *a, const uint8_t *b);
/* image_align.cc:46-84.  cam = fx,fy,u0,v0.  T_io = frame2.pose * frame1.pose^-1.  its[8] per level */
int sdvl_ref_image_align(const uint8_t *img1, const uint8_t *img2, int w, int h, const sdvl_ref_params *p,
                         const double *cam, int n, const double *px, const double *bearing, const double *depth,
                         const uint8_t *valid, double *T_io, int fast, double *error, double *chi2, int *its);
/* matcher.cc:45-121 on pyramids built from the two level-0 images.  corners = [n][3], descs: lazily computed.
 * px_io: in = search centre for fixed points; out = match.  returns found; out_level */
int sdvl_ref_search_point(const uint8_t *ref_img, const uint8_t *cur_img, int w, int h, const sdvl_ref_params *p,
                          const double *cam, const double *ref_pose, const double *cur_pose, const double *feat_px,
                          const double *feat_bearing, int feat_level, const uint8_t *feat_desc, double idepth,
                          double idepth_std, int fixed, int n_corners, const int32_t *corners, double *px_io,
                          int *out_level, uint8_t *out_border_patch, int *out_slevel);
/* matcher.cc:359-445 */
int sdvl_ref_align_patch(const uint8_t *img, int w, int h, int stride, const uint8_t *border_patch,
                         const uint8_t *patch, double *px_io, int max_its);

/* small math, for cross-checks against numpy/scipy */
void sdvl_ref_se3_exp(const double *u6, double *T7);
void sdvl_ref_se3_log(const double *T7, double *u6);
void sdvl_ref_se3_mul(const double *A7, const double *B7, double *C7);
void sdvl_ref_se3_inv(const double *A7, double *B7);
void sdvl_ref_ldlt_solve6(const double *A36, const double *b6, double *x6);
void sdvl_ref_rand_stream(unsigned seed, int n, int *out);

/* Camera::UndistortImage = cv::undistort(in, out, K, D) (camera.cc:39-67,100-105).  Returns 1 if remapped, 0 if cloned */
int sdvl_ref_undistort(const uint8_t *img, int w, int h, int stride, const double *cam4, const double *dist5, uint8_t *out);
void sdvl_ref_remap_weights(int16_t *out4096);

/* FeatureAlign::SelectInliers + OptimizePose (feature_align.cc:73-82,152-243) on a given match list:
 * obs[n][6] = {ax, ay, px, py, pz, level}; the glibc rand stream is seeded with rand_seed and advanced rand_skip draws */
int sdvl_ref_pose_from_matches(const sdvl_ref_params *p, int w, int h, const double *cam, int n, const double *obs,
                               unsigned rand_seed, int rand_skip, double *pose7_io, int *n_draws, int *n_in, int *in_idx,
                               int *n_out, int *out_idx);

/* closed-loop tracker: sdvl.cc:55-130 with the plane map stub (oracle/ref_tracker.h) */
void *sdvl_ref_tracker_create(const sdvl_ref_params *p, int w, int h, const double *cam, const double *plane4,
                              const double *first_pose7);
void sdvl_ref_tracker_destroy(void *t);
void sdvl_ref_tracker_set_max_keyframes(void *t, int max_keyframes);
void sdvl_ref_tracker_use_mapper(void *t, int on, int max_search_keyframes, int max_keyframes, double map_scale, double scale_min_dist);
void sdvl_ref_tracker_map_stats(void *t, int *out6);
int sdvl_ref_tracker_mapper_points(void *t, int cap, double *out_xyzc);
int sdvl_ref_triangulate(const double *pose7, const double *v_ref3, const double *v_cur3, double *depth);
double sdvl_ref_pdf_normal(double mean, double sd, double x);
double sdvl_ref_compute_tau(const double *pose7, const double *v3, double depth, double px_error_angle);
/* the body of Map::UpdateCandidates' loop behind SearchPoint (map.cc:454-497) for ONE candidate, with the tracker's camera and
 * configuration: state = {rho, sigma2, a, b, z_range, cos_alpha, last_distance, p3d[3], fixed, n_failed} in / out (12 doubles);
 * returns 0 not found (Unpromote; | 0x100 deleted), 1 skipped, 2 updated, 3 updated and converged */
int sdvl_ref_depth_filter(void *t, const double *cur_pose7, const double *ref_pose7, const double *bearing3, int found, const double *px2,
                          double depth_mean, double *state12);
int sdvl_ref_tracker_handle_frame(void *t, const uint8_t *img, int stride, sdvl_ref_frame_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* SDVL_ORACLE_H_ */
