// config.h — Config singleton with the reference's getters and defaults (config.h:54-104, config.cc:55-85).
// ReadParameters parses the "key: value" subset of the YAML 1.0 files under config/ (config.cc:88-164 used
// cv::FileStorage).
#ifndef SDVL_HOST_CONFIG_H_
#define SDVL_HOST_CONFIG_H_

#include <string>

namespace sdvl {

struct CameraParameters {
  int width = 640, height = 480;
  double fx = 300.0, fy = 300.0, u0 = 320.0, v0 = 240.0;
  double d1 = 0.0, d2 = 0.0, d3 = 0.0, d4 = 0.0, d5 = 0.0;
};

class Config {
 public:
  static Config &GetInstance() {
    static Config c;
    return c;
  }
  bool ReadParameters(const std::string &filename);
  bool SetParameter(const std::string &key, double value);  // "SDVL.max_matches" style keys
  void Reset() { *this = Config(); }

  static CameraParameters &GetCameraParameters() { return GetInstance().camera_params_; }
  static int PyramidLevels() { return GetInstance().v_.pyramid_levels; }
  static int CellSize() { return GetInstance().v_.cell_size; }
  static int MinAvgShift() { return GetInstance().v_.min_avg_shift; }
  static int MaxMatches() { return GetInstance().v_.max_matches; }
  static int MinMatches() { return GetInstance().v_.min_matches; }
  static int MaxKeyframes() { return GetInstance().v_.max_keyframes; }
  static int MinKeyframeIts() { return GetInstance().v_.min_keyframe_its; }
  static int MaxFailed() { return GetInstance().v_.max_failed; }
  static int MaxSearchKeyframes() { return GetInstance().v_.max_search_keyframes; }
  static int MaxOptimPoseIts() { return GetInstance().v_.max_optim_pose_its; }
  static int MaxRansacPoints() { return GetInstance().v_.max_ransac_points; }
  static int MaxRansacIts() { return GetInstance().v_.max_ransac_its; }
  static double ThresholdConverged() { return GetInstance().v_.threshold_converged; }
  static int MinInitCorners() { return GetInstance().v_.min_init_corners; }
  static double InlierErrorThreshold() { return GetInstance().v_.inlier_error_threshold; }
  static double MapScale() { return GetInstance().v_.map_scale; }
  static int MaxAlignLevel() { return GetInstance().v_.max_align_level; }
  static int MinAlignLevel() { return GetInstance().v_.min_align_level; }
  static int MaxImgAlignIts() { return GetInstance().v_.max_img_align_its; }
  static int AlignPatchSize() { return GetInstance().v_.align_patch_size; }
  static double ScaleMinDist() { return GetInstance().v_.scale_min_dist; }
  static double LostRatio() { return GetInstance().v_.lost_ratio; }
  static int PatchSize() { return GetInstance().v_.patch_size; }
  static int MaxAlignIts() { return GetInstance().v_.max_align_its; }
  static int SearchSize() { return GetInstance().v_.search_size; }
  static bool UseORB() { return GetInstance().v_.use_orb; }
  static int ORBSize() { return GetInstance().v_.orb_size; }
  static int MaxFastLevels() { return GetInstance().v_.max_fast_levels; }
  static int FastThreshold() { return GetInstance().v_.fast_threshold; }
  static int MinFeatureScore() { return GetInstance().v_.min_feature_score; }
  static int NumFeatures() { return GetInstance().v_.num_features; }

 private:
  Config() {}
  CameraParameters camera_params_;
  // the tunables and their defaults (config.cc:55-85), one record: Reset() and the key table of config.cc address it as a whole
  struct Values {
    int pyramid_levels = 5, cell_size = 32, min_avg_shift = 50, max_matches = 150, min_matches = 20, max_keyframes = 100,
        min_keyframe_its = 30, max_failed = 15, max_search_keyframes = 5, max_optim_pose_its = 10, max_ransac_points = 5,
        max_ransac_its = 100, min_init_corners = 50, max_align_level = 4, min_align_level = 2, max_img_align_its = 30,
        align_patch_size = 4, patch_size = 8, max_align_its = 10, search_size = 6, orb_size = 31, max_fast_levels = 3,
        fast_threshold = 10, min_feature_score = 50, num_features = 1000;
    double threshold_converged = 0.1, inlier_error_threshold = 2.0, map_scale = 1.0, scale_min_dist = 0.25, lost_ratio = 0.7;
    bool use_orb = false;
  };
  Values v_;
};

}  // namespace sdvl

#endif  // SDVL_HOST_CONFIG_H_
