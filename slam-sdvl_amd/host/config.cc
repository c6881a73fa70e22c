// config.cc — key: value reader for the reference's config files (config.cc:88-164 key set).
#include "config.h"

#include <cstdlib>
#include <fstream>
#include <iostream>

namespace sdvl {

bool Config::SetParameter(const std::string &key, double v) {
  CameraParameters &c = camera_params_;
  if (key == "Camera.width") c.width = static_cast<int>(v);
  else if (key == "Camera.height") c.height = static_cast<int>(v);
  else if (key == "Camera.fx") c.fx = v;
  else if (key == "Camera.fy") c.fy = v;
  else if (key == "Camera.u0") c.u0 = v;
  else if (key == "Camera.v0") c.v0 = v;
  else if (key == "Camera.d1") c.d1 = v;
  else if (key == "Camera.d2") c.d2 = v;
  else if (key == "Camera.d3") c.d3 = v;
  else if (key == "Camera.d4") c.d4 = v;
  else if (key == "Camera.d5") c.d5 = v;
  else if (key == "SDVL.pyramid_levels") kPyramidLevels_ = static_cast<int>(v);
  else if (key == "SDVL.cell_size") kCellSize_ = static_cast<int>(v);
  else if (key == "SDVL.min_avg_shift") kMinAvgShift_ = static_cast<int>(v);
  else if (key == "SDVL.max_matches") kMaxMatches_ = static_cast<int>(v);
  else if (key == "SDVL.min_matches") kMinMatches_ = static_cast<int>(v);
  else if (key == "SDVL.max_keyframes") kMaxKeyframes_ = static_cast<int>(v);
  else if (key == "SDVL.min_keyframe_its") kMinKeyframeIts_ = static_cast<int>(v);
  else if (key == "SDVL.max_failed") kMaxFailed_ = static_cast<int>(v);
  else if (key == "SDVL.max_search_keyframes") kMaxSearchKeyframes_ = static_cast<int>(v);
  else if (key == "SDVL.max_optim_pose_its") kMaxOptimPoseIts_ = static_cast<int>(v);
  else if (key == "SDVL.max_ransac_points") kMaxRansacPoints_ = static_cast<int>(v);
  else if (key == "SDVL.max_ransac_its") kMaxRansacIts_ = static_cast<int>(v);
  else if (key == "SDVL.threshold_converged") kThresholdConverged_ = v;
  else if (key == "SDVL.min_init_corners") kMinInitCorners_ = static_cast<int>(v);
  else if (key == "SDVL.inlier_error_threshold") kInlierErrorThreshold_ = v;
  else if (key == "SDVL.map_scale") kMapScale_ = v;
  else if (key == "SDVL.max_alignLevel") kMaxAlignLevel_ = static_cast<int>(v);
  else if (key == "SDVL.min_alignLevel") kMinAlignLevel_ = static_cast<int>(v);
  else if (key == "SDVL.max_img_align_its") kMaxImgAlignIts_ = static_cast<int>(v);
  else if (key == "SDVL.align_patch_size") kAlignPatchSize_ = static_cast<int>(v);
  else if (key == "SDVL.scale_min_dist") kScaleMinDist_ = v;
  else if (key == "SDVL.lost_ratio") kLostRatio_ = v;
  else if (key == "SDVL.patch_size") kPatchSize_ = static_cast<int>(v);
  else if (key == "SDVL.max_align_its") kMaxAlignIts_ = static_cast<int>(v);
  else if (key == "SDVL.search_size") kSearchSize_ = static_cast<int>(v);
  else if (key == "SDVL.use_orb") kUseORB_ = (v != 0.0);
  else if (key == "SDVL.orb_size") kORBSize_ = static_cast<int>(v);
  else if (key == "SDVL.max_fast_levels") kMaxFastLevels_ = static_cast<int>(v);
  else if (key == "SDVL.fast_threshold") kFastThreshold_ = static_cast<int>(v);
  else if (key == "SDVL.min_feature_score") kMinFeatureScore_ = static_cast<int>(v);
  else if (key == "SDVL.num_features") kNumFeatures_ = static_cast<int>(v);
  else return false;
  return true;
}

bool Config::ReadParameters(const std::string &filename) {
  std::ifstream in(filename.c_str());
  if (!in.is_open()) {
    std::cerr << "[ERROR] Failed to open file: " << filename << std::endl;
    return false;
  }
  std::string line;
  while (std::getline(in, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    if (line.empty() || line[0] == '%') continue;
    const size_t colon = line.find(':');
    if (colon == std::string::npos) continue;
    std::string key = line.substr(0, colon), val = line.substr(colon + 1);
    while (!key.empty() && (key.back() == ' ' || key.back() == '\t')) key.pop_back();
    size_t b = val.find_first_not_of(" \t");
    if (b == std::string::npos) continue;
    val = val.substr(b);
    if (val[0] == '"') continue;  // Video.path / Video.filename strings: I/O is out of scope
    char *end = nullptr;
    const double v = std::strtod(val.c_str(), &end);
    if (end == val.c_str()) continue;
    SetParameter(key, v);
  }
  return true;
}

}  // namespace sdvl
