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
  else if (key == "SDVL.pyramid_levels") v_.pyramid_levels = static_cast<int>(v);
  else if (key == "SDVL.cell_size") v_.cell_size = static_cast<int>(v);
  else if (key == "SDVL.min_avg_shift") v_.min_avg_shift = static_cast<int>(v);
  else if (key == "SDVL.max_matches") v_.max_matches = static_cast<int>(v);
  else if (key == "SDVL.min_matches") v_.min_matches = static_cast<int>(v);
  else if (key == "SDVL.max_keyframes") v_.max_keyframes = static_cast<int>(v);
  else if (key == "SDVL.min_keyframe_its") v_.min_keyframe_its = static_cast<int>(v);
  else if (key == "SDVL.max_failed") v_.max_failed = static_cast<int>(v);
  else if (key == "SDVL.max_search_keyframes") v_.max_search_keyframes = static_cast<int>(v);
  else if (key == "SDVL.max_optim_pose_its") v_.max_optim_pose_its = static_cast<int>(v);
  else if (key == "SDVL.max_ransac_points") v_.max_ransac_points = static_cast<int>(v);
  else if (key == "SDVL.max_ransac_its") v_.max_ransac_its = static_cast<int>(v);
  else if (key == "SDVL.threshold_converged") v_.threshold_converged = v;
  else if (key == "SDVL.min_init_corners") v_.min_init_corners = static_cast<int>(v);
  else if (key == "SDVL.inlier_error_threshold") v_.inlier_error_threshold = v;
  else if (key == "SDVL.map_scale") v_.map_scale = v;
  else if (key == "SDVL.max_alignLevel") v_.max_align_level = static_cast<int>(v);
  else if (key == "SDVL.min_alignLevel") v_.min_align_level = static_cast<int>(v);
  else if (key == "SDVL.max_img_align_its") v_.max_img_align_its = static_cast<int>(v);
  else if (key == "SDVL.align_patch_size") v_.align_patch_size = static_cast<int>(v);
  else if (key == "SDVL.scale_min_dist") v_.scale_min_dist = v;
  else if (key == "SDVL.lost_ratio") v_.lost_ratio = v;
  else if (key == "SDVL.patch_size") v_.patch_size = static_cast<int>(v);
  else if (key == "SDVL.max_align_its") v_.max_align_its = static_cast<int>(v);
  else if (key == "SDVL.search_size") v_.search_size = static_cast<int>(v);
  else if (key == "SDVL.use_orb") v_.use_orb = (v != 0.0);
  else if (key == "SDVL.orb_size") v_.orb_size = static_cast<int>(v);
  else if (key == "SDVL.max_fast_levels") v_.max_fast_levels = static_cast<int>(v);
  else if (key == "SDVL.fast_threshold") v_.fast_threshold = static_cast<int>(v);
  else if (key == "SDVL.min_feature_score") v_.min_feature_score = static_cast<int>(v);
  else if (key == "SDVL.num_features") v_.num_features = static_cast<int>(v);
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
