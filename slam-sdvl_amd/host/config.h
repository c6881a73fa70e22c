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
  static int PyramidLevels() { return GetInstance().kPyramidLevels_; }
  static int CellSize() { return GetInstance().kCellSize_; }
  static int MinAvgShift() { return GetInstance().kMinAvgShift_; }
  static int MaxMatches() { return GetInstance().kMaxMatches_; }
  static int MinMatches() { return GetInstance().kMinMatches_; }
  static int MaxKeyframes() { return GetInstance().kMaxKeyframes_; }
  static int MinKeyframeIts() { return GetInstance().kMinKeyframeIts_; }
  static int MaxFailed() { return GetInstance().kMaxFailed_; }
  static int MaxSearchKeyframes() { return GetInstance().kMaxSearchKeyframes_; }
  static int MaxOptimPoseIts() { return GetInstance().kMaxOptimPoseIts_; }
  static int MaxRansacPoints() { return GetInstance().kMaxRansacPoints_; }
  static int MaxRansacIts() { return GetInstance().kMaxRansacIts_; }
  static double ThresholdConverged() { return GetInstance().kThresholdConverged_; }
  static int MinInitCorners() { return GetInstance().kMinInitCorners_; }
  static double InlierErrorThreshold() { return GetInstance().kInlierErrorThreshold_; }
  static double MapScale() { return GetInstance().kMapScale_; }
  static int MaxAlignLevel() { return GetInstance().kMaxAlignLevel_; }
  static int MinAlignLevel() { return GetInstance().kMinAlignLevel_; }
  static int MaxImgAlignIts() { return GetInstance().kMaxImgAlignIts_; }
  static int AlignPatchSize() { return GetInstance().kAlignPatchSize_; }
  static double ScaleMinDist() { return GetInstance().kScaleMinDist_; }
  static double LostRatio() { return GetInstance().kLostRatio_; }
  static int PatchSize() { return GetInstance().kPatchSize_; }
  static int MaxAlignIts() { return GetInstance().kMaxAlignIts_; }
  static int SearchSize() { return GetInstance().kSearchSize_; }
  static bool UseORB() { return GetInstance().kUseORB_; }
  static int ORBSize() { return GetInstance().kORBSize_; }
  static int MaxFastLevels() { return GetInstance().kMaxFastLevels_; }
  static int FastThreshold() { return GetInstance().kFastThreshold_; }
  static int MinFeatureScore() { return GetInstance().kMinFeatureScore_; }
  static int NumFeatures() { return GetInstance().kNumFeatures_; }

 private:
  Config() {}
  CameraParameters camera_params_;
  // defaults: config.cc:55-85
  int kPyramidLevels_ = 5, kCellSize_ = 32, kMinAvgShift_ = 50, kMaxMatches_ = 150, kMinMatches_ = 20, kMaxKeyframes_ = 100,
      kMinKeyframeIts_ = 30, kMaxFailed_ = 15, kMaxSearchKeyframes_ = 5, kMaxOptimPoseIts_ = 10, kMaxRansacPoints_ = 5,
      kMaxRansacIts_ = 100, kMinInitCorners_ = 50, kMaxAlignLevel_ = 4, kMinAlignLevel_ = 2, kMaxImgAlignIts_ = 30,
      kAlignPatchSize_ = 4, kPatchSize_ = 8, kMaxAlignIts_ = 10, kSearchSize_ = 6, kORBSize_ = 31, kMaxFastLevels_ = 3,
      kFastThreshold_ = 10, kMinFeatureScore_ = 50, kNumFeatures_ = 1000;
  double kThresholdConverged_ = 0.1, kInlierErrorThreshold_ = 2.0, kMapScale_ = 1.0, kScaleMinDist_ = 0.25, kLostRatio_ = 0.7;
  bool kUseORB_ = false;
};

}  // namespace sdvl

#endif  // SDVL_HOST_CONFIG_H_
