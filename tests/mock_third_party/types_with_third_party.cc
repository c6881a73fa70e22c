// compile-only (tests/test_frontend_split.py): the whole host layer's headers with types.h's Eigen / OpenCV branch switched on, and
// the conversions a caller written against the reference's headers relies on (cv::Mat frames in, Eigen vectors in and out)
#include "sdvl_host.h"

#if !defined(SDVL_HAVE_EIGEN) || !defined(SDVL_HAVE_OPENCV)
#error "the mock <Eigen/Dense> / <opencv2/core.hpp> were not picked up: add -Itests/mock_third_party"
#endif

namespace {

Eigen::Vector2d RoundTrip(const Eigen::Vector3d &p3, const sdvl::Camera &cam) {
  const sdvl::Vector3d p = p3;                 // Eigen -> Vec
  const sdvl::Vector2d px = cam.Project(p);
  const Eigen::Vector2d out = px;              // Vec -> Eigen
  return out;
}

int FrameFromMat(sdvl::Camera *cam, sdvl::ORBDetector *orb, unsigned char *pixels) {
  cv::Mat m(480, 640, CV_8UC1, pixels);
  const sdvl::Image img(m);                    // sdvl.h:62 / frame.h:45 take a cv::Mat
  const cv::Mat back = img;                    // GetPyramid() users read cv::Mat headers (homography_init.cc:196)
  (void)cam; (void)orb;
  return img.cols + back.rows;
}

}  // namespace

int sdvl_types_with_third_party_anchor() {
  Eigen::Vector3d p;
  p(2) = 1.0;
  sdvl::Camera cam(640, 480, 500, 500, 320, 240);
  unsigned char px[4] = {0, 0, 0, 0};
  return static_cast<int>(RoundTrip(p, cam)(0)) + FrameFromMat(&cam, nullptr, px);
}
