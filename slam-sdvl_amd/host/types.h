// types.h — the small Eigen / OpenCV look-alikes the SDVL API surface is written in (frame.h:45-139,
// feature.h:42-94, sdvl.h:49-69).  Only what the tracking front-end touches: fixed-size vectors with Eigen's
// operator() access, and Image = the cv::Mat (CV_8UC1) subset {data, cols, rows, step} plus a binding to the
// HBM-resident pyramid level it mirrors.  With real Eigen/OpenCV available a maintainer maps these 1:1
// (INTEGRATION.md).
#ifndef SDVL_HOST_TYPES_H_
#define SDVL_HOST_TYPES_H_

#include <stdint.h>

#include <memory>
#include <vector>

// Real Eigen / OpenCV, when the build has them (this image has neither): the look-alikes below then convert to and from the real
// types, so callers written against the reference's headers (sdvl.cc, map.cc, ui/: cv::Mat frames in, Eigen vectors out) pass and
// receive their own types — SDVL::HandleFrame(const cv::Mat&) (sdvl.h:62), Frame(..., const cv::Mat&, ...) (frame.h:45),
// Feature::GetPosition() -> Eigen::Vector2d (feature.h:66).  -DSDVL_NO_THIRD_PARTY_TYPES keeps the build free of both.
#if !defined(SDVL_NO_THIRD_PARTY_TYPES) && defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define SDVL_HAVE_EIGEN 1
#endif
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define SDVL_HAVE_OPENCV 1
#endif
#endif

struct sdvl_frame;

namespace sdvl {

typedef unsigned char uchar;

template <typename T, int N>
struct Vec {
  T v[N];
  Vec() { for (int i = 0; i < N; i++) v[i] = T(); }
  Vec(T a, T b) { static_assert(N == 2, "size"); v[0] = a; v[1] = b; }
  Vec(T a, T b, T c) { static_assert(N == 3, "size"); v[0] = a; v[1] = b; v[2] = c; }
  T &operator()(int i) { return v[i]; }
  const T &operator()(int i) const { return v[i]; }
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
  T x() const { return v[0]; }
  T y() const { return v[1]; }
#ifdef SDVL_HAVE_EIGEN
  // Eigen::Matrix<T, N, 1> in and out (Vector2d / Vector3d / Vector3i / Vector6d of the reference's signatures)
  Vec(const Eigen::Matrix<T, N, 1> &e) { for (int i = 0; i < N; i++) v[i] = e(i); }
  operator Eigen::Matrix<T, N, 1>() const {
    Eigen::Matrix<T, N, 1> e;
    for (int i = 0; i < N; i++) e(i) = v[i];
    return e;
  }
#endif
};
typedef Vec<double, 2> Vector2d;
typedef Vec<double, 3> Vector3d;
typedef Vec<int, 2> Vector2i;
typedef Vec<int, 3> Vector3i;
typedef Vec<double, 6> Vector6d;

// cv::Mat (CV_8UC1) subset.  `data` is a host mirror (may be null until HostData() is called on the owning
// Frame); dev/level bind the image to an HBM-resident pyramid level.
#ifndef SDVL_HAVE_OPENCV
enum { CV_8UC1 = 0 };  // the only cv::Mat type the tracking front-end sees (frame.cc:38-41 converts to grey)
#endif

struct Image {
  const uint8_t *data = nullptr;
  int cols = 0, rows = 0, step = 0;
  Image() {}
  // cv::Mat(rows, cols, type, data, step): a header over caller-owned pixels
  Image(int rows_, int cols_, int /*type*/, const void *pixels, size_t step_ = 0)
      : data(static_cast<const uint8_t *>(pixels)), cols(cols_), rows(rows_), step(step_ ? static_cast<int>(step_) : cols_) {}
  sdvl_frame *dev = nullptr;
  int level = 0;
  std::shared_ptr<std::vector<uint8_t>> owner;  // keeps a host copy alive (cv::Mat ref-count analogue)
  bool empty() const { return cols == 0 || rows == 0; }
  Image clone() const {
    Image r = *this;
    if (data) {
      r.owner = std::make_shared<std::vector<uint8_t>>(static_cast<size_t>(cols) * rows);
      for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) (*r.owner)[static_cast<size_t>(y) * cols + x] = data[static_cast<size_t>(y) * step + x];
      r.data = r.owner->data();
      r.step = cols;
    }
    return r;
  }
#ifdef SDVL_HAVE_OPENCV
  // a cv::Mat frame as the reference passes it (CV_8UC1, main.cc:128-138): a header over its pixels, the Mat kept alive
  Image(const cv::Mat &m) {
    if (!m.empty() && m.type() == CV_8UC1) {
      auto keep = std::make_shared<cv::Mat>(m);
      mat_owner = keep;
      data = keep->data;
      cols = keep->cols;
      rows = keep->rows;
      step = static_cast<int>(keep->step);
    }
  }
  // the host mirror as a cv::Mat header (GetPyramid() users: homography_init.cc:196, ui/drawimage.cc); empty while only the HBM copy exists
  operator cv::Mat() const { return data ? cv::Mat(rows, cols, CV_8UC1, const_cast<uint8_t *>(data), static_cast<size_t>(step)) : cv::Mat(); }
  std::shared_ptr<void> mat_owner;
#endif
  static Image Wrap(const uint8_t *p, int w, int h, int stride) {
    Image r;
    r.data = p; r.cols = w; r.rows = h; r.step = stride;
    return r;
  }
  // an image that already lives in HBM (device pointer), e.g. a decoded camera frame
  const void *dev_src = nullptr;
  std::shared_ptr<void> dev_owner;  // keeps an HBM image produced by this layer alive (Camera::UndistortImage)
  bool borrow = false;  // the HBM image outlives every Frame built from it: alias it instead of copying
  // borrow + transient: the HBM image stays valid for the step that tracks it only (a slot of an input ring): the Frame aliases it
  // and takes a copy if it becomes a keyframe (Frame::OwnImages) — one frame in five in S-A, the other four never copy
  bool transient = false;
  // (an addition) a RAW camera image: the Frame built from it takes Camera::UndistortImage(image) (camera.cc:100-105, main.cc:133) as its
  // level 0 — the undistortion fused into the frame's upload, for callers that hand whole batches of camera frames to SDVLBatch
  bool raw = false;
  static Image WrapDevice(const void *dev_ptr, int w, int h, int stride, bool borrow_storage = false, bool transient_storage = false) {
    Image r;
    r.dev_src = dev_ptr; r.cols = w; r.rows = h; r.step = stride; r.borrow = borrow_storage; r.transient = borrow_storage && transient_storage;
    return r;
  }
};

}  // namespace sdvl

#endif  // SDVL_HOST_TYPES_H_
