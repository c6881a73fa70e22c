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
};
typedef Vec<double, 2> Vector2d;
typedef Vec<double, 3> Vector3d;
typedef Vec<int, 2> Vector2i;
typedef Vec<int, 3> Vector3i;
typedef Vec<double, 6> Vector6d;

// cv::Mat (CV_8UC1) subset.  `data` is a host mirror (may be null until HostData() is called on the owning
// Frame); dev/level bind the image to an HBM-resident pyramid level.
enum { CV_8UC1 = 0 };  // the only cv::Mat type the tracking front-end sees (frame.cc:38-41 converts to grey)

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
  static Image WrapDevice(const void *dev_ptr, int w, int h, int stride, bool borrow_storage = false, bool transient_storage = false) {
    Image r;
    r.dev_src = dev_ptr; r.cols = w; r.rows = h; r.step = stride; r.borrow = borrow_storage; r.transient = borrow_storage && transient_storage;
    return r;
  }
};

}  // namespace sdvl

#endif  // SDVL_HOST_TYPES_H_
