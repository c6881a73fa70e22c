// MOCK of <opencv2/core.hpp> — compile-only stand-in for the cv::Mat members host/types.h touches when the build has OpenCV
// (empty(), type(), data, cols, rows, step, the header-over-pixels constructor, CV_8UC1).  Not a library; see Eigen/Dense beside it.
#ifndef SDVL_MOCK_OPENCV_CORE_
#define SDVL_MOCK_OPENCV_CORE_

#include <stddef.h>

#define CV_8UC1 0

namespace cv {

class Mat {
 public:
  struct Step {
    size_t bytes = 0;
    operator size_t() const { return bytes; }
  };
  Mat() {}
  Mat(int rows_, int cols_, int type, void *pixels, size_t step_ = 0) : data(static_cast<unsigned char *>(pixels)), cols(cols_), rows(rows_), type_(type) {
    step.bytes = step_ ? step_ : static_cast<size_t>(cols_);
  }
  bool empty() const { return data == nullptr || cols == 0 || rows == 0; }
  int type() const { return type_; }
  unsigned char *data = nullptr;
  int cols = 0, rows = 0;
  Step step;

 private:
  int type_ = CV_8UC1;
};

}  // namespace cv

#endif
