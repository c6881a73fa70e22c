// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_math.h header).  PARITY UNPINNED.
//
// ref_detect.h — image pyramid, FAST-9/16 per-cell detection, Shi-Tomasi corner filter.
//   Pyramid:       /root/reference/frame.cc:114-120  (cv::pyrDown)
//   FAST:          /root/reference/extra/fast_detector.cc:58-175 (cv::FAST TYPE_9_16 + cornerScore<16>, A.2;
//                  cv::KeyPointsFilter::retainBest, A.3 — third-party OpenCV 2.4/3.x, version unpinned)
//   FilterCorners: /root/reference/extra/fast_detector.cc:177-218, extra/utils.cc:61-97
#ifndef SDVL_ORACLE_REF_DETECT_H_
#define SDVL_ORACLE_REF_DETECT_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace sdvlref {

// The subset of cv::Mat (CV_8UC1) the path uses: data / cols / rows / step. Non-owning view.
struct Image {
  const uint8_t *data = nullptr;
  int cols = 0, rows = 0, step = 0;
  const uint8_t *ptr(int y) const { return data + static_cast<size_t>(y) * step; }
  uint8_t at(int y, int x) const { return data[static_cast<size_t>(y) * step + x]; }
};

// Tunables: defaults of config.cc:55-85 overridden by config/config_tum_f1.cfg:34-42 (SURVEY §8 header).
struct Params {
  int pyramid_levels = 5;
  int cell_size = 32;
  int max_fast_levels = 3;
  int fast_threshold = 10;
  int num_features = 1000;
  int use_orb = 1;
  int orb_size = 31;
  int patch_size = 8;
  int max_align_its = 10;
  int search_size = 6;
  int align_patch_size = 4;
  int max_align_level = 4;
  int min_align_level = 2;
  int max_img_align_its = 30;
  int min_feature_score = 50;
  int max_matches = 200;
  int min_matches = 20;
  int max_failed = 15;
  int max_optim_pose_its = 10;
  int max_ransac_points = 5;
  int max_ransac_its = 100;
  int min_keyframe_its = 30;
  double inlier_error_threshold = 2.0;
  double lost_ratio = 0.7;
};

// BORDER_REFLECT_101 index, Appendix A.1
inline int Reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    else p = 2 * n - 2 - p;
  }
  return p;
}

// cv::pyrDown(src, dst, Size(src.cols/2, src.rows/2)) for 8UC1: separable [1 4 6 4 1], (sum+128)>>8.
inline void PyrDown(const Image &src, uint8_t *dst, int dst_step) {
  const int dw = src.cols / 2, dh = src.rows / 2;
  std::vector<int> rowbuf(static_cast<size_t>(5) * dw);
  for (int y = 0; y < dh; y++) {
    for (int k = 0; k < 5; k++) {
      const uint8_t *s = src.ptr(Reflect101(2 * y + k - 2, src.rows));
      int *r = &rowbuf[static_cast<size_t>(k) * dw];
      for (int x = 0; x < dw; x++) {
        const int x0 = Reflect101(2 * x - 2, src.cols), x1 = Reflect101(2 * x - 1, src.cols);
        const int x2 = Reflect101(2 * x, src.cols), x3 = Reflect101(2 * x + 1, src.cols);
        const int x4 = Reflect101(2 * x + 2, src.cols);
        r[x] = s[x0] + s[x4] + 4 * (s[x1] + s[x3]) + 6 * s[x2];
      }
    }
    for (int x = 0; x < dw; x++) {
      const int v = rowbuf[x] + rowbuf[4 * dw + x] + 4 * (rowbuf[dw + x] + rowbuf[3 * dw + x]) + 6 * rowbuf[2 * dw + x];
      dst[static_cast<size_t>(y) * dst_step + x] = static_cast<uint8_t>((v + 128) >> 8);
    }
  }
}

struct KeyPoint {
  float x, y, response;
};

// Bresenham circle of radius 3, Appendix A.2
static const int kFastCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                       {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// cornerScore<16>: largest threshold for which the pixel stays a FAST-9 corner (A.2)
inline int FastCornerScore(const uint8_t *ptr, const int pixel[25], int threshold) {
  const int N = 25;
  const int v = ptr[0];
  short d[N];
  for (int k = 0; k < N; k++) d[k] = static_cast<short>(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min<int>(d[k + 1], d[k + 2]);
    a = std::min<int>(a, d[k + 3]);
    if (a <= a0) continue;
    a = std::min<int>(a, d[k + 4]);
    a = std::min<int>(a, d[k + 5]);
    a = std::min<int>(a, d[k + 6]);
    a = std::min<int>(a, d[k + 7]);
    a = std::min<int>(a, d[k + 8]);
    a0 = std::max(a0, std::min<int>(a, d[k]));
    a0 = std::max(a0, std::min<int>(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max<int>(d[k + 1], d[k + 2]);
    b = std::max<int>(b, d[k + 3]);
    b = std::max<int>(b, d[k + 4]);
    b = std::max<int>(b, d[k + 5]);
    if (b >= b0) continue;
    b = std::max<int>(b, d[k + 6]);
    b = std::max<int>(b, d[k + 7]);
    b = std::max<int>(b, d[k + 8]);
    b0 = std::min(b0, std::max<int>(b, d[k]));
    b0 = std::min(b0, std::max<int>(b, d[k + 9]));
  }
  return -b0 - 1;
}

// cv::FAST(img, kps, threshold, nonmaxSuppression) TYPE_9_16 on an ROI view; keypoints in ROI coordinates,
// row-major order.  Three-row rolling score buffer exactly as OpenCV's FAST_t<16> (A.2).
inline void Fast9_16(const Image &img, std::vector<KeyPoint> *kps, int threshold, bool nonmax) {
  const int K = 8, N = 25;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = kFastCircle[k][0] + kFastCircle[k][1] * img.step;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  kps->clear();
  threshold = std::min(std::max(threshold, 0), 255);
  if (img.cols < 7 || img.rows < 7) return;

  std::vector<uint8_t> sbuf(static_cast<size_t>(3) * img.cols, 0);
  std::vector<int> cbuf(static_cast<size_t>(3) * (img.cols + 1), 0);
  uint8_t *buf[3] = {&sbuf[0], &sbuf[img.cols], &sbuf[2 * img.cols]};
  int *cpbuf[3] = {&cbuf[1], &cbuf[img.cols + 2], &cbuf[2 * img.cols + 3]};

  for (int i = 3; i < img.rows - 2; i++) {
    const uint8_t *ptr = img.ptr(i) + 3;
    uint8_t *curr = buf[(i - 3) % 3];
    int *cornerpos = cpbuf[(i - 3) % 3];
    std::fill(curr, curr + img.cols, 0);
    int ncorners = 0;
    if (i < img.rows - 3) {
      for (int j = 3; j < img.cols - 3; j++, ptr++) {
        const int v = ptr[0];
        // threshold_tab[x - v + 255]: 1 if x < v-t (darker), 2 if x > v+t (brighter)
        auto tab = [&](int x) -> int { const int df = x - v; return df < -threshold ? 1 : (df > threshold ? 2 : 0); };
        int d = tab(ptr[pixel[0]]) | tab(ptr[pixel[8]]);
        if (d == 0) continue;
        d &= tab(ptr[pixel[2]]) | tab(ptr[pixel[10]]);
        d &= tab(ptr[pixel[4]]) | tab(ptr[pixel[12]]);
        d &= tab(ptr[pixel[6]]) | tab(ptr[pixel[14]]);
        if (d == 0) continue;
        d &= tab(ptr[pixel[1]]) | tab(ptr[pixel[9]]);
        d &= tab(ptr[pixel[3]]) | tab(ptr[pixel[11]]);
        d &= tab(ptr[pixel[5]]) | tab(ptr[pixel[13]]);
        d &= tab(ptr[pixel[7]]) | tab(ptr[pixel[15]]);
        if (d & 1) {
          const int vt = v - threshold;
          int count = 0;
          for (int k = 0; k < N; k++) {
            const int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = static_cast<uint8_t>(FastCornerScore(ptr, pixel, threshold));
                break;
              }
            } else {
              count = 0;
            }
          }
        }
        if (d & 2) {
          const int vt = v + threshold;
          int count = 0;
          for (int k = 0; k < N; k++) {
            const int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = static_cast<uint8_t>(FastCornerScore(ptr, pixel, threshold));
                break;
              }
            } else {
              count = 0;
            }
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t *prev = buf[(i - 4 + 3) % 3];
    const uint8_t *pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      const int j = cornerpos[k];
      const int score = prev[j];
      if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                      score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        kps->push_back(KeyPoint{static_cast<float>(j), static_cast<float>(i - 1), static_cast<float>(score)});
      }
    }
  }
}

// cv::KeyPointsFilter::retainBest (A.3): libstdc++ nth_element + partition, all boundary ties kept.
inline void RetainBest(std::vector<KeyPoint> *kps, int n_points) {
  if (n_points >= 0 && kps->size() > static_cast<size_t>(n_points)) {
    if (n_points == 0) {
      kps->clear();
      return;
    }
    std::nth_element(kps->begin(), kps->begin() + n_points, kps->end(),
                     [](const KeyPoint &a, const KeyPoint &b) { return a.response > b.response; });
    const float ambiguous = (*kps)[n_points - 1].response;
    auto new_end = std::partition(kps->begin() + n_points, kps->end(),
                                  [ambiguous](const KeyPoint &k) { return k.response >= ambiguous; });
    kps->resize(new_end - kps->begin());
  }
}

struct Corner {
  int x, y, level;
};

inline int DetectMargin(const Params &p) { return p.use_orb ? 4 + p.orb_size / 2 : 1 + p.patch_size / 2; }

// Cell ROI of SelectPixels (fast_detector.cc:79-92): returns false when the margin swallows the cell.
inline bool CellRoi(int cols, int rows, int cell, int margin, int i, int j, int *x0, int *y0, int *x1, int *y1) {
  *y0 = std::max(margin, i * cell);
  *y1 = std::min(rows - margin, i * cell + cell);
  if (*y1 <= *y0) return false;
  *x0 = std::max(margin, j * cell);
  *x1 = std::min(cols - margin, j * cell + cell);
  if (*x1 <= *x0) return false;
  return true;
}

// Per-cell FAST lists in image coordinates (the part of SelectPixels the device kernel replaces),
// fast_detector.cc:79-106.  cell_kps[i*wcells+j]; ran[i*wcells+j] = FAST was invoked for the cell.
inline void FastCells(const Image &src, const Params &p, std::vector<std::vector<KeyPoint>> *cell_kps,
                      std::vector<uint8_t> *ran, int *wcells_out, int *hcells_out) {
  const int margin = DetectMargin(p);
  const int wcells = static_cast<int>(std::ceil(static_cast<double>(src.cols) / static_cast<double>(p.cell_size)));
  const int hcells = static_cast<int>(std::ceil(static_cast<double>(src.rows) / static_cast<double>(p.cell_size)));
  cell_kps->assign(static_cast<size_t>(wcells) * hcells, {});
  ran->assign(static_cast<size_t>(wcells) * hcells, 0);
  for (int i = 0; i < hcells; i++) {
    for (int j = 0; j < wcells; j++) {
      int x0, y0, x1, y1;
      if (!CellRoi(src.cols, src.rows, p.cell_size, margin, i, j, &x0, &y0, &x1, &y1)) continue;
      Image roi;
      roi.data = src.data + static_cast<size_t>(y0) * src.step + x0;
      roi.cols = x1 - x0;
      roi.rows = y1 - y0;
      roi.step = src.step;
      std::vector<KeyPoint> &kps = (*cell_kps)[static_cast<size_t>(i) * wcells + j];
      Fast9_16(roi, &kps, p.fast_threshold, true);
      for (auto &k : kps) {
        k.x += x0;
        k.y += y0;
      }
      (*ran)[static_cast<size_t>(i) * wcells + j] = 1;
    }
  }
  *wcells_out = wcells;
  *hcells_out = hcells;
}

// Quota + retainBest part of SelectPixels, fast_detector.cc:108-151 (also used verbatim in spirit by any
// host that post-processes device per-cell lists: order of cells row-major, lists in FAST scan order).
inline void SelectFromCells(std::vector<std::vector<KeyPoint>> *cell_kps, const std::vector<uint8_t> &ran, int wcells,
                            int hcells, int level, int nfeatures, std::vector<Corner> *pixels) {
  std::vector<int> nleft(static_cast<size_t>(wcells) * hcells, 0), nselected(static_cast<size_t>(wcells) * hcells, 0);
  int nempty = 0;
  for (int c = 0; c < wcells * hcells; c++) {
    if (!ran[c]) continue;  // skipped cells are NOT counted in nempty (Appendix B quirk)
    if (!(*cell_kps)[c].empty()) nleft[c] = static_cast<int>((*cell_kps)[c].size());
    else nempty++;
  }
  const int ncells = hcells * wcells;
  int selected = 0;
  int cells_left = ncells - nempty;
  while ((nfeatures - selected) > 0 && cells_left > 0) {
    const int npercell = static_cast<int>(std::ceil(static_cast<double>(nfeatures - selected) / static_cast<double>(cells_left)));
    cells_left = 0;
    for (int c = 0; c < ncells; c++) {
      if (nleft[c] > 0) {
        if (nleft[c] > npercell) {
          nselected[c] += npercell;
          selected += npercell;
          nleft[c] -= npercell;
          cells_left++;
        } else {
          nselected[c] += nleft[c];
          selected += nleft[c];
          nleft[c] = 0;
        }
      }
    }
  }
  std::vector<KeyPoint> fts;
  for (int c = 0; c < ncells; c++) {
    RetainBest(&(*cell_kps)[c], nselected[c]);
    for (const auto &k : (*cell_kps)[c]) fts.push_back(k);
  }
  if (static_cast<int>(fts.size()) > nfeatures) RetainBest(&fts, nfeatures);
  for (const auto &k : fts) pixels->push_back(Corner{static_cast<int>(k.x), static_cast<int>(k.y), level});
}

// FastDetector::SelectPixels, fast_detector.cc:58-152
inline void SelectPixels(const Image &src, const Params &p, int level, int nfeatures, std::vector<Corner> *pixels) {
  std::vector<std::vector<KeyPoint>> cell_kps;
  std::vector<uint8_t> ran;
  int wcells, hcells;
  FastCells(src, p, &cell_kps, &ran, &wcells, &hcells);
  SelectFromCells(&cell_kps, ran, wcells, hcells, level, nfeatures, pixels);
}

// FastDetector::DetectPyramid, fast_detector.cc:154-175
inline void LevelQuotas(const Params &p, int nfeatures, std::vector<int> *quota) {
  const double scale = 1.2;
  double factor = 1.0, val = 0.0;
  for (int i = 0; i < p.max_fast_levels; i++) {
    val += factor;
    factor /= scale;
  }
  int levelfeatures = static_cast<int>(nfeatures / val);
  for (int i = 0; i < p.max_fast_levels; i++) {
    quota->push_back(levelfeatures);
    levelfeatures = static_cast<int>(levelfeatures / scale);
  }
}

inline void DetectPyramid(const std::vector<Image> &pyramid, const Params &p, int nfeatures, std::vector<Corner> *corners) {
  std::vector<int> quota;
  LevelQuotas(p, nfeatures, &quota);
  for (int i = 0; i < p.max_fast_levels; i++) SelectPixels(pyramid[i], p, i, quota[i], corners);
}

// FindShiTomasiScoreAtPoint, extra/utils.cc:61-97.  Float sums of integer products (exact: every partial
// sum < 2^24); unqualified sqrt(float) frozen to the double overload (DESIGN.md "frozen interpretations").
inline double ShiTomasiScore(const Image &img, int px, int py) {
  float dXX = 0.0, dYY = 0.0, dXY = 0.0;
  const int halfbox_size = 4;
  const int box_size = 2 * halfbox_size;
  const int box_area = box_size * box_size;
  const int x_min = px - halfbox_size, x_max = px + halfbox_size;
  const int y_min = py - halfbox_size, y_max = py + halfbox_size;
  if (x_min < 1 || x_max >= img.cols - 1 || y_min < 1 || y_max >= img.rows - 1) return 0.0;
  const int stride = img.step;
  for (int y = y_min; y < y_max; y++) {
    const uint8_t *ptr_left = img.data + stride * y + x_min - 1;
    const uint8_t *ptr_right = img.data + stride * y + x_min + 1;
    const uint8_t *ptr_top = img.data + stride * (y - 1) + x_min;
    const uint8_t *ptr_bottom = img.data + stride * (y + 1) + x_min;
    for (int x = 0; x < box_size; x++, ptr_left++, ptr_right++, ptr_top++, ptr_bottom++) {
      const float dx = *ptr_right - *ptr_left;
      const float dy = *ptr_bottom - *ptr_top;
      dXX += dx * dx;
      dYY += dy * dy;
      dXY += dx * dy;
    }
  }
  dXX = dXX / (2.0 * box_area);
  dYY = dYY / (2.0 * box_area);
  dXY = dXY / (2.0 * box_area);
  const float disc = (dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY);
  return 0.5 * (dXX + dYY - std::sqrt(static_cast<double>(disc)));
}

// FastDetector with grid (ctor :33-40, LockCell :48-51, FilterCorners :177-218)
struct CornerGrid {
  int cell_size, grid_width, grid_height;
  std::vector<std::pair<int, int>> cgrid;  // (corner index, score truncated to int — fast_detector.h:58)
  std::vector<uint8_t> mask;

  CornerGrid(int width, int height, const Params &p) {
    cell_size = p.cell_size;
    grid_width = static_cast<int>(std::ceil(static_cast<double>(width) / cell_size));
    grid_height = static_cast<int>(std::ceil(static_cast<double>(height) / cell_size));
    cgrid.assign(static_cast<size_t>(grid_width) * grid_height, std::make_pair(0, p.min_feature_score));
    mask.assign(static_cast<size_t>(grid_width) * grid_height, 0);
  }
  void LockCell(double px, double py) {
    const int index = static_cast<int>(py / cell_size) * grid_width + static_cast<int>(px / cell_size);
    mask.at(index) = 1;
  }
  void FilterCorners(const std::vector<Image> &pyramid, const std::vector<Corner> &corners, const Params &p,
                     std::vector<int> *indices) {
    const int margin = DetectMargin(p);
    int index = 0;
    for (auto it = corners.begin(); it != corners.end(); it++, index++) {
      const int px = it->x, py = it->y, level = it->level;
      const int scale = (1 << level);
      if (px < margin || py < margin || px >= pyramid[level].cols - margin || py >= pyramid[level].rows - margin) continue;
      const int pos = static_cast<int>((py * scale) / cell_size) * grid_width + static_cast<int>((px * scale) / cell_size);
      if (mask[pos]) continue;
      const double score = ShiTomasiScore(pyramid[level], px, py);
      if (score > cgrid.at(pos).second) cgrid.at(pos) = std::make_pair(index, static_cast<int>(score));
    }
    for (auto it = cgrid.begin(); it != cgrid.end(); it++)
      if (it->second > p.min_feature_score) indices->push_back(it->first);
  }
};

}  // namespace sdvlref

#endif  // SDVL_ORACLE_REF_DETECT_H_
