// frontend.cc — the hot path's classes of the host layer declared in sdvl_host.h: Device (context + frame pool), Frame, Feature,
// FastDetector, ORBDetector, ImageAlign, Matcher, FeatureAlign.  NOTHING ELSE: Camera, Point, Map, SDVL and the batch driver live in
// standalone.cc — in the reference's tree its own camera.cc / point.cc / map.cc / sdvl.cc stand there (INTEGRATION.md route A; the
// symbols this file leaves undefined are listed in frontend_deps.txt and checked by tests/test_frontend_split.py).
// Reference line numbers are cited at each function; device work goes through the C-ABI (include/sdvl_hip.h) only.
#include "host_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <new>
#include <stdexcept>
#include <chrono>
#include <thread>

#include <sys/mman.h>

namespace sdvl {

using std::shared_ptr;
using std::vector;

// ------------------------------------------------------------------------------------------------------ Device
static thread_local Device *g_current_device = nullptr;

// Descriptors are lazy, as in the reference (matcher.cc:266-269, frame.cc:148-161): a new frame gets corners only; the
// search kernel computes the descriptors of the corners it compares and FilterCorners asks for a keyframe's full set.
static thread_local StageTimes *g_stage_times = nullptr;
StageTimes *&StageTimes::Active() { return g_stage_times; }

Device::Device(int gpu) : gpu_(gpu) {
  const int rc = sdvl_ctx_create(gpu, &ctx_);
  if (rc != SDVL_OK) throw std::runtime_error("sdvl_ctx_create failed (" + std::to_string(rc) + "): no MI355X visible; there is no CPU fallback");
  if (!g_current_device) g_current_device = this;
}

Device::~Device() {
  for (auto &p : pool_) sdvl_frame_destroy(ctx_, p.f);
  if (g_current_device == this) g_current_device = nullptr;
  sdvl_ctx_destroy(ctx_);
}

// ------------------------------------------------------------------------------------------------------ ChunkPool
ChunkPool::~ChunkPool() {
  for (auto &r : regions_) munmap(r.first, r.second);
}
void ChunkPool::Map(size_t bytes) {
  bytes = (bytes + kChunk - 1) / kChunk * kChunk;
  // Round 4: the region starts on a 2-MB boundary and asks for transparent huge pages (the boxes run THP in `madvise` mode): the
  // arenas of the keyframes are memory that is written once and kept, i.e. first-touch faults — one per 2 MB instead of one per 4 KB.
  constexpr bool huge = true;
  constexpr size_t kHuge = 2u << 20;
  const size_t span = huge ? bytes + kHuge : bytes;
  void *raw = mmap(nullptr, span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (raw == MAP_FAILED) throw std::bad_alloc();
  regions_.push_back({raw, span});
  void *p = raw;
  if (huge) {
    p = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(raw) + kHuge - 1) / kHuge * kHuge);
    (void)madvise(p, bytes / kHuge * kHuge, MADV_HUGEPAGE);  // advice only: plain pages if the kernel declines
  }
  char *c = static_cast<char *>(p);
  for (size_t off = bytes; off >= kChunk; off -= kChunk) free_.push_back(c + off - kChunk);  // lowest address handed out first
}
char *ChunkPool::Get() {
  std::lock_guard<std::mutex> lk(m_);
  if (free_.empty()) Map(8u << 20);
  char *c = free_.back();
  free_.pop_back();
  return c;
}
void ChunkPool::Put(char *c) {
  std::lock_guard<std::mutex> lk(m_);
  free_.push_back(c);
}

Device *Device::CurrentOrNull() { return g_current_device; }

Device *Device::Current() {
  if (!g_current_device) throw std::runtime_error("no sdvl::Device bound to this thread (construct one, or Device::SetCurrent)");
  return g_current_device;
}
// binds the calling thread to the device object AND to its GPU: HIP's current device is per thread and starts at 0, so a
// farm worker, fiber or pool helper that steps a group of GPU n must select n before anything it does allocates or launches
void Device::SetCurrent(Device *d) {
  g_current_device = d;
  if (d) d->Check(sdvl_ctx_bind_thread(d->ctx()), "sdvl_ctx_bind_thread");
}

void Device::Check(int rc, const char *what) const {
  if (rc != SDVL_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + sdvl_last_error(ctx_));
}

// corners_ capacity of the frames this device creates: num_features plus the ties retainBest keeps (fast_detector.cc:147-148) fit
// twice num_features with room to spare; a keyframe keeps its frame for good, so the list is not sized for the largest configuration
int Device::CornerCap() const {
  const int want = (2 * Config::NumFeatures() + 63) / 64 * 64;
  return std::min(SDVL_MAX_CORNERS, std::max(1024, want));
}

sdvl_frame *Device::AcquireFrame(int w, int h, int levels) {
  std::lock_guard<std::mutex> lk(pool_mutex_);
  const int cap = CornerCap();
  for (size_t i = 0; i < pool_.size(); i++) {
    if (pool_[i].w == w && pool_[i].h == h && pool_[i].levels == levels && pool_[i].cap == cap) {
      sdvl_frame *f = pool_[i].f;
      pool_[i] = pool_.back();
      pool_.pop_back();
      return f;
    }
  }
  // pool empty: take a slab of frames at once (a farm of trackers keeps turning frames into keyframes).  hipMalloc costs
  // milliseconds, so the slabs grow geometrically; callers that know their keyframe budget call Reserve() up front.
  const int chunk = std::min(512, std::max(32, total_frames_ / 2));
  std::vector<sdvl_frame *> fresh(chunk);
  Check(sdvl_ctx_set_corner_capacity(ctx_, cap), "sdvl_ctx_set_corner_capacity");
  Check(sdvl_frame_create_many(ctx_, w, h, levels, chunk, fresh.data()), "sdvl_frame_create_many");
  total_frames_ += chunk;
  for (int i = 1; i < chunk; i++) pool_.push_back(Pooled{fresh[i], w, h, levels, cap});
  return fresh[0];
}

void Device::Reserve(int w, int h, int levels, int frames) {
  std::lock_guard<std::mutex> lk(pool_mutex_);
  const int cap = CornerCap();
  int have = 0;
  for (const Pooled &p : pool_) have += (p.w == w && p.h == h && p.levels == levels && p.cap == cap) ? 1 : 0;
  Check(sdvl_ctx_set_corner_capacity(ctx_, cap), "sdvl_ctx_set_corner_capacity");
  while (have < frames) {
    const int chunk = std::min(512, frames - have);
    std::vector<sdvl_frame *> fresh(chunk);
    Check(sdvl_frame_create_many(ctx_, w, h, levels, chunk, fresh.data()), "sdvl_frame_create_many");
    total_frames_ += chunk;
    for (int i = 0; i < chunk; i++) pool_.push_back(Pooled{fresh[i], w, h, levels, cap});
    have += chunk;
  }
}

// frames die wherever their last shared_ptr is dropped, including the host worker threads
void Device::ReleaseFrame(sdvl_frame *f, int w, int h, int levels) {
  std::lock_guard<std::mutex> lk(pool_mutex_);
  pool_.push_back(Pooled{f, w, h, levels, sdvl_frame_corner_capacity(f)});
}

// ---------------------------------------------------------------------------------------------------- RandStream
RandStream::RandStream(unsigned seed) {
  if (seed == 0) seed = 1;
  r_[0] = static_cast<int>(seed);
  for (int i = 1; i < 31; i++) {
    const long hi = r_[i - 1] / 127773, lo = r_[i - 1] % 127773;
    long word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    r_[i] = static_cast<int>(word);
  }
  fi_ = 3;
  ri_ = 0;
  for (int i = 0; i < 310; i++) Next();
}

int RandStream::Next() {
  const unsigned val = static_cast<unsigned>(r_[fi_]) + static_cast<unsigned>(r_[ri_]);
  r_[fi_] = static_cast<int>(val);
  if (++fi_ >= 31) fi_ = 0;
  if (++ri_ >= 31) ri_ = 0;
  return static_cast<int>(val >> 1);
}

// ------------------------------------------------------------------------------------------------------- detectors
static int DetectMargin() { return Config::UseORB() ? 4 + Config::ORBSize() / 2 : 1 + Config::PatchSize() / 2; }

static sdvl_detect_params DetectParams() {
  sdvl_detect_params dp;
  dp.cell_size = Config::CellSize();
  dp.max_fast_levels = Config::MaxFastLevels();
  dp.fast_threshold = Config::FastThreshold();
  dp.margin = DetectMargin();
  return dp;
}

bool ORBDetector::GetDescriptor(const Image &src, const Vector2i &pos, std::vector<uchar> *desc) {
  if (!src.dev) {
    std::cerr << "[ERROR] ORBDetector::GetDescriptor needs a pyramid level of a Frame (HBM-resident)" << std::endl;
    return false;
  }
  Device *dev = Device::Current();
  const int32_t xyl[3] = {pos(0), pos(1), src.level};
  desc->resize(32);
  dev->Check(sdvl_orb_describe_points(dev->ctx(), src.dev, 1, xyl, desc->data(), nullptr), "sdvl_orb_describe_points");
  return true;
}

// orb_detector.cc:398-410
int ORBDetector::Distance(const std::vector<uchar> &a, const std::vector<uchar> &b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t va, vb;
    std::memcpy(&va, &a[4 * i], 4);
    std::memcpy(&vb, &b[4 * i], 4);
    uint32_t v = va ^ vb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// fast_detector.cc:33-51
FastDetector::FastDetector(int width, int height, bool grid) {
  cell_size_ = Config::CellSize();
  grid_width_ = static_cast<int>(std::ceil(static_cast<double>(width) / cell_size_));
  grid_height_ = static_cast<int>(std::ceil(static_cast<double>(height) / cell_size_));
  if (grid) {
    cgrid_.resize(static_cast<size_t>(grid_width_) * grid_height_, std::make_pair(0, Config::MinFeatureScore()));
    grid_mask_.resize(static_cast<size_t>(grid_width_) * grid_height_, false);
  }
}
void FastDetector::LockCell(Vector2d p) {
  const int index = static_cast<int>(p(1) / cell_size_) * grid_width_ + static_cast<int>(p(0) / cell_size_);
  grid_mask_.at(index) = true;
}
void FastDetector::UnlockCell(Vector2d p) {
  const int index = static_cast<int>(p(1) / cell_size_) * grid_width_ + static_cast<int>(p(0) / cell_size_);
  grid_mask_.at(index) = false;
}

namespace {
struct KP { float x, y, response; };

// cv::KeyPointsFilter::retainBest: same libstdc++ calls on identically ordered input
void RetainBest(vector<KP> *kps, int n_points) {
  if (n_points >= 0 && kps->size() > static_cast<size_t>(n_points)) {
    if (n_points == 0) { kps->clear(); return; }
    std::nth_element(kps->begin(), kps->begin() + n_points, kps->end(), [](const KP &a, const KP &b) { return a.response > b.response; });
    const float ambiguous = (*kps)[n_points - 1].response;
    auto new_end = std::partition(kps->begin() + n_points, kps->end(), [ambiguous](const KP &k) { return k.response >= ambiguous; });
    kps->resize(new_end - kps->begin());
  }
}
}  // namespace

// fast_detector.cc:108-151 over the per-cell lists the K2 kernel produced (fast_detector.cc:79-106)
void FastDetector::SelectFromCells(const sdvl_keypoint *kps, const int32_t *cell_offsets, int level_cell_begin, int wcells, int hcells,
                                   int level, int level_w, int level_h, int nfeatures, vector<Vector3i> *pixels) {
  const int cell = Config::CellSize(), margin = DetectMargin();
  const int ncells = wcells * hcells;
  vector<vector<KP>> cell_fts(ncells);
  vector<int> nleft(ncells, 0), nselected(ncells, 0);
  int nempty = 0;
  for (int i = 0; i < hcells; i++) {
    const int inity = std::max(margin, i * cell), maxy = std::min(level_h - margin, i * cell + cell);
    if (maxy <= inity) continue;
    for (int j = 0; j < wcells; j++) {
      const int initx = std::max(margin, j * cell), maxx = std::min(level_w - margin, j * cell + cell);
      if (maxx <= initx) continue;
      const int c = i * wcells + j;
      const int b = cell_offsets[level_cell_begin + c], e = cell_offsets[level_cell_begin + c + 1];
      vector<KP> &v = cell_fts[c];
      v.reserve(e - b);
      for (int k = b; k < e; k++) v.push_back(KP{static_cast<float>(kps[k].x), static_cast<float>(kps[k].y), static_cast<float>(kps[k].score)});
      if (!v.empty()) nleft[c] = static_cast<int>(v.size());
      else nempty++;
    }
  }
  int selected = 0;
  int cells_left = ncells - nempty;
  while ((nfeatures - selected) > 0 && cells_left > 0) {
    const int npercell = static_cast<int>(std::ceil(static_cast<double>(nfeatures - selected) / static_cast<double>(cells_left)));
    cells_left = 0;
    for (int c = 0; c < ncells; c++) {
      if (nleft[c] > 0) {
        if (nleft[c] > npercell) {
          nselected[c] += npercell; selected += npercell; nleft[c] -= npercell; cells_left++;
        } else {
          nselected[c] += nleft[c]; selected += nleft[c]; nleft[c] = 0;
        }
      }
    }
  }
  vector<KP> fts;
  for (int c = 0; c < ncells; c++) {
    RetainBest(&cell_fts[c], nselected[c]);
    for (const KP &k : cell_fts[c]) fts.push_back(k);
  }
  if (static_cast<int>(fts.size()) > nfeatures) RetainBest(&fts, nfeatures);
  for (const KP &k : fts) pixels->push_back(Vector3i(static_cast<int>(k.x), static_cast<int>(k.y), level));
}

// fast_detector.cc:154-175.  `pyramid` must be a Frame's pyramid (HBM binding).
void FastDetector::DetectPyramid(const vector<Image> &pyramid, vector<Vector3i> *corners, int nfeatures) {
  if (pyramid.empty() || !pyramid[0].dev) {
    std::cerr << "[ERROR] FastDetector::DetectPyramid needs the pyramid of a Frame (HBM-resident)" << std::endl;
    return;
  }
  Device *dev = Device::Current();
  const sdvl_detect_params dp = DetectParams();
  int cpl[4] = {0, 0, 0, 0}, total = 0;
  dev->Check(sdvl_fast_num_cells(pyramid[0].cols, pyramid[0].rows, &dp, cpl, &total), "sdvl_fast_num_cells");
  const int cap = 32768;
  vector<sdvl_keypoint> kps(cap);
  vector<int32_t> offs(total + 1);
  sdvl_frame *fr = pyramid[0].dev;
  dev->Check(sdvl_fast_cells(dev->ctx(), 1, &fr, &dp, cap, kps.data(), offs.data()), "sdvl_fast_cells");
  const double scale = 1.2;
  double factor = 1.0, val = 0.0;
  for (int i = 0; i < Config::MaxFastLevels(); i++) { val += factor; factor /= scale; }
  int levelfeatures = static_cast<int>(nfeatures / val);
  int begin = 0;
  for (int i = 0; i < Config::MaxFastLevels(); i++) {
    const int wc = (pyramid[i].cols + dp.cell_size - 1) / dp.cell_size, hc = (pyramid[i].rows + dp.cell_size - 1) / dp.cell_size;
    SelectFromCells(kps.data(), offs.data(), begin, wc, hc, i, pyramid[i].cols, pyramid[i].rows, levelfeatures, corners);
    begin += cpl[i];
    levelfeatures = static_cast<int>(levelfeatures / scale);
  }
}

// fast_detector.cc:177-218 with the Shi-Tomasi scores supplied by the K3 kernel
void FastDetector::FilterWithScores(const vector<Image> &pyramid, const vector<Vector3i> &corners, const double *scores,
                                    vector<int> *indices) {
  const int margin = DetectMargin();
  int index = 0;
  for (auto it = corners.begin(); it != corners.end(); it++, index++) {
    const int px = (*it)(0), py = (*it)(1), level = (*it)(2);
    const int scale = (1 << level);
    if (px < margin || py < margin || px >= pyramid[level].cols - margin || py >= pyramid[level].rows - margin) continue;
    const int pos = static_cast<int>((py * scale) / cell_size_) * grid_width_ + static_cast<int>((px * scale) / cell_size_);
    if (grid_mask_[pos]) continue;
    const double score = scores[index];
    if (score > cgrid_.at(pos).second) cgrid_.at(pos) = std::make_pair(index, static_cast<int>(score));
  }
  for (auto it = cgrid_.begin(); it != cgrid_.end(); it++)
    if ((*it).second > Config::MinFeatureScore()) indices->push_back((*it).first);
}

void FastDetector::FilterCorners(const vector<Image> &pyramid, const vector<Vector3i> &corners, vector<int> *indices) {
  if (pyramid.empty() || !pyramid[0].dev) {
    std::cerr << "[ERROR] FastDetector::FilterCorners needs the pyramid of a Frame (HBM-resident)" << std::endl;
    return;
  }
  Device *dev = Device::Current();
  sdvl_frame *fr = pyramid[0].dev;
  int32_t dev_count = 0;
  dev->Check(sdvl_frames_corner_counts(dev->ctx(), 1, &fr, &dev_count), "sdvl_frames_corner_counts");
  if (dev_count != static_cast<int>(corners.size())) {
    std::cerr << "[ERROR] FastDetector::FilterCorners: corners differ from the frame's HBM corner list" << std::endl;
    return;
  }
  vector<double> scores(std::max<size_t>(corners.size(), 1));
  dev->Check(sdvl_shi_tomasi(dev->ctx(), 1, &fr, static_cast<int>(scores.size()), scores.data()), "sdvl_shi_tomasi");
  FilterWithScores(pyramid, corners, scores.data(), indices);
}

// ------------------------------------------------------------------------------------------------ Feature / Point
// feature.cc:28-56
// (the reference sizes descriptor_ to 32 bytes in every ctor; here the storage appears with SetDescriptor — features the
//  tracker creates never get a descriptor, feature_align.cc:132, and there are ~200 of them per frame)
Feature::Feature(const shared_ptr<Frame> &f, const Vector2d &p, int l) : frame_(f), frame_raw_(f.get()), point_(nullptr), p2d_(p), level_(l) {
  v_ = f->GetCamera()->Unproject(p2d_);
  has_descriptor_ = false;
}
Feature::Feature(std::weak_ptr<Frame> &&f, Frame *raw, const Vector2d &p, int l)
    : frame_(std::move(f)), frame_raw_(raw), point_(nullptr), p2d_(p), level_(l) {
  v_ = raw->GetCamera()->Unproject(p2d_);
  has_descriptor_ = false;
}
Feature::Feature(const shared_ptr<Frame> &f, const shared_ptr<Point> &ft, const Vector2d &p, int l)
    : frame_(f), frame_raw_(f.get()), point_(ft), p2d_(p), level_(l) {
  v_ = f->GetCamera()->Unproject(p2d_);
  has_descriptor_ = false;
}
Feature::Feature(const shared_ptr<Frame> &f, const shared_ptr<Point> &ft, const Vector2d &p, const Vector3d &v, int l)
    : frame_(f), frame_raw_(f.get()), point_(ft), p2d_(p), v_(v), level_(l) {
  has_descriptor_ = false;
}

// ---------------------------------------------------------------------------------------------------------- Frame
std::atomic<int> Frame::counter_{0};

void Frame::InitCommon(Camera *camera, ORBDetector *detector, int w, int h) {
  id_ = counter_.fetch_add(1, std::memory_order_relaxed);
  camera_ = camera;
  orb_detector_ = detector;
  pyramid_levels_ = Config::PyramidLevels();
  pose_ = SE3();
  width_ = w;
  height_ = h;
  is_keyframe_ = false;
  owner_ = Device::Current();
  dev_ = owner_->AcquireFrame(w, h, pyramid_levels_);
  pyramid_.resize(pyramid_levels_);
  int lw = w, lh = h;
  for (int l = 0; l < pyramid_levels_; l++) {
    Image im;
    im.cols = lw; im.rows = lh; im.step = lw; im.dev = dev_; im.level = l;
    pyramid_[l] = im;
    lw /= 2;
    lh /= 2;
  }
}

namespace {
// pyramid + FAST + selection + ORB for a set of already-initialised frames (frame.cc:34-56 for each)
void BuildFrames(const vector<Frame *> &frames, const vector<sdvl_frame *> &devs, const vector<Image> &imgs, bool corners, int nfeatures) {
  Device *dev = Device::Current();
  const int n = static_cast<int>(frames.size());
  if (n == 0) return;
  std::unique_ptr<StageClock> clk(new StageClock(ST_UPLOAD_PYR));
  // host images of one shape and row pitch go up together (one gather kernel over pinned memory instead of a copy per image)
  vector<sdvl_frame *> up_f;
  vector<const uint8_t *> up_i;
  int up_step = 0;
  auto flush = [&]() {
    if (up_f.empty()) return;
    dev->Check(sdvl_frames_upload(dev->ctx(), static_cast<int>(up_f.size()), up_f.data(), up_i.data(), up_step), "sdvl_frames_upload");
    up_f.clear();
    up_i.clear();
  };
  // RAW camera images (Image::raw) of a camera with a lens: level 0 = Camera::UndistortImage(image), one remap launch per source kind
  vector<sdvl_frame *> un_f[2];
  vector<const void *> un_i[2];
  int un_step[2] = {0, 0};
  auto flush_raw = [&](int kind) {
    if (un_f[kind].empty()) return;
    const Camera *cam = frames[0]->GetCamera();
    const sdvl_camera c = cam->abi();
    const sdvl_distortion d = cam->distortion();
    dev->Check(sdvl_frames_upload_undistorted(dev->ctx(), static_cast<int>(un_f[kind].size()), un_f[kind].data(), un_i[kind].data(), un_step[kind], kind, &c, &d),
               "sdvl_frames_upload_undistorted");
    un_f[kind].clear();
    un_i[kind].clear();
  };
  for (int i = 0; i < n; i++) {
    frames[i]->SetImageTransient(false);
    if (imgs[i].raw && frames[i]->GetCamera()->HasDistortion()) {
      const int kind = imgs[i].dev_src ? 1 : 0;
      if (!un_f[kind].empty() && (imgs[i].step != un_step[kind] || frames[i]->GetCamera() != frames[0]->GetCamera())) flush_raw(kind);
      un_step[kind] = imgs[i].step;
      un_f[kind].push_back(devs[i]);
      un_i[kind].push_back(imgs[i].dev_src ? imgs[i].dev_src : static_cast<const void *>(imgs[i].data));
      continue;
    }
    if (imgs[i].dev_src && imgs[i].step == imgs[i].cols && imgs[i].borrow) {
      dev->Check(sdvl_frame_borrow_image_device(dev->ctx(), devs[i], imgs[i].dev_src), "sdvl_frame_borrow_image_device");
      frames[i]->SetImageTransient(imgs[i].transient);
    } else {  // host images, and images in HBM that the frame must own a copy of (an input ring the caller overwrites)
      if (!up_f.empty() && (imgs[i].step != up_step || frames[i]->GetWidth() != frames[0]->GetWidth() || frames[i]->GetHeight() != frames[0]->GetHeight()))
        flush();
      up_step = imgs[i].step;
      up_f.push_back(devs[i]);
      up_i.push_back(imgs[i].dev_src ? static_cast<const uint8_t *>(imgs[i].dev_src) : imgs[i].data);
    }
  }
  flush();
  flush_raw(0);
  flush_raw(1);
  dev->Check(sdvl_pyramid_build(dev->ctx(), n, devs.data()), "sdvl_pyramid_build");
  if (!corners) return;
  // FastDetector::DetectPyramid on the device (FAST + quota + retainBest in libstdc++ order); nothing returns to the host
  const sdvl_detect_params dp = DetectParams();
  clk.reset(new StageClock(ST_FAST));
  dev->Check(sdvl_detect_corners(dev->ctx(), n, devs.data(), &dp, nfeatures), "sdvl_detect_corners");
  clk.reset(new StageClock(ST_CORNERS_ORB));
}
}  // namespace

// frame.cc:34-56
Frame::Frame(Camera *camera, ORBDetector *detector, const Image &img, bool corners) {
  InitCommon(camera, detector, img.cols, img.rows);
  BuildFrames({this}, {dev_}, {img}, corners, Config::NumFeatures());
  corners_on_host_ = !corners;
}

void Frame::CreateBatch(Camera *camera, ORBDetector *detector, const vector<Image> &imgs, bool corners, int nfeatures,
                        vector<shared_ptr<Frame>> *out, const std::function<void(int, std::function<void(int)>)> *pfor) {
  const int n = static_cast<int>(imgs.size());
  vector<Frame *> raw(n);
  vector<sdvl_frame *> devs(n);
  out->clear();
  std::unique_ptr<StageClock> clk(new StageClock(ST_PRELUDE));
  for (int i = 0; i < n; i++) {
    shared_ptr<Frame> f(new Frame());
    f->InitCommon(camera, detector, imgs[i].cols, imgs[i].rows);
    raw[i] = f.get();
    devs[i] = f->dev_;
    f->corners_on_host_ = !corners;
    out->push_back(f);
  }
  (void)pfor;
  clk.reset();
  BuildFrames(raw, devs, imgs, corners, nfeatures);
}

void Frame::OwnImages(const vector<shared_ptr<Frame>> &frames) {
  vector<sdvl_frame *> devs;
  Device *dev = nullptr;
  for (const shared_ptr<Frame> &f : frames)
    if (f && f->image_transient_ && f->dev_) {
      devs.push_back(f->dev_);
      f->image_transient_ = false;
      dev = f->owner_;
    }
  if (!devs.empty()) dev->Check(sdvl_frames_own_images(dev->ctx(), static_cast<int>(devs.size()), devs.data()), "sdvl_frames_own_images");
}

FrameArena *Frame::NewArena() const { return new FrameArena(owner_ ? owner_->chunks : nullptr); }

Frame::~Frame() {
  features_.clear();
  DropFlat();
  if (arena_) arena_->Release();  // the arena goes with the last object carved out of it
  if (dev_ && owner_) {
    sdvl_frame *f = dev_;  // hand the HBM frame back to the pool
    dev_ = nullptr;
    owner_->ReleaseFrame(f, width_, height_, pyramid_levels_);
  }
}

// slot of this frame (with its current pose) in the open sdvl_search_begin batch `batch_id` of `ctx`
int Frame::SearchSlot(sdvl_ctx *ctx, uint64_t batch_id) {
  if (search_batch_ != batch_id) {
    double pose[7];
    pose_.ToArray(pose);
    search_slot_ = sdvl_search_slot(ctx, dev_, pose);
    if (search_slot_ < 0) throw std::runtime_error(std::string("sdvl_search_slot failed: ") + sdvl_last_error(ctx));
    search_batch_ = batch_id;
  }
  return search_slot_;
}

// Frame::CreateCorners for many frames at once, queued without waiting (the corner lists stay in HBM)
void Frame::DetectBatch(const vector<shared_ptr<Frame>> &frames, int nfeatures) {
  const int n = static_cast<int>(frames.size());
  if (n == 0) return;
  Device *dev = Device::Current();
  vector<sdvl_frame *> devs(n);
  for (int i = 0; i < n; i++) {
    Frame &f = *frames[i];
    devs[i] = f.dev_;
    f.corners_.clear();
    f.descriptors_.clear();
    f.filt_.clear();
    f.descriptors_on_host_ = false;
    f.corners_on_host_ = false;
  }
  const sdvl_detect_params dp = DetectParams();
  {
    StageClock clk(ST_FAST);
    dev->Check(sdvl_detect_corners(dev->ctx(), n, devs.data(), &dp, nfeatures), "sdvl_detect_corners");
  }
  StageClock clk(ST_CORNERS_ORB);
}

// frame.cc:122-131
void Frame::CreateCorners(int, int nfeatures) {
  Device *dev = Device::Current();
  const sdvl_detect_params dp = DetectParams();
  corners_.clear();
  descriptors_.clear();
  filt_.clear();
  descriptors_on_host_ = false;
  dev->Check(sdvl_detect_corners(dev->ctx(), 1, &dev_, &dp, nfeatures), "sdvl_detect_corners");
  corners_on_host_ = false;
}

vector<Vector3i> &Frame::GetCorners() {
  if (!corners_on_host_) {
    Device *dev = Device::Current();
    vector<int32_t> xyl(static_cast<size_t>(SDVL_MAX_CORNERS) * 3);
    int n = 0;
    dev->Check(sdvl_frame_download_corners(dev->ctx(), dev_, SDVL_MAX_CORNERS, xyl.data(), &n), "sdvl_frame_download_corners");
    corners_.resize(n);
    for (int i = 0; i < n; i++) corners_[i] = Vector3i(xyl[3 * i], xyl[3 * i + 1], xyl[3 * i + 2]);
    corners_on_host_ = true;
  }
  return corners_;
}

int Frame::GetNumCorners() {
  if (corners_on_host_) return static_cast<int>(corners_.size());
  Device *dev = Device::Current();
  int32_t n = 0;
  dev->Check(sdvl_frames_corner_counts(dev->ctx(), 1, &dev_, &n), "sdvl_frames_corner_counts");
  return n;
}

vector<Image> &Frame::GetPyramid() {
  if (!pyramid_on_host_) {
    Device *dev = Device::Current();
    for (int l = 0; l < pyramid_levels_; l++) {
      Image &im = pyramid_[l];
      im.owner = std::make_shared<vector<uint8_t>>(static_cast<size_t>(im.cols) * im.rows);
      dev->Check(sdvl_frame_download_level(dev->ctx(), dev_, l, im.owner->data(), im.cols), "sdvl_frame_download_level");
      im.data = im.owner->data();
    }
    pyramid_on_host_ = true;
  }
  return pyramid_;
}

vector<vector<uchar>> &Frame::GetDescriptors() {
  if (!descriptors_on_host_ && Config::UseORB() && !GetCorners().empty()) {
    Device *dev = Device::Current();
    vector<uint8_t> buf(corners_.size() * 32);
    dev->Check(sdvl_frame_download_descriptors(dev->ctx(), dev_, static_cast<int>(corners_.size()), buf.data()), "sdvl_frame_download_descriptors");
    descriptors_.resize(corners_.size());
    for (size_t i = 0; i < corners_.size(); i++) descriptors_[i].assign(buf.begin() + 32 * i, buf.begin() + 32 * (i + 1));
    descriptors_on_host_ = true;  // every entry mirrored (FilterCorners alone mirrors only the filtered ones)
  }
  return descriptors_;
}

// frame.cc:133-163
void Frame::FilterCorners() { FilterCornersBatch({shared_from_this()}); }

void Frame::FilterCornersBatch(const vector<shared_ptr<Frame>> &frames) {
  FilterCornersBegin(frames);
  FilterCornersEnd(frames);
}

// first half: the device side of FilterCorners for all frames — Shi-Tomasi scores, the per-cell selection of
// FastDetector::FilterCorners (fast_detector.cc:177-218) with the cells of the frame's features locked (frame.cc:139-142),
// ORB descriptors of the corners that survive — is queued; the caller may do host work that does not touch the device
// before FilterCornersEnd
void Frame::FilterCornersBegin(const vector<shared_ptr<Frame>> &frames) {
  const int n = static_cast<int>(frames.size());
  if (n == 0) return;
  Device *dev = Device::Current();
  const int cell = Config::CellSize();
  const int gw = static_cast<int>(std::ceil(static_cast<double>(frames[0]->width_) / cell));
  const int gh = static_cast<int>(std::ceil(static_cast<double>(frames[0]->height_) / cell));
  const int words = (gw * gh + 31) / 32;
  vector<sdvl_frame *> devs(n);
  vector<uint32_t> locked(static_cast<size_t>(n) * words, 0u);
  for (int i = 0; i < n; i++) {
    Frame &f = *frames[i];
    devs[i] = f.dev_;
    f.ForEachFeaturePosition([&](double px, double py) {  // FastDetector::LockCell, fast_detector.cc:48-51
      const int index = static_cast<int>(py / cell) * gw + static_cast<int>(px / cell);
      if (index >= 0 && index < gw * gh) locked[static_cast<size_t>(i) * words + (index >> 5)] |= 1u << (index & 31);
    });
  }
  dev->Check(sdvl_filter_corners_begin(dev->ctx(), n, devs.data(), locked.data(), words, cell, DetectMargin(), Config::MinFeatureScore(),
                                       Config::UseORB() ? 1 : 0), "sdvl_filter_corners_begin");
}

void Frame::FilterCornersEnd(const vector<shared_ptr<Frame>> &frames, const std::function<void(int, const std::function<void(int)> &)> *pfor) {
  const int n = static_cast<int>(frames.size());
  if (n == 0) return;
  Device *dev = Device::Current();
  const int cell = Config::CellSize();
  const int gw = static_cast<int>(std::ceil(static_cast<double>(frames[0]->width_) / cell));
  const int gh = static_cast<int>(std::ceil(static_cast<double>(frames[0]->height_) / cell));
  const int cap = gw * gh;
  vector<int32_t> counts(n);
  vector<sdvl_filtered_corner> &recs = dev->scratch_filtered;  // one row of `cap` records per frame
  if (recs.size() < static_cast<size_t>(n) * cap) recs.resize(static_cast<size_t>(n) * cap);
  dev->Check(sdvl_filter_corners_end(dev->ctx(), n, cap, counts.data(), recs.data()), "sdvl_filter_corners_end");
  (void)pfor;  // a frame's share is a copy of <= one record per grid cell: not worth spreading
  for (int i = 0; i < n; i++) {
    Frame &f = *frames[i];
    f.filt_.assign(recs.begin() + static_cast<size_t>(i) * cap, recs.begin() + static_cast<size_t>(i) * cap + counts[i]);
    f.filtered_corners_.resize(counts[i]);
    for (int k = 0; k < counts[i]; k++) f.filtered_corners_[k] = f.filt_[k].index;
  }
}

// frame.cc:165-179
int Frame::GetNumPoints() const {
  int count = 0;
  if (flat_)  // not materialised yet: a record with a point index is a feature with a point
    for (const sdvl_track_feature_out &f : flat_->feats) count += f.point >= 0 ? 1 : 0;
  for (auto it = features_.begin(); it != features_.end(); it++) {
    if (!(*it)) continue;
    if (!(*it)->GetPointRaw()) continue;
    count++;
  }
  return count;
}

// frame.cc:94-103
bool Frame::Project(const Vector3d &p3D, Vector2d *p2D) {
  const Vector3d rel_p = GetRelativePos(p3D);
  if (rel_p(2) < 0.0) return false;
  camera_->Project(rel_p, p2D);
  return true;
}

// features of a frame the device-resident tables tracked: flat records now, Feature objects on first use
void Frame::SetFlatFeatures(const sdvl_track_feature_out *feats, int n, const shared_ptr<PointTable> &points) {
  features_.clear();
  DropFlat();
  if (!arena_) arena_ = NewArena();
  sdvl_track_feature_out *dst = nullptr;
  if (n > 0) {
    dst = static_cast<sdvl_track_feature_out *>(arena_->Allocate(sizeof(sdvl_track_feature_out) * static_cast<size_t>(n), alignof(sdvl_track_feature_out)));
    std::memcpy(dst, feats, sizeof(sdvl_track_feature_out) * static_cast<size_t>(n));
  }
  flat_store_.feats.data = dst;
  flat_store_.feats.n = n;
  flat_store_.points = points;
  flat_ = &flat_store_;
}

void Frame::DropFlat() {
  if (!flat_) return;
  if (flat_store_.feats.data) arena_->Release();  // the block counted as one object of the arena
  flat_store_.feats.data = nullptr;
  flat_store_.feats.n = 0;
  flat_store_.points.reset();
  flat_ = nullptr;
}

// what SelectPoints + RemoveOutliers leave on the frame (feature_align.cc:127-134,245-256): one Feature per match in match
// order, linked to its point unless the pose stage called it an outlier (those positions also go to the outlier list)
void Frame::MaterializeFeatures() {
  const FlatSpan feats = flat_store_.feats;
  const shared_ptr<PointTable> points = flat_store_.points;
  flat_ = nullptr;  // GetFeatures() below must not come back here
  vector<shared_ptr<Feature>> behind;  // features added while the tracked ones were flat (a keyframe's seeds) stay behind them
  behind.swap(features_);
  features_.reserve(behind.size() + feats.size());
  for (const sdvl_track_feature_out &f : feats) {
    shared_ptr<Feature> feature = NewFeature(Vector2d(f.px[0], f.px[1]), f.level);
    if (f.point >= 0) {
      const shared_ptr<Point> &pt = (*points)[f.point];
      if (!link_on_materialize_) {
        feature->SetPoint(pt);
      } else if (!pt->ToDelete()) {  // (a point deleted since then has lost its features, Map::EmptyTrash, map.cc:207-259)
        feature->SetPoint(pt);
        pt->AddFeature(feature);
      }
    } else {
      outliers_.push_back(feature->GetPosition());
    }
    features_.push_back(std::move(feature));
  }
  for (shared_ptr<Feature> &f : behind) features_.push_back(std::move(f));
  link_on_materialize_ = false;
  flat_ = &flat_store_;
  DropFlat();
}

// ----------------------------------------------------------------------------------------------------- ImageAlign
sdvl_align_params AlignParams(bool fast) {
  sdvl_align_params ap;
  ap.max_level = Config::MaxAlignLevel();
  ap.min_level = Config::MinAlignLevel();
  ap.max_its = Config::MaxImgAlignIts();
  ap.patch_size = Config::AlignPatchSize();
  ap.fast = fast ? 1 : 0;
  return ap;
}

// what ComputePose reads of frame1's features (image_align.cc:147-160,219-236), appended to `feats`
void ImageAlign::PackFeatures(Frame &f1, vector<sdvl_align_feature> *feats) {
  const Vector3d first_pos = f1.GetWorldPosition();
  for (auto &ft : f1.GetFeatures()) {
    sdvl_align_feature a;
    a.px = ft->GetPosition()(0); a.py = ft->GetPosition()(1);
    a.fx = ft->GetVector()(0); a.fy = ft->GetVector()(1); a.fz = ft->GetVector()(2);
    Point *pt = ft->GetPointRaw();
    a.valid = (pt && !pt->ToDelete()) ? 1 : 0;
    a.depth = 0.0;
    if (a.valid) {
      const Vector3d p = pt->GetPosition();
      const double dx = p(0) - first_pos(0), dy = p(1) - first_pos(1), dz = p(2) - first_pos(2);
      a.depth = std::sqrt(dx * dx + dy * dy + dz * dz);
    }
    a.pad_ = 0;
    feats->push_back(a);
  }
}

// image_align.cc:46-84 for n pairs with one launch
void ImageAlign::ComputePoseBatch(const vector<std::pair<shared_ptr<Frame>, shared_ptr<Frame>>> &pairs, bool fast, vector<int> *n_meas,
                                  vector<double> *errors, vector<int> *iters, const vector<SE3> *start_poses, vector<SE3> *out_poses,
                                  const std::function<void()> *between) {
  const int n = static_cast<int>(pairs.size());
  n_meas->assign(n, 0);
  errors->assign(n, 1e10);
  if (iters) iters->assign(n, 0);
  if (out_poses) {
    out_poses->resize(n);
    for (int i = 0; i < n; i++) (*out_poses)[i] = start_poses ? (*start_poses)[i] : pairs[i].second->GetPose();
  }
  vector<sdvl_align_job> jobs;
  vector<int> job_of;
  vector<sdvl_align_feature> feats;
  {
    size_t total = 0;
    for (int i = 0; i < n; i++) total += pairs[i].first->GetFeatures().size();
    feats.reserve(total);
    jobs.reserve(n);
    job_of.reserve(n);
  }
  for (int i = 0; i < n; i++) {
    Frame &f1 = *pairs[i].first, &f2 = *pairs[i].second;
    vector<shared_ptr<Feature>> &features = f1.GetFeatures();
    if (features.empty()) {
      std::cerr << "[ERROR] No points to track!" << std::endl;  // image_align.cc:55-58
      continue;
    }
    sdvl_align_job job;
    job.ref = f1.device();
    job.cur = f2.device();
    job.feat_begin = static_cast<int32_t>(feats.size());
    PackFeatures(f1, &feats);
    job.feat_end = static_cast<int32_t>(feats.size());
    const SE3 T = (start_poses ? (*start_poses)[i] : f2.GetPose()) * f1.GetPose().Inverse();  // image_align.cc:66
    T.ToArray(job.T);
    jobs.push_back(job);
    job_of.push_back(i);
  }
  if (jobs.empty()) {
    if (between) (*between)();
    return;
  }
  Device *dev = Device::Current();
  const sdvl_camera cam = pairs[job_of[0]].second->GetCamera()->abi();
  const sdvl_align_params ap = AlignParams(fast);
  vector<sdvl_align_result> res(jobs.size());
  dev->Check(sdvl_image_align_begin(dev->ctx(), static_cast<int>(jobs.size()), jobs.data(), static_cast<int>(feats.size()), feats.data(), &cam,
                                    &ap), "sdvl_image_align_begin");
  if (between) (*between)();  // device work that can run behind the alignment (the new frames' corner detection)
  dev->Check(sdvl_image_align_end(dev->ctx(), static_cast<int>(jobs.size()), res.data()), "sdvl_image_align_end");
  for (size_t j = 0; j < jobs.size(); j++) {
    const int i = job_of[j];
    const SE3 pose2 = SE3::FromArray(res[j].T) * pairs[i].first->GetPose();  // image_align.cc:79
    if (out_poses) (*out_poses)[i] = pose2;
    else pairs[i].second->SetPose(pose2);
    (*n_meas)[i] = res[j].n_meas;
    (*errors)[i] = res[j].error;
    if (iters) (*iters)[i] = res[j].iters_run;
  }
}

int ImageAlign::ComputePose(const shared_ptr<Frame> &frame1, const shared_ptr<Frame> &frame2, bool fast) {
  vector<int> n;
  vector<double> e;
  ComputePoseBatch({{frame1, frame2}}, fast, &n, &e);
  error_ = e[0];
  return n[0];
}

// -------------------------------------------------------------------------------------------------------- Matcher
sdvl_search_params SearchParams() {
  sdvl_search_params sp;
  sp.patch_size = Config::PatchSize();
  sp.max_align_its = Config::MaxAlignIts();
  sp.search_size = Config::SearchSize();
  sp.max_fast_levels = Config::MaxFastLevels();
  sp.margin = DetectMargin();
  sp.use_orb = Config::UseORB() ? 1 : 0;
  sp.lk_tree_sums = 0;  // sequential-order LK sums: offsets bit-identical to the reference's loop (1: tolerance-class tree sums, include/sdvl_hip.h)
  sp.pad_ = 0;
  return sp;
}

bool Matcher::MakeRequest(const shared_ptr<Frame> &frame, const shared_ptr<Feature> &feature, double idepth, double idepth_std, bool fixed,
                          const Vector2d &px, sdvl_search_req *req) {
  shared_ptr<Frame> ref_frame = feature->GetFrame();
  if (!ref_frame) return false;
  req->cur = frame->device();
  req->ref = ref_frame->device();
  frame->GetPose().ToArray(req->cur_pose);
  ref_frame->GetPose().ToArray(req->ref_pose);
  req->px[0] = feature->GetPosition()(0); req->px[1] = feature->GetPosition()(1);
  req->bearing[0] = feature->GetVector()(0); req->bearing[1] = feature->GetVector()(1); req->bearing[2] = feature->GetVector()(2);
  req->idepth = idepth;
  req->idepth_std = idepth_std;
  req->px0[0] = px(0); req->px0[1] = px(1);
  req->level = feature->GetLevel();
  req->fixed = fixed ? 1 : 0;
  if (feature->HasDescriptor()) std::memcpy(req->desc, feature->DescriptorData().data(), 32);
  else std::memset(req->desc, 0, 32);
  return true;
}

void Matcher::SearchPoints(Device *dev, const vector<sdvl_search_req> &reqs, const Camera &cam, vector<sdvl_search_res> *res) {
  res->resize(reqs.size());
  if (reqs.empty()) return;
  const sdvl_camera c = cam.abi();
  const sdvl_search_params sp = SearchParams();
  dev->Check(sdvl_search_points(dev->ctx(), static_cast<int>(reqs.size()), reqs.data(), &c, &sp, res->data()), "sdvl_search_points");
}

void Matcher::SearchPointsFilter(Device *dev, const vector<sdvl_search_req> &reqs, const vector<sdvl_depth_state> &states, const Camera &cam,
                                 const sdvl_depth_params &fp, sdvl_track_set *set, vector<sdvl_search_res> *res, vector<sdvl_depth_out> *fout) {
  res->resize(reqs.size());
  fout->resize(reqs.size());
  if (reqs.empty()) return;
  if (states.size() != reqs.size()) throw std::runtime_error("SearchPointsFilter: one filter state per request");
  const sdvl_camera c = cam.abi();
  const sdvl_search_params sp = SearchParams();
  dev->Check(sdvl_search_points_filter(dev->ctx(), static_cast<int>(reqs.size()), reqs.data(), &c, &sp, states.data(), &fp, set, res->data(),
                                       fout->data()),
             "sdvl_search_points_filter");
}

// matcher.cc:45-121
bool Matcher::SearchPoint(const shared_ptr<Frame> &frame, const shared_ptr<Feature> &feature, double idepth, double idepth_std, bool fixed,
                          Vector2d *px, int *flevel) {
  vector<sdvl_search_req> reqs(1);
  if (!MakeRequest(frame, feature, idepth, idepth_std, fixed, *px, &reqs[0])) return false;
  vector<sdvl_search_res> res;
  SearchPoints(Device::Current(), reqs, *frame->GetCamera(), &res);
  if (res[0].stage >= 2 || res[0].found) { (*px)(0) = res[0].px[0]; (*px)(1) = res[0].px[1]; }
  if (!res[0].found) return false;
  *flevel = res[0].level;
  return true;
}

// feature_align.h:46: the reference draws from the process-wide rand(); a FeatureAlign built with its signature owns the
// stream a lone reference process would see (glibc TYPE_3, seed 1)
FeatureAlign::FeatureAlign(Map *map, Camera *camera, int max_matches) : FeatureAlign(map, camera, max_matches, nullptr) {}

FeatureAlign::FeatureAlign(Map *map, Camera *camera, int max_matches, RandStream *rng) {
  if (!rng) {
    own_rng_.reset(new RandStream(1));
    rng = own_rng_.get();
  }
  map_ = map;
  camera_ = camera;
  rng_ = rng;
  cell_size_ = Config::CellSize();
  max_matches_ = max_matches;
  matches_ = 0;
  num_attempts_ = 0;
  relocalizing_ = false;
  grid_width_ = static_cast<int>(std::ceil(static_cast<double>(camera->GetWidth()) / cell_size_));
  grid_height_ = static_cast<int>(std::ceil(static_cast<double>(camera->GetHeight()) / cell_size_));
  const int size = grid_width_ * grid_height_;
  grid_.resize(size);
  for (int i = 0; i < size; ++i) cell_order_.push_back(i);
  rng_->Shuffle(&cell_order_);
}

FeatureAlign::~FeatureAlign() {}

// feature_align.cc:285-339 (ResetGrid + ProjectPoints + ProjectPoint)
void FeatureAlign::ProjectPoints(const shared_ptr<Frame> &frame, const shared_ptr<Frame> &last_frame) {
  matches_ = 0;
  num_attempts_ = 0;
  for (auto &c : grid_) c.clear();
  vector<shared_ptr<Feature>> &features = last_frame->GetFeatures();
  const Rigid pose = frame->GetPose().pod();
  const M3 R = se3_rot(pose);  // Frame::Project = pose_ * p3D (frame.cc:94-103), rotation hoisted out of the loop
  const int patch = Config::PatchSize();
  const int frame_id = frame->GetID();
  const int n = static_cast<int>(features.size());
  for (int i = 0; i < n; i++) {
    // features and points are separate heap objects reached through pointers: ask for the ones a few iterations ahead
    if (i + 8 < n && features[i + 8]) __builtin_prefetch(features[i + 8].get());
    if (i + 4 < n && features[i + 4]) {
      const char *pp = reinterpret_cast<const char *>(features[i + 4]->GetPointRaw());
      __builtin_prefetch(pp);
      __builtin_prefetch(pp + 64);
    }
    Feature *ft = features[i].get();
    if (ft == nullptr) continue;
    Point *point = ft->GetPointRaw();
    if (!point || point->ToDelete()) continue;
    if (frame_id == point->GetLastFrame()) continue;
    {  // ProjectPoint
      const Vector3d P = point->GetPosition();
      const V3 rel = vadd(mvec(R, {P(0), P(1), P(2)}), pose.t);
      bool ok = !(rel.z < 0.0);
      Vector2d p;
      if (ok) {
        camera_->Project(Vector3d(rel.x, rel.y, rel.z), &p);
        ok = camera_->IsInsideImage(Vector2i(static_cast<int>(p(0)), static_cast<int>(p(1))), patch);
      }
      if (!ok) {
        point->SetStatus(Point::P_UNSEEN);
      } else {
        const int k = static_cast<int>(p(1) / cell_size_) * grid_width_ + static_cast<int>(p(0) / cell_size_);
        grid_.at(k).push_back(CellEntry{i, p, point->Score()});
        point->SetStatus(Point::P_SEEN);
        // the request loop of PrepareReproject reads the point's first observation (position, bearing, descriptor) next
        const char *fp = reinterpret_cast<const char *>(point->GetInitFeatureRaw());
        __builtin_prefetch(fp);
        __builtin_prefetch(fp + 64);
      }
    }
    if (!relocalizing_) point->SetLastFrame(frame_id);
  }
}

// first half of SelectPoints (feature_align.cc:88-118): project, shuffle, sort every cell, emit ALL candidates
void FeatureAlign::PrepareReproject(const shared_ptr<Frame> &frame, const shared_ptr<Frame> &last_frame, bool reloc,
                                    vector<sdvl_search_req> *reqs) {
  PrepareReprojectImpl(frame, last_frame, reloc, reqs, nullptr);
}

// the packed form: requests go straight into the staging area of an open sdvl_search_begin batch
void FeatureAlign::PrepareReprojectPacked(const shared_ptr<Frame> &frame, const shared_ptr<Frame> &last_frame, bool reloc, PackedSink *sink) {
  PrepareReprojectImpl(frame, last_frame, reloc, nullptr, sink);
}

void FeatureAlign::PrepareReprojectImpl(const shared_ptr<Frame> &frame, const shared_ptr<Frame> &last_frame, bool reloc,
                                        vector<sdvl_search_req> *reqs, PackedSink *sink) {
  inliers_.clear();
  outliers_.clear();
  found_.clear();
  obs_.clear();
  relocalizing_ = reloc;
  last_frame_ = last_frame;
  ProjectPoints(frame, last_frame);
  matches_ = 0;
  num_attempts_ = 0;
  rng_->Shuffle(&cell_order_);
  const int size = static_cast<int>(grid_.size());
  plan_.clear();
  plan_begin_.assign(size + 1, 0);
  vector<shared_ptr<Feature>> &features = last_frame->GetFeatures();
  double cur_pose[7];
  frame->GetPose().ToArray(cur_pose);
  req_base_ = sink ? sink->count : static_cast<int>(reqs->size());  // no exact-size reserve: callers append many trackers to one list
  const int cur_slot = sink ? frame->SearchSlot(sink->ctx, sink->batch_id) : -1;
  for (int i = 0; i < size; i++) {
    plan_begin_[i] = static_cast<int>(plan_.size());
    vector<CellEntry> &cell = grid_[cell_order_[i]];
    // cell->sort(CompareQuality): std::list::sort is a stable merge sort
    // (a stable order is unique, so any stable algorithm gives the list's order; cells hold a handful of entries and
    // std::stable_sort would malloc a merge buffer for each of them)
    if (cell.size() > 32) {
      std::stable_sort(cell.begin(), cell.end(), [](const CellEntry &a, const CellEntry &b) { return a.score > b.score; });
    } else {
      for (size_t a = 1; a < cell.size(); a++) {
        const CellEntry e = cell[a];
        size_t b = a;
        while (b > 0 && cell[b - 1].score < e.score) { cell[b] = cell[b - 1]; b--; }
        cell[b] = e;
      }
    }
    for (size_t ce = 0; ce < cell.size(); ce++) {
      const CellEntry &e = cell[ce];
      Point *point = features[e.src]->GetPointRaw();
      if (point->ToDelete()) continue;
      Feature *feature = point->GetInitFeatureRaw();
      if (!feature) continue;
      Candidate c{e.src, -1};
      Frame *ref_frame = feature->GetFrameRaw();
      if (ref_frame && sink) {
        if (sink->count >= sink->cap) throw std::runtime_error("FeatureAlign: packed request batch too small");
        sdvl_search_req_packed &rq = sink->reqs[sink->count++];
        rq.cur = cur_slot;
        rq.ref = ref_frame->SearchSlot(sink->ctx, sink->batch_id);
        rq.px[0] = feature->GetPosition()(0); rq.px[1] = feature->GetPosition()(1);
        rq.bearing[0] = feature->GetVector()(0); rq.bearing[1] = feature->GetVector()(1); rq.bearing[2] = feature->GetVector()(2);
        rq.idepth = point->GetInverseDepth();
        rq.idepth_std = point->GetStd();
        rq.px0[0] = e.p(0); rq.px0[1] = e.p(1);
        rq.level = feature->GetLevel();
        rq.fixed = point->IsFixed() ? 1 : 0;
        if (feature->HasDescriptor()) std::memcpy(rq.desc, feature->DescriptorData().data(), 32);
        else std::memset(rq.desc, 0, 32);
        if (sink->points) {
          const Vector3d P = point->GetPosition();
          double *dst = sink->points + 3 * static_cast<size_t>(sink->count - 1);
          dst[0] = P(0); dst[1] = P(1); dst[2] = P(2);
        }
        c.req = sink->count - 1 - req_base_;
      } else if (ref_frame) {
        reqs->emplace_back();
        sdvl_search_req &rq = reqs->back();
        rq.cur = frame->device();
        rq.ref = ref_frame->device();
        std::memcpy(rq.cur_pose, cur_pose, sizeof(cur_pose));
        ref_frame->GetPose().ToArray(rq.ref_pose);
        rq.px[0] = feature->GetPosition()(0); rq.px[1] = feature->GetPosition()(1);
        rq.bearing[0] = feature->GetVector()(0); rq.bearing[1] = feature->GetVector()(1); rq.bearing[2] = feature->GetVector()(2);
        rq.idepth = point->GetInverseDepth();
        rq.idepth_std = point->GetStd();
        rq.px0[0] = e.p(0); rq.px0[1] = e.p(1);
        rq.level = feature->GetLevel();
        rq.fixed = point->IsFixed() ? 1 : 0;
        if (feature->HasDescriptor()) std::memcpy(rq.desc, feature->DescriptorData().data(), 32);
        else std::memset(rq.desc, 0, 32);
        c.req = static_cast<int>(reqs->size()) - 1 - req_base_;  // relative: FinishReproject gets res + req_base
      }
      plan_.push_back(c);
    }
  }
  plan_begin_[size] = static_cast<int>(plan_.size());
}

void FeatureAlign::EmitChainCandidates(int req_offset, vector<int32_t> *cand_req, vector<int32_t> *cand_first) const {
  const int size = static_cast<int>(plan_begin_.size()) - 1;
  for (int i = 0; i < size; i++) {
    const int first = static_cast<int>(cand_req->size());
    for (int k = plan_begin_[i]; k < plan_begin_[i + 1]; k++) {
      cand_req->push_back(plan_[k].req >= 0 ? plan_[k].req + req_offset : -1);
      cand_first->push_back(first);
    }
  }
}

void FeatureAlign::PeekRand(int n, vector<int32_t> *out) const {
  RandStream peek = *rng_;
  for (int h = 0; h < n; h++) out->push_back(peek.Next());
}

void FeatureAlign::PeekRand(int n, int32_t *out) const {
  RandStream peek = *rng_;
  for (int h = 0; h < n; h++) out[h] = peek.Next();
}

void FeatureAlign::ShuffleCellRanks(uint16_t *rank_of_cell) {
  rng_->Shuffle(&cell_order_);
  for (size_t i = 0; i < cell_order_.size(); i++) rank_of_cell[cell_order_[i]] = static_cast<uint16_t>(i);
}

void FeatureAlign::AdvanceRand(int n) {
  for (int k = 0; k < n; k++) rng_->Next();
}

void FeatureAlign::SetTrackedCounts(int matches, int attempts, int inliers, int outliers) {
  matches_ = matches;
  num_attempts_ = attempts;
  found_.clear();
  obs_.clear();
  plan_.clear();
  last_frame_.reset();
  inliers_.assign(inliers, 0);
  outliers_.assign(outliers, 0);
}

// feature_align.cc:59-71 tail: SelectPoints replay, then SelectInliers
void FeatureAlign::FinishReproject(const shared_ptr<Frame> &frame, const sdvl_search_res *res) {
  FinishSelect(frame, res);
  SelectInliers(frame);
}

// second half of SelectPoints (feature_align.cc:105-149) replayed over the batch results
void FeatureAlign::FinishSelect(const shared_ptr<Frame> &frame, const sdvl_search_res *res, bool build_obs) {
  const int size = static_cast<int>(plan_begin_.size()) - 1;
  if (!relocalizing_) frame->GetFeatures().reserve(frame->GetFeatures().size() + static_cast<size_t>(max_matches_));
  vector<shared_ptr<Feature>> &src_features = last_frame_->GetFeatures();
  // the loop below touches, per candidate, the feature it came from and that feature's point — last used a whole batch of
  // frames ago.  Two passes of prefetches first (features, then points): independent misses overlap, one dependent chain
  // per candidate does not.
  if (!relocalizing_) {
    const int total = plan_begin_[size];
    for (int k = 0; k < total; k++) __builtin_prefetch(src_features[plan_[k].src].get());
    for (int k = 0; k < total; k++) {
      const char *p = reinterpret_cast<const char *>(src_features[plan_[k].src]->GetPointRaw());
      __builtin_prefetch(p, 1);
      __builtin_prefetch(p + 64, 1);
    }
  }
  for (int i = 0; i < size && matches_ < max_matches_; i++) {
    bool found = false;
    for (int k = plan_begin_[i]; k < plan_begin_[i + 1] && !found; k++) {
      const Candidate &cand = plan_[k];
      num_attempts_++;
      const sdvl_search_res *r = cand.req >= 0 ? &res[cand.req] : nullptr;
      found = r && r->found != 0;
      if (found) {
        if (!relocalizing_) {
          shared_ptr<Point> owner = src_features[cand.src]->GetPoint();
          Point *point = owner.get();
          point->Promote();
          shared_ptr<Feature> feature = frame->NewFeature(Vector2d(r->px[0], r->px[1]), r->level);
          feature->SetPoint(std::move(owner));
          Feature *const raw = feature.get();
          frame->GetFeatures().push_back(std::move(feature));  // Frame::AddFeature without a second reference
          point->SetStatus(Point::P_FOUND);
          if (build_obs) {  // observation records for the pose stage; the chained device path builds its own
            const Vector3d P = point->GetPosition();
            const Vector3d &v = raw->GetVector();
            obs_.push_back(Obs{v(0) / v(2), v(1) / v(2), P(0), P(1), P(2), 1.0 / (1 << r->level)});
          }
          found_.push_back(raw);
        }
        matches_++;
      } else {
        if (!relocalizing_) {
          Point *point = src_features[cand.src]->GetPointRaw();
          if (point->Unpromote()) map_->DeletePoint(src_features[cand.src]->GetPoint());
          point->SetStatus(Point::P_NOT_FOUND);
        }
      }
    }
  }
  plan_.clear();
  last_frame_.reset();
  inliers_.clear();
  outliers_.clear();
}

// ---- device pose stage (sdvl_pose_from_matches): what SelectInliers needs from the host — the rand() draws it would
// make (taken from a COPY of the stream; CommitPose advances the real one by the number actually used) and the
// iteration budget as a function of the supporter count (feature_align.cc:199-207: libm log stays on the host)
void FeatureAlign::PoseBatch::Append(const PoseBatch &o) {
  const int ob = static_cast<int>(obs.size()), rb = static_cast<int>(rand_idx.size()), nb = static_cast<int>(nits.size());
  for (sdvl_pose_job j : o.jobs) {
    j.obs_begin += ob; j.obs_end += ob; j.rand_begin += rb; j.nits_begin += nb;
    jobs.push_back(j);
  }
  obs.insert(obs.end(), o.obs.begin(), o.obs.end());
  rand_idx.insert(rand_idx.end(), o.rand_idx.begin(), o.rand_idx.end());
  nits.insert(nits.end(), o.nits.begin(), o.nits.end());
}

sdvl_pose_params FeatureAlign::PoseParams(const Camera &cam) {
  sdvl_pose_params p;
  p.max_ransac_points = Config::MaxRansacPoints();
  p.max_ransac_its = Config::MaxRansacIts();
  p.max_optim_pose_its = Config::MaxOptimPoseIts();
  p.pad_ = 0;
  p.inlier_threshold = Config::InlierErrorThreshold() / cam.GetFx();
  p.fx = cam.GetFx();
  return p;
}

bool FeatureAlign::EmitPoseJob(const shared_ptr<Frame> &frame, PoseBatch *batch) {
  const int size = static_cast<int>(found_.size());
  if (size > kMaxDevicePoseObs || Config::MaxRansacPoints() > 8) return false;
  sdvl_pose_job job;
  job.obs_begin = static_cast<int>(batch->obs.size());
  job.obs_end = job.obs_begin + size;
  job.rand_begin = static_cast<int>(batch->rand_idx.size());
  job.nits_begin = static_cast<int>(batch->nits.size());
  frame->GetPose().ToArray(job.pose);
  for (const Obs &o : obs_) batch->obs.push_back(sdvl_pose_obs{o.ax, o.ay, o.px, o.py, o.pz, o.inv_cov});
  const int max_its = Config::MaxRansacIts();
  RandStream peek = *rng_;
  for (int h = 0; h < max_its; h++) batch->rand_idx.push_back(size > 0 ? peek.Next() % size : 0);
  // the budget table depends on (size, MaxRansacPoints, MaxRansacIts) only: computed once per size and thread (two
  // log() per entry otherwise, ~20k calls per batch step)
  struct BudgetCache { int points = -1, its = -1; vector<vector<int32_t>> by_size; };
  static thread_local BudgetCache cache;
  if (cache.points != Config::MaxRansacPoints() || cache.its != max_its) {
    cache.points = Config::MaxRansacPoints();
    cache.its = max_its;
    cache.by_size.assign(kMaxDevicePoseObs + 1, vector<int32_t>());
  }
  vector<int32_t> &table = cache.by_size[size];
  if (table.empty()) {
    const int npoints = std::min(Config::MaxRansacPoints(), size);
    const double sprob = 0.99;
    for (int supporters = 0; supporters <= size; supporters++) {
      int nits = max_its;
      if (size > 0) {
        const double epsilon = 1.0 - (static_cast<double>(supporters) / static_cast<double>(size));
        double tmp = 1.0 - epsilon;
        for (int k = 1; k < npoints; k++) tmp *= tmp;
        if (!(tmp < 1e-5)) nits = std::min(max_its, static_cast<int>(std::log(1.0 - sprob) / std::log(1.0 - tmp)));
      }
      table.push_back(nits);
    }
  }
  batch->nits.insert(batch->nits.end(), table.begin(), table.end());
  batch->jobs.push_back(job);
  return true;
}

void FeatureAlign::CommitPose(const shared_ptr<Frame> &frame, const sdvl_pose_result &r, const int32_t *lists) {
  for (int k = 0; k < r.n_draws; k++) rng_->Next();
  if (r.refined) frame->SetPose(SE3(se3_from7(r.pose)));
  inliers_.assign(lists, lists + r.n_inliers);
  outliers_.assign(lists + r.n_inliers, lists + r.n_inliers + r.n_outliers);
  RemoveOutliers(frame);
}

// feature_align.cc:59-71
void FeatureAlign::Reproject(const shared_ptr<Frame> &frame, const shared_ptr<Frame> &last_frame, const shared_ptr<Frame> &, bool reloc) {
  vector<sdvl_search_req> reqs;
  PrepareReproject(frame, last_frame, reloc, &reqs);
  vector<sdvl_search_res> res;
  Matcher::SearchPoints(Device::Current(), reqs, *camera_, &res);
  FinishReproject(frame, res.data() + req_base_);
}

// feature_align.cc:73-82
bool FeatureAlign::OptimizePose(const shared_ptr<Frame> &frame) {
  OptimizePoseOnce(frame);
  if (RescueOutliers(frame)) OptimizePoseOnce(frame);
  RemoveOutliers(frame);
  return true;
}

// feature_align.cc:152-216
void FeatureAlign::SelectInliers(const shared_ptr<Frame> &frame) {
  inliers_.clear();
  outliers_.clear();
  if (found_.empty()) return;
  const int size = static_cast<int>(found_.size());
  const int npoints = std::min(Config::MaxRansacPoints(), size);
  vector<int> all(size), selected(npoints);
  for (int i = 0; i < size; i++) all[i] = i;
  SE3 se3, best_se3;
  const SE3 frame_pose = frame->GetPose();
  const double sprob = 0.99;
  int nits = Config::MaxRansacIts();
  int best_supporters = 0;
  int it = 0;
  const double thr = Config::InlierErrorThreshold() / frame->GetCamera()->GetFx();
  while (it < nits) {
    const int index = rng_->Next() % size;
    for (int i = 0; i < npoints; i++) selected[i] = (index + i) % size;
    if (!ConvergePose(frame_pose, selected.data(), npoints, &se3)) { it++; continue; }
    const int supporters = CheckReprojectionError(all, se3, thr, nullptr, nullptr);
    if (supporters > best_supporters) {
      best_supporters = supporters;
      best_se3 = se3;
      const double epsilon = 1.0 - (static_cast<double>(supporters) / static_cast<double>(size));
      double tmp = 1.0 - epsilon;
      for (int k = 1; k < npoints; k++) tmp *= tmp;
      if (tmp < 1e-5) nits = Config::MaxRansacIts();
      else nits = std::min(Config::MaxRansacIts(), static_cast<int>(std::log(1.0 - sprob) / std::log(1.0 - tmp)));
    }
    it++;
  }
  CheckReprojectionError(all, best_se3, thr, &inliers_, &outliers_);
}

// feature_align.cc:218-230
void FeatureAlign::OptimizePoseOnce(const shared_ptr<Frame> &frame) {
  SE3 se3 = frame->GetPose();
  if (!ConvergePose(frame->GetPose(), inliers_.data(), static_cast<int>(inliers_.size()), &se3)) return;
  frame->SetPose(se3);
  vector<int> cfeatures = inliers_;
  inliers_.clear();
  CheckReprojectionError(cfeatures, frame->GetPose(), Config::InlierErrorThreshold() / frame->GetCamera()->GetFx(), &inliers_, &outliers_);
}

// feature_align.cc:232-243
bool FeatureAlign::RescueOutliers(const shared_ptr<Frame> &frame) {
  const int init_inliers = static_cast<int>(inliers_.size());
  vector<int> cfeatures = outliers_;
  outliers_.clear();
  CheckReprojectionError(cfeatures, frame->GetPose(), 2 * Config::InlierErrorThreshold() / frame->GetCamera()->GetFx(), &inliers_, &outliers_);
  return static_cast<int>(inliers_.size()) > init_inliers;
}

// feature_align.cc:245-256
void FeatureAlign::RemoveOutliers(const shared_ptr<Frame> &frame) {
  for (int i : outliers_) {
    Feature &ft = *found_[i];
    shared_ptr<Point> p = ft.GetPoint();
    if (!p) continue;
    ft.SetPoint(nullptr);
    p->SetStatus(Point::P_NOT_FOUND);
    frame->AddOutlier(ft.GetPosition());
  }
}

// feature_align.cc:258-283 over observation records (every found feature has a point until RemoveOutliers)
int FeatureAlign::CheckReprojectionError(const vector<int> &idx, const SE3 &se3, double threshold, vector<int> *inliers, vector<int> *outliers) {
  int valids = 0;
  const Rigid s = se3.pod();
  const M3 R = se3_rot(s);
  for (int i : idx) {
    const Obs &o = obs_[i];
    const V3 pos = vadd(mvec(R, {o.px, o.py, o.pz}), s.t);
    double ex = o.ax - pos.x / pos.z, ey = o.ay - pos.y / pos.z;
    ex *= o.inv_cov;
    ey *= o.inv_cov;
    if (std::sqrt(ex * ex + ey * ey) <= threshold) {
      valids++;
      if (inliers != NULL) inliers->push_back(i);
    } else {
      if (outliers != NULL) outliers->push_back(i);
    }
  }
  return valids;
}

// feature_align.cc:341-421
bool FeatureAlign::ConvergePose(const SE3 &frame_pose, const int *idx, int n, SE3 *se3) {
  SE3 last_se3 = frame_pose;
  *se3 = last_se3;
  double chi2 = 0.0;
  errors_.clear();
  {
    const Rigid s = se3->pod();
    const M3 R = se3_rot(s);
    for (int q = 0; q < n; q++) {
      const Obs &o = obs_[idx[q]];
      const V3 pos = vadd(mvec(R, {o.px, o.py, o.pz}), s.t);
      double ex = o.ax - pos.x / pos.z, ey = o.ay - pos.y / pos.z;
      ex *= o.inv_cov;
      ey *= o.inv_cov;
      errors_.push_back(std::sqrt(ex * ex + ey * ey));
    }
  }
  if (errors_.empty()) return false;
  auto mid = errors_.begin() + static_cast<long>(std::floor(errors_.size() / 2));  // GetMedianVector, extra/utils.cc:215-220
  std::nth_element(errors_.begin(), mid, errors_.end());
  double scale = KMADNorm * (*mid);
  const double fx = camera_->GetFx();
  for (int i = 0; i < Config::MaxOptimPoseIts(); i++) {
    double A[36], b[6];
    for (int r = 0; r < 6; r++) b[r] = 0.0;
    for (int r = 0; r < 36; r++) A[r] = 0.0;
    double new_chi2 = 0.0;
    if (i == 5) scale = 0.85 / fx;
    const Rigid s = se3->pod();
    const M3 R = se3_rot(s);
    for (int q = 0; q < n; q++) {
      const Obs &o = obs_[idx[q]];
      const V3 pos = vadd(mvec(R, {o.px, o.py, o.pz}), s.t);
      double J[12];
      jacobian_3d_to_plane(pos, J);
      double ex = o.ax - pos.x / pos.z, ey = o.ay - pos.y / pos.z;
      const double sqrt_inv_cov = o.inv_cov;
      ex *= sqrt_inv_cov;
      ey *= sqrt_inv_cov;
      for (int c = 0; c < 12; c++) J[c] *= sqrt_inv_cov;
      const double weight = GetTukeyValue(std::sqrt(ex * ex + ey * ey) / scale);
      for (int r = 0; r < 6; r++) {
        for (int c = 0; c < 6; c++) A[6 * r + c] += (J[r] * J[c] + J[6 + r] * J[6 + c]) * weight;
        b[r] -= (J[r] * ex + J[6 + r] * ey) * weight;
      }
      new_chi2 += (ex * ex + ey * ey) * weight;
    }
    double dT[6];
    ldlt_solve6(A, b, dT);
    if ((i > 0 && new_chi2 > chi2) || std::isnan(dT[0])) {
      *se3 = last_se3;
      break;
    }
    Vector6d d;
    for (int r = 0; r < 6; r++) d[r] = dT[r];
    const SE3 T_new = SE3::Exp(d) * (*se3);
    last_se3 = *se3;
    *se3 = T_new;
    chi2 = new_chi2;
    if (abs_max6(dT) <= 1e-10) break;
  }
  return true;
}

// feature_align.cc:423-431
double FeatureAlign::GetTukeyValue(double x) {
  const double x_square = x * x;
  if (x_square <= KTukeyC) {
    const double tmp = 1.0 - x_square / KTukeyC;
    return tmp * tmp;
  }
  return 0.0;
}
}  // namespace sdvl
