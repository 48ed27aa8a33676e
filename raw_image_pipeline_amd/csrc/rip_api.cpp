// rip_api.cpp -- implementation of the C-ABI declared in include/rip.h: one handle owns the
// module parameters (rip_host.hpp), the device-resident constants and scratch, and enqueues the
// kernels of rip_chain/stats/ccc/remap.hip on the caller's HIP stream.  There is no CPU execution path.
#include "../../include/rip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include <pthread.h>
#include <stdexcept>
#include <string>
#include <vector>

#include "rip_host.hpp"
#include "rip_kernels.hpp"

namespace {

struct InvalidArgument : std::invalid_argument {
  using std::invalid_argument::invalid_argument;
};
struct AssertError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct DeviceError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct CapacityError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                                  \
  do {                                                                                                   \
    hipError_t err_ = (expr);                                                                            \
    if (err_ != hipSuccess)                                                                              \
      throw DeviceError(std::string(#expr) + " failed: " + hipGetErrorString(err_) + " (" + __FILE__ + \
                        ":" + std::to_string(__LINE__) + ")");                                           \
  } while (0)

thread_local std::string g_create_error;

// Grow-only device buffer; frees its memory when it goes out of scope (on the device that is current then: handles
// release theirs explicitly under their own device in ~rip_pipeline)
struct DevBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void reserve(size_t bytes) {
    if (bytes <= cap) return;
    if (ptr) HIP_CHECK(hipFree(ptr));
    ptr = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8;
    HIP_CHECK(hipMalloc(&ptr, want));
    cap = want;
    // RIP_TRACE_ALLOC=1: one line per device allocation on stderr (tools/probes/remap_modes_probe.py relates the per-process
    // modes of the remap's duration to where its buffers landed)
    static const bool trace = std::getenv("RIP_TRACE_ALLOC") != nullptr;
    if (trace) std::fprintf(stderr, "rip alloc %zu bytes at %p\n", want, ptr);
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return static_cast<T*>(ptr);
  }
};

// Selects the handle's device for the duration of one C-ABI call and puts the caller's current device back afterwards
// (a CameraRig thread, or torch, keeps its own current device across calls into handles that live elsewhere).
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    HIP_CHECK(hipSetDevice(device));
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct HostImage {
  std::vector<uint8_t> data;
  int rows = 0, cols = 0, channels = 0;
};

int parse_bayer(const std::string& e, int& ry, int& rx) {
  // position of the R sample in the 2x2 cell for the ROS pattern names (debayer.cpp:48-70)
  if (e == "bayer_rggb8") { ry = 0; rx = 0; return 1; }
  if (e == "bayer_grbg8") { ry = 0; rx = 1; return 1; }
  if (e == "bayer_gbrg8") { ry = 1; rx = 0; return 1; }
  if (e == "bayer_bggr8") { ry = 1; rx = 1; return 1; }
  return 0;
}
bool is_bayer16(const std::string& e) {
  return e == "bayer_bggr16" || e == "bayer_gbrg16" || e == "bayer_grbg16" || e == "bayer_rggb16";
}

// What one frame geometry/encoding turns into
struct Plan {
  int src_kind = rip::SRC_BGR, ry = 0, rx = 0;
  int elem_bytes = 1;     // 2: the 16-bit Bayer extension (debayer + flip only, bgr16 out)
  int channels = 3;       // channels after the debayer stage
  int flip_angle = 0;     // effective
  int mid_rows = 0, mid_cols = 0;  // post-flip geometry (pointwise chain output)
  int out_rows = 0, out_cols = 0;
  bool remap = false;
  int wb_mode = rip::WB_NONE;
  int stage_bits = 0;
  std::string encoding_out;
};

// Host-side copies of whole frames (the deep copies the reference's API promises: process() and every image getter return a
// clone, raw_image_pipeline.cpp:182-236): a 15 MB memcpy out of the pinned result buffer into freshly allocated pages takes
// 0.4-0.7 ms on one thread -- more than the frame's kernels and its PCIe transfer together -- so copies of 4 MB and more are
// split over a few worker threads (page faults of a fresh destination included).  The pool is created on first use, leaked on
// purpose (its threads may outlive static destruction) and serves one copy at a time; a second caller copies on its own thread.
class CopyPool {
 public:
  static CopyPool& get() {
    static CopyPool* pool = new CopyPool();
    return *pool;
  }
  void copy(void* dst, const void* src, size_t bytes) {
    // a forked child has none of the worker threads and may have inherited the mutexes locked: plain memcpy there
    if (bytes < (size_t(4) << 20) || workers_ == 0 || forked().load(std::memory_order_relaxed) || !busy_.try_lock()) {
      std::memcpy(dst, src, bytes);
      return;
    }
    std::lock_guard<std::mutex> whole(busy_, std::adopt_lock);
    std::unique_lock<std::mutex> lk(mu_);
    dst_ = static_cast<uint8_t*>(dst);
    src_ = static_cast<const uint8_t*>(src);
    bytes_ = bytes;
    parts_ = workers_ + 1;
    chunk_ = ((bytes + (size_t)parts_ - 1) / (size_t)parts_ + 4095) & ~size_t(4095);
    next_ = done_ = 0;
    wake_.notify_all();
    work(lk);
    finished_.wait(lk, [&] { return done_ == parts_; });
    parts_ = 0;
  }

 private:
  static std::atomic<bool>& forked() {
    static std::atomic<bool> f{false};
    return f;
  }
  CopyPool() {
    (void)pthread_atfork(nullptr, nullptr, [] { forked().store(true, std::memory_order_relaxed); });
    const char* e = std::getenv("RIP_COPY_THREADS");  // worker threads beside the caller; 0 = plain memcpy
    const int hw = (int)std::thread::hardware_concurrency();
    workers_ = e ? std::max(0, std::min(16, std::atoi(e))) : std::max(0, std::min(3, hw - 1));
    for (int i = 0; i < workers_; i++)
      std::thread([this] {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
          wake_.wait(lk, [&] { return next_ < parts_; });
          work(lk);
        }
      }).detach();
  }
  // claims parts of the current copy until none is left; called and left with the lock held
  void work(std::unique_lock<std::mutex>& lk) {
    while (next_ < parts_) {
      const size_t off = (size_t)next_++ * chunk_;
      uint8_t* d = dst_;
      const uint8_t* s = src_;
      const size_t n = off < bytes_ ? std::min(chunk_, bytes_ - off) : 0;
      lk.unlock();
      if (n) std::memcpy(d + off, s + off, n);
      lk.lock();
      if (++done_ == parts_) finished_.notify_all();
    }
  }
  std::mutex busy_, mu_;
  std::condition_variable wake_, finished_;
  uint8_t* dst_ = nullptr;
  const uint8_t* src_ = nullptr;
  size_t bytes_ = 0, chunk_ = 0;
  int parts_ = 0, next_ = 0, done_ = 0, workers_ = 0;
};

// Device-wide limit on the host frames in flight (rip_submit).  Measured on MI355X / ROCm 7.2 (tools/probes/ring_depth_probe.py,
// rig_ring_probe.py, EXPERIMENTS.md): with three frames' uploads, kernels and downloads enqueued on their streams a 15 MB
// download takes 0.295 ms; from the fourth frame on -- one handle with a deeper ring, or several handles on one device -- some
// downloads take 1.19 ms or several ms (the runtime's handling of cross-stream dependencies of SDMA copies under direct
// dispatch: the effect is gone with AMD_DIRECT_DISPATCH=0), and the whole pipeline runs at half its rate.  So rip_submit
// waits for the oldest frame in flight on the device to finish before it enqueues a fourth one -- work that has to finish
// before the new frame's download can start anyway.  RIP_RING_INFLIGHT changes the limit (0 = none).
struct InflightGate {
  // One mutex and one queue per device: a submit that waits at the limit on device 0 holds device 0's lock only;
  // submits and collects of handles on the other devices of a multi-GPU rig go on.
  std::mutex mu[64];
  std::deque<hipEvent_t> q[64];  // per device: ev_done of the frames enqueued and not yet known to be complete, oldest first
  int cap() {
    static const int c = [] {
      const char* e = std::getenv("RIP_RING_INFLIGHT");
      return e ? std::max(0, std::atoi(e)) : 3;
    }();
    return c;
  }
  void admit(int device) {
    const int c = cap();
    if (c <= 0) return;
    // The wait happens under the DEVICE's lock on purpose: an event in the queue belongs to some handle's slot, and forget() --
    // called before a slot's events are destroyed or recorded again -- must not get past it while it is waited on.  The wait
    // is for a frame that is already enqueued in full (at most one frame time), and whoever else wants this device's lock
    // meanwhile is either about to wait for the same frame (another submit at the limit) or finishes a collect a moment later.
    std::lock_guard<std::mutex> lk(mu[device & 63]);
    auto& d = q[device & 63];
    while ((int)d.size() >= c) {
      hipEvent_t e = d.front();
      d.pop_front();
      (void)hipEventSynchronize(e);
    }
  }
  void enqueued(int device, hipEvent_t e) {
    if (cap() <= 0) return;
    std::lock_guard<std::mutex> lk(mu[device & 63]);
    q[device & 63].push_back(e);
  }
  // the frame is complete, or its event is about to be destroyed / recorded again.  device: where it was enqueued (< 0: never)
  void forget(int device, hipEvent_t e) {
    if (!e || device < 0) return;
    std::lock_guard<std::mutex> lk(mu[device & 63]);
    auto& d = q[device & 63];
    for (auto it = d.begin(); it != d.end();) it = (*it == e) ? d.erase(it) : it + 1;
  }
};
InflightGate& inflight_gate() {
  static InflightGate g;
  return g;
}

// One frame in flight on the asynchronous host path (rip_submit / rip_collect): its own device input / output / tap
// buffers, a pinned result buffer, and the three events that chain upload -> kernels -> download.
struct RingSlot {
  DevBuf d_in, d_out, d_tap_deb, d_tap_col;
  void* h_out = nullptr;  // hipHostMalloc
  size_t h_out_cap = 0;
  void* h_in = nullptr;   // hipHostMalloc: staging copy of a pageable caller frame (the caller's buffer is free again when rip_submit returns)
  size_t h_in_cap = 0;
  void* h_tap[2] = {nullptr, nullptr};  // hipHostMalloc: the debayered / colour taps of the frame, downloaded with the result
  // where this frame's downloads go: the slot's own pinned buffers above, or the page-locked buffers the caller gave rip_submit_to
  void* dst_out = nullptr;
  void* dst_tap[2] = {nullptr, nullptr};
  int gate_device = -1;  // device whose InflightGate queue holds ev_done (set when the frame is enqueued)
  size_t h_tap_cap[2] = {0, 0};
  hipEvent_t ev_up = nullptr, ev_kernels = nullptr, ev_done = nullptr;
  hipEvent_t ev_start = nullptr, ev_dl_start = nullptr;  // RIP_DEBUG_RING only: before the upload / the download (the other three then carry timestamps too)
  uint64_t ticket = 0;
  bool busy = false;  // submitted, not collected yet
  bool held = false;  // collected: the pinned result and the taps stay put until the next collect (or until a submit needs the slot)
  Plan pl;
  bool has_deb = false, has_col = false;  // the taps this frame keeps on the device
  bool dl_deb = false, dl_col = false;    // ... and downloads into h_tap with the result
  void reserve_host(size_t bytes) {
    if (bytes <= h_out_cap) return;
    if (h_out) HIP_CHECK(hipHostFree(h_out));
    h_out = nullptr;
    h_out_cap = 0;
    HIP_CHECK(hipHostMalloc(&h_out, bytes + bytes / 8, hipHostMallocDefault));
    h_out_cap = bytes + bytes / 8;
  }
  static void reserve_pinned(void*& ptr, size_t& cap, size_t bytes) {
    if (bytes <= cap) return;
    if (ptr) HIP_CHECK(hipHostFree(ptr));
    ptr = nullptr;
    cap = 0;
    HIP_CHECK(hipHostMalloc(&ptr, bytes + bytes / 8, hipHostMallocDefault));
    cap = bytes + bytes / 8;
  }
  void reserve_host_in(size_t bytes) { reserve_pinned(h_in, h_in_cap, bytes); }
  void release() {
    if (h_out) (void)hipHostFree(h_out);
    h_out = nullptr;
    h_out_cap = 0;
    if (h_in) (void)hipHostFree(h_in);
    h_in = nullptr;
    h_in_cap = 0;
    for (int i = 0; i < 2; i++) {
      if (h_tap[i]) (void)hipHostFree(h_tap[i]);
      h_tap[i] = nullptr;
      h_tap_cap[i] = 0;
    }
    inflight_gate().forget(gate_device, ev_done);
    gate_device = -1;
    for (hipEvent_t* e : {&ev_up, &ev_kernels, &ev_done, &ev_start, &ev_dl_start}) {
      if (*e) (void)hipEventDestroy(*e);
      *e = nullptr;
    }
    for (DevBuf* b : {&d_in, &d_out, &d_tap_deb, &d_tap_col}) b->release();
  }
};

}  // namespace

struct rip_pipeline {
  int device = 0;
  hipStream_t stream = nullptr;
  rip::Modules m;
  // environment overrides, read once when the handle is created (never on a frame path)
  rip::Tunables tn;
  bool maps_on_host = false;      // RIP_MAPS_ON_HOST
  bool plan_on_host = false;      // RIP_PLAN_ON_HOST: compile the remap plan on the host even when the maps are on the device
  std::string debug_dir = "/tmp"; // RIP_DEBUG_DIR
  std::string ccc_model_env;      // RIP_CCC_MODEL
  mutable std::string last_error;
  int tap_mask = RIP_TAP_DEBAYERED | RIP_TAP_COLOR | RIP_TAP_PROCESSED;
  int tap_download_mask = 0;  // rip_set_tap_download: which of the kept taps rip_submit also downloads with the result
  int fp_contract = 0;        // rip_set_fp_contraction / RIP_FP_CONTRACT: contraction model of the float stages (0 none, 1 fused)

  // constants on the device
  rip::DevTables h_tabs;
  DevBuf d_tabs, d_vig_image;  // d_vig_image: the fused chain's LDS tables as one image (rip::launch_vig_image)
  bool tabs_dirty = true;
  // undistortion maps (interleaved float2), built lazily
  std::vector<float> h_map;
  DevBuf d_map, d_map_ckpt;  // d_map_ckpt: scratch of the map kernels (row accumulators at every 32nd column)
  bool map_dirty = true, map_uploaded = false;
  bool h_map_valid = false;  // device-built maps are copied to the host only when something on the host asks for them
  // vignetting mask plane per geometry (float, rows x cols)
  std::vector<float> h_vig;
  DevBuf d_vig;
  int vig_rows = -1, vig_cols = -1;
  bool vig_dirty = true;
  // ccc
  rip::CccModel ccc;
  DevBuf d_filter_fft, d_bias_fft, d_accum, d_ccc_state, d_geom;
  bool ccc_uploaded = false, ccc_state_init = false, ccc_reset_pending = false, ccc_cfg_dirty = true;
  double kf_h = 0.0, kf_r = 1.0;
  int geom_rows = -1, geom_cols = -1;
  // per-batch scratch
  DevBuf d_stats, d_wb, d_hist, d_work, d_rowbest, d_argmax, d_mid;
  // compiled remap plan (tiled LDS gather), rebuilt when the maps or the source geometry change
  rip::RemapPlan plan;
  DevBuf d_plan_words, d_plan_tiles, d_plan_border, d_plan_counters;
  bool plan_uploaded = false;
  bool plan_on_device = false;  // compiled by remap_plan_kernel: plan.words / tiles / border stay empty on the host
  int plan_n_border = 0;
  bool use_tiled_remap = true;
  int last_batch_frames = 0;
  bool work_enqueued = false;  // some frame call has put work on `stream` (rip_set_stream orders a new stream behind it)
  // prefix of d_stats known to hold zeroed FrameStats records (the grey-world / pca statistics kernels clean up after themselves)
  const void* stats_clean_ptr = nullptr;
  size_t stats_clean_cap = 0, stats_clean_bytes = 0;  // (pointer, capacity) identify the allocation: DevBuf only ever grows
  // the leading bytes of d_hist known to be zero: the ccc estimator's global-atomic histogram (small batches) hands its
  // counters back zeroed, so a stream of single frames pays for one memset, not one per frame
  const void* hist_clean_ptr = nullptr;
  size_t hist_clean_cap = 0, hist_clean_bytes = 0;
  // cross-kernel overlap inside one batch (run_batch): the remap of frame group g runs on this internal stream while the
  // statistics and the fused chain of group g + 1 run on the caller's stream
  hipStream_t aux_stream = nullptr;
  std::vector<hipEvent_t> ovl_events;
  hipEvent_t switch_event = nullptr;  // rip_set_stream: orders the new stream behind the work left on the old one
  // asynchronous host path: frames in flight (rip_submit / rip_collect), upload and download streams
  std::vector<std::unique_ptr<RingSlot>> ring;
  int ring_depth = 3;
  uint64_t next_ticket = 1;
  hipStream_t ul_stream = nullptr, dl_stream = nullptr;
  // optional per-kernel timing with HIP events on the handle's stream (bench.py roofline leg)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_events;  // pairs
  std::vector<int> prof_ids;
  size_t prof_used = 0;
  // host-apply staging and last-frame taps
  DevBuf d_in, d_out, d_tap_deb, d_tap_col, d_dbg;
  int last_rows[3] = {0, 0, 0}, last_cols[3] = {0, 0, 0}, last_cn[3] = {0, 0, 0};
  bool last_valid[3] = {false, false, false};
  DevBuf* last_buf[3] = {nullptr, nullptr, nullptr};
  const void* last_host[3] = {nullptr, nullptr, nullptr};  // pinned host copy of the image (frames that came through rip_collect), else null

  ~rip_pipeline() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    for (hipEvent_t e : prof_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ovl_events) (void)hipEventDestroy(e);
    if (switch_event) (void)hipEventDestroy(switch_event);
    if (aux_stream) (void)hipStreamDestroy(aux_stream);
    if (ul_stream) (void)hipStreamSynchronize(ul_stream);
    if (dl_stream) (void)hipStreamSynchronize(dl_stream);
    for (auto& sl : ring) sl->release();
    if (ul_stream) (void)hipStreamDestroy(ul_stream);
    if (dl_stream) (void)hipStreamDestroy(dl_stream);
    for (DevBuf* b : {&d_tabs, &d_vig_image, &d_map, &d_map_ckpt, &d_filter_fft, &d_bias_fft, &d_accum, &d_ccc_state, &d_geom, &d_stats, &d_wb,
                      &d_hist, &d_work, &d_rowbest, &d_argmax, &d_mid, &d_in, &d_out, &d_tap_deb, &d_tap_col, &d_vig, &d_plan_words,
                      &d_plan_tiles, &d_plan_border, &d_plan_counters, &d_dbg})
      b->release();
  }
};

namespace {

// RAII marker: records an event pair around the launches of one kernel class when profiling is on
struct ProfScope {
  rip_pipeline* p;
  hipStream_t stream;  // the stream the class's kernels are launched on (the handle's, or the internal overlap stream)
  size_t slot = (size_t)-1;
  ProfScope(rip_pipeline* pp, int id, hipStream_t s) : p(pp), stream(s) {
    if (!p->prof_on || p->prof_used + 2 > p->prof_events.size()) return;
    slot = p->prof_used;
    p->prof_used += 2;
    p->prof_ids.push_back(id);
    (void)hipEventRecord(p->prof_events[slot], stream);
  }
  ~ProfScope() {
    if (slot != (size_t)-1) (void)hipEventRecord(p->prof_events[slot + 1], stream);
  }
};

// ------------------------------------------------------------------------------------------------
// undistortion bookkeeping: UndistortionModule::init() (undistortion.cpp:197-238) minus the map
// generation, which is deferred until a frame (or rip_init_undistortion) needs it.
// ------------------------------------------------------------------------------------------------
void und_init(rip_pipeline* p) {
  rip::Modules& m = p->m;
  double newK[9];
  rip::fisheye_estimate_new_camera_matrix(m.dist_K, m.dist_D, m.dist_w, m.dist_h, m.dist_R, m.balance, m.rect_w, m.rect_h,
                                          m.fov_scale, newK);
  std::memcpy(m.rect_K, newK, sizeof(newK));
  for (int i = 0; i < 4; i++) m.rect_D[i] = 0;
  const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(m.rect_R, eye, sizeof(eye));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m.rect_P[i * 4 + j] = m.rect_K[i * 3 + j];
  p->map_dirty = true;
}

// Where the maps are built: on the device for device handles (rip_maps.hip: one thread per map row, FP64, double-double
// atan -- milliseconds instead of 0.25-0.5 s of host threads per calibration change), on the host for RIP_DEVICE_NONE handles
// and when RIP_MAPS_ON_HOST is set (A/B and debugging).  Both produce the same floats (tests/test_parity_gpu.py).
bool maps_on_device(const rip_pipeline* p) { return p->device != RIP_DEVICE_NONE && !p->maps_on_host; }

void ensure_host_maps(rip_pipeline* p) {
  if (!p->map_dirty) return;
  const rip::Modules& m = p->m;
  if (m.dist_w <= 0 || m.dist_h <= 0) throw AssertError("undistortion: image size is not set");
  const size_t n = (size_t)m.dist_w * m.dist_h * 2;
  p->h_map.resize(n);
  // maps have the *dist* image size even after setNewImageSize (undistortion.cpp:216)
  if (maps_on_device(p)) {
    DeviceGuard device_guard(p->device);
    rip::FisheyeMapParams fp = {};
    std::memcpy(fp.K, m.dist_K, sizeof(fp.K));
    std::memcpy(fp.D, m.dist_D, sizeof(fp.D));
    rip::fisheye_inverse_PR(m.rect_K, m.dist_R, fp.iR);
    fp.w = m.dist_w;
    fp.h = m.dist_h;
    p->d_map.reserve(n * sizeof(float));
    fp.map_xy = p->d_map.as<float>();
    p->d_map_ckpt.reserve(rip::fisheye_ckpt_bytes(fp.w, fp.h));
    fp.ckpt = p->d_map_ckpt.as<double>();
    rip::launch_fisheye_maps(fp, p->stream);
    // no host copy yet: the remap-plan compiler runs on the device too; need_host_map() fetches the floats for
    // rip_get_undistortion_maps or for a plan compiled on the host
    p->map_dirty = false;
    p->map_uploaded = true;
    p->h_map_valid = false;
    p->plan.valid = false;
    return;
  }
  rip::fisheye_init_undistort_rectify_map(m.dist_K, m.dist_D, m.dist_R, m.rect_K, m.dist_w, m.dist_h, p->h_map.data());
  p->map_dirty = false;
  p->map_uploaded = false;
  p->h_map_valid = true;
  p->plan.valid = false;
}

// the maps as floats on the host
void need_host_map(rip_pipeline* p) {
  ensure_host_maps(p);
  if (p->h_map_valid) return;
  DeviceGuard device_guard(p->device);
  HIP_CHECK(hipMemcpyAsync(p->h_map.data(), p->d_map.ptr, p->h_map.size() * sizeof(float), hipMemcpyDeviceToHost, p->stream));
  HIP_CHECK(hipStreamSynchronize(p->stream));
  p->h_map_valid = true;
}

void ensure_maps(rip_pipeline* p);

static_assert(rip::kRemapOutside == rip::kPlanOutside && rip::kRemapBorder == rip::kPlanBorder, "plan sentinels");
static_assert(sizeof(rip::RemapTile) == sizeof(rip::RemapTileDesc), "tile descriptor layout");

void ensure_plan(rip_pipeline* p, int src_rows, int src_cols) {
  ensure_maps(p);
  const rip::Modules& m = p->m;
  if (!p->plan.valid || p->plan.src_rows != src_rows || p->plan.src_cols != src_cols || p->plan.drows != m.dist_h ||
      p->plan.dcols != m.dist_w) {
    p->plan_on_device = false;
    if (maps_on_device(p) && !p->plan_on_host) {
      // Compile the plan where the maps are (rip_maps.hip remap_plan_kernel: one workgroup per tile; the same words, tile
      // rectangles and border pixels as rip::compile_remap_plan, the border list in another order): no 8 B/px map read-back,
      // no 4 B/px plan upload, no host threads -- a calibration change costs the map kernel plus ~0.1 ms.
      rip::RemapPlan& pl = p->plan;
      pl = rip::RemapPlan();
      pl.drows = m.dist_h;
      pl.dcols = m.dist_w;
      pl.src_rows = src_rows;
      pl.src_cols = src_cols;
      pl.tiles_x = (pl.dcols + rip::kRemapTileW - 1) / rip::kRemapTileW;
      pl.tiles_y = (pl.drows + rip::kRemapTileH - 1) / rip::kRemapTileH;
      const size_t ntiles = (size_t)pl.tiles_x * pl.tiles_y;
      const unsigned border_cap = 1u << 20;  // pixels; more than that (a map that mostly straddles the border) goes to the host
      p->d_plan_words.reserve(ntiles * rip::kRemapTilePx * sizeof(uint32_t));
      p->d_plan_tiles.reserve(ntiles * sizeof(rip::RemapTileDesc));
      p->d_plan_border.reserve((size_t)border_cap * sizeof(uint32_t));
      p->d_plan_counters.reserve(4 * sizeof(unsigned));
      HIP_CHECK(hipMemsetAsync(p->d_plan_counters.ptr, 0, 4 * sizeof(unsigned), p->stream));
      rip::RemapPlanBuildParams bp = {};
      bp.map_xy = p->d_map.as<float>();
      bp.drows = pl.drows;
      bp.dcols = pl.dcols;
      bp.src_rows = src_rows;
      bp.src_cols = src_cols;
      bp.tiles_x = pl.tiles_x;
      bp.tiles_y = pl.tiles_y;
      bp.words = p->d_plan_words.as<uint32_t>();
      bp.tiles = p->d_plan_tiles.as<rip::RemapTileDesc>();
      bp.border = p->d_plan_border.as<uint32_t>();
      bp.border_cap = border_cap;
      bp.counters = p->d_plan_counters.as<unsigned>();
      rip::launch_remap_plan_build(bp, p->stream);
      unsigned counters[4] = {0, 0, 0, 0};
      HIP_CHECK(hipMemcpyAsync(counters, p->d_plan_counters.ptr, sizeof(counters), hipMemcpyDeviceToHost, p->stream));
      HIP_CHECK(hipStreamSynchronize(p->stream));
      if (counters[0] <= border_cap) {
        p->plan_n_border = (int)counters[0];
        pl.max_lds_bytes = counters[1];
        pl.max_rect_w = (int)counters[2];
        pl.max_rect_h = (int)counters[3];
        pl.valid = true;
        p->plan_on_device = true;
        p->plan_uploaded = true;
      }
    }
    if (!p->plan_on_device) {
      need_host_map(p);
      rip::compile_remap_plan(p->plan, p->h_map.data(), m.dist_h, m.dist_w, src_rows, src_cols);
      p->plan_n_border = (int)p->plan.border.size();
      p->plan_uploaded = false;
    }
  }
  if (!p->plan_uploaded) {
    p->d_plan_words.reserve(p->plan.words.size() * sizeof(uint32_t));
    p->d_plan_tiles.reserve(p->plan.tiles.size() * sizeof(rip::RemapTile));
    HIP_CHECK(hipMemcpyAsync(p->d_plan_words.ptr, p->plan.words.data(), p->plan.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                             p->stream));
    HIP_CHECK(hipMemcpyAsync(p->d_plan_tiles.ptr, p->plan.tiles.data(), p->plan.tiles.size() * sizeof(rip::RemapTile),
                             hipMemcpyHostToDevice, p->stream));
    p->d_plan_border.reserve(std::max<size_t>(4, p->plan.border.size() * sizeof(uint32_t)));
    if (!p->plan.border.empty())
      HIP_CHECK(hipMemcpyAsync(p->d_plan_border.ptr, p->plan.border.data(), p->plan.border.size() * sizeof(uint32_t),
                               hipMemcpyHostToDevice, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
    p->plan_uploaded = true;
  }
}

void ensure_maps(rip_pipeline* p) {
  ensure_host_maps(p);
  if (p->map_uploaded) return;
  p->d_map.reserve(p->h_map.size() * sizeof(float));
  HIP_CHECK(hipMemcpyAsync(p->d_map.ptr, p->h_map.data(), p->h_map.size() * sizeof(float), hipMemcpyHostToDevice, p->stream));
  HIP_CHECK(hipStreamSynchronize(p->stream));
  p->map_uploaded = true;
}

void ensure_tables(rip_pipeline* p) {
  if (!p->tabs_dirty) return;
  rip::DevTables& t = p->h_tabs;
  const rip::ColorTables& c = rip::color_tables();
  rip::build_gamma_lut(p->m.gamma_k, t.gamma_lut);
  for (int i = 0; i < 256; i++) t.lin_tab[i] = c.srgb_gamma[p->m.gamma_enabled ? t.gamma_lut[i] : i];
  std::memcpy(t.cbrt_tab, c.cbrt, sizeof(t.cbrt_tab));
  for (int i = 0; i < 256; i++) t.yf_tab[i] = (uint32_t)c.lab_to_yf[2 * i] | ((uint32_t)c.lab_to_yf[2 * i + 1] << 16);
  for (int i = 0; i < 4096; i++) t.inv_gamma[i] = (uint8_t)std::min<int>(255, c.inv_gamma[i]);
  std::memcpy(t.sdiv, c.sdiv, sizeof(t.sdiv));
  std::memcpy(t.hdiv, c.hdiv180, sizeof(t.hdiv));
  std::memcpy(t.lab_fwd, c.fwd, sizeof(t.lab_fwd));
  std::memcpy(t.lab_inv, c.inv, sizeof(t.lab_inv));
  for (int ch = 0; ch < 3; ch++) {
    if (c.inv[ch * 3] < -32768 || c.inv[ch * 3] > 32767 || c.inv[ch * 3 + 1] < -32768 || c.inv[ch * 3 + 1] > 32767)
      throw std::runtime_error("Lab inverse coefficients do not fit 16 bits");
    t.lab_inv_pk[ch * 2] = (int32_t)(((uint32_t)c.inv[ch * 3] & 0xffffu) | ((uint32_t)c.inv[ch * 3 + 1] << 16));
    t.lab_inv_pk[ch * 2 + 1] = c.inv[ch * 3 + 2];
  }
  std::vector<float> accum;
  rip::ccc_build_scalar_tables(t.log_tab, accum, t.exp_neg_tab);
  rip::fft256_twiddles(t.tw_re, t.tw_im);
  p->d_tabs.reserve(sizeof(rip::DevTables));
  HIP_CHECK(hipMemcpyAsync(p->d_tabs.ptr, &t, sizeof(t), hipMemcpyHostToDevice, p->stream));
  p->d_vig_image.reserve(rip::vig_image_bytes());
  rip::launch_vig_image(p->d_tabs.as<rip::DevTables>(), p->d_vig_image.as<uint32_t>(), p->stream);
  if (!p->d_accum.ptr) {
    p->d_accum.reserve(accum.size() * sizeof(float));
    HIP_CHECK(hipMemcpyAsync(p->d_accum.ptr, accum.data(), accum.size() * sizeof(float), hipMemcpyHostToDevice, p->stream));
  }
  HIP_CHECK(hipStreamSynchronize(p->stream));  // host staging buffers go out of scope
  p->tabs_dirty = false;
}

void ensure_vignette(rip_pipeline* p, int rows, int cols) {
  if (!p->vig_dirty && p->vig_rows == rows && p->vig_cols == cols) return;
  rip::build_vignette_mask(rows, cols, p->m.vig_scale, p->m.vig_a2, p->m.vig_a4, p->h_vig, p->fp_contract);
  p->d_vig.reserve(p->h_vig.size() * sizeof(float));
  HIP_CHECK(hipMemcpyAsync(p->d_vig.ptr, p->h_vig.data(), p->h_vig.size() * sizeof(float), hipMemcpyHostToDevice, p->stream));
  HIP_CHECK(hipStreamSynchronize(p->stream));
  p->vig_rows = rows;
  p->vig_cols = cols;
  p->vig_dirty = false;
}

void ensure_ccc(rip_pipeline* p, int rows, int cols) {
  if (!p->ccc.loaded) {
    if (!p->ccc_model_env.empty()) {
      if (!rip::ccc_load_model_file(p->ccc, p->ccc_model_env)) throw InvalidArgument("RIP_CCC_MODEL: cannot read " + p->ccc_model_env);
      p->ccc_uploaded = false;
    } else {
      throw InvalidArgument(
          "white balance method [ccc] needs a model: call rip_load_ccc_model()/rip_set_ccc_model() or set RIP_CCC_MODEL "
          "(the reference loads raw_image_pipeline_white_balance/model/default.bin)");
    }
  }
  if (!p->ccc_uploaded) {
    size_t bytes = 65536 * 2 * sizeof(float);
    p->d_filter_fft.reserve(bytes);
    p->d_bias_fft.reserve(bytes);
    HIP_CHECK(hipMemcpyAsync(p->d_filter_fft.ptr, p->ccc.filter_fft.data(), bytes, hipMemcpyHostToDevice, p->stream));
    HIP_CHECK(hipMemcpyAsync(p->d_bias_fft.ptr, p->ccc.bias_fft.data(), bytes, hipMemcpyHostToDevice, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
    p->ccc_uploaded = true;
  }
  if (!p->ccc_state_init) {
    rip::CccState s = {};
    s.first_frame = 1;
    s.uv_x = s.uv_y = 128;  // uv_pos_ = (height/2, width/2), :178
    s.st_x = s.st_y = 128.f;
    s.kf_h = (float)p->kf_h;
    s.kf_r = (float)p->kf_r;
    s.temporal = p->m.wb_temporal ? 1 : 0;
    p->d_ccc_state.reserve(sizeof(s));
    HIP_CHECK(hipMemcpyAsync(p->d_ccc_state.ptr, &s, sizeof(s), hipMemcpyHostToDevice, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
    p->ccc_state_init = true;
    p->ccc_reset_pending = false;
    p->ccc_cfg_dirty = false;
  }
  if (p->ccc_reset_pending || p->ccc_cfg_dirty) {
    // patch individual fields, stream-ordered, keeping the filter state
    rip::CccState* d = p->d_ccc_state.as<rip::CccState>();
    if (p->ccc_reset_pending) {
      static const int one = 1;
      HIP_CHECK(hipMemcpyAsync(&d->first_frame, &one, sizeof(int), hipMemcpyHostToDevice, p->stream));
    }
    float hr[2] = {(float)p->kf_h, (float)p->kf_r};
    int temporal = p->m.wb_temporal ? 1 : 0;
    HIP_CHECK(hipMemcpyAsync(&d->kf_h, hr, sizeof(hr), hipMemcpyHostToDevice, p->stream));
    HIP_CHECK(hipMemcpyAsync(&d->temporal, &temporal, sizeof(int), hipMemcpyHostToDevice, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
    p->ccc_reset_pending = false;
    p->ccc_cfg_dirty = false;
  }
  if (p->geom_rows != rows || p->geom_cols != cols) {
    // cv::resize(src, small, Size(360,270)) coefficient tables (imgproc/resize.cpp)
    struct Geom {
      int xofs[360];
      short ialpha[720];
      int yofs[540];
      short ibeta[540];
      int area_fast;
    };
    static_assert(sizeof(Geom) % 4 == 0, "geom");
    std::vector<uint8_t> raw(sizeof(Geom));
    Geom& g = *reinterpret_cast<Geom*>(raw.data());
    double scale_x = (double)cols / 360, scale_y = (double)rows / 270;
    int isx = (int)std::lrint(scale_x), isy = (int)std::lrint(scale_y);
    g.area_fast = (std::fabs(scale_x - isx) < 2.220446049250313e-16 && std::fabs(scale_y - isy) < 2.220446049250313e-16 && isx == 2 && isy == 2) ? 1 : 0;
    auto sat16 = [](int v) { return (short)std::min(32767, std::max(-32768, v)); };
    for (int dx = 0; dx < 360; dx++) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = (int)std::floor(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
      g.xofs[dx] = sx;
      g.ialpha[2 * dx] = sat16((int)std::lrintf((1.f - fx) * 2048));
      g.ialpha[2 * dx + 1] = sat16((int)std::lrintf(fx * 2048));
    }
    for (int dy = 0; dy < 270; dy++) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = (int)std::floor(fy);
      fy -= sy;
      g.ibeta[2 * dy] = sat16((int)std::lrintf((1.f - fy) * 2048));
      g.ibeta[2 * dy + 1] = sat16((int)std::lrintf(fy * 2048));
      g.yofs[2 * dy] = std::min(std::max(sy, 0), rows - 1);
      g.yofs[2 * dy + 1] = std::min(std::max(sy + 1, 0), rows - 1);
    }
    p->d_geom.reserve(sizeof(Geom));
    HIP_CHECK(hipMemcpyAsync(p->d_geom.ptr, raw.data(), sizeof(Geom), hipMemcpyHostToDevice, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
    p->geom_rows = rows;
    p->geom_cols = cols;
  }
}

// ------------------------------------------------------------------------------------------------
// planning: raw_image_pipeline.hpp:143-172 stage gating
// ------------------------------------------------------------------------------------------------
Plan make_plan(const rip_pipeline* p, int rows, int cols, int channels, const std::string& encoding) {
  const rip::Modules& m = p->m;
  Plan pl;
  if (rows < 1 || cols < 1) throw AssertError("empty image");
  // the kernels address one frame with 32-bit byte offsets and 24-bit row multiplies
  if (cols > (1 << 22) || rows > (1 << 22) || (unsigned long long)rows * cols * 3ull >= (1ull << 32))
    throw InvalidArgument("image too large: a frame must stay below 4 GiB and 4 Mpx per side");
  pl.encoding_out = encoding;
  if (parse_bayer(encoding, pl.ry, pl.rx)) {
    if (channels != 1) throw AssertError("cv::demosaicing: Bayer input must have one channel");
    if (rows < 3 || cols < 3) throw AssertError("cv::demosaicing: image too small");
    pl.src_kind = rip::SRC_BAYER;
    pl.channels = 3;
    pl.encoding_out = "bgr8";
  } else if (is_bayer16(encoding)) {
    // debayer.cpp:76-78 throws for these names; rip_set_debayer_16bit(1) opts into the extension instead
    if (!m.debayer_16bit) throw InvalidArgument("Encoding [" + encoding + "] is a valid pattern but is not supported!");
    if (channels != 1) throw AssertError("cv::demosaicing: Bayer input must have one channel");
    if (rows < 3 || cols < 3) throw AssertError("cv::demosaicing: image too small");
    std::string e8 = encoding.substr(0, encoding.size() - 2) + "8";
    parse_bayer(e8, pl.ry, pl.rx);
    pl.src_kind = rip::SRC_BAYER;
    pl.elem_bytes = 2;
    pl.channels = 3;
    pl.encoding_out = "bgr16";
  } else if (encoding == "rgb8") {
    if (channels != 3) throw AssertError("cvtColor(RGB2BGR): rgb8 input must have three channels");
    pl.src_kind = rip::SRC_RGB;  // swapped to BGR; the encoding string stays "rgb8" (debayer.cpp:72-73)
    pl.channels = 3;
  } else if (channels == 3) {
    pl.src_kind = rip::SRC_BGR;
    pl.channels = 3;
  } else if (channels == 1) {
    pl.src_kind = rip::SRC_MONO;
    pl.channels = 1;
  } else {
    throw InvalidArgument("images with " + std::to_string(channels) + " channels are not supported");
  }
  pl.flip_angle = (m.flip_enabled && (m.flip_angle == 90 || m.flip_angle == 180 || m.flip_angle == 270)) ? m.flip_angle : 0;
  const bool swap = pl.flip_angle == 90 || pl.flip_angle == 270;
  pl.mid_rows = swap ? cols : rows;
  pl.mid_cols = swap ? rows : cols;
  if (m.wb_enabled && pl.channels == 3) {
    const std::string& w = m.wb_method;
    if (w == "gray_world" || w == "grey_world")
      pl.wb_mode = rip::WB_Q8;
    else if (w == "ccc")
      pl.wb_mode = rip::WB_FLOAT;
    else if (w == "pca")
      pl.wb_mode = rip::WB_PCA;
    else if (w == "simple")
      pl.wb_mode = rip::WB_SIMPLE;
    else if (w == "learned")
      throw InvalidArgument("White Balance method [learned] (cv::xphoto::LearningBasedWB, a model compiled into opencv_contrib) is not implemented by the MI355X pipeline; use 'simple', 'gray_world', 'ccc' or 'pca'");
    else
      throw InvalidArgument("White Balance method [" + w + "] not supported. Supported algorithms: 'simple', 'gray_world', 'learned', 'ccc', 'pca'");
  }
  if (m.cc_enabled && pl.channels == 3 && m.cc_available) pl.stage_bits |= rip::ST_CC;
  if (m.gamma_enabled) pl.stage_bits |= rip::ST_GAMMA;
  if (m.vig_enabled) {
    if (pl.channels != 3) throw AssertError("cvtColor(BGR2Lab): vignetting correction needs a 3-channel image");
    pl.stage_bits |= rip::ST_VIG;
  }
  if (m.ce_enabled && pl.channels == 3) pl.stage_bits |= rip::ST_HSV;
  pl.remap = m.und_enabled && m.und_available && m.dist_model != "none";
  if (pl.elem_bytes == 2 && (pl.wb_mode != rip::WB_NONE || pl.stage_bits != 0 || pl.remap))
    // every later module of the reference works on 8-bit images (cv::LUT, xphoto white balance, 8-bit Lab / HSV tables)
    // and would assert on CV_16UC3
    throw AssertError("16-bit Bayer frames go through debayer and flip only: disable white balance, colour calibration, gamma, "
                      "vignetting, colour enhancer and undistortion (they are 8-bit stages)");
  pl.out_rows = pl.remap ? m.dist_h : pl.mid_rows;
  pl.out_cols = pl.remap ? m.dist_w : pl.mid_cols;
  return pl;
}

// Enqueues the whole chain for n frames.  d_out rows of out_step bytes.  Taps may be null.
// reuse_wb: the white-balance gains of the previous launch (same frames) are applied again and no estimator runs -- the
// debug stage dumps re-run prefixes of the chain without advancing the ccc Kalman state.
void run_batch(rip_pipeline* p, const Plan& pl, const uint8_t* d_in, size_t in_step, size_t in_frame_stride, int n, int rows,
               int cols, uint8_t* d_out, size_t out_step, size_t out_frame_stride, uint8_t* d_tap_deb, uint8_t* d_tap_col,
               bool reuse_wb = false) {
  p->work_enqueued = true;  // from here on something may sit on p->stream
  DeviceGuard device_guard(p->device);
  if (pl.elem_bytes == 2) {  // 16-bit Bayer extension: one kernel, no taps
    rip::Debayer16Params d = {};
    d.src = d_in;
    d.src_step = in_step;
    d.src_frame_stride = in_frame_stride;
    d.rows = rows;
    d.cols = cols;
    d.bayer_ry = pl.ry;
    d.bayer_rx = pl.rx;
    d.dst = d_out;
    d.dst_step = out_step ? out_step : (size_t)pl.out_cols * 6;
    d.dst_frame_stride = out_frame_stride ? out_frame_stride : d.dst_step * pl.out_rows;
    d.drows = pl.out_rows;
    d.dcols = pl.out_cols;
    d.flip_angle = pl.flip_angle;
    d.n_frames = n;
    ProfScope ps(p, RIP_KERNEL_CHAIN, p->stream);
    rip::launch_debayer16(d, p->stream);
    hipError_t le16 = hipGetLastError();
    if (le16 != hipSuccess) throw DeviceError(std::string("kernel launch failed: ") + hipGetErrorString(le16));
    p->last_batch_frames = 0;  // no white balance ran: rip_get_white_balance_info must not hand out an earlier batch's gains
    return;
  }
  ensure_tables(p);
  const size_t tap_pitch = (size_t)pl.mid_cols * pl.channels;  // taps are tightly packed API outputs
  const size_t tap_frame = tap_pitch * pl.mid_rows;
  // the internal pre-undistortion image uses 16-byte aligned rows (the tiled remap stages it with
  // aligned 16-byte loads); when that image is an API output the caller's tight pitch is used
  const size_t mid_pitch = d_tap_col ? tap_pitch : ((tap_pitch + 15) & ~(size_t)15);
  const size_t mid_frame = mid_pitch * pl.mid_rows;
  if (out_step == 0) out_step = (size_t)pl.out_cols * pl.channels;
  if (out_frame_stride == 0) out_frame_stride = out_step * pl.out_rows;

  // ---- everything that allocates, uploads or synchronises happens before the first launch -------------
  p->d_wb.reserve(sizeof(rip::FrameWb) * (size_t)n);
  const bool sums = pl.wb_mode == rip::WB_Q8 || pl.wb_mode == rip::WB_PCA || pl.wb_mode == rip::WB_SIMPLE;
  if (reuse_wb) {
    if (p->last_batch_frames != n) throw DeviceError("internal: white-balance gains of another batch");
  } else if (sums) {
    p->d_stats.reserve(sizeof(rip::FrameStats) * (size_t)n);
    if (pl.wb_mode == rip::WB_SIMPLE) p->d_hist.reserve((size_t)n * 768 * sizeof(unsigned));
  } else if (pl.wb_mode == rip::WB_FLOAT) {
    ensure_ccc(p, pl.mid_rows, pl.mid_cols);
    p->d_hist.reserve((size_t)n * rip::ccc_hist_split(n) * 65536 * sizeof(unsigned));
    p->d_work.reserve((size_t)n * 65536 * 2 * sizeof(float));
    p->d_rowbest.reserve((size_t)n * 256 * 2 * sizeof(float));
    p->d_argmax.reserve((size_t)n * 2 * sizeof(int));
  }
  uint8_t* chain_dst = d_out;
  size_t chain_step = out_step, chain_stride = out_frame_stride;
  bool tiled = false, fused = false, direct = false, mono_direct = false;
  // the remap's view of one group of frames: plan, destination, and -- `src` -- either the intermediate image or, on the
  // fused path, the Bayer frames themselves
  auto tiled_params = [&](const uint8_t* src, size_t src_step, size_t src_frame_stride, int src_rows, int src_cols, int f0, int ng) {
    rip::RemapTiledParams tp = {};
    rip::RemapParams& r = tp.base;
    r.src = src;
    r.src_step = src_step;
    r.src_frame_stride = src_frame_stride;
    r.rows = src_rows;
    r.cols = src_cols;
    r.channels = pl.channels;
    r.map_xy = p->d_map.as<float>();
    r.dst = d_out + (size_t)f0 * out_frame_stride;
    r.dst_step = out_step;
    r.dst_frame_stride = out_frame_stride;
    r.drows = pl.out_rows;
    r.dcols = pl.out_cols;
    r.n_frames = ng;
    tp.words = p->d_plan_words.as<uint32_t>();
    tp.tiles = p->d_plan_tiles.as<rip::RemapTileDesc>();
    tp.tiles_x = p->plan.tiles_x;
    tp.tiles_y = p->plan.tiles_y;
    tp.border_list = p->d_plan_border.as<uint32_t>();
    tp.n_border = p->plan_n_border;
    tp.lds_bytes = (unsigned)p->plan.max_lds_bytes;
    return tp;
  };
  auto mono_ops = [&](rip::RemapTiledParams tp) {
    tp.mono_lut = (pl.stage_bits & rip::ST_GAMMA) ? p->d_tabs.as<uint8_t>() + offsetof(rip::DevTables, gamma_lut) : nullptr;
    tp.mono_flip180 = pl.flip_angle == 180 ? 1 : 0;
    return tp;
  };
  // the chain's parameters for one group of frames (dst / taps filled in by the caller)
  auto chain_params = [&](const uint8_t* in_g, rip::FrameWb* wb_g, int ng) {
    rip::ChainParams c = {};
    c.src = in_g;
    c.src_step = in_step;
    c.src_frame_stride = in_frame_stride;
    c.rows = rows;
    c.cols = cols;
    c.src_kind = pl.src_kind;
    c.bayer_ry = pl.ry;
    c.bayer_rx = pl.rx;
    c.drows = pl.mid_rows;
    c.dcols = pl.mid_cols;
    c.channels = pl.channels;
    c.flip_angle = pl.flip_angle;
    c.n_frames = ng;
    c.wb_mode = pl.wb_mode;
    c.wb = wb_g;
    c.stage_bits = pl.stage_bits;
    for (int i = 0; i < 9; i++) c.cc_m[i] = p->m.cc_matrix[i];
    for (int i = 0; i < 3; i++) c.cc_bias[i] = (float)p->m.cc_bias[i];
    if (pl.stage_bits & rip::ST_VIG) c.vig_mask = p->d_vig.as<float>();
    // cv::Scalar(hue_gain_, saturation_gain_, value_gain_) on (H,S,V), color_enhancer.cpp:42
    c.hsv_gain[0] = (float)p->m.ce_hue_gain;
    c.hsv_gain[1] = (float)p->m.ce_saturation_gain;
    c.hsv_gain[2] = (float)p->m.ce_value_gain;
    c.tabs = p->d_tabs.as<rip::DevTables>();
    c.vig_image = p->d_vig_image.as<uint32_t>();
    c.fp_contract = p->fp_contract;
    return c;
  };
  if (pl.remap) {
    ensure_maps(p);
    if (p->use_tiled_remap && (pl.channels == 3 || pl.channels == 1)) {
      ensure_plan(p, pl.mid_rows, pl.mid_cols);
      tiled = true;
    }
    // Memory-rate stage sets with neither tap requested: the remap's tiles demosaic and colour their own source rectangles
    // out of the Bayer frames (rip_fused.hip) -- no intermediate image is written, read or even allocated.
    if (tiled && !d_tap_col && !d_tap_deb && pl.src_kind == rip::SRC_BAYER)
      fused = rip::launch_remap_fused(tiled_params(d_in, in_step, in_frame_stride, rows, cols, 0, n), chain_params(d_in, p->d_wb.as<rip::FrameWb>(), n),
                                      p->plan.max_rect_w, p->plan.max_rect_h, p->tn, p->stream, /*dry_run=*/true);
    // bgr8 / mono8 frames with nothing to do before the undistortion (no flip, no white balance, no stage, no tap): the
    // chain would be a copy -- the remap gathers from the caller's frames as they lie
    direct = !fused && !d_tap_col && !d_tap_deb && (pl.src_kind == rip::SRC_BGR || pl.src_kind == rip::SRC_MONO) && pl.flip_angle == 0 &&
             pl.wb_mode == rip::WB_NONE && pl.stage_bits == 0;
    // mono8: the whole chain is a 180-degree flip and the gamma table -- the ring kernel addresses the mirrored rectangle
    // and maps the taps through the table as it gathers them
    if (!direct && !fused && tiled && !d_tap_col && !d_tap_deb && pl.src_kind == rip::SRC_MONO && (pl.flip_angle == 0 || pl.flip_angle == 180) &&
        p->tn.remap_fused) {
      rip::RemapTiledParams tp = mono_ops(tiled_params(d_in, in_step, in_frame_stride, rows, cols, 0, n));
      direct = mono_direct = rip::launch_remap_tiled(tp, p->tn, p->stream, /*dry_run=*/true);
    }
    if (fused || direct) {
    } else if (d_tap_col) {  // the pre-undistortion image is an API output: write it once, gather from it
      chain_dst = d_tap_col;
    } else {
      p->d_mid.reserve(mid_frame * (size_t)n);
      chain_dst = p->d_mid.as<uint8_t>();
    }
    chain_step = mid_pitch;
    chain_stride = mid_frame;
  }
  if (pl.stage_bits & rip::ST_VIG) ensure_vignette(p, pl.mid_rows, pl.mid_cols);

  // ---- frame groups -------------------------------------------------------------------------------------
  // Optional (tunable overlap_groups > 1; OFF by default): with the batch cut into G groups of frames, remap(g) runs on the
  // handle's internal stream beside stats(g + 1) and chain(g + 1) on the caller's (overlap_mode 1), or beside stats(g + 1)
  // only (mode 2: the chain waits for the remap).  raw_image_pipeline.hpp:143-172 only orders the stages of ONE frame, and
  // everything that carries state from frame to frame -- the ccc Kalman filter -- stays on the caller's stream in frame order;
  // the caller's stream waits for the internal one before this function returns, so the batch is complete in stream order
  // exactly as without the split.  Measured on config2 (256 frames, one box, round 3): 4.87 ms per step unsplit; mode 1 with
  // 2 / 4 / 8 / 16 groups 4.98 / 5.00 / 5.09 / 5.70; mode 2 with 2 / 4 groups 4.94 / 5.05 -- the three kernels lean on the
  // same VALU issue slots and LDS, and the shorter launches pay their tails (docs/experiments_r1-3.md), so the default stays 1.
  bool stats_cleared_here = false;
  const size_t stats_clean_before = (p->stats_clean_ptr == p->d_stats.ptr && p->stats_clean_cap == p->d_stats.cap) ? p->stats_clean_bytes : 0;
  int groups = 1;
  if (pl.remap && !reuse_wb && p->tn.overlap_groups > 1) groups = std::min(p->tn.overlap_groups, n);
  hipStream_t front = p->stream, back = p->stream;
  if (groups > 1) {
    if (!p->aux_stream) HIP_CHECK(hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
    while (p->ovl_events.size() < 2 * (size_t)groups + 1) {
      hipEvent_t e;
      HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      p->ovl_events.push_back(e);
    }
    back = p->aux_stream;
  }
  const int per_group = (n + groups - 1) / groups;
  // Whatever was enqueued on the internal stream is joined into the caller's stream when this function is left -- also by
  // an exception: the batch is complete, in the caller's stream order, once the last remap is.
  struct Join {
    rip_pipeline* p;
    int slot;
    bool used = false;
    ~Join() {
      if (!used) return;
      (void)hipEventRecord(p->ovl_events[slot], p->aux_stream);
      (void)hipStreamWaitEvent(p->stream, p->ovl_events[slot], 0);
    }
  } join{p, 2 * groups};
  bool& back_used = join.used;
  for (int g = 0; g < groups; g++) {
    const int f0 = g * per_group, ng = std::min(per_group, n - f0);
    if (ng <= 0) break;
    const uint8_t* in_g = d_in + (size_t)f0 * in_frame_stride;
    rip::FrameWb* wb_g = p->d_wb.as<rip::FrameWb>() + f0;
    // ---- white-balance statistics -------------------------------------------------------------
    if (reuse_wb) {
    } else if (sums) {
      rip::FrameStats* stats_g = p->d_stats.as<rip::FrameStats>() + f0;
      unsigned* hist_g = pl.wb_mode == rip::WB_SIMPLE ? p->d_hist.as<unsigned>() + (size_t)f0 * 768 : nullptr;
      if (hist_g) p->hist_clean_bytes = 0;  // the same buffer serves SimpleWB's histograms
      // grey-world / pca: the statistics kernel itself finishes a frame (gains written by the workgroup that ends last) and
      // hands its FrameStats back zeroed, so neither a memset nor a finalisation launch separates the batches -- two
      // dependent launches less on the single-frame path.  The records are cleared here only when they are not known to be
      // clean: fresh memory, or a batch that did not run to its end.
      const bool fused_finalize = pl.wb_mode != rip::WB_SIMPLE;
      const size_t stats_bytes = sizeof(rip::FrameStats) * (size_t)n;
      if (!fused_finalize || stats_clean_before < stats_bytes) {
        HIP_CHECK(hipMemsetAsync(stats_g, 0, sizeof(rip::FrameStats) * (size_t)ng, front));
        stats_cleared_here = true;
      }
      p->stats_clean_bytes = 0;  // until this batch has been enqueued completely
      if (hist_g) HIP_CHECK(hipMemsetAsync(hist_g, 0, (size_t)ng * 768 * sizeof(unsigned), front));
      rip::StatsParams sp = {};
      sp.src = in_g;
      sp.src_step = in_step;
      sp.src_frame_stride = in_frame_stride;
      sp.rows = rows;
      sp.cols = cols;
      sp.src_kind = pl.src_kind;
      sp.bayer_ry = pl.ry;
      sp.bayer_rx = pl.rx;
      sp.n_frames = ng;
      sp.mode = pl.wb_mode;
      sp.thresh255 = (unsigned)(uint16_t)std::lrintf((float)p->m.wb_bright_thr * 255);
      sp.stats = stats_g;
      sp.hist3 = hist_g;
      sp.wb_out = fused_finalize ? wb_g : nullptr;
      {
        ProfScope ps(p, RIP_KERNEL_STATS, front);
        rip::launch_stats(sp, p->tn, front);
      }
      // SimpleWB::setP(clipping_percentile_) (white_balance.cpp:55); total = pixels per channel plane
      if (!fused_finalize)
        rip::launch_wb_finalize(pl.wb_mode, sp.stats, nullptr, nullptr, p->d_tabs.as<rip::DevTables>(), wb_g, ng, front, sp.hist3,
                                (float)p->m.wb_percentile, rows * cols);
    } else if (pl.wb_mode == rip::WB_FLOAT) {
      rip::CccParams cp = {};  // the launcher zeroes the histogram when its kernel accumulates in HBM
      cp.src = in_g;
      cp.src_step = in_step;
      cp.src_frame_stride = in_frame_stride;
      cp.rows = rows;
      cp.cols = cols;
      cp.src_kind = pl.src_kind;
      cp.bayer_ry = pl.ry;
      cp.bayer_rx = pl.rx;
      cp.flip_angle = pl.flip_angle;
      cp.drows = pl.mid_rows;
      cp.dcols = pl.mid_cols;
      cp.n_frames = ng;
      const uint8_t* gm = p->d_geom.as<uint8_t>();
      cp.geom.xofs = reinterpret_cast<const int*>(gm);
      cp.geom.ialpha = reinterpret_cast<const short*>(gm + 360 * 4);
      cp.geom.yofs = reinterpret_cast<const int*>(gm + 360 * 4 + 720 * 2);
      cp.geom.ibeta = reinterpret_cast<const short*>(gm + 360 * 4 + 720 * 2 + 540 * 4);
      cp.geom.area_fast = ((double)pl.mid_cols / 360 == 2.0 && (double)pl.mid_rows / 270 == 2.0) ? 1 : 0;
      // setSaturationThreshold(float, float): thresholds are held as float (:437-440); 255 * thr in float
      cp.upper = 255 * (float)p->m.wb_bright_thr;
      cp.lower = 255 * (float)p->m.wb_dark_thr;
      cp.hist_split = rip::ccc_hist_split(n);  // of the whole batch: what d_hist was sized for
      cp.hist_counts = p->d_hist.as<unsigned>() + (size_t)f0 * cp.hist_split * 65536;
      cp.accum_tab = p->d_accum.as<float>();
      cp.work = p->d_work.as<float>() + (size_t)f0 * 65536 * 2;
      cp.filter_fft = p->d_filter_fft.as<float>();
      cp.bias_fft = p->d_bias_fft.as<float>();
      cp.row_best = p->d_rowbest.as<float>() + (size_t)f0 * 256 * 2;
      cp.argmax = p->d_argmax.as<int>() + (size_t)f0 * 2;
      cp.tabs = p->d_tabs.as<rip::DevTables>();
      const size_t hist_bytes = (size_t)ng * 65536 * sizeof(unsigned);  // what the global-atomic kernel accumulates into
      cp.hist_is_clean = (groups == 1 && p->hist_clean_ptr == p->d_hist.ptr && p->hist_clean_cap == p->d_hist.cap && p->hist_clean_bytes >= hist_bytes) ? 1 : 0;
      p->hist_clean_bytes = 0;  // until the estimator has been enqueued completely
      bool estimated;
      int left_clean = 0;
      {
        ProfScope ps(p, RIP_KERNEL_CCC, front);
        estimated = rip::launch_ccc_estimate(cp, p->tn, front, &left_clean);
      }
      // no histogram, no estimate: fail before the finalisation advances the persistent Kalman state on stale data
      if (!estimated) throw DeviceError("ccc white balance: a kernel of the estimator could not be launched");
      if (left_clean && groups == 1) {
        p->hist_clean_ptr = p->d_hist.ptr;
        p->hist_clean_cap = p->d_hist.cap;
        p->hist_clean_bytes = hist_bytes;
      }
      const bool inline_argmax = rip::ccc_argmax_in_finalize(ng);
      rip::launch_wb_finalize(rip::WB_FLOAT, nullptr, cp.argmax, p->d_ccc_state.as<rip::CccState>(), cp.tabs, wb_g, ng, front, nullptr, 0.f, 0,
                              inline_argmax ? cp.row_best : nullptr, inline_argmax ? cp.argmax : nullptr);
    }

    // ---- chain + remap in one kernel (memory-rate stage sets, no taps) ---------------------------------
    if (fused) {
      ProfScope ps(p, RIP_KERNEL_REMAP, front);
      rip::ChainParams fc = chain_params(in_g, wb_g, ng);
      fc.dst_streaming = (n >= 8 && p->tn.chain_nt != 0) ? 1 : 0;  // the kernel's output is the batch's final image: non-temporal stores for batches
      if (!rip::launch_remap_fused(tiled_params(in_g, in_step, in_frame_stride, rows, cols, f0, ng), fc, p->plan.max_rect_w,
                                   p->plan.max_rect_h, p->tn, front, /*dry_run=*/false))
        throw DeviceError("internal: the fused remap refused a geometry it had accepted");
      continue;
    }
    // ---- fused chain -----------------------------------------------------------------------------
    rip::ChainParams c = chain_params(in_g, wb_g, ng);
    if (direct) {  // no chain at all: the remap below reads the input frames
      ProfScope ps(p, RIP_KERNEL_REMAP, front);
      rip::RemapTiledParams tp = tiled_params(in_g, in_step, in_frame_stride, rows, cols, f0, ng);
      if (mono_direct) {
        if (!rip::launch_remap_tiled(mono_ops(tp), p->tn, front)) throw DeviceError("internal: the ring remap refused a geometry it had accepted");
        continue;
      }
      if (!(tiled && rip::launch_remap_tiled(tp, p->tn, front)) && !rip::launch_remap(tp.base, front))
        throw InvalidArgument("undistortion: frame geometry exceeds the kernels' 32-bit addressing");
      continue;
    }
    c.dst = chain_dst + (size_t)f0 * chain_stride;
    c.dst_step = chain_step;
    c.dst_frame_stride = chain_stride;
    // Non-temporal stores for batches: the image is far larger than the L2s, so lines the chain leaves there only get in the
    // way.  Rounds 3-4 kept them for images no kernel of the batch reads again (debayer-only, 256 frames: 1.08 against 1.16
    // ms; the remap of that time lost 19 % behind them); with the LDS-DMA ring remap it is the other way round (round 5, config 2:
    // remap 1.94-1.97 -> 1.83-1.85 ms behind a chain that stores non-temporally) -- tunable chain_nt
    c.dst_streaming = (n >= 8 && p->tn.chain_nt != 0 && (!pl.remap || p->tn.chain_nt < 0)) ? 1 : 0;
    c.tap = d_tap_deb ? d_tap_deb + (size_t)f0 * tap_frame : nullptr;
    c.tap_frame_stride = tap_frame;
    c.deal = pl.remap ? -1 : 0;  // hint for launch_chain: the remap gathers from this image next (Tunables::chain_deal)
    // overlap_mode 2: only the statistics of this group share the chip with the remap of the previous one; the chain waits
    if (back != front && p->tn.overlap_mode == 2 && g > 0) HIP_CHECK(hipStreamWaitEvent(front, p->ovl_events[groups + g - 1], 0));
    {
      ProfScope ps(p, RIP_KERNEL_CHAIN, front);
      rip::launch_chain(c, p->tn, front);
    }
    if (!pl.remap && d_tap_col) {
      // pre-undistortion copy == final image when no remap follows
      for (int f = f0; f < f0 + ng; f++)
        HIP_CHECK(hipMemcpy2DAsync(d_tap_col + (size_t)f * tap_frame, tap_pitch, d_out + (size_t)f * out_frame_stride, out_step, tap_pitch,
                                   (size_t)pl.mid_rows, hipMemcpyDeviceToDevice, front));
    }
    // ---- remap -----------------------------------------------------------------------------------
    if (pl.remap) {
      if (back != front) {  // remap(g) starts when chain(g) is done; the caller's stream goes on with group g + 1
        HIP_CHECK(hipEventRecord(p->ovl_events[g], front));
        HIP_CHECK(hipStreamWaitEvent(back, p->ovl_events[g], 0));
        back_used = true;
      }
      rip::RemapTiledParams tp = tiled_params(c.dst, mid_pitch, mid_frame, pl.mid_rows, pl.mid_cols, f0, ng);
      const rip::RemapParams& r = tp.base;
      bool done = false;
      if (tiled) {
        ProfScope ps(p, RIP_KERNEL_REMAP, back);
        done = rip::launch_remap_tiled(tp, p->tn, back);
      }
      if (!done) {
        ProfScope ps(p, RIP_KERNEL_REMAP, back);
        if (!rip::launch_remap(r, back)) throw InvalidArgument("undistortion: frame geometry exceeds the kernels' 32-bit addressing");
      }
      if (back != front && p->tn.overlap_mode == 2) HIP_CHECK(hipEventRecord(p->ovl_events[groups + g], back));
    }
  }
  p->last_batch_frames = n;
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) throw DeviceError(std::string("kernel launch failed: ") + hipGetErrorString(le));
  if (sums && !reuse_wb && pl.wb_mode != rip::WB_SIMPLE) {  // every statistics launch went out: its records come back zeroed
    p->stats_clean_ptr = p->d_stats.ptr;
    p->stats_clean_cap = p->d_stats.cap;
    p->stats_clean_bytes = std::max(stats_clean_before, sizeof(rip::FrameStats) * (size_t)n);
  }
  (void)stats_cleared_here;
}

// setDebug(true): raw_image_pipeline.hpp:143-172 writes the image after EVERY module -- enabled or not -- to
// /tmp/0N_<module>.png through saveDebugImage (:179-186: copy, cv::normalize(0, 255, NORM_MINMAX), cv::imwrite).  The modules
// are one fused kernel here, so the image after module k is produced by running the chain once more with the modules after k
// switched off (same input frame still in d_in, same white-balance gains: reuse_wb).  RIP_DEBUG_DIR replaces /tmp.
void write_debug_dumps(rip_pipeline* p, const Plan& pl, size_t in_pitch, size_t in_bytes, int rows, int cols, const uint8_t* final_image) {
  static const char* const kNames[8] = {"00_debayer", "01_flip", "02_white_balancing", "03_color_calibration", "04_gamma_correction",
                                        "05_vignetting_correction", "06_color_enhancer", "07_undistortion"};
  static const int kStages[8] = {0, 0, 0, rip::ST_CC, rip::ST_CC | rip::ST_GAMMA, rip::ST_CC | rip::ST_GAMMA | rip::ST_VIG,
                                 rip::ST_CC | rip::ST_GAMMA | rip::ST_VIG | rip::ST_HSV, rip::ST_CC | rip::ST_GAMMA | rip::ST_VIG | rip::ST_HSV};
  const std::string& dir = p->debug_dir;
  std::vector<uint8_t> host;
  // the re-runs below are not launches of the caller's frame: keep them out of an active rip_profile_begin/end session
  // (they would skew its per-class averages and use up its event slots)
  struct ProfPause {
    rip_pipeline* p;
    bool was;
    explicit ProfPause(rip_pipeline* pp) : p(pp), was(pp->prof_on) { p->prof_on = false; }
    ~ProfPause() { p->prof_on = was; }
  } prof_pause(p);
  std::string failed;
  for (int k = 0; k < 8; k++) {
    int r, c;
    if (k == 7) {  // after the undistortion module: the output of this call
      r = pl.out_rows;
      c = pl.out_cols;
      host.assign(final_image, final_image + (size_t)r * c * pl.channels);
    } else {
      Plan s = pl;
      s.remap = false;
      if (k < 1) s.flip_angle = 0;
      const bool swap = s.flip_angle == 90 || s.flip_angle == 270;
      s.mid_rows = swap ? cols : rows;
      s.mid_cols = swap ? rows : cols;
      if (k < 2) s.wb_mode = rip::WB_NONE;
      s.stage_bits &= kStages[k];
      s.out_rows = r = s.mid_rows;
      s.out_cols = c = s.mid_cols;
      const size_t bytes = (size_t)r * c * s.channels;
      p->d_dbg.reserve(bytes);
      run_batch(p, s, p->d_in.as<uint8_t>(), in_pitch, in_bytes, 1, rows, cols, p->d_dbg.as<uint8_t>(), 0, 0, nullptr, nullptr, true);
      host.resize(bytes);
      HIP_CHECK(hipMemcpyAsync(host.data(), p->d_dbg.ptr, bytes, hipMemcpyDeviceToHost, p->stream));
      HIP_CHECK(hipStreamSynchronize(p->stream));
    }
    rip::normalize_minmax_u8(host.data(), host.size());
    const std::string path = dir + "/" + kNames[k] + ".png";
    if (!rip::write_png(path, host.data(), r, c, pl.channels)) {
      std::fprintf(stderr, "raw_image_pipeline: could not write %s\n", path.c_str());
      failed += (failed.empty() ? "" : ", ") + path;
    }
  }
  // cv::imwrite's failure does not fail apply() in the reference either; the message stays readable through rip_last_error()
  if (!failed.empty()) p->last_error = "debug dumps not written: " + failed;
}

template <typename F>
rip_status guarded(const rip_pipeline* p, F&& fn) {
  try {
    fn();
    return RIP_OK;
  } catch (const InvalidArgument& e) {
    if (p) p->last_error = e.what();
    return RIP_ERR_INVALID_ARGUMENT;
  } catch (const std::invalid_argument& e) {
    if (p) p->last_error = e.what();
    return RIP_ERR_INVALID_ARGUMENT;
  } catch (const AssertError& e) {
    if (p) p->last_error = e.what();
    return RIP_ERR_ASSERT;
  } catch (const rip::YamlError& e) {
    if (p) p->last_error = e.what();
    return RIP_ERR_IO;
  } catch (const CapacityError& e) {
    if (p) p->last_error = e.what();
    return RIP_ERR_CAPACITY;
  } catch (const DeviceError& e) {
    if (p) p->last_error = e.what();
    return RIP_ERR_DEVICE;
  } catch (const std::exception& e) {
    if (p) p->last_error = e.what();
    return RIP_ERR_DEVICE;
  }
}

void need(const rip_pipeline* p) {
  if (!p) throw InvalidArgument("null pipeline handle");
}

void need_device(const rip_pipeline* p) {
  need(p);
  if (p->device == RIP_DEVICE_NONE)
    throw DeviceError("this handle was created with RIP_DEVICE_NONE (parameter handling only): frames can only be "
                      "processed on a HIP device; there is no CPU execution path");
}

void copy_string(const std::string& s, char* out, size_t cap) {
  if (!out || cap == 0) throw InvalidArgument("null output buffer");
  if (s.size() + 1 > cap) throw CapacityError("string buffer too small");
  std::memcpy(out, s.c_str(), s.size() + 1);
}

}  // namespace

namespace rip {
Tunables tunables_from_env() {
  Tunables t;
  auto positive = [](const char* name, int dflt) {
    const char* e = std::getenv(name);
    if (!e || !*e) return dflt;
    const int v = std::atoi(e);
    return v > 0 ? v : dflt;
  };
  t.chain_blocks = positive("RIP_CHAIN_BLOCKS", t.chain_blocks);
  t.chain_frames = positive("RIP_CHAIN_FRAMES", t.chain_frames);
  t.stats_blocks = positive("RIP_STATS_BLOCKS", t.stats_blocks);
  if (const char* e = std::getenv("RIP_REMAP_RING")) t.remap_ring = std::atoi(e) != 0;
  t.remap_stages = positive("RIP_REMAP_STAGES", t.remap_stages);
  t.remap_per_cu = positive("RIP_REMAP_PER_CU", t.remap_per_cu);
  t.remap_frames = positive("RIP_REMAP_FRAMES", t.remap_frames);
  t.remap_exp = positive("RIP_REMAP_EXP", t.remap_exp);
  if (const char* e = std::getenv("RIP_REMAP_DEAL")) t.remap_deal = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("RIP_CHAIN_DEAL")) t.chain_deal = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("RIP_REMAP_FUSED")) t.remap_fused = std::atoi(e) != 0;
  if (const char* e = std::getenv("RIP_CHAIN_NT")) t.chain_nt = std::atoi(e);
  t.ccc_lds_hist_min = positive("RIP_CCC_LDS_HIST_MIN", t.ccc_lds_hist_min);
  t.overlap_groups = positive("RIP_OVERLAP_GROUPS", t.overlap_groups);
  t.overlap_mode = positive("RIP_OVERLAP_MODE", t.overlap_mode);
  t.debug_occupancy = std::getenv("RIP_DEBUG_OCC") != nullptr;
  return t;
}
}  // namespace rip

#pragma GCC visibility push(default)
extern "C" {

const char* rip_version(void) { return "raw_image_pipeline_amd 0.1 (gfx950)"; }

rip_status rip_create(int device, int use_gpu, const char* params_path, const char* calibration_path,
                      const char* color_calibration_path, rip_pipeline** out) {
  if (!out) {
    g_create_error = "rip_create: out is null";
    return RIP_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  rip_pipeline* p = nullptr;
  try {
    if (device != RIP_DEVICE_NONE) {
      int count = 0;
      hipError_t e = hipGetDeviceCount(&count);
      if (e != hipSuccess || count <= 0)
        throw DeviceError("no HIP device available: this library has no CPU execution path (hipGetDeviceCount: " +
                          std::string(hipGetErrorString(e)) + ")");
      if (device < 0 || device >= count) throw InvalidArgument("device ordinal out of range");
      HIP_CHECK(hipSetDevice(device));
    }
    p = new rip_pipeline();
    p->device = device;
    p->m.use_gpu = use_gpu != 0;
    // RawImagePipeline(use_gpu, params, calib, color_calib), raw_image_pipeline.cpp:23-40
    if (!params_path || !*params_path) {
      rip::apply_example_params(p->m);  // DEFAULT_PARAMS_PATH = config/pipeline_params_example.yaml
    } else if (!rip::load_params_file(p->m, params_path)) {
      std::fprintf(stderr, "Warning: parameters file doesn't exist\n");
    }
    if (calibration_path && *calibration_path) {
      if (!rip::load_camera_calibration_file(p->m, calibration_path)) std::fprintf(stderr, "Warning: Calibration file doesn't exist\n");
    }
    if (!color_calibration_path || !*color_calibration_path) {
      rip::apply_example_color_calibration(p->m);  // DEFAULT_COLOR_CALIBRATION_PATH
    } else if (!rip::load_color_calibration_file(p->m, color_calibration_path)) {
      std::fprintf(stderr, "Warning: Color calibration file doesn't exist\n");
    }
    und_init(p);
    // every environment override is read here, once per handle
    p->tn = rip::tunables_from_env();
    if (const char* t = std::getenv("RIP_REMAP_TILED")) p->use_tiled_remap = std::atoi(t) != 0;
    if (const char* e = std::getenv("RIP_MAPS_ON_HOST")) p->maps_on_host = *e && *e != '0';
    if (const char* e = std::getenv("RIP_PLAN_ON_HOST")) p->plan_on_host = *e && *e != '0';
    if (const char* e = std::getenv("RIP_DEBUG_DIR")) if (*e) p->debug_dir = e;
    if (const char* e = std::getenv("RIP_CCC_MODEL")) if (*e) p->ccc_model_env = e;
    if (const char* e = std::getenv("RIP_FP_CONTRACT")) {
      // the same values rip_set_fp_contraction accepts; anything else ("2", "true", a typo) fails the create instead of silently
      // selecting the x86 model
      if ((e[0] != '0' && e[0] != '1') || e[1] != 0) throw rip::YamlError(std::string("RIP_FP_CONTRACT=[") + e + "]: 0 (uncontracted) or 1 (contracted) expected");
      p->fp_contract = e[0] == '1' ? 1 : 0;
    }
    if (!p->ccc_model_env.empty()) rip::ccc_load_model_file(p->ccc, p->ccc_model_env);
    *out = p;
    return RIP_OK;
  } catch (const rip::YamlError& e) {
    g_create_error = e.what();
    delete p;
    return RIP_ERR_IO;
  } catch (const std::invalid_argument& e) {
    g_create_error = e.what();
    delete p;
    return RIP_ERR_INVALID_ARGUMENT;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    delete p;
    return RIP_ERR_DEVICE;
  }
}

rip_status rip_create_default(int device, int use_gpu, rip_pipeline** out) {
  // RawImagePipeline(bool use_gpu), raw_image_pipeline.cpp:16-21: example params, example camera
  // calibration and example colour calibration
  rip_status st = rip_create(device, use_gpu, nullptr, nullptr, nullptr, out);
  if (st != RIP_OK) return st;
  rip::apply_example_camera_calibration((*out)->m);
  und_init(*out);
  return RIP_OK;
}

void rip_destroy(rip_pipeline* p) { delete p; }

const char* rip_last_error(const rip_pipeline* p) { return p ? p->last_error.c_str() : g_create_error.c_str(); }

rip_status rip_set_stream(rip_pipeline* p, void* s) {
  return guarded(p, [&] {
    need_device(p);
    hipStream_t next = static_cast<hipStream_t>(s);
    if (next != p->stream && p->work_enqueued) {
      // The handle's scratch state crosses frame calls in stream order (statistics records and ccc histogram counters handed
      // back zeroed by the kernels of the call before, the ccc Kalman state, the gains the debug dumps reuse): work left on
      // the old stream must come first on the new one too.  Best effort -- a stream the caller has destroyed already has
      // nothing pending and the runtime refuses the record, which is ignored.
      DeviceGuard device_guard(p->device);
      if (!p->switch_event && hipEventCreateWithFlags(&p->switch_event, hipEventDisableTiming) != hipSuccess) p->switch_event = nullptr;
      if (p->switch_event && hipEventRecord(p->switch_event, p->stream) == hipSuccess) (void)hipStreamWaitEvent(next, p->switch_event, 0);
      (void)hipGetLastError();
    }
    p->stream = next;
  });
}

rip_status rip_query_output(rip_pipeline* p, int rows, int cols, int channels, const char* encoding, int* out_rows,
                            int* out_cols, int* out_channels, char encoding_out[32]) {
  return guarded(p, [&] {
    need(p);
    if (!encoding) throw InvalidArgument("encoding is null");
    Plan pl = make_plan(p, rows, cols, channels, encoding);
    if (out_rows) *out_rows = pl.out_rows;
    if (out_cols) *out_cols = pl.out_cols;
    if (out_channels) *out_channels = pl.channels;
    if (encoding_out) copy_string(pl.encoding_out, encoding_out, 32);
  });
}

rip_status rip_query_taps(rip_pipeline* p, int rows, int cols, int channels, const char* encoding, int* tap_rows, int* tap_cols,
                          int* tap_channels) {
  return guarded(p, [&] {
    need(p);
    if (!encoding) throw InvalidArgument("encoding is null");
    Plan pl = make_plan(p, rows, cols, channels, encoding);
    if (tap_rows) *tap_rows = pl.mid_rows;
    if (tap_cols) *tap_cols = pl.mid_cols;
    if (tap_channels) *tap_channels = pl.channels;
  });
}

rip_status rip_apply_device(rip_pipeline* p, const void* d_in, size_t in_step, size_t in_frame_stride, int n_frames, int rows,
                            int cols, int channels, const char* encoding, void* d_out, size_t out_step,
                            size_t out_frame_stride, void* d_tap_debayered, void* d_tap_color) {
  return guarded(p, [&] {
    need_device(p);
    if (!d_in || !d_out || !encoding) throw InvalidArgument("null buffer or encoding");
    if (n_frames < 0) throw InvalidArgument("negative frame count");
    if (n_frames == 0) return;
    Plan pl = make_plan(p, rows, cols, channels, encoding);
    const size_t eb = (size_t)pl.elem_bytes;
    if (in_step == 0) in_step = (size_t)cols * channels * eb;
    if (in_frame_stride == 0) in_frame_stride = in_step * rows;
    if (in_step < (size_t)cols * channels * eb) throw InvalidArgument("input row pitch smaller than a row");
    if (eb == 2 && (d_tap_debayered || d_tap_color)) throw InvalidArgument("16-bit Bayer frames have no taps");
    // several kernels put the frame index on gridDim.y (<= 65535): longer batches go through in slices.  The
    // frames of a stream are processed in order either way (the ccc Kalman state lives on the device).
    constexpr int kMaxFramesPerLaunch = 16384;
    const size_t o_step = out_step ? out_step : (size_t)pl.out_cols * pl.channels * eb;
    const size_t o_stride = out_frame_stride ? out_frame_stride : o_step * pl.out_rows;
    // the kernels address one frame with 32-bit byte offsets and 24-bit row multiplies: refuse pitches they cannot
    // express (and pitches that would make rows or frames overlap) instead of writing somewhere else
    if (o_step < (size_t)pl.out_cols * pl.channels * eb) throw InvalidArgument("output row pitch smaller than a row");
    if (o_stride < o_step * (size_t)pl.out_rows) throw InvalidArgument("output frame stride smaller than a frame");
    if (in_frame_stride < in_step * (size_t)(rows - 1) + (size_t)cols * channels * eb) throw InvalidArgument("input frame stride smaller than a frame");
    if (in_step >= (1u << 24) || o_step >= (1u << 24) || (unsigned long long)in_step * rows >= (1ull << 32) ||
        (unsigned long long)o_step * pl.out_rows >= (1ull << 32))
      throw InvalidArgument("row pitch too large: pitches must stay below 16 MiB and a frame below 4 GiB");
    const size_t tap_frame = (size_t)pl.mid_rows * pl.mid_cols * pl.channels;
    for (int f0 = 0; f0 < n_frames; f0 += kMaxFramesPerLaunch) {
      const int n = std::min(kMaxFramesPerLaunch, n_frames - f0);
      uint8_t* tap_deb = d_tap_debayered ? static_cast<uint8_t*>(d_tap_debayered) + (size_t)f0 * tap_frame : nullptr;
      uint8_t* tap_col = d_tap_color ? static_cast<uint8_t*>(d_tap_color) + (size_t)f0 * tap_frame : nullptr;
      run_batch(p, pl, static_cast<const uint8_t*>(d_in) + (size_t)f0 * in_frame_stride, in_step, in_frame_stride, n, rows, cols,
                static_cast<uint8_t*>(d_out) + (size_t)f0 * o_stride, o_step, o_stride, tap_deb, tap_col);
    }
  });
}

rip_status rip_apply(rip_pipeline* p, const uint8_t* image, int rows, int cols, int channels, size_t step, const char* encoding,
                     uint8_t* out, size_t out_capacity, int* out_rows, int* out_cols, int* out_channels,
                     char encoding_out[32]) {
  return guarded(p, [&] {
    need_device(p);
    if (!image || !out || !encoding) throw InvalidArgument("null buffer or encoding");
    Plan pl = make_plan(p, rows, cols, channels, encoding);
    DeviceGuard device_guard(p->device);
    const size_t eb = (size_t)pl.elem_bytes;
    if (step == 0) step = (size_t)cols * channels * eb;
    const size_t in_pitch = ((size_t)cols * channels * eb + 3) & ~(size_t)3;  // dword-aligned rows on the device
    const size_t in_bytes = in_pitch * rows;
    const size_t out_bytes = (size_t)pl.out_rows * pl.out_cols * pl.channels * eb;
    const size_t mid_bytes = (size_t)pl.mid_rows * pl.mid_cols * pl.channels;
    if (out_capacity < out_bytes) throw CapacityError("output buffer too small: need " + std::to_string(out_bytes) + " bytes");
    p->d_in.reserve(in_bytes);
    p->d_out.reserve(out_bytes);
    uint8_t* tap_deb = nullptr;
    uint8_t* tap_col = nullptr;
    if ((p->tap_mask & RIP_TAP_DEBAYERED) && eb == 1) {
      p->d_tap_deb.reserve(mid_bytes);
      tap_deb = p->d_tap_deb.as<uint8_t>();
    }
    if ((p->tap_mask & RIP_TAP_COLOR) && eb == 1) {
      p->d_tap_col.reserve(mid_bytes);
      tap_col = p->d_tap_col.as<uint8_t>();
    }
    HIP_CHECK(hipMemcpy2DAsync(p->d_in.ptr, in_pitch, image, step, (size_t)cols * channels * eb, (size_t)rows, hipMemcpyHostToDevice, p->stream));
    run_batch(p, pl, p->d_in.as<uint8_t>(), in_pitch, in_bytes, 1, rows, cols, p->d_out.as<uint8_t>(), 0, 0, tap_deb, tap_col);
    HIP_CHECK(hipMemcpyAsync(out, p->d_out.ptr, out_bytes, hipMemcpyDeviceToHost, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
    if (p->m.debug && eb == 1) write_debug_dumps(p, pl, in_pitch, in_bytes, rows, cols, out);
    for (int i = 0; i < 3; i++) p->last_valid[i] = false;
    auto remember = [&](int which, DevBuf* buf, int r, int c, bool on) {
      p->last_valid[which] = on;
      p->last_buf[which] = buf;
      p->last_host[which] = nullptr;
      p->last_rows[which] = r;
      p->last_cols[which] = c;
      p->last_cn[which] = pl.channels;
    };
    remember(RIP_IMAGE_DEBAYERED, &p->d_tap_deb, pl.mid_rows, pl.mid_cols, tap_deb != nullptr);
    remember(RIP_IMAGE_COLOR, &p->d_tap_col, pl.mid_rows, pl.mid_cols, tap_col != nullptr);
    remember(RIP_IMAGE_PROCESSED, &p->d_out, pl.out_rows, pl.out_cols, (p->tap_mask & RIP_TAP_PROCESSED) != 0 && eb == 1);
    if (out_rows) *out_rows = pl.out_rows;
    if (out_cols) *out_cols = pl.out_cols;
    if (out_channels) *out_channels = pl.channels;
    if (encoding_out) copy_string(pl.encoding_out, encoding_out, 32);
  });
}

namespace {
// true for hipHostMalloc'd / hipHostRegister'ed memory (the runtime can DMA from it without a staging copy)
bool host_pointer_is_pinned(const void* ptr) {
  hipPointerAttribute_t at;
  std::memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
    (void)hipGetLastError();  // older runtimes report an ordinary malloc'd pointer as an error
    return false;
  }
  return at.type == hipMemoryTypeHost;
}
}  // namespace

namespace {
rip_status submit_impl(rip_pipeline* p, const uint8_t* image, int rows, int cols, int channels, size_t step, const char* encoding,
                       uint8_t* ext_out, size_t ext_out_capacity, uint8_t* ext_deb, uint8_t* ext_col, size_t ext_tap_capacity,
                       uint64_t* ticket) {
  return guarded(p, [&] {
    need_device(p);
    if (!image || !encoding || !ticket) throw InvalidArgument("null buffer, encoding or ticket");
    Plan pl = make_plan(p, rows, cols, channels, encoding);
    {  // destinations given by the caller (rip_submit_to): checked before anything is enqueued or any slot is touched
      const size_t eb0 = (size_t)pl.elem_bytes;
      const size_t out_need = (size_t)pl.out_rows * pl.out_cols * pl.channels * eb0, mid_need = (size_t)pl.mid_rows * pl.mid_cols * pl.channels;
      if (ext_out && ext_out_capacity < out_need) throw CapacityError("rip_submit_to: result buffer too small: need " + std::to_string(out_need) + " bytes");
      if ((ext_deb || ext_col) && ext_tap_capacity < mid_need) throw CapacityError("rip_submit_to: tap buffer too small: need " + std::to_string(mid_need) + " bytes");
      if (ext_deb && !((p->tap_mask & RIP_TAP_DEBAYERED) && eb0 == 1)) throw InvalidArgument("rip_submit_to: the debayered tap is not kept (rip_set_taps)");
      if (ext_col && !((p->tap_mask & RIP_TAP_COLOR) && eb0 == 1)) throw InvalidArgument("rip_submit_to: the colour tap is not kept (rip_set_taps)");
      for (const void* ptr : {(const void*)ext_out, (const void*)ext_deb, (const void*)ext_col})
        if (ptr && !host_pointer_is_pinned(ptr))
          throw InvalidArgument("rip_submit_to: destination buffers must be page-locked (rip_host_alloc, hipHostMalloc, hipHostRegister)");
    }
    DeviceGuard device_guard(p->device);
    if (!p->ul_stream) HIP_CHECK(hipStreamCreateWithFlags(&p->ul_stream, hipStreamNonBlocking));
    if (!p->dl_stream) HIP_CHECK(hipStreamCreateWithFlags(&p->dl_stream, hipStreamNonBlocking));
    while ((int)p->ring.size() < p->ring_depth) {
      std::unique_ptr<RingSlot> sl(new RingSlot());
      static const bool debug_ring = std::getenv("RIP_DEBUG_RING") != nullptr;  // development aid: per-frame upload / kernel / download times on stderr
      for (hipEvent_t* e : {&sl->ev_up, &sl->ev_kernels, &sl->ev_done}) HIP_CHECK(hipEventCreateWithFlags(e, debug_ring ? hipEventDefault : hipEventDisableTiming));
      if (debug_ring) HIP_CHECK(hipEventCreateWithFlags(&sl->ev_start, hipEventDefault));
      if (debug_ring) HIP_CHECK(hipEventCreateWithFlags(&sl->ev_dl_start, hipEventDefault));
      p->ring.push_back(std::move(sl));
    }
    // a free slot; failing that the slot of the frame collected last (its view and taps end here); failing that: full
    RingSlot* pick = nullptr;
    uint64_t oldest = ~0ull;
    for (auto& c : p->ring)
      if (!c->busy && !c->held) pick = c.get();
    if (!pick)
      for (auto& c : p->ring) {
        if (c->held) pick = c.get();
        if (c->busy) oldest = std::min(oldest, c->ticket);
      }
    if (!pick)
      throw CapacityError("rip_submit: " + std::to_string(p->ring_depth) + " frames are in flight; collect ticket " +
                          std::to_string(oldest) + " first (or raise rip_set_ring_depth)");
    RingSlot& sl = *pick;
    if (sl.held) {
      sl.held = false;
      for (int i = 0; i < 3; i++)
        if (p->last_buf[i] == &sl.d_tap_deb || p->last_buf[i] == &sl.d_tap_col || p->last_buf[i] == &sl.d_out) p->last_valid[i] = false;
    }
    const size_t eb = (size_t)pl.elem_bytes;
    if (step == 0) step = (size_t)cols * channels * eb;
    const size_t in_pitch = ((size_t)cols * channels * eb + 3) & ~(size_t)3;  // dword-aligned rows on the device
    const size_t in_bytes = in_pitch * rows;
    const size_t out_bytes = (size_t)pl.out_rows * pl.out_cols * pl.channels * eb;
    const size_t mid_bytes = (size_t)pl.mid_rows * pl.mid_cols * pl.channels;
    sl.d_in.reserve(in_bytes);
    sl.d_out.reserve(out_bytes);
    if (!ext_out) sl.reserve_host(out_bytes);
    sl.has_deb = (p->tap_mask & RIP_TAP_DEBAYERED) && eb == 1;
    sl.has_col = (p->tap_mask & RIP_TAP_COLOR) && eb == 1;
    sl.dl_deb = sl.has_deb && ((p->tap_download_mask & RIP_TAP_DEBAYERED) || ext_deb);
    sl.dl_col = sl.has_col && ((p->tap_download_mask & RIP_TAP_COLOR) || ext_col);
    if (sl.has_deb) sl.d_tap_deb.reserve(mid_bytes);
    if (sl.has_col) sl.d_tap_col.reserve(mid_bytes);
    if (sl.dl_deb && !ext_deb) RingSlot::reserve_pinned(sl.h_tap[0], sl.h_tap_cap[0], mid_bytes);
    if (sl.dl_col && !ext_col) RingSlot::reserve_pinned(sl.h_tap[1], sl.h_tap_cap[1], mid_bytes);
    sl.dst_out = ext_out ? (void*)ext_out : sl.h_out;
    sl.dst_tap[0] = ext_deb ? (void*)ext_deb : sl.h_tap[0];
    sl.dst_tap[1] = ext_col ? (void*)ext_col : sl.h_tap[1];
    // upload (its own stream: it overlaps the kernels of the frame before) -> kernels on the handle's stream, in submission
    // order -> download into the slot's pinned buffer (its own stream: it overlaps the kernels of the frame after)
    // A frame in pinned memory (rip_host_alloc, hipHostMalloc, hipHostRegister) is DMA'd from where it lies and must stay
    // untouched until its ticket is collected.  Anything else is copied into the slot's pinned staging buffer first, so the
    // caller's buffer is free again when this call returns whatever the runtime does with an asynchronous 2-D copy from
    // pageable memory (above its staging threshold it pins the pages in place and copies after the call has returned).
    const size_t row_bytes = (size_t)cols * channels * eb;
    inflight_gate().forget(sl.gate_device, sl.ev_done);  // the slot's previous frame (collected, or it would not have been picked)
    inflight_gate().admit(p->device);
    if (sl.ev_start) HIP_CHECK(hipEventRecord(sl.ev_start, p->ul_stream));
    if (host_pointer_is_pinned(image)) {
      HIP_CHECK(hipMemcpy2DAsync(sl.d_in.ptr, in_pitch, image, step, row_bytes, (size_t)rows, hipMemcpyHostToDevice, p->ul_stream));
    } else {
      sl.reserve_host_in(in_bytes);
      uint8_t* stage = static_cast<uint8_t*>(sl.h_in);
      if (step == in_pitch) {
        CopyPool::get().copy(stage, image, in_pitch * (size_t)(rows - 1) + row_bytes);
      } else {
        for (int r = 0; r < rows; r++) std::memcpy(stage + (size_t)r * in_pitch, image + (size_t)r * step, row_bytes);
      }
      HIP_CHECK(hipMemcpyAsync(sl.d_in.ptr, stage, in_bytes, hipMemcpyHostToDevice, p->ul_stream));
    }
    HIP_CHECK(hipEventRecord(sl.ev_up, p->ul_stream));
    HIP_CHECK(hipStreamWaitEvent(p->stream, sl.ev_up, 0));
    run_batch(p, pl, sl.d_in.as<uint8_t>(), in_pitch, in_bytes, 1, rows, cols, sl.d_out.as<uint8_t>(), 0, 0,
              sl.has_deb ? sl.d_tap_deb.as<uint8_t>() : nullptr, sl.has_col ? sl.d_tap_col.as<uint8_t>() : nullptr);
    HIP_CHECK(hipEventRecord(sl.ev_kernels, p->stream));
    HIP_CHECK(hipStreamWaitEvent(p->dl_stream, sl.ev_kernels, 0));
    if (sl.ev_dl_start) HIP_CHECK(hipEventRecord(sl.ev_dl_start, p->dl_stream));
    HIP_CHECK(hipMemcpyAsync(sl.dst_out, sl.d_out.ptr, out_bytes, hipMemcpyDeviceToHost, p->dl_stream));
    // the taps rip_set_tap_download names travel with the result: a per-frame caller that publishes them
    // (raw_image_pipeline_ros.cpp:245-287: up to three images per callback) gets them from pinned host memory instead of
    // one synchronous device read each (rip_get_image / rip_get_image_view after rip_collect).  Off by default: a caller that
    // only wants the final image must not pay 30 MB more PCIe traffic per 2448 x 2048 frame.
    if (sl.dl_deb) HIP_CHECK(hipMemcpyAsync(sl.dst_tap[0], sl.d_tap_deb.ptr, mid_bytes, hipMemcpyDeviceToHost, p->dl_stream));
    if (sl.dl_col) HIP_CHECK(hipMemcpyAsync(sl.dst_tap[1], sl.d_tap_col.ptr, mid_bytes, hipMemcpyDeviceToHost, p->dl_stream));
    HIP_CHECK(hipEventRecord(sl.ev_done, p->dl_stream));
    inflight_gate().enqueued(p->device, sl.ev_done);
    sl.gate_device = p->device;
    sl.pl = pl;
    sl.ticket = p->next_ticket++;
    sl.busy = true;
    *ticket = sl.ticket;
  });
}
}  // namespace

rip_status rip_submit(rip_pipeline* p, const uint8_t* image, int rows, int cols, int channels, size_t step, const char* encoding,
                      uint64_t* ticket) {
  return submit_impl(p, image, rows, cols, channels, step, encoding, nullptr, 0, nullptr, nullptr, 0, ticket);
}
rip_status rip_submit_to(rip_pipeline* p, const uint8_t* image, int rows, int cols, int channels, size_t step, const char* encoding,
                         uint8_t* out, size_t out_capacity, uint8_t* tap_debayered, uint8_t* tap_color, size_t tap_capacity, uint64_t* ticket) {
  return submit_impl(p, image, rows, cols, channels, step, encoding, out, out_capacity, tap_debayered, tap_color, tap_capacity, ticket);
}

rip_status rip_collect(rip_pipeline* p, uint64_t ticket, uint8_t* out, size_t out_capacity, const uint8_t** out_view, int* out_rows,
                       int* out_cols, int* out_channels, char encoding_out[32]) {
  return guarded(p, [&] {
    need_device(p);
    RingSlot* sl = nullptr;
    for (auto& s : p->ring)
      if (s->busy && s->ticket == ticket) sl = s.get();
    if (!sl) throw InvalidArgument("rip_collect: ticket " + std::to_string(ticket) + " is not in flight");
    const Plan& pl = sl->pl;
    const size_t out_bytes = (size_t)pl.out_rows * pl.out_cols * pl.channels * (size_t)pl.elem_bytes;
    if (out && out_capacity < out_bytes) throw CapacityError("output buffer too small: need " + std::to_string(out_bytes) + " bytes");
    DeviceGuard device_guard(p->device);
    HIP_CHECK(hipEventSynchronize(sl->ev_done));
    inflight_gate().forget(sl->gate_device, sl->ev_done);
    if (sl->ev_start) {
      float up = 0, kern = 0, down = 0, all = 0, copy = 0;
      (void)hipEventElapsedTime(&copy, sl->ev_dl_start, sl->ev_done);
      (void)hipEventElapsedTime(&up, sl->ev_start, sl->ev_up);
      (void)hipEventElapsedTime(&kern, sl->ev_up, sl->ev_kernels);
      (void)hipEventElapsedTime(&down, sl->ev_kernels, sl->ev_done);
      (void)hipEventElapsedTime(&all, sl->ev_start, sl->ev_done);
      int idx = 0;
      for (size_t i = 0; i < p->ring.size(); i++)
        if (p->ring[i].get() == sl) idx = (int)i;
      std::fprintf(stderr, "rip ring: ticket %llu slot %d upload %.3f ms, kernels (incl. waiting for the frame before) %.3f, download (incl. waiting) %.3f of which the copies %.3f, total %.3f\n",
                   (unsigned long long)ticket, idx, up, kern, down, copy, all);
    }
    if (out && out != sl->dst_out) CopyPool::get().copy(out, sl->dst_out, out_bytes);
    if (out_view) *out_view = static_cast<const uint8_t*>(sl->dst_out);
    for (auto& c : p->ring) c->held = false;  // the frame collected before this one lets go of its slot
    sl->busy = false;
    sl->held = true;
    const bool eb1 = pl.elem_bytes == 1;
    auto remember = [&](int which, DevBuf* buf, const void* host, int r, int c, bool on) {
      p->last_valid[which] = on;
      p->last_buf[which] = buf;
      p->last_host[which] = host;
      p->last_rows[which] = r;
      p->last_cols[which] = c;
      p->last_cn[which] = pl.channels;
    };
    // The getters may read a host copy only where it is the HANDLE's pinned buffer.  Destinations the caller named
    // (rip_submit_to) are the caller's again from this point on -- rip.h only asks for them until the ticket is collected:
    // they may be freed or edited, so the getters go back to the device image in the slot this frame keeps held.
    remember(RIP_IMAGE_DEBAYERED, &sl->d_tap_deb, sl->dl_deb && sl->dst_tap[0] == sl->h_tap[0] ? sl->dst_tap[0] : nullptr, pl.mid_rows, pl.mid_cols, sl->has_deb);
    remember(RIP_IMAGE_COLOR, &sl->d_tap_col, sl->dl_col && sl->dst_tap[1] == sl->h_tap[1] ? sl->dst_tap[1] : nullptr, pl.mid_rows, pl.mid_cols, sl->has_col);
    remember(RIP_IMAGE_PROCESSED, &sl->d_out, sl->dst_out == sl->h_out ? sl->dst_out : nullptr, pl.out_rows, pl.out_cols, (p->tap_mask & RIP_TAP_PROCESSED) != 0 && eb1);
    if (out_rows) *out_rows = pl.out_rows;
    if (out_cols) *out_cols = pl.out_cols;
    if (out_channels) *out_channels = pl.channels;
    if (encoding_out) copy_string(pl.encoding_out, encoding_out, 32);
  });
}

rip_status rip_set_ring_depth(rip_pipeline* p, int depth) {
  return guarded(p, [&] {
    need(p);
    if (depth < 1 || depth > 16) throw InvalidArgument("ring depth must be in 1..16");
    for (auto& s : p->ring)
      if (s->busy) throw InvalidArgument("rip_set_ring_depth: frames are in flight");
    if (depth == p->ring_depth) return;
    if (!p->ring.empty()) {
      DeviceGuard device_guard(p->device);
      for (int i = 0; i < 3; i++)  // the getters must not look into a slot that is about to go
        for (auto& s : p->ring)
          if (p->last_buf[i] == &s->d_tap_deb || p->last_buf[i] == &s->d_tap_col || p->last_buf[i] == &s->d_out) p->last_valid[i] = false;
      if (p->dl_stream) HIP_CHECK(hipStreamSynchronize(p->dl_stream));
      for (auto& s : p->ring) s->release();
      p->ring.clear();
    }
    p->ring_depth = depth;
  });
}

void* rip_host_alloc(size_t bytes) {
  void* ptr = nullptr;
  if (bytes == 0 || hipHostMalloc(&ptr, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return ptr;
}
void rip_host_free(void* ptr) {
  if (ptr) (void)hipHostFree(ptr);
}
void rip_copy_host(void* dst, const void* src, size_t bytes) {
  if (dst && src && bytes) CopyPool::get().copy(dst, src, bytes);
}

rip_status rip_get_image(rip_pipeline* p, int which, uint8_t* out, size_t out_capacity, int* rows, int* cols, int* channels) {
  return guarded(p, [&] {
    need(p);
    if (which == RIP_IMAGE_RECT_MASK || which < 0 || which > 3 || !p->last_valid[which]) {
      // rect_mask_ is never written by the reference; taps that were not kept are empty too
      if (which < 0 || which > 3) throw InvalidArgument("unknown image id");
      if (rows) *rows = 0;
      if (cols) *cols = 0;
      if (channels) *channels = 0;
      return;
    }
    size_t bytes = (size_t)p->last_rows[which] * p->last_cols[which] * p->last_cn[which];
    if (rows) *rows = p->last_rows[which];
    if (cols) *cols = p->last_cols[which];
    if (channels) *channels = p->last_cn[which];
    if (!out) return;  // size query
    if (out_capacity < bytes) throw CapacityError("image buffer too small");
    need_device(p);
    if (p->last_host[which]) {  // a frame that came through rip_collect: the image is in pinned host memory already
      CopyPool::get().copy(out, p->last_host[which], bytes);
      return;
    }
    DeviceGuard device_guard(p->device);
    HIP_CHECK(hipMemcpyAsync(out, p->last_buf[which]->ptr, bytes, hipMemcpyDeviceToHost, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
  });
}

rip_status rip_get_image_view(rip_pipeline* p, int which, const uint8_t** view, int* rows, int* cols, int* channels) {
  return guarded(p, [&] {
    need(p);
    if (!view) throw InvalidArgument("null view");
    if (which < 0 || which > 3) throw InvalidArgument("unknown image id");
    *view = nullptr;
    const bool have = which != RIP_IMAGE_RECT_MASK && p->last_valid[which];
    if (rows) *rows = have ? p->last_rows[which] : 0;
    if (cols) *cols = have ? p->last_cols[which] : 0;
    if (channels) *channels = have ? p->last_cn[which] : 0;
    if (have) *view = static_cast<const uint8_t*>(p->last_host[which]);
  });
}

rip_status rip_set_taps(rip_pipeline* p, int mask) {
  return guarded(p, [&] {
    need(p);
    p->tap_mask = mask & 7;
  });
}

rip_status rip_set_tap_download(rip_pipeline* p, int mask) {
  return guarded(p, [&] {
    need(p);
    p->tap_download_mask = mask & (RIP_TAP_DEBAYERED | RIP_TAP_COLOR);
  });
}

// ---- loaders -----------------------------------------------------------------------------------
rip_status rip_load_params(rip_pipeline* p, const char* path) {
  return guarded(p, [&] {
    need(p);
    if (!path) throw InvalidArgument("path is null");
    std::fprintf(stderr, "Loading raw_image_pipeline params from file %s\n", path);
    if (!rip::load_params_file(p->m, path)) {
      std::fprintf(stderr, "Warning: parameters file doesn't exist\n");  // nothing else happens (:162-164)
    } else {
      // loadParams re-creates every module (std::make_unique, raw_image_pipeline.cpp:56-160): whatever
      // loadColorCalibration / loadCameraCalibration had loaded is gone (identity matrix, calibration not available --
      // the constructors reload them afterwards, :27-29), and the white balancer starts over with a fresh ccc
      // estimator (first frame, new Kalman filter)
      const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      for (int i = 0; i < 9; i++) p->m.cc_matrix[i] = (float)eye[i];
      for (int i = 0; i < 4; i++) p->m.cc_bias[i] = 0;
      p->m.cc_available = false;
      p->m.und_available = false;
      p->ccc_state_init = false;
    }
    p->tabs_dirty = p->vig_dirty = p->ccc_cfg_dirty = true;
    und_init(p);  // setBalance / setFovScale re-run init()
  });
}
rip_status rip_load_camera_calibration(rip_pipeline* p, const char* path) {
  return guarded(p, [&] {
    need(p);
    if (!path) throw InvalidArgument("path is null");
    std::fprintf(stderr, "Loading camera calibration from file %s\n", path);
    if (!rip::load_camera_calibration_file(p->m, path)) std::fprintf(stderr, "Warning: Calibration file doesn't exist\n");
    und_init(p);
  });
}
rip_status rip_load_color_calibration(rip_pipeline* p, const char* path) {
  return guarded(p, [&] {
    need(p);
    if (!path) throw InvalidArgument("path is null");
    std::fprintf(stderr, "Loading color calibration from file %s\n", path);
    if (!rip::load_color_calibration_file(p->m, path)) std::fprintf(stderr, "Warning: Color calibration file doesn't exist\n");
  });
}
rip_status rip_init_undistortion(rip_pipeline* p) {
  return guarded(p, [&] {
    need(p);
    und_init(p);
    ensure_host_maps(p);
    if (p->device != RIP_DEVICE_NONE) {
      DeviceGuard device_guard(p->device);
      ensure_maps(p);
    }
  });
}
rip_status rip_load_ccc_model(rip_pipeline* p, const char* path) {
  return guarded(p, [&] {
    need(p);
    if (!path) throw InvalidArgument("path is null");
    if (!rip::ccc_load_model_file(p->ccc, path)) throw rip::YamlError(std::string("cannot read CCC model ") + path);
    p->ccc_uploaded = false;
  });
}
rip_status rip_set_ccc_model(rip_pipeline* p, int w, int h, const float* filter, const float* bias) {
  return guarded(p, [&] {
    need(p);
    if (!filter || !bias) throw InvalidArgument("null model planes");
    rip::ccc_build_model(p->ccc, w, h, filter, bias);
    p->ccc_uploaded = false;
  });
}
rip_status rip_set_ccc_kalman_model(rip_pipeline* p, double h, double r) {
  return guarded(p, [&] {
    need(p);
    p->kf_h = h;
    p->kf_r = r;
    p->ccc_cfg_dirty = true;
  });
}

rip_status rip_reset_white_balance_temporal_consistency(rip_pipeline* p) {
  return guarded(p, [&] {
    need(p);
    if (p->m.wb_method == "ccc") p->ccc_reset_pending = true;  // white_balance.cpp:42-47
  });
}
rip_status rip_set_gpu(rip_pipeline* p, int v) {
  return guarded(p, [&] {
    need(p);
    p->m.use_gpu = v != 0;
  });
}
rip_status rip_set_debug(rip_pipeline* p, int v) {
  return guarded(p, [&] {
    need(p);
    p->m.debug = v != 0;
  });
}

// ---- setters -----------------------------------------------------------------------------------
#define RIP_SETTER(name, args, body)          \
  rip_status name args {                      \
    return guarded(p, [&] {                   \
      need(p);                                \
      body;                                   \
    });                                       \
  }

RIP_SETTER(rip_set_debayer, (rip_pipeline * p, int v), p->m.debayer_enabled = v != 0)
RIP_SETTER(rip_set_debayer_16bit, (rip_pipeline * p, int v), p->m.debayer_16bit = v != 0)
RIP_SETTER(rip_set_debayer_encoding, (rip_pipeline * p, const char* s), if (!s) throw InvalidArgument("null string"); p->m.debayer_encoding = s)
RIP_SETTER(rip_set_flip, (rip_pipeline * p, int v), p->m.flip_enabled = v != 0)
RIP_SETTER(rip_set_flip_angle, (rip_pipeline * p, int a), p->m.flip_angle = a)
RIP_SETTER(rip_set_white_balance, (rip_pipeline * p, int v), p->m.wb_enabled = v != 0)
RIP_SETTER(rip_set_white_balance_method, (rip_pipeline * p, const char* s), if (!s) throw InvalidArgument("null string"); p->m.wb_method = s)
RIP_SETTER(rip_set_white_balance_percentile, (rip_pipeline * p, double v), p->m.wb_percentile = v)
RIP_SETTER(rip_set_white_balance_saturation_threshold, (rip_pipeline * p, double b, double d), p->m.wb_bright_thr = b; p->m.wb_dark_thr = d)
RIP_SETTER(rip_set_white_balance_temporal_consistency, (rip_pipeline * p, int v), p->m.wb_temporal = v != 0; p->ccc_cfg_dirty = true)
RIP_SETTER(rip_set_color_calibration, (rip_pipeline * p, int v), p->m.cc_enabled = v != 0)
RIP_SETTER(rip_set_color_calibration_matrix, (rip_pipeline * p, const double* v, int n),
           if (!v || n != 9) throw InvalidArgument("color calibration matrix needs 9 values");
           for (int i = 0; i < 9; i++) p->m.cc_matrix[i] = (float)v[i])  // Matx33d -> Matx33f, color_calibration.cpp:79
RIP_SETTER(rip_set_color_calibration_bias, (rip_pipeline * p, const double* v, int n),
           if (!v || n < 3) throw InvalidArgument("color calibration bias needs 3 values");  // vector::at throws
           for (int i = 0; i < 3; i++) p->m.cc_bias[i] = v[i]; p->m.cc_bias[3] = 0)
RIP_SETTER(rip_set_gamma_correction, (rip_pipeline * p, int v), p->m.gamma_enabled = v != 0; p->tabs_dirty = true)
RIP_SETTER(rip_set_gamma_correction_method, (rip_pipeline * p, const char* s), if (!s) throw InvalidArgument("null string"); p->m.gamma_method = s)
RIP_SETTER(rip_set_gamma_correction_k, (rip_pipeline * p, double k), p->m.gamma_k = k; p->tabs_dirty = true)
RIP_SETTER(rip_set_fp_contraction, (rip_pipeline * p, int mode),
           if (mode != 0 && mode != 1) throw InvalidArgument("fp contraction model must be 0 (none) or 1 (fused)");
           if (mode != p->fp_contract) p->vig_dirty = true;  // the mask plane's k = r^2 a2 + r^4 a4 is one of the contracted expressions
           p->fp_contract = mode)
RIP_SETTER(rip_set_vignetting_correction, (rip_pipeline * p, int v), p->m.vig_enabled = v != 0)
RIP_SETTER(rip_set_vignetting_correction_parameters, (rip_pipeline * p, double s, double a2, double a4),
           p->m.vig_scale = s; p->m.vig_a2 = a2; p->m.vig_a4 = a4; p->vig_dirty = true)
RIP_SETTER(rip_set_color_enhancer, (rip_pipeline * p, int v), p->m.ce_enabled = v != 0)
// cross-wired exactly as color_enhancer.cpp:23-33
RIP_SETTER(rip_set_color_enhancer_hue_gain, (rip_pipeline * p, double g), p->m.ce_value_gain = g)
RIP_SETTER(rip_set_color_enhancer_saturation_gain, (rip_pipeline * p, double g), p->m.ce_saturation_gain = g)
RIP_SETTER(rip_set_color_enhancer_value_gain, (rip_pipeline * p, double g), p->m.ce_hue_gain = g)
RIP_SETTER(rip_set_undistortion, (rip_pipeline * p, int v), p->m.und_enabled = v != 0)
RIP_SETTER(rip_set_undistortion_image_size, (rip_pipeline * p, int w, int h), p->m.dist_w = p->m.rect_w = w; p->m.dist_h = p->m.rect_h = h; und_init(p))
RIP_SETTER(rip_set_undistortion_new_image_size, (rip_pipeline * p, int w, int h), p->m.rect_w = w; p->m.rect_h = h; und_init(p))
RIP_SETTER(rip_set_undistortion_balance, (rip_pipeline * p, double b), p->m.balance = b; und_init(p))
RIP_SETTER(rip_set_undistortion_fov_scale, (rip_pipeline * p, double f), p->m.fov_scale = f; und_init(p))
RIP_SETTER(rip_set_undistortion_camera_matrix, (rip_pipeline * p, const double* v, int n),
           if (!v || n < 9) throw InvalidArgument("camera matrix needs 9 values");
           for (int i = 0; i < 9; i++) p->m.dist_K[i] = p->m.rect_K[i] = v[i]; und_init(p))
RIP_SETTER(rip_set_undistortion_distortion_coefficients, (rip_pipeline * p, const double* v, int n),
           if (!v || n < 4) throw InvalidArgument("distortion coefficients need 4 values");
           for (int i = 0; i < 4; i++) p->m.dist_D[i] = p->m.rect_D[i] = v[i]; und_init(p))
RIP_SETTER(rip_set_undistortion_distortion_model, (rip_pipeline * p, const char* s), if (!s) throw InvalidArgument("null string");
           p->m.dist_model = p->m.rect_model = s; und_init(p))
RIP_SETTER(rip_set_undistortion_rectification_matrix, (rip_pipeline * p, const double* v, int n),
           if (!v || n < 9) throw InvalidArgument("rectification matrix needs 9 values");
           for (int i = 0; i < 9; i++) p->m.dist_R[i] = p->m.rect_R[i] = v[i]; und_init(p))
RIP_SETTER(rip_set_undistortion_projection_matrix, (rip_pipeline * p, const double* v, int n),
           if (!v || n < 12) throw InvalidArgument("projection matrix needs 12 values");
           for (int i = 0; i < 12; i++) p->m.dist_P[i] = p->m.rect_P[i] = v[i]; und_init(p))

// ---- getters -----------------------------------------------------------------------------------
int rip_is_debayer_enabled(const rip_pipeline* p) { return p && p->m.debayer_enabled; }
int rip_is_flip_enabled(const rip_pipeline* p) { return p && p->m.flip_enabled; }
int rip_is_white_balance_enabled(const rip_pipeline* p) { return p && p->m.wb_enabled; }
int rip_is_color_calibration_enabled(const rip_pipeline* p) { return p && p->m.cc_enabled; }
int rip_is_gamma_correction_enabled(const rip_pipeline* p) { return p && p->m.gamma_enabled; }
int rip_is_vignetting_correction_enabled(const rip_pipeline* p) { return p && p->m.vig_enabled; }
int rip_is_color_enhancer_enabled(const rip_pipeline* p) { return p && p->m.ce_enabled; }
int rip_is_undistortion_enabled(const rip_pipeline* p) { return p && p->m.und_enabled; }
int rip_get_dist_image_height(const rip_pipeline* p) { return p ? p->m.dist_h : 0; }
int rip_get_dist_image_width(const rip_pipeline* p) { return p ? p->m.dist_w : 0; }
int rip_get_rect_image_height(const rip_pipeline* p) { return p ? p->m.rect_h : 0; }
int rip_get_rect_image_width(const rip_pipeline* p) { return p ? p->m.rect_w : 0; }

rip_status rip_get_dist_distortion_model(const rip_pipeline* p, char* out, size_t cap) {
  return guarded(p, [&] {
    need(p);
    copy_string(p->m.und_available ? p->m.dist_model : std::string("none"), out, cap);  // undistortion.cpp:106-112
  });
}
rip_status rip_get_rect_distortion_model(const rip_pipeline* p, char* out, size_t cap) {
  return guarded(p, [&] {
    need(p);
    // undistortion.cpp:94-104: "none" once undistortion is enabled (the published image is rectified)
    std::string s = "none";
    if (p->m.und_available && !p->m.und_enabled) s = p->m.rect_model;
    copy_string(s, out, cap);
  });
}

#define RIP_GET_VEC(name, field, n)                              \
  rip_status name(const rip_pipeline* p, double* out) {          \
    return guarded(p, [&] {                                      \
      need(p);                                                   \
      if (!out) throw InvalidArgument("null output");            \
      for (int i = 0; i < n; i++) out[i] = (double)p->m.field[i]; \
    });                                                          \
  }
RIP_GET_VEC(rip_get_color_calibration_matrix, cc_matrix, 9)
RIP_GET_VEC(rip_get_color_calibration_bias, cc_bias, 4)
RIP_GET_VEC(rip_get_dist_camera_matrix, dist_K, 9)
RIP_GET_VEC(rip_get_dist_distortion_coefficients, dist_D, 4)
RIP_GET_VEC(rip_get_dist_rectification_matrix, dist_R, 9)
RIP_GET_VEC(rip_get_dist_projection_matrix, dist_P, 12)
RIP_GET_VEC(rip_get_rect_camera_matrix, rect_K, 9)
RIP_GET_VEC(rip_get_rect_distortion_coefficients, rect_D, 4)
RIP_GET_VEC(rip_get_rect_rectification_matrix, rect_R, 9)
RIP_GET_VEC(rip_get_rect_projection_matrix, rect_P, 12)

// ---- introspection -------------------------------------------------------------------------------
rip_status rip_get_undistortion_maps(rip_pipeline* p, float* map_x, float* map_y, size_t cap, int* rows, int* cols) {
  return guarded(p, [&] {
    need(p);
    if (rows) *rows = p->m.dist_h;
    if (cols) *cols = p->m.dist_w;
    if (!map_x || !map_y) return;
    need_host_map(p);
    size_t n = (size_t)p->m.dist_w * p->m.dist_h;
    if (cap < n) throw CapacityError("map buffers too small");
    for (size_t i = 0; i < n; i++) {
      map_x[i] = p->h_map[2 * i];
      map_y[i] = p->h_map[2 * i + 1];
    }
  });
}

rip_status rip_get_white_balance_info(rip_pipeline* p, float* out, int n_frames) {
  return guarded(p, [&] {
    need(p);
    if (!out || n_frames <= 0) throw InvalidArgument("bad arguments");
    if (n_frames > p->last_batch_frames || !p->d_wb.ptr) throw InvalidArgument("no white-balance results for that many frames");
    need_device(p);
    DeviceGuard device_guard(p->device);
    std::vector<rip::FrameWb> h(n_frames);
    HIP_CHECK(hipMemcpyAsync(h.data(), p->d_wb.ptr, sizeof(rip::FrameWb) * n_frames, hipMemcpyDeviceToHost, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
    for (int f = 0; f < n_frames; f++) {
      float* o = out + 8 * f;
      o[0] = h[f].fg[0]; o[1] = h[f].fg[1]; o[2] = h[f].fg[2];
      o[3] = (float)h[f].q8[0]; o[4] = (float)h[f].q8[1]; o[5] = (float)h[f].q8[2];
      o[6] = (float)h[f].uv[0]; o[7] = (float)h[f].uv[1];
    }
  });
}

rip_status rip_get_ccc_track(rip_pipeline* p, int* out, int n_frames) {
  return guarded(p, [&] {
    need(p);
    if (!out || n_frames <= 0) throw InvalidArgument("bad arguments");
    if (n_frames > p->last_batch_frames || !p->d_wb.ptr) throw InvalidArgument("no white-balance results for that many frames");
    need_device(p);
    DeviceGuard device_guard(p->device);
    std::vector<rip::FrameWb> h(n_frames);
    HIP_CHECK(hipMemcpyAsync(h.data(), p->d_wb.ptr, sizeof(rip::FrameWb) * n_frames, hipMemcpyDeviceToHost, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
    for (int f = 0; f < n_frames; f++) {
      out[4 * f + 0] = h[f].uv_raw[0];
      out[4 * f + 1] = h[f].uv_raw[1];
      out[4 * f + 2] = h[f].uv[0];
      out[4 * f + 3] = h[f].uv[1];
    }
  });
}

rip_status rip_profile_begin(rip_pipeline* p, int max_records) {
  return guarded(p, [&] {
    need_device(p);
    DeviceGuard device_guard(p->device);
    if (max_records < 0) throw InvalidArgument("negative record count");
    while (p->prof_events.size() < (size_t)max_records * 2) {
      hipEvent_t e;
      HIP_CHECK(hipEventCreate(&e));
      p->prof_events.push_back(e);
    }
    p->prof_used = 0;
    p->prof_ids.clear();
    p->prof_on = max_records > 0;
  });
}

rip_status rip_profile_end(rip_pipeline* p, double ms_sum[RIP_KERNEL_COUNT], int count[RIP_KERNEL_COUNT]) {
  return guarded(p, [&] {
    need_device(p);
    DeviceGuard device_guard(p->device);
    HIP_CHECK(hipStreamSynchronize(p->stream));
    for (int i = 0; i < RIP_KERNEL_COUNT; i++) {
      if (ms_sum) ms_sum[i] = 0;
      if (count) count[i] = 0;
    }
    for (size_t r = 0; r < p->prof_ids.size(); r++) {
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, p->prof_events[2 * r], p->prof_events[2 * r + 1]));
      if (ms_sum) ms_sum[p->prof_ids[r]] += ms;
      if (count) count[p->prof_ids[r]]++;
    }
    p->prof_on = false;
    p->prof_used = 0;
    p->prof_ids.clear();
  });
}

rip_status rip_debug_write_png(rip_pipeline* p, const char* path, const uint8_t* image, int rows, int cols, int channels, int normalize) {
  return guarded(p, [&] {
    if (!path || !image) throw InvalidArgument("null path or image");
    if (rows < 1 || cols < 1 || (channels != 1 && channels != 3)) throw InvalidArgument("PNG dumps hold 1- or 3-channel 8-bit images");
    std::vector<uint8_t> tmp(image, image + (size_t)rows * cols * channels);
    if (normalize) rip::normalize_minmax_u8(tmp.data(), tmp.size());
    if (!rip::write_png(path, tmp.data(), rows, cols, channels)) throw rip::YamlError(std::string("cannot write ") + path);
  });
}

rip_status rip_debug_plan_info(rip_pipeline* p, int src_rows, int src_cols, int info[9]) {
  return guarded(p, [&] {
    need_device(p);
    if (!info) throw InvalidArgument("null info");
    DeviceGuard device_guard(p->device);
    ensure_plan(p, src_rows, src_cols);
    info[0] = p->plan.tiles_x;
    info[1] = p->plan.tiles_y;
    info[2] = p->plan_n_border;
    info[3] = (int)p->plan.max_lds_bytes;
    info[4] = p->plan.max_rect_w;
    info[5] = p->plan.max_rect_h;
    info[6] = p->plan_on_device ? 1 : 0;
    info[7] = rip::kRemapTileW;
    info[8] = rip::kRemapTileH;
  });
}

rip_status rip_debug_atan(rip_pipeline* p, const double* in, double* out, int n) {
  if (!p) return RIP_ERR_INVALID_ARGUMENT;
  return guarded(p, [&] {
    need_device(p);
    if (!in || !out || n < 0) throw InvalidArgument("bad arguments");
    DeviceGuard device_guard(p->device);
    DevBuf a, b;
    a.reserve((size_t)n * 8 + 8);
    b.reserve((size_t)n * 8 + 8);
    HIP_CHECK(hipMemcpyAsync(a.ptr, in, (size_t)n * 8, hipMemcpyHostToDevice, p->stream));
    rip::launch_atan_probe(a.as<double>(), b.as<double>(), n, p->stream);
    HIP_CHECK(hipMemcpyAsync(out, b.ptr, (size_t)n * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_CHECK(hipStreamSynchronize(p->stream));
  });
}

rip_status rip_get_vignetting_mask(rip_pipeline* p, int rows, int cols, float* out, size_t capacity_floats) {
  if (!p) return RIP_ERR_INVALID_ARGUMENT;
  return guarded(p, [&] {
    if (rows < 1 || cols < 1 || !out || capacity_floats < (size_t)rows * cols) throw InvalidArgument("vignetting mask: bad size / buffer");
    std::vector<float> m;
    rip::build_vignette_mask(rows, cols, p->m.vig_scale, p->m.vig_a2, p->m.vig_a4, m, p->fp_contract);
    std::memcpy(out, m.data(), m.size() * sizeof(float));
  });
}

rip_status rip_set_tunable(rip_pipeline* p, const char* name, int value) {
  return guarded(p, [&] {
    need(p);
    if (!name) throw InvalidArgument("tunable name is null");
    if (value < 0) throw InvalidArgument("tunable values are non-negative");
    const std::string n = name;
    const rip::Tunables dflt;
    rip::Tunables& t = p->tn;
    if (n == "chain_blocks") t.chain_blocks = value;
    else if (n == "chain_frames") t.chain_frames = value;
    else if (n == "stats_blocks") t.stats_blocks = value > 0 ? value : dflt.stats_blocks;
    else if (n == "remap_ring") t.remap_ring = value;
    else if (n == "remap_stages") t.remap_stages = value > 0 ? value : dflt.remap_stages;
    else if (n == "remap_per_cu") t.remap_per_cu = value;
    else if (n == "remap_frames") t.remap_frames = value;
    else if (n == "remap_fused") t.remap_fused = value;
    else if (n == "remap_exp") t.remap_exp = value;
    else if (n == "remap_deal") t.remap_deal = value;
    else if (n == "chain_deal") t.chain_deal = value;
    else if (n == "chain_nt") t.chain_nt = value;
    else if (n == "remap_tiled") p->use_tiled_remap = value != 0;
    else if (n == "ccc_lds_hist_min") t.ccc_lds_hist_min = value > 0 ? value : dflt.ccc_lds_hist_min;
    else if (n == "overlap_groups") t.overlap_groups = value;
    else if (n == "overlap_mode") t.overlap_mode = value;
    else throw InvalidArgument("unknown tunable [" + n + "]");
  });
}

rip_status rip_debug_hbm_probe(rip_pipeline* p, int kind, size_t bytes, int reps, double* gbps) {
  return guarded(p, [&] {
    need_device(p);
    if (!gbps) throw InvalidArgument("null result pointer");
    if (kind < RIP_PROBE_COPY || kind > RIP_PROBE_EXPAND13_COALESCED_NT) throw InvalidArgument("unknown probe kind");
    bytes = bytes / 48 * 48;
    if (kind == RIP_PROBE_EXPAND13_COALESCED || kind == RIP_PROBE_EXPAND13_COALESCED_NT) bytes = bytes / 3072 * 3072;  // whole waves of 16-byte lanes
    if (bytes < 48 || reps < 1) throw InvalidArgument("rip_debug_hbm_probe: at least 48 bytes and one repetition");
    DeviceGuard device_guard(p->device);
    DevBuf src, dst;
    const bool expands = kind == RIP_PROBE_EXPAND13 || kind == RIP_PROBE_EXPAND13_NT || kind == RIP_PROBE_EXPAND13_WIDE || kind == RIP_PROBE_EXPAND13_WIDE_NT ||
                         kind == RIP_PROBE_EXPAND13_COALESCED || kind == RIP_PROBE_EXPAND13_COALESCED_NT;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    try {
      if (kind != RIP_PROBE_FILL) {
        src.reserve(bytes);
        HIP_CHECK(hipMemsetAsync(src.ptr, 0x3c, bytes, p->stream));
      }
      dst.reserve(expands ? 3 * bytes : bytes);
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      size_t moved = rip::launch_hbm_probe(kind, src.ptr, dst.ptr, bytes, p->stream);  // warm-up: page tables, caches, clocks
      HIP_CHECK(hipGetLastError());
      float best = 0.f;
      for (int r = 0; r < reps; r++) {
        HIP_CHECK(hipEventRecord(e0, p->stream));
        moved = rip::launch_hbm_probe(kind, src.ptr, dst.ptr, bytes, p->stream);
        HIP_CHECK(hipEventRecord(e1, p->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms > 0.f && (best == 0.f || ms < best)) best = ms;
      }
      *gbps = best > 0.f ? (double)moved / (best * 1e-3) / 1e9 : 0.0;
    } catch (...) {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      src.release();
      dst.release();
      throw;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    src.release();
    dst.release();
  });
}

int rip_get_table(rip_pipeline* p, int which, int32_t* out, int cap) {
  if (!p || !out) return -1;
  const rip::ColorTables& c = rip::color_tables();
  int n = 0;
  auto put = [&](auto* tab, int cnt) {
    n = cnt;
    for (int i = 0; i < cnt && i < cap; i++) out[i] = (int32_t)tab[i];
  };
  switch (which) {
    case 0: put(c.srgb_gamma, 256); break;
    case 1: put(c.cbrt, 3072); break;
    case 2: put(c.lab_to_yf, 512); break;
    case 3: put(c.inv_gamma, 4096); break;
    case 4: put(c.fwd, 9); break;
    case 5: put(c.inv, 9); break;
    case 6: put(c.sdiv, 256); break;
    case 7: put(c.hdiv180, 256); break;
    case 8: {
      uint8_t lut[256];
      rip::build_gamma_lut(p->m.gamma_k, lut);
      put(lut, 256);
      break;
    }
    default: return -1;
  }
  return n;
}

}  // extern "C"
#pragma GCC visibility pop
