// camera_rig.hpp -- one process, several cameras, several GPUs: the C++-side counterpart of the reference's deployment of
// one raw_image_pipeline_ros node per camera (raw_image_pipeline_ros/launch/raw_image_pipeline_node.launch:85; the Alphasense
// rig of BASELINE.json config 4 has eight).  Header-only, C++14, on top of the facade (raw_image_pipeline.hpp) and therefore
// of the C-ABI (rip.h).
//
// Sharding rule: camera c lives on devices[c % devices.size()] for its whole life -- its parameters,
// undistortion plan, vignetting plane and ccc Kalman state are resident there -- and cameras never exchange data: there is
// no collective on the data path, only independent streams.  Every camera owns one RawImagePipeline; the frames of one camera
// are processed strictly in order, different cameras concurrently.
//
// How the cameras overlap: process() submits every camera's frame from the calling thread (RawImagePipeline::submit =
// rip_submit: upload, kernels and download are only enqueued, on that camera's own streams) and then collects them in camera
// order -- camera c's download runs while camera c + 1 uploads and computes, on one device or on several, without a single
// thread hand-over.  The round-3 shape, one worker thread per camera around the synchronous process(), is kept as
// processThreaded() / submit(): measured on the GPU box it LOSES to plain sequential calls at small frames (4 cameras
// 640x480: 1 613 vs 3 708 frames/s -- the wake-ups cost more than the overlap gives), so it is an option, not the default.
#pragma once

#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

#include "raw_image_pipeline.hpp"

namespace raw_image_pipeline {

class CameraRig {
 public:
  struct Result {
    Mat image;
    std::string encoding;
  };

  // `devices`: HIP device ordinals to spread the cameras over (camera c -> devices[c % devices.size()]).  The three paths
  // have the meaning of the reference's four-argument constructor and apply to every camera; per-camera settings go through
  // camera(c).
  CameraRig(int n_cameras, const std::vector<int>& devices, bool use_gpu = false, const std::string& params_path = "",
            const std::string& calibration_path = "", const std::string& color_calibration_path = "") {
    if (n_cameras < 1) throw std::invalid_argument("CameraRig: at least one camera");
    if (devices.empty()) throw std::invalid_argument("CameraRig: at least one device");
    for (int c = 0; c < n_cameras; c++) {
      const int dev = devices[(size_t)c % devices.size()];
      cams_.emplace_back(new Camera(use_gpu, params_path, calibration_path, color_calibration_path, dev));
    }
  }
  ~CameraRig() = default;  // every Camera joins its own worker
  CameraRig(const CameraRig&) = delete;
  CameraRig& operator=(const CameraRig&) = delete;

  int size() const { return (int)cams_.size(); }
  int deviceOf(int camera) const { return cams_.at((size_t)camera)->device; }
  // The camera's pipeline object, for setters / getters / loaders.  Do not call its frame methods while frames of this
  // camera are queued in the rig (a RawImagePipeline is not re-entrant, like the reference's).
  RawImagePipeline& camera(int c) { return cams_.at((size_t)c)->pipe; }

  // One frame per camera, all cameras overlapped from THIS thread (rip_submit x N, then rip_collect x N in camera order);
  // returns when every camera is done.  frames.size() == size().  An exception of one camera (a bad encoding, say) is
  // rethrown after the frames already in flight have been collected, so no ticket is left behind.
  std::vector<Result> process(const std::vector<Mat>& frames, const std::vector<std::string>& encodings) {
    if ((int)frames.size() != size() || (int)encodings.size() != size()) throw std::invalid_argument("CameraRig::process: one frame and one encoding per camera");
    std::vector<uint64_t> tickets((size_t)size(), 0);
    std::vector<Result> out((size_t)size());
    std::exception_ptr failed;
    int sent = 0;
    try {
      for (; sent < size(); sent++) tickets[(size_t)sent] = cams_[(size_t)sent]->pipe.submit(frames[(size_t)sent], encodings[(size_t)sent]);
    } catch (...) {
      failed = std::current_exception();
    }
    for (int c = 0; c < sent; c++) {
      try {
        out[(size_t)c].encoding = encodings[(size_t)c];
        out[(size_t)c].image = cams_[(size_t)c]->pipe.collect(tickets[(size_t)c], out[(size_t)c].encoding);
      } catch (...) {
        if (!failed) failed = std::current_exception();
      }
    }
    if (failed) std::rethrow_exception(failed);
    return out;
  }

  // The threaded option: queues one frame of `camera` on that camera's worker thread (started on first use); the future
  // delivers the processed image and the rewritten encoding (or rethrows what apply() would have thrown).  The Mat header
  // is captured by value (reference-counted, like cv::Mat): the pixel memory must stay valid until then, the header not.
  std::future<Result> submit(int camera, const Mat& image, const std::string& encoding) {
    return cams_.at((size_t)camera)->enqueue(image, encoding);
  }
  std::vector<Result> processThreaded(const std::vector<Mat>& frames, const std::vector<std::string>& encodings) {
    if ((int)frames.size() != size() || (int)encodings.size() != size()) throw std::invalid_argument("CameraRig::processThreaded: one frame and one encoding per camera");
    std::vector<std::future<Result>> pending;
    for (int c = 0; c < size(); c++) pending.push_back(submit(c, frames[(size_t)c], encodings[(size_t)c]));
    std::vector<Result> out;
    for (auto& f : pending) out.push_back(f.get());
    return out;
  }

 private:
  struct Camera {
    RawImagePipeline pipe;
    int device;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    bool quit = false;

    Camera(bool use_gpu, const std::string& a, const std::string& b, const std::string& c, int dev) : pipe(use_gpu, a, b, c, dev), device(dev) {}
    // a Camera destroyed with its worker still joinable (a later camera's constructor threw) would end in std::terminate
    ~Camera() { stop(); }
    Camera(const Camera&) = delete;
    Camera& operator=(const Camera&) = delete;
    std::future<Result> enqueue(const Mat& image, const std::string& encoding) {
      auto task = std::make_shared<std::packaged_task<Result()>>([this, image, encoding] {
        Result r;
        r.encoding = encoding;
        r.image = pipe.process(image, r.encoding);
        return r;
      });
      std::future<Result> f = task->get_future();
      {
        std::lock_guard<std::mutex> lk(m);
        if (!worker.joinable()) worker = std::thread([this] { run(); });  // the threaded mode pays for its thread only when used
        jobs.emplace_back([task] { (*task)(); });
      }
      cv.notify_one();
      return f;
    }
    void run() {
      for (;;) {
        std::function<void()> job;
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [this] { return quit || !jobs.empty(); });
          if (jobs.empty()) return;  // quit and drained
          job = std::move(jobs.front());
          jobs.pop_front();
        }
        job();
      }
    }
    void stop() {
      {
        std::lock_guard<std::mutex> lk(m);
        quit = true;
      }
      cv.notify_one();
      if (worker.joinable()) worker.join();
    }
  };
  std::vector<std::unique_ptr<Camera>> cams_;
};

}  // namespace raw_image_pipeline
