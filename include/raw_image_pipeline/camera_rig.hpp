// camera_rig.hpp -- one process, several cameras, several GPUs: the C++-side counterpart of the reference's deployment of
// one raw_image_pipeline_ros node per camera (raw_image_pipeline_ros/launch/raw_image_pipeline_node.launch:85; the Alphasense
// rig of BASELINE.json config 4 has eight).  Header-only, C++14, on top of the facade (raw_image_pipeline.hpp) and therefore
// of the C-ABI (rip.h).
//
// Sharding rule: camera c lives on devices[c % devices.size()] for its whole life -- its parameters,
// undistortion plan, vignetting plane and ccc Kalman state are resident there -- and cameras never exchange data: there is
// no collective on the data path, only independent streams.  Every camera owns one RawImagePipeline and one worker thread;
// the frames of one camera are processed strictly in order, different cameras concurrently (their HIP work is enqueued from
// different threads on different handles, so uploads, kernels and downloads of different cameras overlap on one device and
// run fully in parallel on different devices).
#pragma once

#include <condition_variable>
#include <deque>
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
  ~CameraRig() {
    for (auto& c : cams_) c->stop();
  }
  CameraRig(const CameraRig&) = delete;
  CameraRig& operator=(const CameraRig&) = delete;

  int size() const { return (int)cams_.size(); }
  int deviceOf(int camera) const { return cams_.at((size_t)camera)->device; }
  // The camera's pipeline object, for setters / getters / loaders.  Do not call its frame methods while frames of this
  // camera are queued in the rig (a RawImagePipeline is not re-entrant, like the reference's).
  RawImagePipeline& camera(int c) { return cams_.at((size_t)c)->pipe; }

  // Queues one frame of `camera`; the future delivers the processed image and the rewritten encoding (or rethrows what
  // apply() would have thrown).  `image` must stay valid until then.
  std::future<Result> submit(int camera, const Mat& image, const std::string& encoding) {
    return cams_.at((size_t)camera)->enqueue(image, encoding);
  }
  // One frame per camera, all cameras concurrently; returns when every camera is done.  frames.size() == size().
  std::vector<Result> process(const std::vector<Mat>& frames, const std::vector<std::string>& encodings) {
    if ((int)frames.size() != size() || (int)encodings.size() != size()) throw std::invalid_argument("CameraRig::process: one frame and one encoding per camera");
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

    Camera(bool use_gpu, const std::string& a, const std::string& b, const std::string& c, int dev) : pipe(use_gpu, a, b, c, dev), device(dev) {
      worker = std::thread([this] { run(); });
    }
    std::future<Result> enqueue(const Mat& image, const std::string& encoding) {
      auto task = std::make_shared<std::packaged_task<Result()>>([this, &image, encoding] {
        Result r;
        r.encoding = encoding;
        r.image = pipe.process(image, r.encoding);
        return r;
      });
      std::future<Result> f = task->get_future();
      {
        std::lock_guard<std::mutex> lk(m);
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
