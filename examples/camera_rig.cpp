// camera_rig.cpp -- several cameras in one C++ process on the MI355X pipeline (include/raw_image_pipeline/camera_rig.hpp).
//
//   g++ -std=c++14 -O2 -Iinclude examples/camera_rig.cpp -o camera_rig -Lraw_image_pipeline_amd -l:librip_hip.so
//       -Wl,-rpath,$PWD/raw_image_pipeline_amd -Wl,--allow-shlib-undefined -pthread          (one command line)
//   ./camera_rig <n_cameras> <width> <height> <frames_per_camera> <out_prefix> [device ...]
//
// Every camera gets its own parameters (here: a different gamma and white-balance method per camera, the way the eight
// cameras of a rig carry their own calibration), its own device (camera c -> devices[c % n]) and its own ordered stream of
// seeded synthetic Bayer frames.  The last processed frame of camera c is written to <out_prefix><c>.bin (raw BGR bytes) so
// that a test can compare it with the CPU oracle (tests/test_cpp_facade.py); the frame rate over all cameras is printed.
#include <raw_image_pipeline/camera_rig.hpp>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <fstream>

using raw_image_pipeline::CameraRig;
using raw_image_pipeline::Mat;

#ifdef RIP_HAVE_OPENCV
static Mat make_u8(int rows, int cols, int channels) { return Mat(rows, cols, CV_8UC(channels)); }
#else
static Mat make_u8(int rows, int cols, int channels) { return Mat(rows, cols, channels); }
#endif

// the test regenerates the same bytes: a 32-bit LCG seeded per (camera, frame), top byte of every state
static void fill_frame(Mat& m, uint32_t seed) {
  uint32_t s = seed;
  for (int y = 0; y < m.rows; y++)
    for (int x = 0; x < m.cols; x++) {
      s = s * 1664525u + 1013904223u;
      m.data[(size_t)y * m.step + (size_t)x] = (uint8_t)(s >> 24);
    }
}

int main(int argc, char** argv) {
  if (argc < 6) {
    std::printf("usage: %s <n_cameras> <width> <height> <frames_per_camera> <out_prefix> [device ...]\n", argv[0]);
    return 2;
  }
  const int n_cam = std::atoi(argv[1]), w = std::atoi(argv[2]), h = std::atoi(argv[3]), n_frames = std::atoi(argv[4]);
  const std::string prefix = argv[5];
  std::vector<int> devices;
  for (int i = 6; i < argc; i++) devices.push_back(std::atoi(argv[i]));
  if (devices.empty()) devices.push_back(0);
  try {
    CameraRig rig(n_cam, devices);
    for (int c = 0; c < n_cam; c++) {
      auto& p = rig.camera(c);
      p.setFlip(true);
      p.setFlipAngle(180);
      p.setWhiteBalance(true);
      p.setWhiteBalanceMethod(c % 2 ? "pca" : "gray_world");
      p.setWhiteBalanceSaturationThreshold(0.8, 0.2);
      p.setColorCalibration(true);
      p.setColorCalibrationMatrix({1.5, -0.25, 0.0, 0.125, 1.0, -0.125, 0.0, -0.5, 1.75});
      p.setColorCalibrationBias({0.0, 0.0, 0.0});
      p.setGammaCorrection(true);
      p.setGammaCorrectionMethod("custom");
      p.setGammaCorrectionK(0.7 + 0.05 * c);
      p.setVignettingCorrection(true);
      p.setVignettingCorrectionParameters(1.5, 1e-3, 1e-6);
      p.setColorEnhancer(false);
      p.setUndistortion(false);
    }
    std::vector<Mat> frames;
    std::vector<std::string> enc((size_t)n_cam, "bayer_rggb8");
    for (int c = 0; c < n_cam; c++) frames.push_back(make_u8(h, w, 1));
    std::vector<CameraRig::Result> last, last_threaded;
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < n_frames; f++) {
      for (int c = 0; c < n_cam; c++) fill_frame(frames[(size_t)c], 1000u * (uint32_t)c + (uint32_t)f + 1u);
      last = rig.process(frames, enc);  // all cameras overlapped from this thread (submit x N, collect x N), each camera's frames in order
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // the same stream once more through the threaded option (one worker per camera): identical images, rate for the record
    const auto t1 = std::chrono::steady_clock::now();
    for (int f = 0; f < n_frames; f++) {
      for (int c = 0; c < n_cam; c++) fill_frame(frames[(size_t)c], 1000u * (uint32_t)c + (uint32_t)f + 1u);
      last_threaded = rig.processThreaded(frames, enc);
    }
    const double sec_threaded = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    for (int c = 0; c < n_cam; c++) {
      const Mat &a = last[(size_t)c].image, &b = last_threaded[(size_t)c].image;
      bool same = a.rows == b.rows && a.cols == b.cols && last[(size_t)c].encoding == last_threaded[(size_t)c].encoding;
      for (int y = 0; same && y < a.rows; y++) same = std::memcmp(a.data + (size_t)y * a.step, b.data + (size_t)y * b.step, (size_t)a.cols * 3) == 0;
      if (!same) throw std::runtime_error("camera " + std::to_string(c) + ": threaded and pipelined results differ");
    }
    for (int c = 0; c < n_cam; c++) {
      const Mat& o = last[(size_t)c].image;
      std::ofstream out(prefix + std::to_string(c) + ".bin", std::ios::binary);
      for (int y = 0; y < o.rows; y++) out.write(reinterpret_cast<const char*>(o.data + (size_t)y * o.step), (std::streamsize)o.cols * 3);
      std::printf("camera %d on device %d: %dx%d %s\n", c, rig.deviceOf(c), o.cols, o.rows, last[(size_t)c].encoding.c_str());
    }
    std::printf("camera rig OK: %d cameras x %d frames in %.3f s (%.1f frames/s pipelined, %.1f frames/s threaded; host frames incl. generation)\n",
                n_cam, n_frames, sec, n_cam * n_frames / sec, n_cam * n_frames / sec_threaded);
  } catch (const std::exception& e) {
    std::printf("FAIL: %s\n", e.what());
    return 1;
  }
  return 0;
}
