// Exercises include/raw_image_pipeline/raw_image_pipeline.hpp the way a reference caller would
// (raw_image_pipeline_ros.cpp:36-182 setters, :237 apply(), getters).  Built as C++14 like the reference.
// usage: facade_test host | facade_test gpu <width> <height> <out.bin>
#include <raw_image_pipeline/raw_image_pipeline.hpp>

#include <cstdio>
#include <fstream>
#include <iostream>

using raw_image_pipeline::Mat;
using raw_image_pipeline::RawImagePipeline;

// cv::Mat(rows, cols, type) with OpenCV headers, the stand-in's (rows, cols, channels) without
#ifdef RIP_HAVE_OPENCV
static Mat make_u8(int rows, int cols, int channels) { return Mat(rows, cols, CV_8UC(channels)); }
static Mat wrap_u8(int rows, int cols, int channels, uint8_t* ptr) { return Mat(rows, cols, CV_8UC(channels), ptr); }
typedef cv::Exception AssertType;
#else
static Mat make_u8(int rows, int cols, int channels) { return Mat(rows, cols, channels); }
static Mat wrap_u8(int rows, int cols, int channels, uint8_t* ptr) { return Mat(rows, cols, channels, ptr); }
typedef raw_image_pipeline::AssertionError AssertType;
#endif

static int fail(const char* what) {
  std::printf("FAIL: %s\n", what);
  return 1;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "host";
  RawImagePipeline proc(false, "", "", "");
  // setters as the ROS wrapper calls them
  proc.setDebayer(true);
  proc.setDebayerEncoding("auto");
  proc.setFlip(true);
  proc.setFlipAngle(180);
  proc.setWhiteBalance(true);
  proc.setWhiteBalanceMethod("gray_world");
  proc.setWhiteBalancePercentile(10.0);
  proc.setWhiteBalanceSaturationThreshold(0.8, 0.2);
  proc.setWhiteBalanceTemporalConsistency(false);
  proc.setColorCalibration(true);
  proc.setColorCalibrationMatrix({1.5, -0.25, 0.0, 0.125, 1.0, -0.125, 0.0, -0.5, 1.75});
  proc.setColorCalibrationBias({1.0, -2.0, 3.5});
  proc.setGammaCorrection(true);
  proc.setGammaCorrectionMethod("custom");
  proc.setGammaCorrectionK(0.8);
  proc.setVignettingCorrection(true);
  proc.setVignettingCorrectionParameters(1.5, 1e-3, 1e-6);
  proc.setColorEnhancer(false);
  proc.setUndistortion(false);
  if (!proc.isFlipEnabled() || !proc.isGammaCorrectionEnabled() || proc.isUndistortionEnabled()) return fail("flags");
  Mat m = proc.getColorCalibrationMatrix();
  if (m.rows != 3 || m.cols != 3 || m.at<double>(2, 2) != 1.75) return fail("colour matrix getter");
  if (proc.getDistDistortionModel() != "none") return fail("dist model without calibration");
  // exceptions: same types as the reference
  try {
    proc.setColorCalibrationMatrix({1.0, 2.0});
    return fail("short matrix accepted");
  } catch (const std::invalid_argument&) {
  }
  if (mode == "host") {
    try {
      Mat img = make_u8(8, 8, 1);
      std::string enc = "bayer_rggb8";
      proc.apply(img, enc);
      return fail("frame processed without a device");
    } catch (const std::runtime_error& e) {
      std::printf("expected failure: %s\n", e.what());
    }
    std::printf("facade host OK\n");
    return 0;
  }
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  Mat bayer = make_u8(h, w, 1);
  unsigned s = 12345u;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      s = s * 1664525u + 1013904223u;  // LCG, reproduced by the Python side of the test
      bayer.data[(size_t)y * bayer.step + x] = (uint8_t)(s >> 24);
    }
  std::string enc = "bayer_rggb8";
  Mat out = proc.process(bayer, enc);
  if (enc != "bgr8" || out.rows != h || out.cols != w || out.channels() != 3) return fail("process geometry");
  std::string enc2 = "bayer_rggb8";
  Mat inplace = bayer.clone();
  if (!proc.apply(inplace, enc2) || enc2 != "bgr8" || inplace.channels() != 3) return fail("apply re-seat");
  if (std::memcmp(inplace.data, out.data, (size_t)w * h * 3) != 0) return fail("apply != process");
  Mat tap = proc.getDistDebayeredImage(), col = proc.getDistColorImage(), fin = proc.getProcessedImage();
  if (tap.rows != h || col.rows != h || fin.rows != h || !proc.getRectMask().empty()) return fail("taps");
  if (std::memcmp(fin.data, out.data, (size_t)w * h * 3) != 0) return fail("processed tap");
  // streaming extension: submit() / collect(): two frames in flight, a view of the pinned result and a copy
  {
    proc.setTapDownload(RIP_TAP_DEBAYERED | RIP_TAP_COLOR);  // the taps travel with the result of the frames below
    const uint64_t t1 = proc.submit(bayer, "bayer_rggb8"), t2 = proc.submit(bayer, "bayer_rggb8");
    std::string e1, e2;
    Mat v1 = proc.collectView(t1, e1);
    if (e1 != "bgr8" || v1.rows != h || v1.cols != w || v1.channels() != 3) return fail("collectView geometry");
    if (std::memcmp(v1.data, out.data, (size_t)w * h * 3) != 0) return fail("collectView != process");
    // the taps of the collected frame as views of the slot's pinned buffers: same pixels as the copies of the process() frame
    Mat tv = proc.getDistDebayeredImageView(), cv2 = proc.getDistColorImageView(), pv = proc.getProcessedImageView();
    if (tv.rows != h || cv2.rows != h || pv.data != v1.data) return fail("tap views after collectView");
    if (std::memcmp(tv.data, tap.data, (size_t)w * h * 3) != 0 || std::memcmp(cv2.data, col.data, (size_t)w * h * 3) != 0) return fail("tap views != taps");
    Mat c2 = proc.collect(t2, e2);
    if (e2 != "bgr8" || std::memcmp(c2.data, out.data, (size_t)w * h * 3) != 0) return fail("collect != process");
    try {
      proc.collect(t2, e2);
      return fail("ticket collected twice");
    } catch (const std::invalid_argument&) {
    }
    // submitTo(): the result and the colour tap land in page-locked memory of the caller's
    uint8_t* pin_out = static_cast<uint8_t*>(rip_host_alloc((size_t)w * h * 3));
    uint8_t* pin_col = static_cast<uint8_t*>(rip_host_alloc((size_t)w * h * 3));
    if (!pin_out || !pin_col) return fail("rip_host_alloc");
    Mat mo = wrap_u8(h, w, 3, pin_out), mc = wrap_u8(h, w, 3, pin_col);
    std::string e3;
    Mat v3 = proc.collectView(proc.submitTo(bayer, "bayer_rggb8", mo, nullptr, &mc), e3);
    if (v3.data != pin_out || std::memcmp(pin_out, out.data, (size_t)w * h * 3) != 0) return fail("submitTo result");
    if (std::memcmp(pin_col, col.data, (size_t)w * h * 3) != 0) return fail("submitTo colour tap");
    // the caller's buffers are the caller's again once the ticket is collected (rip.h): the getters must not alias them --
    // they hand out the image from the slot the frame keeps held, even after the caller has scribbled over its own copy
    std::memset(pin_col, 0x5a, (size_t)w * h * 3);
    Mat after = proc.getDistColorImageView();
    if (after.data == pin_col || after.rows != h || std::memcmp(after.data, col.data, (size_t)w * h * 3) != 0) return fail("getter after submitTo");
    try {  // a pageable destination is refused
      Mat pageable = make_u8(h, w, 3);
      proc.submitTo(bayer, "bayer_rggb8", pageable);
      return fail("pageable destination accepted");
    } catch (const std::invalid_argument&) {
    }
    rip_host_free(pin_out);
    rip_host_free(pin_col);
  }
  try {  // cvtColor(BGR2Lab) on one channel: cv::Exception with OpenCV, AssertionError without
    Mat mono = make_u8(h, w, 1);
    std::memset(mono.data, 7, (size_t)w * h);
    std::string em = "mono8";
    proc.process(mono, em);
    return fail("vignetting on mono8 accepted");
  } catch (const AssertType&) {
  }
  try {
    std::string e16 = "bayer_rggb16";
    proc.process(bayer, e16);
    return fail("16-bit bayer accepted");
  } catch (const std::invalid_argument&) {
  }
  std::ofstream f(argv[4], std::ios::binary);
  f.write(reinterpret_cast<const char*>(out.data), (std::streamsize)w * h * 3);
  std::printf("facade gpu OK\n");
  return 0;
}
