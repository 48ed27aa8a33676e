// raw_image_pipeline.hpp -- header-only C++ facade with the reference's class surface
// (raw_image_pipeline/include/raw_image_pipeline/raw_image_pipeline.hpp:36-137 of
// leggedrobotics/raw_image_pipeline) on top of the C-ABI of librip_hip.so (include/rip.h).
//
// raw_image_pipeline_ros / raw_image_pipeline_python compile against this header unchanged: same
// namespace, class, method names and argument meaning.  With OpenCV headers present images are
// cv::Mat; without them (this repository's build image) a minimal raw_image_pipeline::Mat stands in.
//
// Device selection: HIP device ordinal from the environment variable RIP_DEVICE (default 0;
// -1 = parameter handling only, every frame call throws).  `use_gpu` is recorded for getters only:
// frames are always processed by the HIP kernels and follow the reference's CPU/OpenCV arithmetic.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../rip.h"

#if !defined(RIP_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define RIP_HAVE_OPENCV 1
#endif
#endif

namespace raw_image_pipeline {

#ifdef RIP_HAVE_OPENCV
using Mat = cv::Mat;
namespace detail {
inline Mat make_u8(int rows, int cols, int channels) { return Mat(rows, cols, CV_8UC(channels)); }
inline Mat make_f64(int rows, int cols) { return Mat(rows, cols, CV_64F); }
inline Mat wrap_u8(int rows, int cols, int channels, uint8_t* ptr) { return Mat(rows, cols, CV_8UC(channels), ptr); }
inline uint8_t* bytes(Mat& m) { return m.data; }
inline const uint8_t* bytes(const Mat& m) { return m.data; }
inline double* doubles(Mat& m) { return m.ptr<double>(); }
inline size_t step_of(const Mat& m) { return m.step; }
[[noreturn]] inline void throw_assert(const std::string& msg) { CV_Error(cv::Error::StsAssert, msg); }
}  // namespace detail
#else
// Minimal stand-in for cv::Mat: dense uint8 images (interleaved channels) and float64 matrices.
class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;  // bytes per row
  uint8_t* data = nullptr;
  Mat() = default;
  Mat(int r, int c, int channels, bool f64 = false)
      : rows(r), cols(c), step((size_t)c * channels * (f64 ? 8 : 1)), channels_(channels), f64_(f64),
        buf_(std::make_shared<std::vector<uint8_t>>((size_t)r * c * channels * (f64 ? 8 : 1))) {
    data = buf_->data();
  }
  // wraps caller memory without taking ownership (like cv::Mat(rows, cols, type, ptr, step))
  Mat(int r, int c, int channels, uint8_t* ptr, size_t step_bytes = 0)
      : rows(r), cols(c), step(step_bytes ? step_bytes : (size_t)c * channels), data(ptr), channels_(channels) {}
  int channels() const { return channels_; }
  bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
  bool isFloat64() const { return f64_; }
  Mat clone() const {
    Mat m(rows, cols, channels_, f64_);
    const size_t row_bytes = (size_t)cols * channels_ * (f64_ ? 8 : 1);
    for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, row_bytes);
    return m;
  }
  template <typename T>
  T& at(int r, int c) {
    return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T));
  }
  template <typename T>
  const T& at(int r, int c) const {
    return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T));
  }

 private:
  int channels_ = 1;
  bool f64_ = false;
  std::shared_ptr<std::vector<uint8_t>> buf_;
};
// what a failed OpenCV assertion (cv::Exception) maps to when OpenCV is absent
struct AssertionError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
namespace detail {
inline Mat make_u8(int rows, int cols, int channels) { return Mat(rows, cols, channels); }
inline Mat make_f64(int rows, int cols) { return Mat(rows, cols, 1, true); }
inline Mat wrap_u8(int rows, int cols, int channels, uint8_t* ptr) { return Mat(rows, cols, channels, ptr); }
inline uint8_t* bytes(Mat& m) { return m.data; }
inline const uint8_t* bytes(const Mat& m) { return m.data; }
inline double* doubles(Mat& m) { return reinterpret_cast<double*>(m.data); }
inline size_t step_of(const Mat& m) { return m.step; }
[[noreturn]] inline void throw_assert(const std::string& msg) { throw AssertionError(msg); }
}  // namespace detail
#endif

class RawImagePipeline {
 public:
  // Constructor & destructor (reference hpp:39-42)
  explicit RawImagePipeline(bool use_gpu) {
    check_create(rip_create_default(device_from_env(), use_gpu ? 1 : 0, &h_));
  }
  RawImagePipeline(bool use_gpu, const std::string& params_path, const std::string& calibration_path,
                   const std::string& color_calibration_path) {
    check_create(rip_create(device_from_env(), use_gpu ? 1 : 0, params_path.c_str(), calibration_path.c_str(),
                            color_calibration_path.c_str(), &h_));
  }
  // Not in the reference: the same two constructors with an explicit HIP device ordinal instead of $RIP_DEVICE (one process
  // that drives several cameras on several GPUs: camera_rig.hpp).
  RawImagePipeline(bool use_gpu, int device) { check_create(rip_create_default(device, use_gpu ? 1 : 0, &h_)); }
  RawImagePipeline(bool use_gpu, const std::string& params_path, const std::string& calibration_path,
                   const std::string& color_calibration_path, int device) {
    check_create(rip_create(device, use_gpu ? 1 : 0, params_path.c_str(), calibration_path.c_str(), color_calibration_path.c_str(), &h_));
  }
  ~RawImagePipeline() { rip_destroy(h_); }
  RawImagePipeline(const RawImagePipeline&) = delete;
  RawImagePipeline& operator=(const RawImagePipeline&) = delete;

  //-----------------------------------------------------------------------------
  // Main interfaces (hpp:47-50)
  //-----------------------------------------------------------------------------
  // In place: `image` is re-seated to the processed image (possibly other size / channel count),
  // `encoding` is rewritten ("bayer_*8" -> "bgr8").  Always returns true, like the reference.
  bool apply(Mat& image, std::string& encoding) {
    image = run(image, encoding);
    return true;
  }
  // Alternative pipeline that returns a copy
  Mat process(const Mat& image, std::string& encoding) { return run(image, encoding); }

  // Not in the reference: apply() split in two for a streaming caller (an image callback like
  // raw_image_pipeline_ros.cpp:219-288).  submit() enqueues upload, chain and download of one frame and returns at once;
  // collect() waits for that frame.  With one frame kept in flight -- collect(previous) after submit(current) -- the PCIe
  // transfers of neighbouring frames and the kernels overlap (rip.h: rip_submit / rip_collect).  Frames are processed in
  // submission order.  collect() returns a copy; collectView() a Mat header over the handle's pinned result buffer (no
  // copy), valid until the next collect*() on this object.
  uint64_t submit(const Mat& image, const std::string& encoding) {
    uint64_t ticket = 0;
    check(rip_submit(h_, detail::bytes(image), image.rows, image.cols, image.channels(), detail::step_of(image), encoding.c_str(), &ticket));
    return ticket;
  }
  // submit() with the destinations named by the caller (rip_submit_to): `out` -- and the taps the object keeps, when given --
  // are continuous Mats over PAGE-LOCKED memory (rip_host_alloc, cv::cuda::HostMem, cudaHostRegister'ed pages) of the
  // result's / the taps' size; the downloads are written into them directly and collectView() / the *View() getters return
  // headers over these very buffers: the clone the reference hands out without a copy.  They must stay untouched until the
  // ticket is collected.
  uint64_t submitTo(const Mat& image, const std::string& encoding, Mat& out, Mat* tap_debayered = nullptr, Mat* tap_color = nullptr) {
    uint64_t ticket = 0;
    const Mat* tap = tap_debayered ? tap_debayered : tap_color;
    check(rip_submit_to(h_, detail::bytes(image), image.rows, image.cols, image.channels(), detail::step_of(image), encoding.c_str(),
                        detail::bytes(out), (size_t)out.rows * out.cols * out.channels(), tap_debayered ? detail::bytes(*tap_debayered) : nullptr,
                        tap_color ? detail::bytes(*tap_color) : nullptr, tap ? (size_t)tap->rows * tap->cols * tap->channels() : 0, &ticket));
    return ticket;
  }
  Mat collect(uint64_t ticket, std::string& encoding) {
    // the clone the reference's process() / getters promise, through the library's split copy (rip_copy_host): a 15 MB
    // Mat::clone() on one thread takes longer than the frame's kernels and PCIe transfers together
    const Mat view = collectView(ticket, encoding);
    Mat out = detail::make_u8(view.rows, view.cols, view.channels());
    rip_copy_host(detail::bytes(out), detail::bytes(view), (size_t)view.rows * view.cols * view.channels());
    return out;
  }
  Mat collectView(uint64_t ticket, std::string& encoding) {
    int rows = 0, cols = 0, cn = 0;
    char enc[32] = {0};
    const uint8_t* view = nullptr;
    check(rip_collect(h_, ticket, nullptr, 0, &view, &rows, &cols, &cn, enc));
    encoding = enc;
    return detail::wrap_u8(rows, cols, cn, const_cast<uint8_t*>(view));
  }
  void setRingDepth(int depth) { check(rip_set_ring_depth(h_, depth)); }

  // Loaders (hpp:53-56)
  void loadParams(const std::string& file_path) { check(rip_load_params(h_, file_path.c_str())); }
  void loadCameraCalibration(const std::string& file_path) { check(rip_load_camera_calibration(h_, file_path.c_str())); }
  void loadColorCalibration(const std::string& file_path) { check(rip_load_color_calibration(h_, file_path.c_str())); }
  void initUndistortion() { check(rip_init_undistortion(h_)); }
  // not in the reference (its CCC model path is baked in at compile time)
  void loadWhiteBalanceModel(const std::string& file_path) { check(rip_load_ccc_model(h_, file_path.c_str())); }

  // Other interfaces (hpp:59-61)
  void resetWhiteBalanceTemporalConsistency() { check(rip_reset_white_balance_temporal_consistency(h_)); }
  void setGpu(bool use_gpu) { check(rip_set_gpu(h_, use_gpu)); }
  void setDebug(bool debug) { check(rip_set_debug(h_, debug)); }
  // not in the reference: contraction model of the float stages (0: baseline x86-64 OpenCV, 1: FMA-target build), rip.h
  void setFpContraction(int mode) { check(rip_set_fp_contraction(h_, mode)); }

  //-----------------------------------------------------------------------------
  // Setters (hpp:66-104)
  //-----------------------------------------------------------------------------
  void setDebayer(bool enabled) { check(rip_set_debayer(h_, enabled)); }
  void setDebayerEncoding(const std::string& encoding) { check(rip_set_debayer_encoding(h_, encoding.c_str())); }

  void setFlip(bool enabled) { check(rip_set_flip(h_, enabled)); }
  void setFlipAngle(int angle) { check(rip_set_flip_angle(h_, angle)); }

  void setWhiteBalance(bool enabled) { check(rip_set_white_balance(h_, enabled)); }
  void setWhiteBalanceMethod(const std::string& method) { check(rip_set_white_balance_method(h_, method.c_str())); }
  void setWhiteBalancePercentile(const double& percentile) { check(rip_set_white_balance_percentile(h_, percentile)); }
  void setWhiteBalanceSaturationThreshold(const double& bright_thr, const double& dark_thr) {
    check(rip_set_white_balance_saturation_threshold(h_, bright_thr, dark_thr));
  }
  void setWhiteBalanceTemporalConsistency(bool enabled) { check(rip_set_white_balance_temporal_consistency(h_, enabled)); }
  void setColorCalibration(bool enabled) { check(rip_set_color_calibration(h_, enabled)); }
  void setColorCalibrationMatrix(const std::vector<double>& m) { check(rip_set_color_calibration_matrix(h_, m.data(), (int)m.size())); }
  void setColorCalibrationBias(const std::vector<double>& b) { check(rip_set_color_calibration_bias(h_, b.data(), (int)b.size())); }
  Mat getColorCalibrationMatrix() const { return matrix(&rip_get_color_calibration_matrix, 3, 3); }
  Mat getColorCalibrationBias() const { return matrix(&rip_get_color_calibration_bias, 4, 1); }

  void setGammaCorrection(bool enabled) { check(rip_set_gamma_correction(h_, enabled)); }
  void setGammaCorrectionMethod(const std::string& method) { check(rip_set_gamma_correction_method(h_, method.c_str())); }
  void setGammaCorrectionK(const double& k) { check(rip_set_gamma_correction_k(h_, k)); }

  void setVignettingCorrection(bool enabled) { check(rip_set_vignetting_correction(h_, enabled)); }
  void setVignettingCorrectionParameters(const double& scale, const double& a2, const double& a4) {
    check(rip_set_vignetting_correction_parameters(h_, scale, a2, a4));
  }

  void setColorEnhancer(bool enabled) { check(rip_set_color_enhancer(h_, enabled)); }
  void setColorEnhancerHueGain(const double& gain) { check(rip_set_color_enhancer_hue_gain(h_, gain)); }
  void setColorEnhancerSaturationGain(const double& gain) { check(rip_set_color_enhancer_saturation_gain(h_, gain)); }
  void setColorEnhancerValueGain(const double& gain) { check(rip_set_color_enhancer_value_gain(h_, gain)); }

  void setUndistortion(bool enabled) { check(rip_set_undistortion(h_, enabled)); }
  void setUndistortionImageSize(int width, int height) { check(rip_set_undistortion_image_size(h_, width, height)); }
  void setUndistortionNewImageSize(int width, int height) { check(rip_set_undistortion_new_image_size(h_, width, height)); }
  void setUndistortionBalance(double balance) { check(rip_set_undistortion_balance(h_, balance)); }
  void setUndistortionFovScale(double fov_scale) { check(rip_set_undistortion_fov_scale(h_, fov_scale)); }
  void setUndistortionCameraMatrix(const std::vector<double>& m) { check(rip_set_undistortion_camera_matrix(h_, m.data(), (int)m.size())); }
  void setUndistortionDistortionCoefficients(const std::vector<double>& c) {
    check(rip_set_undistortion_distortion_coefficients(h_, c.data(), (int)c.size()));
  }
  void setUndistortionDistortionModel(const std::string& model) { check(rip_set_undistortion_distortion_model(h_, model.c_str())); }
  void setUndistortionRectificationMatrix(const std::vector<double>& m) {
    check(rip_set_undistortion_rectification_matrix(h_, m.data(), (int)m.size()));
  }
  void setUndistortionProjectionMatrix(const std::vector<double>& m) {
    check(rip_set_undistortion_projection_matrix(h_, m.data(), (int)m.size()));
  }

  //-----------------------------------------------------------------------------
  // Getters (hpp:109-137)
  //-----------------------------------------------------------------------------
  bool isDebayerEnabled() const { return rip_is_debayer_enabled(h_) != 0; }
  bool isFlipEnabled() const { return rip_is_flip_enabled(h_) != 0; }
  bool isWhiteBalanceEnabled() const { return rip_is_white_balance_enabled(h_) != 0; }
  bool isColorCalibrationEnabled() const { return rip_is_color_calibration_enabled(h_) != 0; }
  bool isGammaCorrectionEnabled() const { return rip_is_gamma_correction_enabled(h_) != 0; }
  bool isVignettingCorrectionEnabled() const { return rip_is_vignetting_correction_enabled(h_) != 0; }
  bool isColorEnhancerEnabled() const { return rip_is_color_enhancer_enabled(h_) != 0; }
  bool isUndistortionEnabled() const { return rip_is_undistortion_enabled(h_) != 0; }

  int getDistImageHeight() const { return rip_get_dist_image_height(h_); }
  int getDistImageWidth() const { return rip_get_dist_image_width(h_); }
  std::string getDistDistortionModel() const { return str(&rip_get_dist_distortion_model); }
  Mat getDistCameraMatrix() const { return matrix(&rip_get_dist_camera_matrix, 3, 3); }
  Mat getDistDistortionCoefficients() const { return matrix(&rip_get_dist_distortion_coefficients, 1, 4); }
  Mat getDistRectificationMatrix() const { return matrix(&rip_get_dist_rectification_matrix, 3, 3); }
  Mat getDistProjectionMatrix() const { return matrix(&rip_get_dist_projection_matrix, 3, 4); }

  int getRectImageHeight() const { return rip_get_rect_image_height(h_); }
  int getRectImageWidth() const { return rip_get_rect_image_width(h_); }
  std::string getRectDistortionModel() const { return str(&rip_get_rect_distortion_model); }
  Mat getRectCameraMatrix() const { return matrix(&rip_get_rect_camera_matrix, 3, 3); }
  Mat getRectDistortionCoefficients() const { return matrix(&rip_get_rect_distortion_coefficients, 1, 4); }
  Mat getRectRectificationMatrix() const { return matrix(&rip_get_rect_rectification_matrix, 3, 3); }
  Mat getRectProjectionMatrix() const { return matrix(&rip_get_rect_projection_matrix, 3, 4); }

  Mat getDistDebayeredImage() const { return image(RIP_IMAGE_DEBAYERED); }
  Mat getDistColorImage() const { return image(RIP_IMAGE_COLOR); }
  Mat getRectMask() const { return image(RIP_IMAGE_RECT_MASK); }
  Mat getProcessedImage() const { return image(RIP_IMAGE_PROCESSED); }
  // Not in the reference: the same three images without a copy for a frame that came through collect() / collectView() --
  // the taps travel with the result into the handle's pinned host memory (rip_get_image_view), so a publisher that serialises
  // the image at once needs no second device read and no memcpy.  Valid until the next collect*() on this object; frames of
  // apply() / process() only exist on the device and come back as a copy, like the getters above.
  // setTapDownload(RIP_TAP_DEBAYERED | RIP_TAP_COLOR) makes submit() download those taps with the result (default: none).
  void setTapDownload(int mask) { check(rip_set_tap_download(h_, mask)); }
  Mat getDistDebayeredImageView() const { return image_view(RIP_IMAGE_DEBAYERED); }
  Mat getDistColorImageView() const { return image_view(RIP_IMAGE_COLOR); }
  Mat getProcessedImageView() const { return image_view(RIP_IMAGE_PROCESSED); }

  // access for callers that want the device-resident batch API (rip_apply_device)
  rip_pipeline* handle() const { return h_; }

 private:
  static int device_from_env() {
    const char* e = std::getenv("RIP_DEVICE");
    return e && *e ? std::atoi(e) : 0;
  }
  [[noreturn]] static void raise(rip_status st, const std::string& msg) {
    switch (st) {
      case RIP_ERR_INVALID_ARGUMENT: throw std::invalid_argument(msg);
      case RIP_ERR_ASSERT: detail::throw_assert(msg);
      default: throw std::runtime_error(msg);
    }
  }
  void check_create(rip_status st) {
    if (st != RIP_OK) {
      h_ = nullptr;
      raise(st, rip_last_error(nullptr));
    }
  }
  void check(rip_status st) const {
    if (st != RIP_OK) raise(st, rip_last_error(h_));
  }
  Mat run(const Mat& in, std::string& encoding) {
    int rows = 0, cols = 0, cn = 0;
    char enc[32] = {0};
    check(rip_query_output(h_, in.rows, in.cols, in.channels(), encoding.c_str(), &rows, &cols, &cn, enc));
    Mat out = detail::make_u8(rows, cols, cn);
    check(rip_apply(h_, detail::bytes(in), in.rows, in.cols, in.channels(), detail::step_of(in), encoding.c_str(), detail::bytes(out),
                    (size_t)rows * cols * cn, &rows, &cols, &cn, enc));
    encoding = enc;
    return out;
  }
  Mat image(int which) const {
    int rows = 0, cols = 0, cn = 0;
    check(rip_get_image(h_, which, nullptr, 0, &rows, &cols, &cn));
    if (rows == 0 || cols == 0) return Mat();
    Mat out = detail::make_u8(rows, cols, cn);
    check(rip_get_image(h_, which, detail::bytes(out), (size_t)rows * cols * cn, &rows, &cols, &cn));
    return out;
  }
  Mat image_view(int which) const {
    int rows = 0, cols = 0, cn = 0;
    const uint8_t* view = nullptr;
    check(rip_get_image_view(h_, which, &view, &rows, &cols, &cn));
    if (rows == 0 || cols == 0) return Mat();
    if (!view) return image(which);
    return detail::wrap_u8(rows, cols, cn, const_cast<uint8_t*>(view));
  }
  Mat matrix(rip_status (*fn)(const rip_pipeline*, double*), int rows, int cols) const {
    Mat m = detail::make_f64(rows, cols);
    check(fn(h_, detail::doubles(m)));
    return m;
  }
  std::string str(rip_status (*fn)(const rip_pipeline*, char*, size_t)) const {
    char buf[64] = {0};
    check(fn(h_, buf, sizeof(buf)));
    return buf;
  }

  rip_pipeline* h_ = nullptr;
};

}  // namespace raw_image_pipeline
