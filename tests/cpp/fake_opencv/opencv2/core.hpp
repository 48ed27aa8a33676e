// TEST-ONLY stand-in for <opencv2/core.hpp>: the few cv:: names include/raw_image_pipeline/raw_image_pipeline.hpp touches on
// its RIP_HAVE_OPENCV branch (cv::Mat with rows / cols / data / step / channels() / ptr<T>() / at<T>() / clone() / empty(),
// CV_8UC(n), CV_64F, CV_Error, cv::Exception), with OpenCV's signatures.  OpenCV is not installable in this image, so this
// is what lets the branch a ROS / pybind11 workspace takes be compiled and exercised here (tests/test_cpp_facade.py).
// It is NOT a reference build aid and nothing outside tests/ includes it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_CN_SHIFT 3
#define CV_8U 0
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC(n) CV_MAKETYPE(CV_8U, (n))
#define CV_8UC1 CV_8UC(1)
#define CV_8UC3 CV_8UC(3)

namespace cv {
namespace Error {
enum Code { StsAssert = -215 };
}
class Exception : public std::exception {
 public:
  Exception(int c, const std::string& m) : code(c), msg(m) {}
  const char* what() const noexcept override { return msg.c_str(); }
  int code;
  std::string msg;
};
struct MatStep {
  size_t v = 0;
  operator size_t() const { return v; }
};
class Mat {
 public:
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  MatStep step;
  Mat() = default;
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type), buf_(std::make_shared<std::vector<uint8_t>>((size_t)r * c * elem())) {
    data = buf_->data();
    step.v = (size_t)c * elem();
  }
  Mat(int r, int c, int type, void* ptr, size_t step_bytes = 0) : rows(r), cols(c), data(static_cast<uint8_t*>(ptr)), type_(type) {
    step.v = step_bytes ? step_bytes : (size_t)c * elem();
  }
  int type() const { return type_; }
  int depth() const { return type_ & 7; }
  int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
  bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * elem());
    return m;
  }
  template <typename T>
  T* ptr(int r = 0) {
    return reinterpret_cast<T*>(data + (size_t)r * step);
  }
  template <typename T>
  const T* ptr(int r = 0) const {
    return reinterpret_cast<const T*>(data + (size_t)r * step);
  }
  template <typename T>
  T& at(int r, int c) {
    return ptr<T>(r)[c];
  }
  template <typename T>
  const T& at(int r, int c) const {
    return ptr<T>(r)[c];
  }

 private:
  size_t elem() const { return (size_t)channels() * (depth() == CV_64F ? 8 : 1); }
  int type_ = 0;
  std::shared_ptr<std::vector<uint8_t>> buf_;
};
}  // namespace cv
#define CV_Error(code, msg) throw cv::Exception(code, msg)
