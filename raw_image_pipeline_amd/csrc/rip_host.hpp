// rip_host.hpp -- host-side state of one pipeline: module parameters (mirroring the reference's
// eight modules), the YAML-subset reader for the three config files, and the builders of the
// constant tables / undistortion maps the HIP kernels consume.  No device code here.
#pragma once
#include "rip_tile.hpp"

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rip {

// ---------------------------------------------------------------------------------------------
// Minimal YAML reader: nested block maps, scalars, flow sequences of scalars ("[a, b, c]",
// possibly spanning lines), comments.  Enough for the reference's three file kinds
// (config/pipeline_params_example.yaml, alphasense_calib_example.yaml,
// alphasense_color_calib_example.yaml).  Throws YamlError on malformed input.
// ---------------------------------------------------------------------------------------------
struct YamlError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct YamlNode {
  enum Kind { Null, Scalar, Sequence, Map } kind = Null;
  std::string scalar;
  std::vector<std::string> seq;
  std::map<std::string, YamlNode> map;

  const YamlNode& operator[](const std::string& key) const;  // Null node when absent
  bool defined() const { return kind != Null; }
  // utils::get<T>(node, key, default) (reference utils.hpp:61-74): value or default
  bool get(const std::string& key, bool dflt) const;
  int get(const std::string& key, int dflt) const;
  double get(const std::string& key, double dflt) const;
  std::string get(const std::string& key, const std::string& dflt) const;
  std::vector<double> get_vector(const std::string& key) const;  // empty when absent / not a sequence
};
YamlNode yaml_parse(const std::string& text);
YamlNode yaml_load_file(const std::string& path);  // throws YamlError when unreadable
bool file_exists(const std::string& path);

// ---------------------------------------------------------------------------------------------
// Module parameters.  Field comments cite the reference member they mirror.
// ---------------------------------------------------------------------------------------------
struct Modules {
  bool use_gpu = false, debug = false;
  // DebayerModule (debayer.hpp:70-72): enable flag stored but ignored by apply (:38-40)
  bool debayer_enabled = true;
  std::string debayer_encoding = "auto";
  bool debayer_16bit = false;  // extension (rip_set_debayer_16bit): accept bayer_*16 instead of throwing like the reference
  // FlipModule (flip.hpp:63-66)
  bool flip_enabled = false;
  int flip_angle = 0;
  // WhiteBalanceModule (white_balance.hpp:108-116)
  bool wb_enabled = false;
  std::string wb_method = "ccc";
  double wb_percentile = 20.0, wb_bright_thr = 0.8, wb_dark_thr = 0.1;
  bool wb_temporal = true;
  // ColorCalibrationModule (color_calibration.hpp:76-81): matrix held as Matx33f
  bool cc_enabled = false, cc_available = false;
  float cc_matrix[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double cc_bias[4] = {0, 0, 0, 0};
  // GammaCorrectionModule (gamma_correction.hpp:66-70)
  bool gamma_enabled = false;
  std::string gamma_method = "custom";
  double gamma_k = 1.0;
  // VignettingCorrectionModule (vignetting_correction.hpp:52-55)
  bool vig_enabled = false;
  double vig_scale = 1.5, vig_a2 = 1e-3, vig_a4 = 1e-6;
  // ColorEnhancerModule (color_enhancer.hpp:58-60): members, after the cross-wired setters
  bool ce_enabled = false;
  double ce_value_gain = 1.0, ce_saturation_gain = 1.0, ce_hue_gain = 1.0;
  // UndistortionModule (undistortion.hpp:100-125)
  bool und_enabled = false, und_available = false;
  std::string dist_model = "none", rect_model = "none";
  double dist_K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, rect_K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double dist_D[4] = {0, 0, 0, 0}, rect_D[4] = {0, 0, 0, 0};
  double dist_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, rect_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double dist_P[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, rect_P[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  int dist_w = 320, dist_h = 240, rect_w = 320, rect_h = 240;
  double balance = 0.0, fov_scale = 1.0;
};

// Parameter loaders (throw YamlError on malformed files; a missing file is a soft failure
// that returns false, as the reference's std::cout warning paths do).
void apply_example_params(Modules& m);  // the values of config/pipeline_params_example.yaml
bool load_params_file(Modules& m, const std::string& path);             // raw_image_pipeline.cpp:44-165
bool load_camera_calibration_file(Modules& m, const std::string& path); // undistortion.cpp:157-195
bool load_color_calibration_file(Modules& m, const std::string& path);  // color_calibration.cpp:52-76
void apply_example_camera_calibration(Modules& m);  // values of config/alphasense_calib_example.yaml
void apply_example_color_calibration(Modules& m);   // values of config/alphasense_color_calib_example.yaml

// ---------------------------------------------------------------------------------------------
// Constant tables
// ---------------------------------------------------------------------------------------------
// gamma_correction.cpp:35-43
void build_gamma_lut(double k, uint8_t lut[256]);

// 8-bit Lab (OpenCV 4.2 imgproc/color_lab.cpp initLabTabs, RGB2Lab_b, Lab2RGBinteger) and
// 8-bit HSV (color_hsv.cpp RGB2HSV_b) tables.
struct ColorTables {
  uint16_t srgb_gamma[256];   // sRGBGammaTab_b
  uint16_t cbrt[3072];        // LabCbrtTab_b
  uint16_t lab_to_yf[512];    // LabToYF_b: {y, ify} pairs
  uint16_t inv_gamma[4096];   // sRGBInvGammaTab_b
  int32_t fwd[9];             // BGR -> XYZ/white, Q12, memory order B,G,R per row
  int32_t inv[9];             // XYZ -> B,G,R rows, Q12
  int32_t sdiv[256], hdiv180[256];
};
const ColorTables& color_tables();  // built once

// vignetting_correction.cpp:32-63: the float mask plane (rows x cols, tightly packed), evaluated with the
// reference's operation order (sqrt, pow(r, 2), pow(r, 4) in double); uploaded once per geometry / parameter set
void build_vignette_mask(int rows, int cols, double scale, double a2, double a4, std::vector<float>& mask, int fp_contract = 0);

// ---------------------------------------------------------------------------------------------
// Debug stage dumps (raw_image_pipeline.hpp:179-186 saveDebugImage): cv::normalize(NORM_MINMAX, 0..255) over all
// channels of the 8-bit image, then cv::imwrite to a .png
// ---------------------------------------------------------------------------------------------
// scale = 255 / (max - min) (0 when max == min), shift = -min * scale in double; per byte
// saturate_cast<uchar>(v * (float)scale + (float)shift), float multiply then add, round half to even (convertTo 8U -> 8U)
void normalize_minmax_u8(uint8_t* data, size_t n);
// 8-bit PNG, grey (channels 1) or colour (channels 3, BGR in memory -> RGB in the file, as cv::imwrite does); the zlib
// stream uses stored blocks (no compression: a debugging aid, not an encoder).  false when the file cannot be written.
bool write_png(const std::string& path, const uint8_t* data, int rows, int cols, int channels);

// ---------------------------------------------------------------------------------------------
// Fisheye undistortion (undistortion.cpp:197-238 -> OpenCV 4.2 calib3d/fisheye.cpp), double
// ---------------------------------------------------------------------------------------------
void fisheye_estimate_new_camera_matrix(const double K[9], const double D[4], int w, int h, const double R[9],
                                        double balance, int new_w, int new_h, double fov_scale, double newK[9]);
// Interleaved float2 map (x,y per destination pixel), w*h*2 floats.
// iR = (P R)^-1 of initUndistortRectifyMap (adjugate inverse), shared by the host builder and the device kernel
void fisheye_inverse_PR(const double P[9], const double R[9], double iR_out[9]);
void fisheye_init_undistort_rectify_map(const double K[9], const double D[4], const double R[9],
                                        const double P[9], int w, int h, float* map_xy);

// ---------------------------------------------------------------------------------------------
// Compiled remap plan.  cv::remap(INTER_LINEAR) quantises every map entry to 1/32 px before
// interpolating; the plan stores exactly that quantised form, organised in destination tiles of
// kRemapTileW x kRemapTileH pixels whose source footprint (a rectangle) is staged in LDS by the
// kernel.  Per destination pixel one 32-bit word: relx | rely << 11 | fx << 22 | fy << 27 with
// (relx, rely) relative to the tile's source rectangle; kRemapOutside: the pixel maps entirely
// outside the source (border constant 0); kRemapBorder: some taps fall outside -- the kernel
// takes the per-tap path on the float map for it.
// ---------------------------------------------------------------------------------------------
// kRemapTileW, kRemapTileH: rip_tile.hpp
constexpr uint32_t kRemapOutside = 0xFFFFFFFFu, kRemapBorder = 0xFFFFFFFEu;
struct RemapTile {
  int x0, y0, w, h;  // source rectangle in pixels (w == 0: no interior pixel in this tile)
};
struct RemapPlan {
  int drows = 0, dcols = 0, src_rows = 0, src_cols = 0;
  int tiles_x = 0, tiles_y = 0;
  std::vector<uint32_t> words;   // tile-major: tile t occupies words[t*1024 .. t*1024+1023], row-major inside
  std::vector<RemapTile> tiles;
  std::vector<uint32_t> border;  // (y << 16 | x) of every destination pixel marked kRemapBorder
  int max_rect_w = 0, max_rect_h = 0;
  size_t max_lds_bytes = 0;      // over tiles, for a source pixel size of 3 bytes
  bool valid = false;
};
// LDS bytes the kernel needs for a source rectangle w x h of 3-byte pixels
size_t remap_tile_lds_bytes(int x0, int w, int h);
void compile_remap_plan(RemapPlan& plan, const float* map_xy, int drows, int dcols, int src_rows, int src_cols);

// ---------------------------------------------------------------------------------------------
// CCC model (convolutional_color_constancy.cpp:116-207)
// ---------------------------------------------------------------------------------------------
struct CccModel {
  bool loaded = false;
  std::vector<float> filter_fft;  // 256*256 complex (re,im interleaved), FFT of the transposed filter
  std::vector<float> bias_fft;    // idem for the bias
  std::vector<float> filter_t, bias_t;  // transposed spatial copies (tests)
};
void ccc_build_model(CccModel& m, int w, int h, const float* filter, const float* bias);  // throws std::invalid_argument
bool ccc_load_model_file(CccModel& m, const std::string& path);                           // false: unreadable
// 128 complex twiddles of the radix-2 256-point FFT (shared with the kernels)
void fft256_twiddles(float re[128], float im[128]);
// Tables used by the device-side ccc estimator so that its float results do not depend on
// device transcendental implementations: ln(i) for i in 0..255 (ln 0 = -inf), the value of
// a histogram bin after n sequential float additions of 1/97200, and exp(-(k/64 + uv0)).
void ccc_build_scalar_tables(float log_tab[256], std::vector<float>& accum_tab, float exp_neg_tab[256]);

}  // namespace rip
