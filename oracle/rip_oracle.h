/*
 * rip_oracle.h -- CPU restatement ("oracle") of the per-frame image chain behind
 * RawImagePipeline::apply()/process() of leggedrobotics/raw_image_pipeline.
 *
 * THIS IS TEST INFRASTRUCTURE.  It is the checker for the HIP product path, never the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under raw_image_pipeline_amd/ links, imports or executes it.
 *
 * PARITY UNPINNED: the reference's arithmetic lives in OpenCV 4.2.0 + opencv_contrib
 * 4.2.0 (xphoto), pinned only by prose (reference README.md:180, jenkins-pipeline:2),
 * which is neither vendored under /root/reference nor installed in this image, and the
 * reference ships no tests or golden vectors for this path.  Every
 * function below restates the reference call site it cites (file:line relative to
 * /root/reference) together with the published OpenCV 4.2 algorithm for that call
 * (integer/fixed-point semantics, rounding modes, table construction).  What pins it
 * instead: closed-form known answers derivable from the reference text (gamma LUT
 * entries, flip permutations, identity no-ops, remap weights, mask values) -- see
 * tests/test_oracle_known_answers.py and tests/golden/.
 *
 * All images are tightly packed, row-major, channel-interleaved uint8 unless noted.
 * Colour order is BGR as in the reference.
 */
#ifndef RIP_ORACLE_H
#define RIP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Bayer patterns (ROS encoding names, reference debayer.cpp:45-79) ---- */
enum {
  RIPO_BAYER_RGGB = 0,
  RIPO_BAYER_GRBG = 1,
  RIPO_BAYER_GBRG = 2,
  RIPO_BAYER_BGGR = 3
};

/* Returns the pattern id for "bayer_xxxx8", -1 for anything else. */
/* Floating-point contraction model of the float stages (colour matrix, pca map, HSV inverse, vignetting mask plane):
 * 0 = none (default; what the HIP kernels implement), 1 = fused as GCC / Clang contract the source expressions on FMA
 * targets, 2 = the other association of the 3-term dot product.  Process-wide; see rip_oracle.c. */
void ripo_set_fp_contraction(int mode);
int ripo_get_fp_contraction(void);
int ripo_bayer_pattern(const char* encoding);

/* cv::demosaicing(COLOR_BayerXX2BGR) + cvtColor(RGB2BGR): net true-colour bilinear
 * demosaic (debayer.cpp:48-70).  rows, cols >= 3. */
void ripo_debayer_bilinear(const uint8_t* bayer, int rows, int cols, int pattern, uint8_t* bgr);

/* cvtColor(RGB2BGR) on a 3-channel image (debayer.cpp:72-73). In place allowed. */
/* Extension (the reference rejects bayer_*16): the same bilinear demosaic on 16-bit samples, BGR out. */
void ripo_debayer_bilinear16(const uint16_t* bayer, int rows, int cols, int pattern, uint16_t* bgr);
void ripo_swap_rb(const uint8_t* src, size_t npix, uint8_t* dst);

/* flip.cpp:37-58.  angle in {90,180,270}; any other value copies.  dst has
 * out_rows x out_cols (swapped for 90/270). */
void ripo_flip(const uint8_t* src, int rows, int cols, int cn, int angle, uint8_t* dst, int* out_rows,
               int* out_cols);

/* cv::xphoto::GrayworldWB (white_balance.cpp:59-64).  In place.  Optionally returns the
 * three channel sums and the Q8 gains (B,G,R). */
void ripo_wb_grayworld(uint8_t* bgr, size_t npix, double saturation_thr, uint64_t sums_out[3],
                       int igains_out[3]);

/* cv::xphoto::SimpleWB (white_balance.cpp:52-57), p = clipping percentile.  ab_out = {alpha,beta} x B,G,R. */
void ripo_wb_simple(uint8_t* bgr, size_t npix, double percentile, float ab_out[6]);
void ripo_simple_wb_stretch(const uint32_t hist256[256], int total, float p, float* alpha_out, float* beta_out);

/* white_balance.cpp:73-136 ("pca").  In place.  coeffs_out = {b_c0,b_c1,r_c0,r_c1}. */
void ripo_wb_pca(uint8_t* bgr, size_t npix, float coeffs_out[4]);

/* ---- Convolutional colour constancy (convolutional_color_constancy.cpp) ---- */
typedef struct ripo_ccc ripo_ccc;
/* filter/bias: row-major height x width float32 exactly as stored in the model file
 * (loadModel :116-132 transposes them; so does this constructor). width==height==256. */
ripo_ccc* ripo_ccc_create(int width, int height, const float* filter, const float* bias);
void ripo_ccc_destroy(ripo_ccc*);
void ripo_ccc_set_thresholds(ripo_ccc*, float bright_thr, float dark_thr);
void ripo_ccc_set_temporal_consistency(ripo_ccc*, int enabled);
/* Kalman measurement model: the pipeline's one-argument constructor (:43-46) replaces the
 * configured filter by a default cv::KalmanFilter (H = 0, R = I); the two-argument
 * constructor keeps loadModel's H = I, R = 10 I.  h in {0,1}. Default h=0, r=1. */
void ripo_ccc_set_kalman_model(ripo_ccc*, float h, float r);
void ripo_ccc_reset(ripo_ccc*);
/* Full balanceWhite(src,dst) (:91-113), in place on bgr.  info_out (may be NULL):
 * {argmax_x, argmax_y, used_x, used_y}; gains_out: {gain_b, gain_g, gain_r}. */
void ripo_ccc_balance(ripo_ccc*, uint8_t* bgr, int rows, int cols, int info_out[4], float gains_out[3]);
/* Pieces, exposed for tests. */
void ripo_resize_linear_8u(const uint8_t* src, int rows, int cols, int cn, uint8_t* dst, int drows,
                           int dcols);
void ripo_ccc_histogram(const ripo_ccc*, const uint8_t* small_bgr, int rows, int cols,
                        float* hist /* h*w */);
void ripo_ccc_response(const ripo_ccc*, const float* hist, float* response /* h*w */);
/* reference double-precision direct circular convolution (sanity check of the FFT path) */
void ripo_ccc_response_direct(const ripo_ccc*, const float* hist, double* response);
void ripo_ccc_gains_from_uv(int u_x, int u_y, float gains_bgr[3]);
/* The radix-2 twiddle table (128 complex float) used by ripo_ccc_response. */
void ripo_fft256_twiddles(float* re128, float* im128);

/* color_calibration.cpp:91-104.  M row-major 3x3 (doubles narrowed to float once),
 * bias BGR.  In place. */
void ripo_color_matrix(uint8_t* bgr, size_t npix, const double m[9], const double bias[3]);

/* gamma_correction.cpp:35-43 (LUT build) and :54-60 (cv::LUT). */
void ripo_gamma_lut(double k, uint8_t lut[256]);
void ripo_apply_lut(uint8_t* data, size_t nbytes, const uint8_t lut[256]);

/* vignetting_correction.cpp:32-63 (mask, float32 rows x cols) and :68-93. */
void ripo_vignetting_mask(int rows, int cols, double scale, double a2, double a4, float* mask);
void ripo_vignetting(uint8_t* bgr, int rows, int cols, const float* mask);
/* 8-bit Lab both ways (cvtColor BGR2Lab / Lab2BGR), exposed for tests. */
void ripo_bgr2lab(const uint8_t* bgr, size_t npix, uint8_t* lab);
void ripo_lab2bgr(const uint8_t* lab, size_t npix, uint8_t* bgr);
/* Table access for cross-checking the product's host-side table builder. which:
 * 0 sRGBGammaTab_b[256] u16, 1 LabCbrtTab_b[3072] u16, 2 LabToYF_b[512] u16,
 * 3 sRGBInvGammaTab_b[4096] u16, 4 fwd coeffs[9] i32, 5 inv coeffs[9] i32,
 * 6 sdiv[256] i32, 7 hdiv180[256] i32.  Returns element count, copies as int32. */
int ripo_table(int which, int32_t* out, int capacity);
int ripo_ab_to_xz(int i); /* abToXZ_b[i - minABvalue] */

/* color_enhancer.cpp:38-47: BGR2HSV (H in [0,180)), per-channel gain, HSV2BGR. */
void ripo_bgr2hsv(const uint8_t* bgr, size_t npix, uint8_t* hsv);
void ripo_hsv2bgr(const uint8_t* hsv, size_t npix, uint8_t* bgr);
void ripo_color_enhance(uint8_t* bgr, size_t npix, double h_gain, double s_gain, double v_gain);

/* undistortion.cpp:197-238: cv::fisheye::estimateNewCameraMatrixForUndistortRectify and
 * cv::fisheye::initUndistortRectifyMap (CV_32F planes). K, R 3x3 row-major, D[4]. */
void ripo_fisheye_new_camera_matrix(const double K[9], const double D[4], int img_w, int img_h,
                                    const double R[9], double balance, int new_w, int new_h,
                                    double fov_scale, double newK[9]);
void ripo_fisheye_maps(const double K[9], const double D[4], const double R[9], const double P[9],
                       int w, int h, float* map_x, float* map_y);
/* undistortion.cpp:240-245: cv::remap(INTER_LINEAR, BORDER_CONSTANT 0); dst has the map size. */
void ripo_remap_linear(const uint8_t* src, int rows, int cols, int cn, const float* map_x,
                       const float* map_y, int drows, int dcols, uint8_t* dst);

/* ---- Whole chain (raw_image_pipeline.hpp:143-172) ---- */
typedef struct {
  /* debayer: enable flag is ignored by the reference (debayer.hpp:38-40) */
  int flip_enabled, flip_angle;
  int wb_enabled;
  int wb_method; /* 0 simple(unsupported) 1 grey_world 2 learned(unsupported) 3 ccc 4 pca */
  double wb_bright_thr, wb_dark_thr;
  double wb_percentile;
  int wb_temporal_consistency;
  int cc_enabled, cc_available;
  double cc_matrix[9], cc_bias[3];
  int gamma_enabled;
  double gamma_k;
  int vig_enabled;
  double vig_scale, vig_a2, vig_a4;
  const float* vig_mask; /* optional precomputed mask (rows x cols after flip); NULL = build per call */
  int ce_enabled;
  double ce_h_gain, ce_s_gain, ce_v_gain; /* gains on the H,S,V channels as applied */
  int und_enabled; /* enabled && calibration available && model != "none" */
  const float* map_x;
  const float* map_y; /* size = rows x cols of the image entering the stage */
  int map_rows, map_cols;
  /* 1: keep the reference's schedule (4 full-frame copies, mask rebuilt when W != H) for the
   * CPU baseline; 0: same results without the redundant passes. */
  int reference_schedule;
} ripo_params;

/* Runs the chain on one frame.  out must hold rows*cols*3 bytes (or rows*cols*cn for
 * pass-through encodings).  tap_debayered / tap_color (may be NULL) receive the post-flip
 * image (flip.cpp:60-62) and the pre-undistortion image (undistortion.cpp:247-249).
 * Returns 0, or -1 for an unsupported encoding (debayer.cpp:76-78). */
int ripo_pipeline(const ripo_params* p, ripo_ccc* ccc, const uint8_t* in, int rows, int cols, int cn,
                  const char* encoding, uint8_t* out, int* out_rows, int* out_cols, int* out_cn,
                  char encoding_out[32], uint8_t* tap_debayered, uint8_t* tap_color);

#ifdef __cplusplus
}
#endif
#endif
