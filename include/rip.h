/*
 * rip.h -- C-ABI of the MI355X-native RAW image pipeline (librip_hip.so).
 *
 * Drop-in boundary for raw_image_pipeline::RawImagePipeline (reference
 * raw_image_pipeline/include/raw_image_pipeline/raw_image_pipeline.hpp:36-137, "hpp" below;
 * implementation raw_image_pipeline/src/raw_image_pipeline/raw_image_pipeline.cpp, "cpp").
 * One rip_pipeline == one reference RawImagePipeline object == one camera stream; like the
 * reference it is not re-entrant (raw_image_pipeline_ros.cpp:14: one spinner thread per
 * node).  Plain pointers and sizes only; no C++/torch/OpenCV types cross this boundary.
 *
 * Every per-frame stage runs as hand-written HIP kernels on gfx950; there is no CPU
 * fallback.  Results follow the reference's CPU/OpenCV path (the parity target named by
 * BASELINE.json), whatever `use_gpu` says.
 *
 * Errors: every function returns a rip_status; rip_last_error() gives the message.  The
 * C++ facade (include/raw_image_pipeline/raw_image_pipeline.hpp) maps
 * RIP_ERR_INVALID_ARGUMENT -> std::invalid_argument (reference debayer.cpp:76-78,
 * white_balance.hpp:82-84), RIP_ERR_ASSERT -> the cv::Exception an OpenCV assert would
 * have raised, RIP_ERR_IO -> YAML::Exception-class failures, the rest -> std::runtime_error.
 */
#ifndef RIP_H
#define RIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rip_pipeline rip_pipeline;

typedef enum {
  RIP_OK = 0,
  RIP_ERR_INVALID_ARGUMENT = 1, /* unsupported encoding / method / bad vector length */
  RIP_ERR_ASSERT = 2,           /* input the reference's OpenCV call would assert on */
  RIP_ERR_IO = 3,               /* malformed YAML / model file */
  RIP_ERR_DEVICE = 4,           /* HIP runtime failure, no device */
  RIP_ERR_CAPACITY = 5          /* caller's output buffer too small */
} rip_status;

/* Image taps kept per frame (reference keeps all three, unconditionally):
 *   DEBAYERED  FlipModule::image_  (flip.cpp:60-62)   -> getDistDebayeredImage (cpp:222-224)
 *   COLOR      UndistortionModule::dist_image_ (undistortion.cpp:247-249) -> getDistColorImage
 *   PROCESSED  RawImagePipeline::image_ (hpp:174-177) -> getProcessedImage */
enum { RIP_TAP_DEBAYERED = 1, RIP_TAP_COLOR = 2, RIP_TAP_PROCESSED = 4 };
enum { RIP_IMAGE_DEBAYERED = 0, RIP_IMAGE_COLOR = 1, RIP_IMAGE_PROCESSED = 2, RIP_IMAGE_RECT_MASK = 3 };

/* Device ordinal for a handle that only manages parameters (setters, getters, loaders, host-side
 * tables and maps) -- used by the CPU-only tests.  Every frame call on it fails with
 * RIP_ERR_DEVICE: there is no CPU execution path. */
#define RIP_DEVICE_NONE (-1)

/* ---- lifetime ------------------------------------------------------------------------ */
/* RawImagePipeline(bool use_gpu) / RawImagePipeline(use_gpu, params, calib, color_calib)
 * (hpp:39-41, cpp:16-40).  NULL or "" paths: parameters take the loader defaults of
 * cpp:44-165, no camera calibration, identity colour calibration *not available* (the
 * reference bakes source-tree default paths in at compile time, cpp:9-12; this library
 * ships no data files, so an empty params / colour-calibration path selects the VALUES of the
 * reference's example files instead).  `device` is the HIP device ordinal. */
rip_status rip_create(int device, int use_gpu, const char* params_path, const char* calibration_path,
                      const char* color_calibration_path, rip_pipeline** out);
/* RawImagePipeline(bool use_gpu) (hpp:39, cpp:16-21): the values of the reference's three example
 * config files (params, 720x540 equidistant camera, colour matrix). */
rip_status rip_create_default(int device, int use_gpu, rip_pipeline** out);
void rip_destroy(rip_pipeline* p);
/* Message of the last failure on this handle (p == NULL: of the last failed rip_create on
 * this thread).  Valid until the next call on the handle. */
const char* rip_last_error(const rip_pipeline* p);
/* HIP stream (hipStream_t) all device work of this handle is enqueued on; default: the
 * null stream.  The caller keeps ownership.  A switch orders the new stream behind the work this handle left on the old
 * one (its scratch state crosses frame calls in stream order): an event is recorded on the OLD stream during this call, so the
 * stream passed to the previous rip_set_stream must still exist when the next one is made -- destroy it only afterwards. */
rip_status rip_set_stream(rip_pipeline* p, void* hip_stream);

/* ---- frame API ----------------------------------------------------------------------- */
/* bool apply(cv::Mat& image, std::string& encoding) (hpp:47, cpp:190-205) on host memory:
 * uploads `image` (rows x cols x channels, `step` bytes per row), runs the chain, downloads
 * into `out` (tightly packed, capacity in bytes) and rewrites the encoding ("bayer_*8" ->
 * "bgr8").  The reference re-seats the caller's Mat; here the caller passes the output
 * buffer, whose geometry is returned (90/270 flips swap rows/cols).  Synchronous. */
rip_status rip_apply(rip_pipeline* p, const uint8_t* image, int rows, int cols, int channels, size_t step,
                     const char* encoding, uint8_t* out, size_t out_capacity, int* out_rows, int* out_cols,
                     int* out_channels, char encoding_out[32]);
/* cv::Mat process(const cv::Mat&, std::string&) (hpp:50, cpp:182-188) is rip_apply with
 * distinct in/out buffers -- which rip_apply already requires; the facade maps both. */

/* Geometry/encoding rip_apply would produce for such an input (no device work). */
rip_status rip_query_output(rip_pipeline* p, int rows, int cols, int channels, const char* encoding,
                            int* out_rows, int* out_cols, int* out_channels, char encoding_out[32]);
/* Geometry of the two tap images for such an input: the post-flip debayered image (flip.cpp:60-62, what
 * getDistDebayeredImage returns) and the pre-undistortion colour image (undistortion.cpp:247-249) share it;
 * rows * cols * channels tightly packed bytes per frame is what rip_apply_device writes to each tap buffer. */
rip_status rip_query_taps(rip_pipeline* p, int rows, int cols, int channels, const char* encoding,
                          int* tap_rows, int* tap_cols, int* tap_channels);

/* Device-resident, batched form of apply(): n_frames consecutive frames of THIS stream,
 * already in HBM (frame f at d_in + f*in_frame_stride, rows of in_step bytes), results to
 * d_out (frame stride out_frame_stride, row pitch out_step bytes; 0 = tightly packed).
 * Asynchronous on the handle's stream; frames are processed in order (the ccc Kalman state
 * advances frame by frame).  d_tap_debayered / d_tap_color (may be NULL) receive the
 * per-frame taps, tightly packed, same frame count. */
rip_status rip_apply_device(rip_pipeline* p, const void* d_in, size_t in_step, size_t in_frame_stride,
                            int n_frames, int rows, int cols, int channels, const char* encoding, void* d_out,
                            size_t out_step, size_t out_frame_stride, void* d_tap_debayered, void* d_tap_color);

/* ---- asynchronous host-memory frames: the path a streaming caller (raw_image_pipeline_ros.cpp:219-288: one host frame
 * per image callback) uses to keep the GPU and both PCIe directions busy at once --------------------------------------
 * rip_submit() is apply() split in two.  It uploads `image` (same arguments as rip_apply), enqueues the chain and the
 * download into a PINNED result buffer owned by the handle, and returns a ticket without waiting for any of it: upload of
 * frame f + 1, kernels of frame f and download of frame f - 1 run concurrently (three HIP streams, events in between).
 * Frames are processed in submission order (the ccc Kalman state advances frame by frame, exactly as with rip_apply).
 * The handle owns `depth` frame slots (rip_set_ring_depth, default 3, 1..16): with all of them in flight one more
 * rip_submit fails with RIP_ERR_CAPACITY and changes nothing.  A pageable `image` is read before rip_submit returns
 * (the library copies it into a pinned staging buffer of the frame's slot -- it does not rely on what the HIP runtime does
 * with an asynchronous copy from pageable memory); pinned memory (rip_host_alloc(), hipHostMalloc, hipHostRegister) is read
 * asynchronously, with no staging copy, and must stay untouched until the frame's rip_collect().  Debug dumps (rip_set_debug) are written by rip_apply only.
 * Whatever the depth, at most three host frames are in flight per DEVICE (all handles together; RIP_RING_INFLIGHT, 0 = no
 * limit): rip_submit waits for the oldest one to finish first -- beyond three the runtime's downloads slow down fourfold
 * (measured: tools/probes/ring_depth_probe.py), so a deeper ring only adds slots whose results can be held longer. */
rip_status rip_submit(rip_pipeline* p, const uint8_t* image, int rows, int cols, int channels, size_t step,
                      const char* encoding, uint64_t* ticket);
/* rip_submit with the destinations given by the caller: `out` (and, for the taps the handle keeps -- rip_set_taps --
 * `tap_debayered` / `tap_color`; NULL = not wanted here) are page-locked buffers (rip_host_alloc, hipHostMalloc,
 * hipHostRegister) the downloads are written into directly.  A caller that needs every frame in memory of its own -- the
 * deep copies the reference's process() and getters hand out (cpp:182-236) -- gets them without a pinned buffer of the handle
 * in between and without a memcpy: rip_collect only waits (its *out_view is `out`; the taps' views are the buffers given
 * here).  The buffers must stay valid and untouched until the ticket is collected.  Capacities in bytes; a buffer that is
 * too small, not page-locked, or a tap the handle does not keep: an error and nothing is enqueued. */
rip_status rip_submit_to(rip_pipeline* p, const uint8_t* image, int rows, int cols, int channels, size_t step,
                         const char* encoding, uint8_t* out, size_t out_capacity, uint8_t* tap_debayered, uint8_t* tap_color,
                         size_t tap_capacity, uint64_t* ticket);
/* Waits for the frame of `ticket` (tickets of one handle may be collected in any order) and hands over the result:
 * copied into `out` (tightly packed, `out_capacity` bytes) when out != NULL, and / or as a pointer to the handle's pinned
 * buffer in *out_view when out_view != NULL -- no copy.  The view, and the taps rip_get_image returns afterwards (those of
 * the collected frame), stay valid until the next rip_collect on this handle -- or until a rip_submit finds every other slot
 * in flight and takes this one: keep at most depth - 1 frames in flight to hold on to a result while submitting.  Geometry /
 * encoding as rip_apply.  Unknown or already collected ticket: RIP_ERR_INVALID_ARGUMENT. */
rip_status rip_collect(rip_pipeline* p, uint64_t ticket, uint8_t* out, size_t out_capacity, const uint8_t** out_view,
                       int* out_rows, int* out_cols, int* out_channels, char encoding_out[32]);
/* Frames the handle keeps in flight (1..16; default 3).  Only while nothing is in flight. */
rip_status rip_set_ring_depth(rip_pipeline* p, int depth);
/* Page-locked host memory for frames handed to rip_submit / rip_apply (hipHostMalloc): uploads from it are asynchronous
 * and run at the full PCIe rate.  NULL when the allocation fails. */
void* rip_host_alloc(size_t bytes);
void rip_host_free(void* ptr);
/* memcpy for whole frames: the deep copies the reference's API hands out (process(), every image getter: cpp:182-236) of a
 * 15 MB result are slower on one host thread than the frame's kernels and PCIe transfers together; copies of 4 MB and more
 * are split over a few library threads (RIP_COPY_THREADS, default 3 beside the caller; 0 = plain memcpy).  rip_collect and
 * rip_get_image copy this way themselves; this entry is for callers that copy out of a view. */
void rip_copy_host(void* dst, const void* src, size_t bytes);

/* Image getters (hpp:134-137, cpp:222-236): copy of the tap of the most recent rip_apply
 * frame.  RIP_IMAGE_RECT_MASK is always empty (rows = cols = 0): the reference never writes
 * rect_mask_ (undistortion.cpp:150-152). */
rip_status rip_get_image(rip_pipeline* p, int which, uint8_t* out, size_t out_capacity, int* rows, int* cols,
                         int* channels);
/* The same image without a copy, for frames that came through rip_collect: the final image always, the taps named by
 * rip_set_tap_download are downloaded together with the result, so after rip_collect *view points into the handle's pinned host memory (valid as long as the
 * rip_collect view: until the next rip_collect on the handle or until a rip_submit takes the slot).  *view is NULL -- with
 * the geometry still reported -- when the image only exists on the device (frames of rip_apply; images whose download went
 * into a buffer the caller named with rip_submit_to: that buffer is the caller's again once the ticket is collected and the
 * library does not look at it any more): use rip_get_image then.
 * hpp:134-137 (getDistDebayeredImage / getDistColorImage / getProcessedImage return Mat headers, no copy either). */
rip_status rip_get_image_view(rip_pipeline* p, int which, const uint8_t** view, int* rows, int* cols, int* channels);
/* Which of the kept taps (RIP_TAP_DEBAYERED | RIP_TAP_COLOR) rip_submit ALSO downloads into pinned host memory together
 * with the result, for rip_get_image_view / a copy-free rip_get_image after rip_collect.  Default 0: the taps stay on the
 * device until a getter asks (one synchronous device read each), and a caller that only wants the final image moves no
 * extra bytes.  A front end that publishes the taps of every frame (raw_image_pipeline_ros.cpp:245-287) sets the bits of
 * the images it publishes. */
rip_status rip_set_tap_download(rip_pipeline* p, int tap_mask);
/* Which taps rip_apply materialises (default: all three, as the reference does). */
rip_status rip_set_taps(rip_pipeline* p, int tap_mask);

/* ---- loaders (hpp:53-56) ---------------------------------------------------------------- */
rip_status rip_load_params(rip_pipeline* p, const char* file_path);             /* cpp:44-165 */
rip_status rip_load_camera_calibration(rip_pipeline* p, const char* file_path); /* undistortion.cpp:157-195 */
rip_status rip_load_color_calibration(rip_pipeline* p, const char* file_path);  /* color_calibration.cpp:52-76 */
rip_status rip_init_undistortion(rip_pipeline* p);                              /* undistortion.cpp:197-238 */
/* CCC model ("default.bin" layout: int32 w, int32 h, float32 filter[h*w], bias[h*w];
 * convolutional_color_constancy.cpp:116-132).  The reference loads a file baked in at
 * compile time; this library ships none, so method "ccc" needs one of these calls. */
rip_status rip_load_ccc_model(rip_pipeline* p, const char* file_path);
rip_status rip_set_ccc_model(rip_pipeline* p, int width, int height, const float* filter, const float* bias);
/* Kalman measurement model of the ccc temporal filter.  Default (h=0, r=1) is what the
 * pipeline's one-argument ConvolutionalColorConstancyWB constructor leaves behind
 * (convolutional_color_constancy.cpp:43-46 replaces the configured filter by a default
 * cv::KalmanFilter: H = 0); (h=1, r=10) is loadModel's configuration (:186-202). */
rip_status rip_set_ccc_kalman_model(rip_pipeline* p, double h, double r);

/* ---- other interfaces (hpp:59-61) -------------------------------------------------------- */
rip_status rip_reset_white_balance_temporal_consistency(rip_pipeline* p); /* cpp:218-220 */
rip_status rip_set_gpu(rip_pipeline* p, int use_gpu);                     /* cpp:210-212; recorded only */
/* cpp:214-216.  While on, every rip_apply() of an 8-bit frame also writes the image after each of the eight modules --
 * enabled or not -- as the reference's pipeline() does (raw_image_pipeline.hpp:143-172 -> saveDebugImage :179-186: copy,
 * cv::normalize(0, 255, NORM_MINMAX), cv::imwrite): /tmp/00_debayer.png, 01_flip, 02_white_balancing, 03_color_calibration,
 * 04_gamma_correction, 05_vignetting_correction, 06_color_enhancer, 07_undistortion (.png).  The environment variable
 * RIP_DEBUG_DIR (read when the handle is created) replaces /tmp.  The fused kernel is re-run once per module prefix with the
 * gains of the real pass (outside any rip_profile_begin/end session).  The PNGs (stored, not compressed) hold the reference's
 * normalisation formula, scale / shift in double applied in float as separate multiply and add; an OpenCV built with FMA3
 * contracts that pair, so single values on a rounding tie may differ by 1 LSB from such a build.  The file names are fixed,
 * as in the reference: handles that share a directory overwrite each other's dumps (give each camera its own RIP_DEBUG_DIR).
 * A file that cannot be written does not fail the frame (cv::imwrite's result is ignored by the reference too); the paths are
 * left in rip_last_error().  rip_apply_device() never dumps. */
rip_status rip_set_debug(rip_pipeline* p, int debug);
/* Not in the reference: which OpenCV BUILD the float stages reproduce.  The colour matrix (color_calibration.cpp:93-103 ->
 * cv::gemm), the pca map (white_balance.cpp:122-127 -> cv::addWeighted), the HSV inverse (color_enhancer.cpp:44 ->
 * HSV2RGB_f) and the vignetting mask (vignetting_correction.cpp:42-43) are C expressions of the form a*b + c, which GCC and
 * Clang contract into fused multiply-adds wherever the target has them: every aarch64 build (the reference's Jetson
 * deployment, README.md:191-201), never the baseline x86-64 one.  mode 0 (default; environment RIP_FP_CONTRACT): every
 * product and every sum rounded; mode 1: the fused forms (oracle/rip_oracle.c contraction model 1).  The two differ by at
 * most 1 LSB per stage on rounding ties (PARITY.md).  Anything else: RIP_ERR_INVALID_ARGUMENT. */
rip_status rip_set_fp_contraction(rip_pipeline* p, int mode);

/* ---- setters (hpp:66-104; cpp:241-383) ---------------------------------------------------- */
rip_status rip_set_debayer(rip_pipeline* p, int enabled);                         /* hpp:66 */
rip_status rip_set_debayer_encoding(rip_pipeline* p, const char* encoding);       /* hpp:67 */
/* Extension beyond the reference, off by default.  DebayerModule lists bayer_{rggb,bggr,gbrg,grbg}16 (debayer.hpp:73-80)
 * and throws for them (debayer.cpp:76-78; so does this library).  With the extension on such frames (one channel of
 * uint16, pitches in bytes) are demosaiced with the 8-bit path's formulas on 16-bit samples -- what cv::demosaicing does
 * for CV_16UC1 -- and flipped; the result is 3 x uint16 per pixel, encoding "bgr16", rows * cols * 6 bytes.  Every other
 * stage is an 8-bit stage in the reference and must be disabled (RIP_ERR_ASSERT otherwise); no taps are kept. */
rip_status rip_set_debayer_16bit(rip_pipeline* p, int enabled);
rip_status rip_set_flip(rip_pipeline* p, int enabled);                            /* hpp:69 */
rip_status rip_set_flip_angle(rip_pipeline* p, int angle);                        /* hpp:70 */
rip_status rip_set_white_balance(rip_pipeline* p, int enabled);                   /* hpp:72 */
rip_status rip_set_white_balance_method(rip_pipeline* p, const char* method);     /* hpp:73 */
rip_status rip_set_white_balance_percentile(rip_pipeline* p, double percentile);  /* hpp:74 */
rip_status rip_set_white_balance_saturation_threshold(rip_pipeline* p, double bright_thr, double dark_thr); /* hpp:75 */
rip_status rip_set_white_balance_temporal_consistency(rip_pipeline* p, int enabled); /* hpp:76 */
rip_status rip_set_color_calibration(rip_pipeline* p, int enabled);               /* hpp:77 */
rip_status rip_set_color_calibration_matrix(rip_pipeline* p, const double* m, int n); /* hpp:78, n == 9 */
rip_status rip_set_color_calibration_bias(rip_pipeline* p, const double* b, int n);   /* hpp:79, n == 3 */
rip_status rip_set_gamma_correction(rip_pipeline* p, int enabled);                /* hpp:83 */
rip_status rip_set_gamma_correction_method(rip_pipeline* p, const char* method);  /* hpp:84 */
rip_status rip_set_gamma_correction_k(rip_pipeline* p, double k);                 /* hpp:85 */
rip_status rip_set_vignetting_correction(rip_pipeline* p, int enabled);           /* hpp:87 */
rip_status rip_set_vignetting_correction_parameters(rip_pipeline* p, double scale, double a2, double a4); /* hpp:88 */
rip_status rip_set_color_enhancer(rip_pipeline* p, int enabled);                  /* hpp:90 */
/* The reference's setters are cross-wired (color_enhancer.cpp:23-33): "hue" scales the V
 * channel and "value" scales H.  Reproduced, so a drop-in caller sees the same pixels. */
rip_status rip_set_color_enhancer_hue_gain(rip_pipeline* p, double gain);         /* hpp:91 */
rip_status rip_set_color_enhancer_saturation_gain(rip_pipeline* p, double gain);  /* hpp:92 */
rip_status rip_set_color_enhancer_value_gain(rip_pipeline* p, double gain);       /* hpp:93 */
rip_status rip_set_undistortion(rip_pipeline* p, int enabled);                    /* hpp:95 */
rip_status rip_set_undistortion_image_size(rip_pipeline* p, int width, int height);     /* hpp:96 */
rip_status rip_set_undistortion_new_image_size(rip_pipeline* p, int width, int height); /* hpp:97 */
rip_status rip_set_undistortion_balance(rip_pipeline* p, double balance);         /* hpp:98 */
rip_status rip_set_undistortion_fov_scale(rip_pipeline* p, double fov_scale);     /* hpp:99 */
rip_status rip_set_undistortion_camera_matrix(rip_pipeline* p, const double* k, int n);            /* hpp:100, n >= 9 */
rip_status rip_set_undistortion_distortion_coefficients(rip_pipeline* p, const double* d, int n);  /* hpp:101, n >= 4 */
rip_status rip_set_undistortion_distortion_model(rip_pipeline* p, const char* model);              /* hpp:102 */
rip_status rip_set_undistortion_rectification_matrix(rip_pipeline* p, const double* r, int n);     /* hpp:103, n >= 9 */
rip_status rip_set_undistortion_projection_matrix(rip_pipeline* p, const double* pm, int n);       /* hpp:104, n >= 12 */

/* ---- getters (hpp:80-81, 109-132; cpp:388-489) ---------------------------------------------- */
int rip_is_debayer_enabled(const rip_pipeline* p);               /* hpp:109 */
int rip_is_flip_enabled(const rip_pipeline* p);                  /* hpp:110 */
int rip_is_white_balance_enabled(const rip_pipeline* p);         /* hpp:111 */
int rip_is_color_calibration_enabled(const rip_pipeline* p);     /* hpp:112 */
int rip_is_gamma_correction_enabled(const rip_pipeline* p);      /* hpp:113 */
int rip_is_vignetting_correction_enabled(const rip_pipeline* p); /* hpp:114 */
int rip_is_color_enhancer_enabled(const rip_pipeline* p);        /* hpp:115 */
int rip_is_undistortion_enabled(const rip_pipeline* p);          /* hpp:116 */
int rip_get_dist_image_height(const rip_pipeline* p);            /* hpp:118 */
int rip_get_dist_image_width(const rip_pipeline* p);             /* hpp:119 */
int rip_get_rect_image_height(const rip_pipeline* p);            /* hpp:126 */
int rip_get_rect_image_width(const rip_pipeline* p);             /* hpp:127 */
/* Strings are copied into out[capacity] (NUL-terminated). */
rip_status rip_get_dist_distortion_model(const rip_pipeline* p, char* out, size_t capacity); /* hpp:120 */
rip_status rip_get_rect_distortion_model(const rip_pipeline* p, char* out, size_t capacity); /* hpp:128 */
/* Matrices are written row-major as float64, like the CV_64F Mats the reference returns:
 * 3x3 -> out[9], 1x4 -> out[4], 3x4 -> out[12]; colour calibration 3x3 / 4x1 (cv::Scalar). */
rip_status rip_get_color_calibration_matrix(const rip_pipeline* p, double out[9]);   /* hpp:80 */
rip_status rip_get_color_calibration_bias(const rip_pipeline* p, double out[4]);     /* hpp:81 */
rip_status rip_get_dist_camera_matrix(const rip_pipeline* p, double out[9]);          /* hpp:121 */
rip_status rip_get_dist_distortion_coefficients(const rip_pipeline* p, double out[4]);/* hpp:122 */
rip_status rip_get_dist_rectification_matrix(const rip_pipeline* p, double out[9]);   /* hpp:123 */
rip_status rip_get_dist_projection_matrix(const rip_pipeline* p, double out[12]);     /* hpp:124 */
rip_status rip_get_rect_camera_matrix(const rip_pipeline* p, double out[9]);          /* hpp:129 */
rip_status rip_get_rect_distortion_coefficients(const rip_pipeline* p, double out[4]);/* hpp:130 */
rip_status rip_get_rect_rectification_matrix(const rip_pipeline* p, double out[9]);   /* hpp:131 */
rip_status rip_get_rect_projection_matrix(const rip_pipeline* p, double out[12]);     /* hpp:132 */

/* ---- introspection used by tests / bench (no reference counterpart) --------------------------- */
/* Host copy of the undistortion maps (float32, map_rows x map_cols each); NULL pointers
 * just query the size. */
rip_status rip_get_undistortion_maps(rip_pipeline* p, float* map_x, float* map_y, size_t capacity_floats,
                                     int* map_rows, int* map_cols);
/* Per-frame white-balance results of the most recent device batch (D2H, synchronises):
 * for frame f, out[f*8 ..] = {gain_b, gain_g, gain_r, q8_b, q8_g, q8_r, uv_x, uv_y}. */
rip_status rip_get_white_balance_info(rip_pipeline* p, float* out, int n_frames);
/* The ccc estimator's track over the most recent device batch (D2H, synchronises): for frame f,
 * out[f*4 ..] = {raw_x, raw_y, x, y} -- the argmax of the response (convolutional_color_constancy.cpp:283-299) and the
 * position after the temporal filter that the gains were taken at (:300-340); equal without temporal consistency. */
rip_status rip_get_ccc_track(rip_pipeline* p, int* out, int n_frames);
/* Per-kernel-class timing with HIP events recorded on the handle's stream around each launch
 * (what bench.py's roofline leg reads).  rip_profile_begin arms up to max_records event pairs;
 * rip_profile_end synchronises the stream and returns, per class, the summed elapsed
 * milliseconds and the number of launches recorded. */
enum { RIP_KERNEL_STATS = 0, RIP_KERNEL_CCC = 1, RIP_KERNEL_CHAIN = 2, RIP_KERNEL_REMAP = 3, RIP_KERNEL_COUNT = 4 };
rip_status rip_profile_begin(rip_pipeline* p, int max_records);
rip_status rip_profile_end(rip_pipeline* p, double ms_sum[4], int count[4]);
/* Host-built tables the kernels use (same ids as oracle ripo_table, plus 8: gamma LUT). */
int rip_get_table(rip_pipeline* p, int which, int32_t* out, int capacity);
/* The vignetting mask plane the kernels multiply L by (VignettingCorrectionModule::precomputeVignettingMask,
 * vignetting_correction.cpp:32-63) for a rows x cols frame with the handle's current scale / a2 / a4:
 * rows * cols floats, row-major.  Host computation only; works on RIP_DEVICE_NONE handles. */
rip_status rip_get_vignetting_mask(rip_pipeline* p, int rows, int cols, float* out, size_t capacity_floats);
/* Test hook: the double-double atan the device map builder uses (rip_maps.hip), evaluated on the device for n doubles;
 * the parity tests compare it with libm's over the range fisheye maps reach. */
rip_status rip_debug_atan(rip_pipeline* p, const double* in, double* out, int n);
/* Test hook: the compiled remap plan of the current geometry (compiled now if it is not yet): info = {tiles_x, tiles_y,
 * border pixels, largest LDS footprint of a tile's source rectangle in bytes, largest rectangle width, height, 1 if the
 * plan was compiled on the device, tile width, tile height}.  Needs a loaded calibration and a device. */
rip_status rip_debug_plan_info(rip_pipeline* p, int src_rows, int src_cols, int info[9]);
/* Test hook for the debug dumps: writes image (rows x cols x channels bytes, channels 1 or 3 = BGR) to path as the PNG
 * writer of rip_set_debug does, after the reference's min-max normalisation when normalize != 0.  No device needed;
 * p may be NULL. */
rip_status rip_debug_write_png(rip_pipeline* p, const char* path, const uint8_t* image, int rows, int cols, int channels, int normalize);
/* Launch tunables of this handle (development / test hook; no reference counterpart).  The library reads its environment
 * overrides once, in rip_create(); this sets one of them afterwards.  Names: "chain_blocks", "chain_frames", "stats_blocks",
 * "remap_ring", "remap_stages", "remap_per_cu", "remap_frames", "remap_tiled", "ccc_lds_hist_min", "overlap_groups";
 * value 0 restores the built-in default where the tunable has one.  Unknown names: RIP_ERR_INVALID_ARGUMENT. */
rip_status rip_set_tunable(rip_pipeline* p, const char* name, int value);
/* Streaming microbenchmarks on the handle's device and stream (measurement hook, no reference counterpart): what this box's
 * memory system delivers to the access shapes the pipeline's kernels are made of -- bench.py reports them as
 * roofline.empirical beside the 8 TB/s spec figure.  `bytes` = size of the source stream (of the destination for FILL),
 * rounded down to a multiple of 48; the call allocates its buffers, runs one warm-up launch and `reps` timed ones (HIP events
 * on the handle's stream) and returns the BEST launch as GB/s (1e9) of bytes moved, read + written. */
enum {
  RIP_PROBE_COPY = 0,        /* 16 B per lane in, 16 B out (1 : 1) */
  RIP_PROBE_READ = 1,        /* 16 B per lane in, one dword per wave out */
  RIP_PROBE_FILL = 2,        /* 16 B per lane out */
  RIP_PROBE_EXPAND13 = 3,    /* 4 B per lane in, 12 B out: the fused chain's shape */
  RIP_PROBE_EXPAND13_NT = 4, /* the same with non-temporal stores (how the chain writes an image nothing reads again) */
  RIP_PROBE_COPY12 = 5,      /* 12 B per lane in and out: the remap's store shape fed by a contiguous read */
  RIP_PROBE_EXPAND13_WIDE = 6,    /* 16 B per lane in, 48 contiguous B out (three 16-byte stores) */
  RIP_PROBE_EXPAND13_WIDE_NT = 7, /* the same with non-temporal stores */
  RIP_PROBE_READ_NT = 8,          /* read only, eight 16-byte non-temporal loads in flight per lane */
  RIP_PROBE_EXPAND13_COALESCED = 9,    /* 16 B per lane in, three 16-byte stores out, each store instruction of a wave contiguous (1 KB) */
  RIP_PROBE_EXPAND13_COALESCED_NT = 10 /* the same with non-temporal stores */
};
rip_status rip_debug_hbm_probe(rip_pipeline* p, int kind, size_t bytes, int reps, double* gbps);
const char* rip_version(void);

#ifdef __cplusplus
}
#endif
#endif
