// rip_kernels.hpp -- POD parameter blocks and launch entry points of the gfx950 kernels.
// Everything here is plain data: the API layer (rip_api.cpp) fills the structs, the launchers
// (rip_chain.hip, rip_stats.hip, rip_ccc.hip, rip_remap.hip; shared device code in rip_device.hpp) enqueue on the caller's stream.  No allocation, no synchronisation.
#pragma once

#include <hip/hip_runtime.h>

#include "rip_tile.hpp"

#include <cstddef>
#include <cstdint>

namespace rip {

enum SrcKind : int { SRC_BAYER = 0, SRC_BGR = 1, SRC_RGB = 2, SRC_MONO = 3 };
enum WbMode : int { WB_NONE = 0, WB_Q8 = 1, WB_FLOAT = 2, WB_PCA = 3, WB_SIMPLE = 4 };
// Stage bits of the fused chain (compile-time specialisation key of the fast kernel)
enum StageBits : int { ST_CC = 1, ST_GAMMA = 2, ST_VIG = 4, ST_HSV = 8 };

// Constant tables in HBM (one blob per pipeline handle; copied to LDS by each workgroup).
struct DevTables {
  uint8_t gamma_lut[256];       // gamma_correction.cpp:35-43
  uint16_t lin_tab[256];        // sRGBGammaTab_b[gamma_on ? gamma_lut[v] : v]
  uint16_t cbrt_tab[3072];      // LabCbrtTab_b
  uint32_t yf_tab[256];         // LabToYF_b: y | ify << 16
  uint8_t inv_gamma[4096];      // sRGBInvGammaTab_b (values <= 255)
  int32_t sdiv[256];            // RGB2HSV_b
  int32_t hdiv[256];
  int32_t lab_fwd[9];
  int32_t lab_inv[9];
  int32_t lab_inv_pk[6];  // per output channel: (c_x & 0xffff) | (c_y << 16), c_z -- operands of v_dot2_i32_i16
  // ccc estimator
  float log_tab[256];
  float exp_neg_tab[256];
  float tw_re[128], tw_im[128];
};

// Per-frame white-balance parameters, produced on the device by the statistics kernels.
struct FrameWb {
  int q8[3];      // grey-world Q8 gains (B,G,R)
  float fg[3];    // float gains (B,G,R): ccc; SimpleWB alpha
  float pca[4];   // pca: b_c0, b_c1, r_c0, r_c1; SimpleWB: beta (B,G,R)
  int uv[2];      // ccc (x, y) used for the gains
  int uv_raw[2];  // ccc argmax before temporal filtering
  int pad[2];
};

// Raw per-frame statistics (zeroed before the first batch; the statistics kernels hand them back zeroed).  Every frame has
// kStatShards records, one 128-byte line each: a workgroup adds its part to shard blockIdx.x % 8 (the XCD it runs on, by
// dispatch order), so at most an eighth of a frame's workgroups queue up on one line's atomics -- a single frame's
// 260 workgroups x 3 sums on ONE line were most of that kernel's 12 us.
constexpr int kStatShards = 8;
struct StatShard {
  unsigned long long sum[5];  // grey-world: B,G,R ; pca: B, B^2, R, R^2, G
  unsigned int mx[3];         // pca: max B, R, G
  unsigned int done;          // shard 0 only: workgroups that have added their part (fused finalisation, StatsParams::wb_out)
  unsigned long long pad[9];
};
static_assert(sizeof(StatShard) == 128, "one shard per 128-byte line");
struct FrameStats {
  StatShard shard[kStatShards];
};

// Persistent ccc temporal state of one stream (convolutional_color_constancy.cpp:300-340).
struct CccState {
  int first_frame;
  float st_x, st_y;  // statePost
  float p_x, p_y;    // errorCovPost diagonal
  int uv_x, uv_y;
  float kf_h, kf_r;
  int temporal;
};

struct ChainParams {
  // source
  const uint8_t* src;
  size_t src_step, src_frame_stride;
  int rows, cols;  // source geometry
  int src_kind;    // SrcKind
  int bayer_ry, bayer_rx;  // position of the R sample inside the 2x2 cell
  // destination of the pointwise chain (post-flip geometry)
  uint8_t* dst;
  size_t dst_step, dst_frame_stride;
  int drows, dcols;
  int channels;  // 3, or 1 for mono pass-through
  // dst is the batch's final image and no kernel of this batch reads it again: the fast kernel stores it non-temporally
  // (+1.7 % on the chain without a remap behind it; with the remap gathering from dst the same stores cost the remap 19 %)
  int dst_streaming;
  // optional post-flip, pre-WB tap (tightly packed rows), may be null
  uint8_t* tap;
  size_t tap_frame_stride;
  int flip_angle;  // 0, 90, 180, 270
  int n_frames;
  // stages
  int wb_mode;
  const FrameWb* wb;  // [n_frames]
  int stage_bits;     // StageBits
  float cc_m[9], cc_bias[3];
  const float* vig_mask;  // [drows][dcols] vignetting mask plane (vignetting_correction.cpp:32-63), ST_VIG only
  float hsv_gain[3];  // applied to H, S, V
  const DevTables* tabs;
  // the LDS tables of the vignetting variants as one ready-made image (launch_vig_image): a workgroup copies it with 16-byte
  // loads instead of rebuilding 54 KB of tables from DevTables; null: build them in the kernel
  const uint32_t* vig_image;
  // floating-point contraction model of the float stages (rip_device.hpp RIP_FP_CONTRACT): 0 = none, 1 = fused as on FMA targets
  int fp_contract;
  // in: -1 = the remap gathers from dst next (run_batch's hint); set by launch_chain (Tunables::chain_deal): 1 = the fast kernel deals its chunks round-robin to the XCDs (fast_chunks kDeal), 0 = one contiguous range per XCD
  int deal;
};

// 16-bit Bayer extension (the reference lists bayer_*16 and rejects them, debayer.hpp:73-80 / debayer.cpp:76-78):
// bilinear demosaic on 16-bit samples + flip, BGR16 out.  Pitches and strides in BYTES.
struct Debayer16Params {
  const uint8_t* src;
  size_t src_step, src_frame_stride;
  int rows, cols, bayer_ry, bayer_rx;
  uint8_t* dst;
  size_t dst_step, dst_frame_stride;
  int drows, dcols, flip_angle, n_frames;
};

struct StatsParams {
  const uint8_t* src;
  size_t src_step, src_frame_stride;
  int rows, cols, src_kind, bayer_ry, bayer_rx;
  int n_frames;
  int mode;  // WB_Q8: grey-world sums; WB_PCA: pca sums
  unsigned thresh255;  // grey-world: cvRound(255 * thr)
  FrameStats* stats;   // [n_frames], zeroed
  unsigned* hist3;     // WB_SIMPLE: [n_frames][3][256] per-channel histograms, zeroed
  // grey-world / pca: when set, the workgroup that finishes a frame last turns its sums into the frame's gains (wb_out[frame])
  // and leaves the frame's FrameStats zeroed for the next batch -- no finalisation launch, no memset between batches
  FrameWb* wb_out;
};

struct CccGeom {
  // cv::resize(..., Size(360,270)) tables for the current post-flip geometry
  const int* xofs;       // [360]
  const short* ialpha;   // [360][2]
  const int* yofs;       // [270][2] clamped row indices
  const short* ibeta;    // [270][2]
  int area_fast;         // both scales exactly 2 -> 2x2 mean
};

struct CccParams {
  const uint8_t* src;
  size_t src_step, src_frame_stride;
  int rows, cols, src_kind, bayer_ry, bayer_rx;
  int flip_angle, drows, dcols;  // post-flip geometry the resize samples
  int n_frames;
  CccGeom geom;
  float upper, lower;            // 255*bright_thr, 255*dark_thr
  unsigned int* hist_counts;     // [n_frames][hist_split][65536]
  int hist_is_clean;             // the first n_frames * 65536 counters are known to be zero (the atomic kernel's launcher then skips its memset)
  int hist_zero_after;           // set by the launcher: the row transforms hand the counters they read back zeroed (atomic kernel)
  int hist_split;                // partial histograms per frame the LDS-histogram kernel writes (ccc_hist_split of the batch the buffer was sized for)
  const float* accum_tab;        // [97201]
  float* work;                   // [n_frames][65536] complex
  const float* filter_fft;       // [65536] complex
  const float* bias_fft;
  float* row_best;               // [n_frames][256][2] (value, col)
  int* argmax;                   // [n_frames][2] (x, y)
  const DevTables* tabs;
};

struct RemapParams {
  const uint8_t* src;
  size_t src_step, src_frame_stride;
  int rows, cols, channels;
  const float* map_xy;  // interleaved (x,y), drows x dcols
  uint8_t* dst;
  size_t dst_step, dst_frame_stride;
  int drows, dcols;
  int n_frames;
};

// Tiled remap over a compiled plan (rip_host.hpp RemapPlan): 64x16 destination tiles, source
// rectangle staged in LDS, 4 B/px plan word instead of the 8 B/px float2 map.
constexpr uint32_t kPlanOutside = 0xFFFFFFFFu, kPlanBorder = 0xFFFFFFFEu;  // == rip_host.hpp kRemap*
struct RemapTileDesc {
  int x0, y0, w, h;
};
struct RemapTiledParams {
  RemapParams base;            // src/dst geometry; base.map_xy is used for border pixels only
  const uint32_t* words;       // [tiles][1024]
  const RemapTileDesc* tiles;  // [tiles_y * tiles_x]
  int tiles_x, tiles_y;
  const uint32_t* border_list; // [n_border] (yd << 16 | xd) of the pixels marked kPlanBorder
  int n_border;
  unsigned lds_bytes;          // dynamic LDS per staging buffer (>= max over tiles)
  int double_buffer;           // set by the launcher: two staging buffers, loads of frame f+1 overlap the gather of f
  int stages;                  // set by the launcher (ring kernel): LDS ring size, prefetch distance = stages - 1
  // one-channel frames gathered straight from the caller's frames (ring kernel only): the taps go through this 256-byte
  // table (the gamma LUT; null = none) and / or are addressed in the 180-degree-flipped frame -- the whole mono8 chain
  const uint8_t* mono_lut;
  int mono_flip180;
  int deal_run;                // set by the launcher (remap_deal_run): tiles per run of the round-robin deal to the XCDs; 0 = one contiguous range per XCD
  int exp;                     // timing-only experiment switches (Tunables::remap_exp); read by -DRIP_EXPERIMENTS builds only
};

// Fisheye maps on the device (rip_maps.hip): iR = (P R)^-1 from the host, K / D of the distorted camera.
struct FisheyeMapParams {
  double K[9], D[4], iR[9];
  int w, h;
  float* map_xy;  // [h][w] interleaved (x, y)
  double* ckpt;   // scratch: fisheye_ckpt_bytes(w, h) bytes -- the row accumulators (X, Y, W) at every 32nd column
};
size_t fisheye_ckpt_bytes(int w, int h);
void launch_fisheye_maps(const FisheyeMapParams& p, hipStream_t stream);
void launch_atan_probe(const double* in, double* out, int n, hipStream_t stream);  // test hook: the kernel's atan

// Remap-plan compiler on the device (host counterpart and format: rip_host.cpp compile_remap_plan): one workgroup per
// destination tile turns the float2 map into the tile's source rectangle and its 1024 plan words; pixels whose taps
// straddle the source border are appended to `border` (any order).  counters: [0] border pixels found (may exceed
// border_cap: the caller then falls back to the host compiler), [1] largest LDS footprint of a tile's rectangle in bytes
// (remap_tile_lds_bytes), [2] / [3] largest rectangle width / height.  counters must be zeroed before the launch.
struct RemapPlanBuildParams {
  const float* map_xy;  // [drows][dcols] interleaved (x, y)
  int drows, dcols, src_rows, src_cols, tiles_x, tiles_y;
  uint32_t* words;      // [tiles][kRemapTilePx]
  RemapTileDesc* tiles; // [tiles]
  uint32_t* border;     // [border_cap] (yd << 16 | xd)
  unsigned border_cap;
  unsigned* counters;   // [4]
};
void launch_remap_plan_build(const RemapPlanBuildParams& p, hipStream_t stream);

// Launch tunables.  The defaults are the measured optima (sweeps in EXPERIMENTS.md); the environment variables named beside them
// override them for experiments, and are read ONCE, by tunables_from_env() when a handle is created -- never on a launch path.
struct Tunables {
  int chain_blocks = 0;       // RIP_CHAIN_BLOCKS: persistent 256-thread workgroups per launch; 0 = 4096
  int chain_frames = 0;       // RIP_CHAIN_FRAMES: frames per item visit; 0 = 16 for the VALU-bound stage sets, 1 for the HBM-bound ones
  int stats_blocks = 2048;    // RIP_STATS_BLOCKS
  int remap_ring = 1;         // RIP_REMAP_RING=0: register-pipelined tiled kernel instead of the LDS-DMA ring
  int remap_stages = 3;       // RIP_REMAP_STAGES: LDS ring size
  int remap_per_cu = 0;       // RIP_REMAP_PER_CU: resident workgroups per CU; 0 = 6 (ring) / 8 (tiled)
  int remap_frames = 0;       // RIP_REMAP_FRAMES: frames per tile visit (0: by the size of a source frame, 4 .. 12)
  int chain_deal = 1;         // RIP_CHAIN_DEAL: the fast chain kernel deals runs of 3 x 512 items (of 4 x 2 pixels; 2.5 row pairs of a 2448-wide frame) round-robin to the XCDs, so that the whole chip reads and writes one band of the frame (round 6: default stage set 1.270 -> 1.204 ms, Lab chain alone 2.216 -> 2.158; runs of 1 .. 6 chunks run alike, 10 / 24 lose; the run length is a compile-time constant of the kernel) -- 1: when no remap gathers from the image afterwards (in front of the remap it gains nothing and fetches a third more input), 2: always, 0: one contiguous range of chunks per XCD (rounds 1-5)
  int remap_deal = 4;         // RIP_REMAP_DEAL: how the ring kernels deal the tiles to the eight XCDs -- k > 0: runs of about k tile rows round-robin (all XCDs work on one band of the image; round 6: -5 % at 4 frames per visit, -12 % with 6-8; 1, 2 and 4 tile rows run alike, 4 fetches least: 7.58 / 7.11 / 6.54 GB per 256 frames at the L2s), 0: one contiguous range of tiles per XCD (rounds 1-5)
  int remap_exp = 0;          // RIP_REMAP_EXP: bit mask of timing-only experiments (wrong pixels), honoured by -DRIP_EXPERIMENTS builds only (tools/probes/remap_exp_probe.py)
  int remap_fused = 1;        // RIP_REMAP_FUSED=0: never run the chain inside the remap's tiles (rip_fused.hip)
  int chain_nt = -1;          // RIP_CHAIN_NT: non-temporal stores of the fused chain for batches of >= 8 frames; -1 = always (round 5: also when the remap reads the image back), 0 = never, 1 = only when no kernel of the batch reads the image again (rounds 3-4)
  int ccc_lds_hist_min = 12;  // RIP_CCC_LDS_HIST_MIN: smallest batch that takes the LDS histogram (round 4, with 4 workgroups per frame: ms per batch of 8 / 16 / 32 / 47 frames, atomic kernel vs LDS: 0.069 / 0.093 / 0.142 / 0.191 vs 0.077 / 0.081 / 0.093 / 0.108)
  int overlap_groups = 0;     // RIP_OVERLAP_GROUPS: frame groups a batch is split into on the handle's internal streams (rip_api.cpp run_batch); 0 / 1 = off (the measured optimum)
  int overlap_mode = 1;       // RIP_OVERLAP_MODE: 1 = remap(g) beside stats(g+1) + chain(g+1); 2 = beside stats(g+1) only (the chain waits)
  int debug_occupancy = 0;    // RIP_DEBUG_OCC: print the residency of every chain variant launched (development aid)
};
Tunables tunables_from_env();  // rip_api.cpp; called by rip_create

// ---- launchers (asynchronous on `stream`) -------------------------------------------------------
// Returns false (and launches nothing) when the geometry does not qualify for the tiled kernel.
// dry_run: only answer (the API decides with it whether a mono8 chain can be folded into the gather).
bool launch_remap_tiled(const RemapTiledParams& p, const Tunables& tn, hipStream_t stream, bool dry_run = false);
// tiles per run of the round-robin deal (RemapTiledParams::deal_run) for a plan of tiles_x x tiles_y tiles; 0 = contiguous ranges
int remap_deal_run(int tiles_x, int tiles_y, const Tunables& tn);
// rip_fused.hip: debayer + memory-rate stages + remap in one kernel over the Bayer frames (p.base.src); false when the
// configuration / geometry does not qualify (nothing launched).  dry_run: only answer.
bool launch_remap_fused(const RemapTiledParams& p, const ChainParams& c, int max_rect_w, int max_rect_h, const Tunables& tn, hipStream_t stream,
                        bool dry_run);
void launch_chain(const ChainParams& p, const Tunables& tn, hipStream_t stream);
// the same launchers compiled under the contracted model (the translation units built with -DRIP_FP_CONTRACT=1);
// launch_chain / launch_remap_fused hand over to them when ChainParams::fp_contract == 1
void launch_chain_fc1(const ChainParams& p, const Tunables& tn, hipStream_t stream);
bool launch_remap_fused_fc1(const RemapTiledParams& p, const ChainParams& c, int max_rect_w, int max_rect_h, const Tunables& tn, hipStream_t stream,
                            bool dry_run);
void launch_debayer16(const Debayer16Params& p, hipStream_t stream);
// builds the image ChainParams::vig_image points to (vig_image_bytes() bytes) from the handle's tables
size_t vig_image_bytes();
void launch_vig_image(const DevTables* tabs, uint32_t* image, hipStream_t stream);
void launch_stats(const StatsParams& p, const Tunables& tn, hipStream_t stream);
// Returns false when the histogram launch failed: nothing after it was enqueued and the caller must not run the
// state-advancing finalisation on stale data.
// hist_left_clean (optional): set to 1 when the histogram counters of this batch are zero again once the launches have run
// (the global-atomic kernel's path: the row transforms zero what they read), else 0.
bool launch_ccc_estimate(const CccParams& p, const Tunables& tn, hipStream_t stream, int* hist_left_clean = nullptr);
// Small batches (single frames above all: every dependent launch costs ~5 us of an estimator that takes ~45): the per-frame
// argmax over the 256 row maxima is done by the finalisation kernel itself instead of a launch of its own.
inline bool ccc_argmax_in_finalize(int n_frames) { return n_frames <= 8; }
// Partial histograms per frame the LDS-histogram kernel may write for a batch of n frames (one 1024-thread workgroup each:
// fewer than 128 frames would leave most CUs idle with one workgroup per frame); the histogram buffer holds n * this * 65536 counters.
inline int ccc_hist_split(int n_frames) { return n_frames >= 192 ? 1 : (n_frames >= 96 ? 2 : 4); }
// rip_probe.hip: one launch of a streaming microbenchmark (rip_debug_hbm_probe); returns the bytes it moves, 0 = unknown kind
size_t launch_hbm_probe(int kind, const void* src, void* dst, size_t bytes, hipStream_t stream);
// Turns raw statistics into FrameWb (grey-world / pca) or runs the ccc temporal filter + gains.
void launch_wb_finalize(int mode, const FrameStats* stats, const int* ccc_argmax, CccState* ccc_state,
                        const DevTables* tabs, FrameWb* out, int n_frames, hipStream_t stream,
                        const unsigned* simple_hist = nullptr, float simple_p = 0.f, int simple_total = 0,
                        const float* ccc_row_best = nullptr, int* ccc_argmax_out = nullptr);
// Returns false (and launches nothing) when a pitch or frame size exceeds the kernels' 32-bit addressing.
bool launch_remap(const RemapParams& p, hipStream_t stream);
// Which code path launch_chain would pick (for tests / DESIGN.md): 1 fast, 0 generic.
int chain_uses_fast_path(const ChainParams& p);
// bgr8 / rgb8 frames that qualify for the 4-px-per-lane colour kernels (rip_chain.hip)
bool color_fast_geometry(const uint8_t* src, size_t step, size_t frame_stride, int rows, int cols, int kind);

}  // namespace rip
