// rip_chain.hip -- the fused per-pixel chain (debayer -> flip -> white-balance gains -> 3x3 -> gamma -> vignetting -> HSV):
// ONE kernel, 1 B/px read, 3 B/px written, tables in LDS, no intermediate image between the stages.
// Shared device code and the stage-by-stage reference citations: rip_device.hpp.
#define RIP_GAMMA_FOLD 1
#include "rip_device.hpp"
#include "rip_chain_dev.hpp"

#include <cstdio>

namespace rip {
namespace {

// ------------------------------------------------------------------------------------------------
// generic chain kernel: one thread per destination pixel; any input kind, size, pitch, flip
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void chain_generic_kernel(ChainParams p) {
  const int frame = blockIdx.y;
  const long long npix = (long long)p.drows * p.dcols;
  SrcView s{p.src + (size_t)frame * p.src_frame_stride, p.src_step, p.rows, p.cols, p.src_kind, p.bayer_ry, p.bayer_rx};
  GlobalTabs tb{p.tabs};
  FrameWb w;
  if (p.wb_mode != WB_NONE) w = p.wb[frame];
  float fwdf[9];
#pragma unroll
  for (int k = 0; k < 9; k++) fwdf[k] = (float)p.tabs->lab_fwd[k];
  uint8_t* dst = p.dst + (size_t)frame * p.dst_frame_stride;
  uint8_t* tap = p.tap ? p.tap + (size_t)frame * p.tap_frame_stride : nullptr;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < npix; i += (long long)gridDim.x * kBlock) {
    int yd = (int)(i / p.dcols), xd = (int)(i - (long long)yd * p.dcols);
    int ys, xs;
    unflip(p.flip_angle, p.rows, p.cols, yd, xd, ys, xs);
    int b, g, r;
    fetch_src(s, ys, xs, b, g, r);
    if (p.channels == 1) {
      // mono pass-through: only flip, gamma (cv::LUT is channel-agnostic) and remap apply
      if (tap) tap[(size_t)yd * p.dcols + xd] = (uint8_t)g;
      if (p.stage_bits & ST_GAMMA) g = p.tabs->gamma_lut[g];
      dst[(size_t)yd * p.dst_step + xd] = (uint8_t)g;
      continue;
    }
    if (tap) {
      uint8_t* t = tap + ((size_t)yd * p.dcols + xd) * 3;
      t[0] = (uint8_t)b;
      t[1] = (uint8_t)g;
      t[2] = (uint8_t)r;
    }
    pointwise<-1, -1>(p, w, tb, fwdf, p.tabs->lab_inv_pk, (p.stage_bits & ST_VIG) ? p.vig_mask[(size_t)yd * p.dcols + xd] : 1.0f, b, g, r);
    uint8_t* o = dst + (size_t)yd * p.dst_step + (size_t)xd * 3;
    o[0] = (uint8_t)b;
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)r;
  }
}


// ------------------------------------------------------------------------------------------------
// 16-bit Bayer extension: one thread per destination pixel (a feature path, not a tuned one): the same two-tap / four-tap
// rounding averages and border rule as debayer_at on 16-bit samples (cv::demosaicing's Bayer2RGB_Invoker<ushort>), flip.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void debayer16_kernel(Debayer16Params p) {
  const int frame = blockIdx.y;
  const uint8_t* src = p.src + (size_t)frame * p.src_frame_stride;
  uint8_t* dst = p.dst + (size_t)frame * p.dst_frame_stride;
  const long long npix = (long long)p.drows * p.dcols;
  auto at = [&](int y, int x) { return (int)*reinterpret_cast<const uint16_t*>(src + (size_t)y * p.src_step + (size_t)x * 2); };
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < npix; i += (long long)gridDim.x * kBlock) {
    const int yd = (int)(i / p.dcols), xd = (int)(i - (long long)yd * p.dcols);
    int ys, xs;
    unflip(p.flip_angle, p.rows, p.cols, yd, xd, ys, xs);
    const int y = clampi(ys, 1, p.rows - 2), x = clampi(xs, 1, p.cols - 2);  // border: the interior formula at the clamped position
    const int dy = (y - p.bayer_ry) & 1, dx = (x - p.bayer_rx) & 1;           // (0,0): R site, (1,1): B site
    const int c = at(y, x);
    int b, g, r;
    if (dy != dx) {
      const int h = (at(y, x - 1) + at(y, x + 1) + 1) >> 1, v = (at(y - 1, x) + at(y + 1, x) + 1) >> 1;
      g = c;
      r = dy == 0 ? h : v;
      b = dy == 0 ? v : h;
    } else {
      const int x4 = (at(y, x - 1) + at(y, x + 1) + at(y - 1, x) + at(y + 1, x) + 2) >> 2;
      const int d4 = (at(y - 1, x - 1) + at(y - 1, x + 1) + at(y + 1, x - 1) + at(y + 1, x + 1) + 2) >> 2;
      g = x4;
      r = dy == 0 ? c : d4;
      b = dy == 0 ? d4 : c;
    }
    uint16_t* o = reinterpret_cast<uint16_t*>(dst + (size_t)yd * p.dst_step + (size_t)xd * 6);
    o[0] = (uint16_t)b;
    o[1] = (uint16_t)g;
    o[2] = (uint16_t)r;
  }
}

// The chunk and frame loops of chain_fast_kernel.  PLAIN (decided once per launch, wave-uniform): no debayered tap is
// requested and the colour bias is all zero -- the configuration every benchmark and the reference's default parameter sets
// run; the loop body then carries no test for either.  Round 5: the per-frame body used to re-derive five wave-uniform
// conditions through v_cndmask / v_cmp pairs, evaluate the byte merges of two Bayer patterns behind a four-way switch and
// reverse the six planar dwords for the 180-degree flip -- 96 issue cycles per item that are gone: the pattern's column
// parity and the flip are selectors of the demosaic's own v_perm_b32 (debayer_tile_sel), the row parity is one scalar
// branch around four v_swap_b32.
template <int BITS, int WB, int NT, bool PLAIN>
__device__ __forceinline__ void fast_chunks(const ChainParams& p, const ItemMap& im, const int items_per_frame, const FastTabs<BITS>& tb,
                                            const CcRegs& cc, const HsvRegs& hr) {
  // Persistent workgroups: the LDS tables are loaded once and amortised over many chunks of
  // NT items.  Block b runs on XCD b % 8 (observed dispatch order; speed only), so each
  // XCD walks its own contiguous range of chunks and vertically adjacent row pairs -- which
  // share two halo rows -- hit the same L2.  Frames are the innermost loop: everything that
  // depends only on the position (item split, vignetting mask, addresses) is fetched / computed
  // once per item and reused for every frame of the batch.
  const int chunks_per_frame = (items_per_frame + NT - 1) / NT;
  // small frames do not fill the chip with one frame's chunks: blockIdx.y splits the batch
  const int f_per_group = (p.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(p.n_frames, f_begin + f_per_group);
  const int per_xcd = (chunks_per_frame + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int flip180 = p.flip_angle == 180 ? 1 : 0;
  const DemosaicSel ds = demosaic_selectors(p.bayer_ry, p.bayer_rx, flip180);
  // Round 6: runs of p.deal chunks dealt round-robin to the XCDs -- the chunks in flight on the whole chip then form ONE band of
  // the frame (as the ring remap's tiles do, rip_remap_dev.hpp TileDeal), which the memory system serves 3-5 % faster than eight
  // bands; vertically adjacent row pairs, which share two halo rows, still meet in one L2 inside a run.  p.deal == 0: contiguous.
  // (the dealt chunk number grows with ci, so the first one past the frame ends the workgroup's walk in both deals)
  // The run length is a compile-time constant (3 x 512 items in chunks of this variant; a run-time divisor costs the Lab variant
  // of config 2 eleven extra waits in its frame loop and 1.5 % of its time: the scheduler's doing, found by diffing the
  // listings); p.deal only switches the deal on (RIP_CHAIN_DEAL != 0) or back to contiguous ranges.
  constexpr int kDeal = 3 * 512 / NT;
  const bool dealt = p.deal != 0;
  for (int ci = blockIdx.x >> 3;; ci += gridDim.x >> 3) {
    int chunk;
    if (dealt) {
      const int r = ci / kDeal;
      chunk = (r * 8 + xcd) * kDeal + (ci - r * kDeal);
    } else {
      if (ci >= per_xcd) break;
      chunk = xcd * per_xcd + ci;
    }
    if (chunk >= chunks_per_frame) break;
    const int item = chunk * NT + threadIdx.x;
    if (item >= items_per_frame) continue;
    int pair, grp;
    im.split(item, pair, grp);
    const int y0 = pair * 2, x0 = grp * 4;
    const int xbase = flip180 ? p.cols - 4 - x0 : x0;
    float mask[2][4];
    unsigned dst_off[2], tap_off[2];
#pragma unroll
    for (int ly = 0; ly < 2; ly++) {
      const int yd = flip180 ? p.rows - 1 - (y0 + ly) : y0 + ly;
      dst_off[ly] = __umul24((unsigned)yd, (unsigned)p.dst_step) + (unsigned)xbase * 3u;
      tap_off[ly] = (__umul24((unsigned)yd, (unsigned)p.dcols) + (unsigned)xbase) * 3u;
      if constexpr ((BITS & ST_VIG) != 0) {
        // four consecutive floats of the mask plane: 16-byte aligned (dcols % 4 == 0, xbase % 4 == 0)
        const float4 m = *reinterpret_cast<const float4*>(p.vig_mask + (size_t)yd * p.dcols + xbase);
        mask[ly][0] = m.x;
        mask[ly][1] = m.y;
        mask[ly][2] = m.z;
        mask[ly][3] = m.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) mask[ly][k] = 1.0f;
      }
    }
    const WindowOffsets wo = window_offsets((unsigned)p.src_step, p.rows, p.cols, y0, x0);
    const unsigned src_bytes = __umul24((unsigned)(p.rows - 1), (unsigned)p.src_step) + (unsigned)p.cols;
    const unsigned dst_bytes = __umul24((unsigned)(p.drows - 1), (unsigned)p.dst_step) + (unsigned)p.dcols * 3u;
    const unsigned tap_bytes = __umul24((unsigned)p.drows, (unsigned)p.dcols) * 3u;
    const unsigned long long edge_lanes = debayer_edge_lanes(y0, x0, p.rows, p.cols);
    for (int frame = f_begin; frame < f_end; frame++) {
      const __amdgpu_buffer_rsrc_t src = frame_rsrc(p.src + (size_t)frame * p.src_frame_stride, src_bytes);
      const __amdgpu_buffer_rsrc_t dst = frame_rsrc(p.dst + (size_t)frame * p.dst_frame_stride, dst_bytes);
      FrameWb w;
      if (WB != WB_NONE) w = p.wb[frame];
      if constexpr (WB == WB_FLOAT || WB == WB_SIMPLE || WB == WB_PCA) {
        // the per-frame float gains arrive through scalar loads; as SGPR operands they would make every multiply of the
        // white balance issue at 4.3 cycles instead of 2.45 (once per frame: a handful of v_mov)
#pragma unroll
        for (int c = 0; c < 3; c++) asm volatile("" : "+v"(w.fg[c]));
#pragma unroll
        for (int c = 0; c < 4; c++) asm volatile("" : "+v"(w.pca[c]));
      }
      Window win;
      load_window(src, wo, win);
      Planar rowpx[2];
      debayer_tile_sel(win, ds, y0, x0, p.rows, p.cols, edge_lanes, rowpx);  // already mirrored for the 180-degree flip
#pragma unroll
      for (int ly = 0; ly < 2; ly++) {
        Planar v = rowpx[ly];
        constexpr bool kRawOut = BITS == 0 && WB == WB_NONE;  // pure demosaic: no per-pixel stage
        if constexpr (!PLAIN || kRawOut) {
          Pack3 raw;
          interleave4(v, raw.a, raw.b, raw.c);
          if constexpr (!PLAIN) {
            if (p.tap != nullptr) store12(frame_rsrc(p.tap + (size_t)frame * p.tap_frame_stride, tap_bytes), tap_off[ly], raw);
          }
          if constexpr (kRawOut) {
            store12(dst, dst_off[ly], raw, p.dst_streaming);
            continue;
          }
        }
        if (WB == WB_Q8) {  // grey-world gains on the packed bytes, two pixels per multiply
          v.b = gains_q8_swar(v.b, (unsigned)w.q8[0]);
          v.g = gains_q8_swar(v.g, (unsigned)w.q8[1]);
          v.r = gains_q8_swar(v.r, (unsigned)w.q8[2]);
        }
        int q[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          q[k][0] = (int)((v.b >> (8 * k)) & 0xFFu);
          q[k][1] = (int)((v.g >> (8 * k)) & 0xFFu);
          q[k][2] = (int)((v.r >> (8 * k)) & 0xFFu);
        }
        store12(dst, dst_off[ly], pointwise4<BITS, WB == WB_Q8 ? WB_NONE : WB, PLAIN ? 0 : 1>(p, w, tb, cc, hr, mask[ly], q), p.dst_streaming);
      }
    }
  }
}

template <int BITS, int WB, int NT>
__global__ __launch_bounds__(NT, fast_waves_per_simd<BITS>()) void chain_fast_kernel(ChainParams p, ItemMap im, int items_per_frame) {
  __shared__ FastTabs<BITS> tb;
  tb.template load<NT>(p.tabs, p.vig_image, p.hsv_gain);
  CcRegs cc = {};
  if constexpr ((BITS & ST_CC) != 0) cc.load(p);
  HsvRegs hr = {};
  if constexpr ((BITS & ST_HSV) != 0) hr.load(p);
  __syncthreads();
  const bool bias = (BITS & ST_CC) != 0 && (p.cc_bias[0] != 0.f || p.cc_bias[1] != 0.f || p.cc_bias[2] != 0.f);
  if (p.tap == nullptr && !bias)
    fast_chunks<BITS, WB, NT, true>(p, im, items_per_frame, tb, cc, hr);
  else
    fast_chunks<BITS, WB, NT, false>(p, im, items_per_frame, tb, cc, hr);
}


// ------------------------------------------------------------------------------------------------
// colour-input path (bgr8 / rgb8 frames: the reference's Python demo, and every compressed transport of the ROS node, which
// cv_bridge converts to bgr8, raw_image_pipeline_ros.cpp:226-231): 4 px per lane, 12-byte loads and stores, flip 0 / 180.
// Round 4: the compile-time stage sets and tables of chain_fast_kernel (FastTabs, pointwise4) instead of the round-1
// per-pixel stage functions with run-time stage tests.
// ------------------------------------------------------------------------------------------------
template <int BITS, int WB, int NT>
__global__ __launch_bounds__(NT, fast_waves_per_simd<BITS>()) void chain_color_kernel(ChainParams p, ItemMap im, int items_per_frame) {
  __shared__ FastTabs<BITS> tb;
  tb.template load<NT>(p.tabs, p.vig_image, p.hsv_gain);
  CcRegs cc = {};
  if constexpr ((BITS & ST_CC) != 0) cc.load(p);
  HsvRegs hr = {};
  if constexpr ((BITS & ST_HSV) != 0) hr.load(p);
  __syncthreads();
  const int chunks_per_frame = (items_per_frame + NT - 1) / NT;
  const int f_per_group = (p.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(p.n_frames, f_begin + f_per_group);
  const bool flip180 = p.flip_angle == 180;
  const bool rgb = p.src_kind == SRC_RGB;
  const bool has_tap = p.tap != nullptr;
  const bool dst_nt = p.dst_streaming != 0;
  const unsigned dst_bytes = __umul24((unsigned)(p.drows - 1), (unsigned)p.dst_step) + (unsigned)p.dcols * 3u;
  for (int chunk = blockIdx.x; chunk < chunks_per_frame; chunk += gridDim.x) {
    const int item = chunk * NT + threadIdx.x;
    if (item >= items_per_frame) continue;
    int ys, grp;
    im.split(item, ys, grp);
    const int x0 = grp * 4;
    const int yd = flip180 ? p.rows - 1 - ys : ys;
    const int xbase = flip180 ? p.cols - 4 - x0 : x0;
    float mask[4];
    if constexpr ((BITS & ST_VIG) != 0) {
      const float4 m = *reinterpret_cast<const float4*>(p.vig_mask + (size_t)yd * p.dcols + xbase);  // dcols % 4 == 0, xbase % 4 == 0
      mask[0] = m.x;
      mask[1] = m.y;
      mask[2] = m.z;
      mask[3] = m.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) mask[k] = 1.0f;
    }
    const unsigned src_off = __umul24((unsigned)ys, (unsigned)p.src_step) + (unsigned)x0 * 3u;
    const unsigned dst_off = __umul24((unsigned)yd, (unsigned)p.dst_step) + (unsigned)xbase * 3u;
    const unsigned tap_off = (__umul24((unsigned)yd, (unsigned)p.dcols) + (unsigned)xbase) * 3u;
    for (int frame = f_begin; frame < f_end; frame++) {
      const uint3 in = *reinterpret_cast<const uint3*>(p.src + (size_t)frame * p.src_frame_stride + src_off);
      FrameWb w;
      if (WB != WB_NONE) w = p.wb[frame];
      if constexpr (WB == WB_FLOAT || WB == WB_SIMPLE || WB == WB_PCA) {
#pragma unroll
        for (int c = 0; c < 3; c++) asm volatile("" : "+v"(w.fg[c]));
#pragma unroll
        for (int c = 0; c < 4; c++) asm volatile("" : "+v"(w.pca[c]));
      }
      int s[4][3], q[4][3];
      unpack12(in, rgb, s);
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) q[k][c] = flip180 ? s[3 - k][c] : s[k][c];
      if (has_tap) store12(p.tap + (size_t)frame * p.tap_frame_stride + tap_off, pack4(q));
      store12(frame_rsrc(p.dst + (size_t)frame * p.dst_frame_stride, dst_bytes), dst_off, pointwise4<BITS, WB>(p, w, tb, cc, hr, mask, q), dst_nt);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// mono8 input (the reference passes single-channel frames through: debayer.cpp:45-79 only converts Bayer and rgb8, flip.cpp
// and cv::LUT are channel-agnostic, every colour module is skipped or asserts on one channel): flip 0 / 180 + gamma LUT, four
// pixels (one dword) per lane in and out, the 256-byte table in LDS.  1 B/px in, 1 B/px out.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void chain_mono_kernel(ChainParams p, ItemMap im, int items_per_frame) {
  __shared__ uint8_t s_gamma[256];
  const bool gam = (p.stage_bits & ST_GAMMA) != 0;
  s_gamma[threadIdx.x] = gam ? p.tabs->gamma_lut[threadIdx.x] : (uint8_t)threadIdx.x;
  __syncthreads();
  const int chunks_per_frame = (items_per_frame + kBlock - 1) / kBlock;
  const int f_per_group = (p.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(p.n_frames, f_begin + f_per_group);
  const bool flip180 = p.flip_angle == 180;
  const bool dst_nt = p.dst_streaming != 0;
  const unsigned src_bytes = __umul24((unsigned)(p.rows - 1), (unsigned)p.src_step) + (unsigned)p.cols;
  const unsigned dst_bytes = __umul24((unsigned)(p.drows - 1), (unsigned)p.dst_step) + (unsigned)p.dcols;
  const unsigned tap_bytes = __umul24((unsigned)p.drows, (unsigned)p.dcols);
  for (int chunk = blockIdx.x; chunk < chunks_per_frame; chunk += gridDim.x) {
    const int item = chunk * kBlock + threadIdx.x;
    if (item >= items_per_frame) continue;
    int ys, grp;
    im.split(item, ys, grp);
    const int x0 = grp * 4;
    const int yd = flip180 ? p.rows - 1 - ys : ys;
    const int xbase = flip180 ? p.cols - 4 - x0 : x0;
    const unsigned src_off = __umul24((unsigned)ys, (unsigned)p.src_step) + (unsigned)x0;
    const unsigned dst_off = __umul24((unsigned)yd, (unsigned)p.dst_step) + (unsigned)xbase;
    const unsigned tap_off = __umul24((unsigned)yd, (unsigned)p.dcols) + (unsigned)xbase;
    for (int frame = f_begin; frame < f_end; frame++) {
      const __amdgpu_buffer_rsrc_t src = frame_rsrc(p.src + (size_t)frame * p.src_frame_stride, src_bytes);
      const __amdgpu_buffer_rsrc_t dst = frame_rsrc(p.dst + (size_t)frame * p.dst_frame_stride, dst_bytes);
      uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(src, (int)src_off, 0, 0);
      if (flip180) v = __builtin_bswap32(v);  // the group is written mirrored: reverse the four pixels
      if (p.tap) __builtin_amdgcn_raw_buffer_store_b32(v, frame_rsrc(p.tap + (size_t)frame * p.tap_frame_stride, tap_bytes), (int)tap_off, 0, 0);
      if (gam) {
        const uint32_t a = s_gamma[v & 0xFFu], b = s_gamma[(v >> 8) & 0xFFu], c = s_gamma[(v >> 16) & 0xFFu], d = s_gamma[v >> 24];
        v = a | (b << 8) | (c << 16) | (d << 24);
      }
      if (dst_nt)
        __builtin_amdgcn_raw_buffer_store_b32(v, dst, (int)dst_off, 0, 2);
      else
        __builtin_amdgcn_raw_buffer_store_b32(v, dst, (int)dst_off, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Bayer input with a 90 / 270 degree flip (flip.cpp:45-60: transpose + flip == cv::rotate).  Same window, v_lerp_u8
// demosaic and compile-time stage sets (FastTabs, pointwise4) as chain_fast_kernel -- round 4; until then this kernel
// carried the round-1 per-pixel stage functions and ran 1.65 x slower than the unrotated chain.  A 4x2 item lands as four
// 2-pixel (6-byte) pieces in four output rows, so the lanes of a workgroup are laid out 4 column groups x NT / 4 row pairs:
// for one output row the 16 row pairs a wave holds write 96 contiguous bytes, while each source row is still read in
// 16..24-byte runs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store6(__amdgpu_buffer_rsrc_t frame, unsigned off, uint32_t first, uint32_t second) {
  // two packed pixels (b | g << 8 | r << 16) as three 16-bit stores: the address is only 2-byte aligned
  const uint32_t lo = first | (second << 24), hi = second >> 8;
  __builtin_amdgcn_raw_buffer_store_b16((short)(lo & 0xffffu), frame, (int)off, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b16((short)(lo >> 16), frame, (int)off + 2, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b16((short)(hi & 0xffffu), frame, (int)off + 4, 0, 0);
}
// the four pixels of twelve interleaved bytes as b | g << 8 | r << 16 each
__device__ __forceinline__ void unpack3(const Pack3& v, uint32_t (&px)[4]) {
  px[0] = v.a & 0xFFFFFFu;
  px[1] = (v.a >> 24) | ((v.b & 0xFFFFu) << 8);
  px[2] = (v.b >> 16) | ((v.c & 0xFFu) << 16);
  px[3] = v.c >> 8;
}

template <int BITS, int WB, int NT>
__global__ __launch_bounds__(NT, fast_waves_per_simd<BITS>()) void chain_rot_kernel(ChainParams p, int tiles_x, int tiles_per_frame) {
  __shared__ FastTabs<BITS> tb;
  tb.template load<NT>(p.tabs, p.vig_image, p.hsv_gain);
  CcRegs cc = {};
  if constexpr ((BITS & ST_CC) != 0) cc.load(p);
  HsvRegs hr = {};
  if constexpr ((BITS & ST_HSV) != 0) hr.load(p);
  __syncthreads();
  constexpr int kPairsPerTile = NT / 4;
  const int f_per_group = (p.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(p.n_frames, f_begin + f_per_group);
  const bool rot90 = p.flip_angle == 90;
  const int groups = p.cols >> 2, pairs = p.rows >> 1;
  const unsigned src_bytes = __umul24((unsigned)(p.rows - 1), (unsigned)p.src_step) + (unsigned)p.cols;
  const unsigned dst_bytes = __umul24((unsigned)(p.drows - 1), (unsigned)p.dst_step) + (unsigned)p.dcols * 3u;
  const unsigned tap_bytes = __umul24((unsigned)p.drows, (unsigned)p.dcols) * 3u;
  const bool has_tap = p.tap != nullptr;
  for (int tile = blockIdx.x; tile < tiles_per_frame; tile += gridDim.x) {
    const int tpy = tile / tiles_x, tgx = tile - tpy * tiles_x;
    const int grp = tgx * 4 + (int)(threadIdx.x & 3u), pair = tpy * kPairsPerTile + (int)(threadIdx.x >> 2);
    if (grp >= groups || pair >= pairs) continue;
    const int y0 = pair * 2, x0 = grp * 4;
    // source (ys, xs) -> 90: (xs, R-1-ys);  270: (C-1-xs, ys)   [oracle/rip_oracle.c ripo_flip]
    const int col_d = rot90 ? p.rows - 2 - y0 : y0;  // left one of the two destination columns
    unsigned dst_off[4], tap_off[4];
    float mask[2][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int row_d = rot90 ? x0 + k : p.cols - 1 - (x0 + k);
      dst_off[k] = __umul24((unsigned)row_d, (unsigned)p.dst_step) + (unsigned)col_d * 3u;
      tap_off[k] = (__umul24((unsigned)row_d, (unsigned)p.dcols) + (unsigned)col_d) * 3u;
#pragma unroll
      for (int ly = 0; ly < 2; ly++) {
        if constexpr ((BITS & ST_VIG) != 0)
          mask[ly][k] = p.vig_mask[(size_t)row_d * p.dcols + (rot90 ? col_d + 1 - ly : col_d + ly)];
        else
          mask[ly][k] = 1.0f;
      }
    }
    const WindowOffsets wo = window_offsets((unsigned)p.src_step, p.rows, p.cols, y0, x0);
    for (int frame = f_begin; frame < f_end; frame++) {
      const __amdgpu_buffer_rsrc_t src = frame_rsrc(p.src + (size_t)frame * p.src_frame_stride, src_bytes);
      const __amdgpu_buffer_rsrc_t dst = frame_rsrc(p.dst + (size_t)frame * p.dst_frame_stride, dst_bytes);
      const __amdgpu_buffer_rsrc_t tap = frame_rsrc(has_tap ? p.tap + (size_t)frame * p.tap_frame_stride : nullptr, has_tap ? tap_bytes : 0u);
      FrameWb w;
      if (WB != WB_NONE) w = p.wb[frame];
      if constexpr (WB == WB_FLOAT || WB == WB_SIMPLE || WB == WB_PCA) {
#pragma unroll
        for (int c = 0; c < 3; c++) asm volatile("" : "+v"(w.fg[c]));
#pragma unroll
        for (int c = 0; c < 4; c++) asm volatile("" : "+v"(w.pca[c]));
      }
      Window win;
      load_window(src, wo, win);
      Planar rowpx[2];
      debayer_tile_any(win, p.bayer_ry, p.bayer_rx, y0, x0, p.rows, p.cols, rowpx);
      uint32_t raw[2][4], pix[2][4];  // b | g << 8 | r << 16
#pragma unroll
      for (int ly = 0; ly < 2; ly++) {
        Planar v = rowpx[ly];
        Pack3 rawp;
        const bool need_raw = has_tap || (BITS == 0 && WB == WB_NONE);
        if (need_raw) {
          interleave4(v, rawp.a, rawp.b, rawp.c);
          unpack3(rawp, raw[ly]);
        }
        if (BITS == 0 && WB == WB_NONE) {  // pure demosaic: no per-pixel stage
#pragma unroll
          for (int k = 0; k < 4; k++) pix[ly][k] = raw[ly][k];
          continue;
        }
        if (WB == WB_Q8) {  // grey-world gains on the packed bytes, two pixels per multiply
          v.b = gains_q8_swar(v.b, (unsigned)w.q8[0]);
          v.g = gains_q8_swar(v.g, (unsigned)w.q8[1]);
          v.r = gains_q8_swar(v.r, (unsigned)w.q8[2]);
        }
        int q[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          q[k][0] = (int)((v.b >> (8 * k)) & 0xFFu);
          q[k][1] = (int)((v.g >> (8 * k)) & 0xFFu);
          q[k][2] = (int)((v.r >> (8 * k)) & 0xFFu);
        }
        unpack3(pointwise4<BITS, WB == WB_Q8 ? WB_NONE : WB>(p, w, tb, cc, hr, mask[ly], q), pix[ly]);
      }
      const int first = rot90 ? 1 : 0;  // which source row lands in the left destination column
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (has_tap) store6(tap, tap_off[k], first ? raw[1][k] : raw[0][k], first ? raw[0][k] : raw[1][k]);
        store6(dst, dst_off[k], first ? pix[1][k] : pix[0][k], first ? pix[0][k] : pix[1][k]);
      }
    }
  }
}


// the LDS tables of the vignetting variants, built once with the table address left out (load_image adds the kernel's own)
__global__ __launch_bounds__(512) void vig_image_kernel(const DevTables* tabs, uint32_t* image) {
  __shared__ VigTabs t;
  t.template load<512>(tabs, 0u);
  __syncthreads();
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&t);
  for (int i = threadIdx.x; i < (int)(sizeof(VigTabs) / 4); i += 512) image[i] = src[i];
}

template <int BITS, int WB>
void launch_fast(const ChainParams& p, const ItemMap& im, int items, dim3 grid, hipStream_t stream, bool debug_occupancy) {
  constexpr int NT = fast_threads<BITS>();
  if (debug_occupancy) {  // development aid: resident workgroups per CU the runtime computes for this variant
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, chain_fast_kernel<BITS, WB, NT>, NT, 0);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(chain_fast_kernel<BITS, WB, NT>));
    std::fprintf(stderr, "[rip] chain_fast_kernel<%d,%d,%d>: %d workgroups/CU, %d VGPR, %zu B LDS, grid %u x %u\n", BITS, WB, NT, nb,
                 fa.numRegs, fa.sharedSizeBytes, grid.x, grid.y);
  }
  hipLaunchKernelGGL((chain_fast_kernel<BITS, WB, NT>), grid, dim3(NT), 0, stream, p, im, items);
}

template <int BITS, int WB>
void launch_color(const ChainParams& p, const ItemMap& im, int items, dim3 grid, hipStream_t stream) {
  constexpr int NT = fast_threads<BITS>();
  hipLaunchKernelGGL((chain_color_kernel<BITS, WB, NT>), grid, dim3(NT), 0, stream, p, im, items);
}
template <int BITS>
void launch_color_wb(const ChainParams& p, const ItemMap& im, int items, dim3 grid, hipStream_t stream) {
  switch (p.wb_mode) {
    case WB_Q8: launch_color<BITS, WB_Q8>(p, im, items, grid, stream); break;
    case WB_FLOAT: launch_color<BITS, WB_FLOAT>(p, im, items, grid, stream); break;
    case WB_PCA: launch_color<BITS, WB_PCA>(p, im, items, grid, stream); break;
    case WB_SIMPLE: launch_color<BITS, WB_SIMPLE>(p, im, items, grid, stream); break;
    default: launch_color<BITS, WB_NONE>(p, im, items, grid, stream); break;
  }
}

template <int BITS, int WB>
void launch_rot(const ChainParams& p, int tiles_x, int tiles, dim3 grid, hipStream_t stream) {
  constexpr int NT = fast_threads<BITS>();
  hipLaunchKernelGGL((chain_rot_kernel<BITS, WB, NT>), grid, dim3(NT), 0, stream, p, tiles_x, tiles);
}
template <int BITS>
void launch_rot_wb(const ChainParams& p, int tiles_x, int tiles, dim3 grid, hipStream_t stream) {
  switch (p.wb_mode) {
    case WB_Q8: launch_rot<BITS, WB_Q8>(p, tiles_x, tiles, grid, stream); break;
    case WB_FLOAT: launch_rot<BITS, WB_FLOAT>(p, tiles_x, tiles, grid, stream); break;
    case WB_PCA: launch_rot<BITS, WB_PCA>(p, tiles_x, tiles, grid, stream); break;
    case WB_SIMPLE: launch_rot<BITS, WB_SIMPLE>(p, tiles_x, tiles, grid, stream); break;
    default: launch_rot<BITS, WB_NONE>(p, tiles_x, tiles, grid, stream); break;
  }
}

template <int BITS>
void launch_fast_wb(const ChainParams& p, const ItemMap& im, int items, dim3 grid, hipStream_t stream, bool occ) {
  switch (p.wb_mode) {
    case WB_Q8: launch_fast<BITS, WB_Q8>(p, im, items, grid, stream, occ); break;
    case WB_FLOAT: launch_fast<BITS, WB_FLOAT>(p, im, items, grid, stream, occ); break;
    case WB_PCA: launch_fast<BITS, WB_PCA>(p, im, items, grid, stream, occ); break;
    case WB_SIMPLE: launch_fast<BITS, WB_SIMPLE>(p, im, items, grid, stream, occ); break;
    default: launch_fast<BITS, WB_NONE>(p, im, items, grid, stream, occ); break;
  }
}

}  // namespace

#if !RIP_FP_CONTRACT
bool color_fast_geometry(const uint8_t* src, size_t step, size_t frame_stride, int rows, int cols, int kind) {
  return (kind == SRC_BGR || kind == SRC_RGB) && cols % 4 == 0 && step % 4 == 0 && frame_stride % 4 == 0 && aligned4(src) &&
         step < (1u << 24) && rows < (1 << 23) && (unsigned long long)step * (unsigned long long)rows < (1ull << 32);
}

bool chain_uses_color_path(const ChainParams& p) {
  return color_fast_geometry(p.src, p.src_step, p.src_frame_stride, p.rows, p.cols, p.src_kind) &&
         (p.flip_angle == 0 || p.flip_angle == 180) && p.channels == 3 && p.dst_step % 4 == 0 && p.dst_step < (1u << 24) &&
         (unsigned long long)p.dst_step * (unsigned long long)p.drows < (1ull << 32) && p.dst_frame_stride % 4 == 0 &&
         aligned4(p.dst) && (!p.tap || (aligned4(p.tap) && p.tap_frame_stride % 4 == 0));
}

int chain_uses_fast_path(const ChainParams& p) {
  return bayer_fast_geometry(p.src, p.src_step, p.src_frame_stride, p.rows, p.cols, p.src_kind) &&
         (p.flip_angle == 0 || p.flip_angle == 180) && p.channels == 3 && p.dst_step % 4 == 0 &&
         p.dst_step < (1u << 24) && (unsigned long long)p.dst_step * (unsigned long long)p.drows < (1ull << 32) &&
         p.dst_frame_stride % 4 == 0 && aligned4(p.dst) && (!p.tap || (aligned4(p.tap) && p.tap_frame_stride % 4 == 0));
}

bool chain_uses_mono_path(const ChainParams& p) {
  return p.src_kind == SRC_MONO && p.channels == 1 && (p.flip_angle == 0 || p.flip_angle == 180) && p.cols % 4 == 0 && p.src_step % 4 == 0 &&
         p.src_frame_stride % 4 == 0 && aligned4(p.src) && p.src_step < (1u << 24) && p.rows < (1 << 23) &&
         (unsigned long long)p.src_step * (unsigned long long)p.rows < (1ull << 32) && p.dst_step % 4 == 0 && p.dst_step < (1u << 24) &&
         (unsigned long long)p.dst_step * (unsigned long long)p.drows < (1ull << 32) && p.dst_frame_stride % 4 == 0 && aligned4(p.dst) &&
         (!p.tap || (aligned4(p.tap) && p.tap_frame_stride % 4 == 0));
}

bool chain_uses_rot_path(const ChainParams& p) {
  return bayer_fast_geometry(p.src, p.src_step, p.src_frame_stride, p.rows, p.cols, p.src_kind) &&
         (p.flip_angle == 90 || p.flip_angle == 270) && p.channels == 3 && p.drows == p.cols && p.dcols == p.rows &&
         p.dst_step % 2 == 0 && p.dst_step < (1u << 24) && p.cols < (1 << 23) &&
         (unsigned long long)p.dst_step * (unsigned long long)p.drows < (1ull << 32) && p.dst_frame_stride % 2 == 0 &&
         (reinterpret_cast<uintptr_t>(p.dst) & 1u) == 0 &&
         (!p.tap || ((reinterpret_cast<uintptr_t>(p.tap) & 1u) == 0 && p.tap_frame_stride % 2 == 0));
}

#endif  // !RIP_FP_CONTRACT

// gridDim.y: how many groups of frames the batch is split into (every kernel here walks the frames of its group
// innermost).  At most 16 frames per item visit (RIP_CHAIN_FRAMES) for the VALU-bound stage sets: 2448 chunks on 2048
// persistent workgroups would leave most of the chip idle while a fifth of them does a second chunk; four times as many,
// shorter units let the dispatcher even that out (config2: 0.80 -> 0.75 ms per 64 frames) and 16 frames still amortise the
// per-item setup (mask, addresses).  The stage sets without Lab / HSV round trip run at the memory rate and want few
// frames per visit -- a frame is contiguous, the next frame of the batch is megabytes away -- but not one: round 3,
// 2448x2048, 256 frames, ms per launch at 1 / 2 / 4 / 8 / 16 frames per visit: demosaic only 1.17 / 1.09 / 1.09 / 1.26 / 1.35,
// + grey-world gains . / 1.05 / 1.07 / 1.23 / 1.43, + gamma . / 1.15 / 1.09 / 1.23 / 1.39, + gains + colour matrix + gamma
// 1.62 / 1.39 / 1.27 / 1.28 / 1.41; 3840x2160 demosaic only 1.76 / 1.70 / 1.80.  (Requesting the windows of two frames before
// the first is demosaiced, or walking the (frame, chunk) pairs of a group in address order with long-lived workgroups,
// measured equal or slower.)
static int frame_groups(const ChainParams& p, const Tunables& tn, int cap, int blocks) {
  const bool valu_bound = (p.stage_bits & (ST_VIG | ST_HSV)) != 0;
  // memory-rate stage sets: two frames per visit (four once the colour matrix or the gamma table add per-pixel work) --
  // the per-item setup is shared and a workgroup lives a little longer, while the frames it touches stay few
  // (round 6, with the round-robin deal of the chunks -- p.deal > 0 -- and 256 frames of 2448x2048, gains + colour matrix + gamma at
  // 2 / 4 / 6 / 8 / 16 frames per visit: 1.461 / 1.240 / 1.213 / 1.247 / 1.286 ms)
  const int streaming = (p.stage_bits & (ST_CC | ST_GAMMA)) ? (p.deal > 0 ? 6 : 4) : 2;
  const int frames_per_visit = tn.chain_frames > 0 ? tn.chain_frames : (valu_bound ? 16 : streaming);
  const int groups = std::max(cap / std::max(blocks, 1), (p.n_frames + frames_per_visit - 1) / frames_per_visit);
  return std::max(1, std::min(p.n_frames, groups));
}

#if RIP_FP_CONTRACT
// declared by the uncontracted translation unit's twin (same definitions, one copy in the library)
int chain_uses_fast_path(const ChainParams& p);
bool chain_uses_color_path(const ChainParams& p);
bool chain_uses_mono_path(const ChainParams& p);
bool chain_uses_rot_path(const ChainParams& p);
#define launch_chain launch_chain_fc1
#else
size_t vig_image_bytes() { return sizeof(VigTabs); }
void launch_vig_image(const DevTables* tabs, uint32_t* image, hipStream_t stream) {
  hipLaunchKernelGGL(vig_image_kernel, dim3(1), dim3(512), 0, stream, tabs, image);
}

void launch_debayer16(const Debayer16Params& p, hipStream_t stream) {
  if (p.n_frames <= 0) return;
  const long long npix = (long long)p.drows * p.dcols;
  hipLaunchKernelGGL(debayer16_kernel, dim3(grid_blocks_for(npix, 4096), p.n_frames), dim3(kBlock), 0, stream, p);
}

#endif  // RIP_FP_CONTRACT

void launch_chain(const ChainParams& p_in, const Tunables& tn, hipStream_t stream) {
  ChainParams p = p_in;
  const bool remap_behind = p_in.deal < 0;  // run_batch's hint: the image this launch writes is the intermediate the remap gathers from
  p.deal = 0;                               // the fast path below sets it
  if (p.n_frames <= 0) return;
#if !RIP_FP_CONTRACT
  if (p.fp_contract == 1) return launch_chain_fc1(p_in, tn, stream);  // with run_batch's hint intact
#endif
  if (chain_uses_rot_path(p)) {
    const int nt = (p.stage_bits & ST_VIG) ? fast_threads<ST_VIG>() : fast_threads<0>();
    const int tiles_x = (p.cols / 4 + 3) / 4, tiles_y = (p.rows / 2 + nt / 4 - 1) / (nt / 4);
    const int tiles = tiles_x * tiles_y;
    const int dflt_blocks = nt == kBlock ? 2048 : 4096;
    const int cap = std::max(8, grid_multiple_of_8(tn.chain_blocks > 0 ? tn.chain_blocks : dflt_blocks) * kBlock / nt / 8 * 8);
    const int blocks = std::min(cap, tiles);
    const dim3 grid(blocks, frame_groups(p, tn, cap, blocks));
    switch (p.stage_bits & 15) {
#define RIP_CASE(B) case B: launch_rot_wb<B>(p, tiles_x, tiles, grid, stream); break;
      RIP_CASE(0) RIP_CASE(1) RIP_CASE(2) RIP_CASE(3) RIP_CASE(4) RIP_CASE(5) RIP_CASE(6) RIP_CASE(7)
      RIP_CASE(8) RIP_CASE(9) RIP_CASE(10) RIP_CASE(11) RIP_CASE(12) RIP_CASE(13) RIP_CASE(14) RIP_CASE(15)
#undef RIP_CASE
    }
    return;
  }
  if (chain_uses_fast_path(p)) {
    ItemMap im{p.cols / 4, 1.0f / (float)(p.cols / 4)};
    const int items = (p.rows / 2) * (p.cols / 4);
    const int nt = (p.stage_bits & ST_VIG) ? fast_threads<ST_VIG>() : fast_threads<0>();
    const long long chunks = (long long)((items + nt - 1) / nt);
    // persistent grid: at most 256 CUs x 8 x 256 threads, a multiple of 8 workgroups (one share per XCD; the kernel
    // strides its chunk loop by gridDim.x / 8)
    // (the 512-thread vignetting variants run ~3 % faster with one chunk per workgroup than with a 768-workgroup persistent grid)
    // a frame or two on the vignetting variants: three persistent workgroups per CU build the 54 KB of tables once each
    // (67.4 us against 70.3 for the whole single-frame call with one workgroup per chunk: tools/probes/graph_probe.py)
    // (round 5 sweep of the memory-rate variant, debayer + gains + matrix + gamma, 256 frames: 1.245 / 1.242 / 1.224 / 1.219 ms at
    // 1024 / 2048 / 4096 / 8192 persistent workgroups: the cap is 4096 since then)
    const int dflt_blocks = nt == kBlock ? 4096 : (p.n_frames <= 2 ? 1536 : 4096);
    const int cap = std::max(8, grid_multiple_of_8(tn.chain_blocks > 0 ? tn.chain_blocks : dflt_blocks) * kBlock / nt / 8 * 8);
    int blocks = (int)std::min<long long>(cap, (chunks + 7) / 8 * 8);
    // batches only (a single frame keeps the halo rows of neighbouring chunks in one L2), and by default not in front of the
    // remap: there the deal buys nothing (config 2's chain inside the step, deal off -> on, three boxes: 2.193 -> 2.158, 2.215 ->
    // 2.229, 2.169 -> 2.182 ms) while the input fetched at the L2s rises from 1.62 to 2.17 GB per 256 frames; alone the chain gains
    // 3-6 %.  RIP_CHAIN_DEAL: 0 never, 1 when no remap follows, 2 always.  Runs of 3 x 512 items (fast_chunks kDeal)
    p.deal = p.n_frames >= 4 && (tn.chain_deal >= 2 || (tn.chain_deal == 1 && !remap_behind)) ? 1 : 0;
    dim3 grid(blocks, frame_groups(p, tn, cap, blocks));
    switch (p.stage_bits & 15) {
#define RIP_CASE(B) case B: launch_fast_wb<B>(p, im, items, grid, stream, tn.debug_occupancy != 0); break;
      RIP_CASE(0) RIP_CASE(1) RIP_CASE(2) RIP_CASE(3) RIP_CASE(4) RIP_CASE(5) RIP_CASE(6) RIP_CASE(7)
      RIP_CASE(8) RIP_CASE(9) RIP_CASE(10) RIP_CASE(11) RIP_CASE(12) RIP_CASE(13) RIP_CASE(14) RIP_CASE(15)
#undef RIP_CASE
    }
    return;
  }
  if (chain_uses_color_path(p)) {
    ItemMap im{p.cols / 4, 1.0f / (float)(p.cols / 4)};
    const int items = p.rows * (p.cols / 4);
    const int nt = (p.stage_bits & ST_VIG) ? fast_threads<ST_VIG>() : fast_threads<0>();
    const int chunks = (items + nt - 1) / nt;
    const int dflt_blocks = nt == kBlock ? 2048 : 4096;
    const int cap = std::max(8, grid_multiple_of_8(tn.chain_blocks > 0 ? tn.chain_blocks : dflt_blocks) * kBlock / nt / 8 * 8);
    const int blocks = std::min(cap, chunks);
    const dim3 grid(blocks, frame_groups(p, tn, cap, blocks));
    switch (p.stage_bits & 15) {
#define RIP_CASE(B) case B: launch_color_wb<B>(p, im, items, grid, stream); break;
      RIP_CASE(0) RIP_CASE(1) RIP_CASE(2) RIP_CASE(3) RIP_CASE(4) RIP_CASE(5) RIP_CASE(6) RIP_CASE(7)
      RIP_CASE(8) RIP_CASE(9) RIP_CASE(10) RIP_CASE(11) RIP_CASE(12) RIP_CASE(13) RIP_CASE(14) RIP_CASE(15)
#undef RIP_CASE
    }
    return;
  }
  if (chain_uses_mono_path(p)) {
    ItemMap im{p.cols / 4, 1.0f / (float)(p.cols / 4)};
    const int items = p.rows * (p.cols / 4);
    const int chunks = (items + kBlock - 1) / kBlock;
    const int blocks = std::min(8192, chunks);
    const int groups = frame_groups(p, tn, 8192, blocks);  // memory-rate: two (four with the gamma table) frames per visit
    hipLaunchKernelGGL(chain_mono_kernel, dim3(blocks, groups), dim3(kBlock), 0, stream, p, im, items);
    return;
  }
  long long npix = (long long)p.drows * p.dcols;
  dim3 grid(grid_blocks_for(npix, 2048), p.n_frames);
  hipLaunchKernelGGL(chain_generic_kernel, grid, dim3(kBlock), 0, stream, p);
}

}  // namespace rip
