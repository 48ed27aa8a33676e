// rip_fused.hip -- the per-pixel chain INSIDE the remap's tiles, for the stage sets that run at the memory rate.
//
// debayer -> (white-balance gains, colour matrix, gamma LUT) -> undistortion as two kernels moves 1 + 3 B/px through the
// chain and 3 x 1.15 + 3 B/px (+ plan) through the remap: 11 B/px, seven of them for an intermediate image nobody asked for
// (undistortion.cpp:240-249 gathers from the image flip.cpp / the colour modules left behind; the reference keeps it as
// its dist_image_ tap).  When that tap is not requested, a remap tile can make its own source rectangle: the Bayer bytes
// under the rectangle (1 B/px, a one-pixel halo) come in through the LDS-DMA ring, the workgroup demosaics and colours
// them LDS -> LDS with the fused chain's own stage code (debayer_tile_any, pointwise4: same functions, same bits), and the
// gather runs on that LDS image exactly as remap_ring_kernel's does.  Compulsory traffic: 1 B/px of Bayer (x the
// rectangles' overlap) + 3 B/px written + the plan words -- 4 + 4 / n B/px instead of 11.  The demosaic of a rectangle
// covers ~1.6 x the tile's pixels (overlap + alignment to 4 x 2 Bayer items); at ~5 VALU instructions per pixel that is
// small beside the gather.
//
// Applies to: Bayer input on the fast geometry, flip 0 / 180, no vignetting / enhancer (those stage sets are VALU-bound: the
// 1.6 x recompute would cost more than the bytes save), tiled plan available, 16-byte-aligned Bayer pitch, and neither
// the debayered nor the colour tap requested.  Everything else takes the two-kernel path.  Border pixels (taps straddling
// the image edge) are patched per tap from the Bayer frame by remap_border_bayer_kernel.
#include "rip_chain_dev.hpp"
#include "rip_remap_dev.hpp"

#include <cstdio>

namespace rip {
namespace {

#ifndef RIP_FUSED_WAVES
// waves per SIMD the register allocation must allow.  config5, ms per 256 frames: 4 waves (100-127 VGPRs as hipcc likes it) 3.44,
// 5 (96 VGPRs, no spill in the frame loop) 3.20-3.25, 6 (80 VGPRs, 52-116 bytes of scratch) 4.40
#define RIP_FUSED_WAVES 5
#endif
// LDS image of the visit's per-frame gains (FrameWb is read by every lane of every frame: one global read per visit)
constexpr int kFusedMaxFrames = 16;
#ifndef RIP_FUSED_PAD
#define RIP_FUSED_PAD 8
#endif
constexpr unsigned kFusedRowPad = RIP_FUSED_PAD;  // bytes added to every row of the LDS colour image (a multiple of 8)

// s_waitcnt for this wave's LDS writes, then the workgroup barrier: the raw s_barrier builtin does not wait for them, and a
// __syncthreads() would also drain vmcnt, i.e. the ring's prefetched frames
__device__ __forceinline__ void lds_write_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int PRE, int BITS, int WB>
__global__ __launch_bounds__(kRemapTileThreads, RIP_FUSED_WAVES) void remap_bayer_ring_kernel(RemapTiledParams p, ChainParams c, unsigned bgr_off, unsigned bgr_bytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  __shared__ FastTabs<BITS> tb;
  __shared__ FrameWb s_wb[kFusedMaxFrames];
  constexpr unsigned kStage = (unsigned)PRE * kRemapTileThreads * 16u;  // bytes per Bayer stage
  const RemapParams& b = p.base;  // b.src: the BAYER frames (1 B/px), b.rows / b.cols their geometry
  tb.template load<kRemapTileThreads>(c.tabs, nullptr);
  CcRegs cc = {};
  if constexpr ((BITS & ST_CC) != 0) cc.load(c);
  HsvRegs hr = {};
  const int ntiles = p.tiles_x * p.tiles_y;
  const TileDeal deal(ntiles, p.deal_run);
  const int xcd = blockIdx.x & 7;
  const int tid = threadIdx.x;
  const int lrow = tid / kRemapGroupsPerRow, lgrp = tid % kRemapGroupsPerRow;
  const unsigned step = (unsigned)b.src_step;
  const int f_per_group = (b.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(b.n_frames, f_begin + f_per_group);
  if (f_begin >= f_end) return;  // uniform for the workgroup
  if (WB != WB_NONE)
    for (int i = tid; i < f_end - f_begin; i += kRemapTileThreads) s_wb[i] = c.wb[f_begin + i];
  __syncthreads();
  const int nb = p.stages, dist = nb - 1;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);
  uint8_t* const bgr = lds + bgr_off;
  const unsigned wave_chunk0 = (unsigned)__builtin_amdgcn_readfirstlane(tid & ~63);
  const unsigned dst_bytes = __umul24((unsigned)(b.drows - 1), (unsigned)b.dst_step) + (unsigned)b.dcols * 3u;
  const unsigned src_bytes = __umul24((unsigned)(b.rows - 1), step) + (unsigned)b.cols;
  const float ones[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  for (int ti = blockIdx.x >> 3; ti < deal.per_xcd; ti += gridDim.x >> 3) {
    const int tile = deal.tile(ti, xcd);
    if (tile == -1) continue;
    if (tile < 0) break;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const RemapTileDesc d = p.tiles[tile];
    const uint4 wd = reinterpret_cast<const uint4*>(p.words + (size_t)tile * kRemapTilePx)[tid];
    const uint32_t words[4] = {wd.x, wd.y, wd.z, wd.w};
    const int yd = ty * kRemapTileH + lrow, xd = tx * kRemapTileW + lgrp * 4;
    const bool in_image = yd < b.drows && xd < b.dcols;
    // The rectangle widened to whole 4 x 2 Bayer items, and the Bayer bytes those items read (a dword left and right, a row
    // above and below, clamped to the frame exactly as window_offsets clamps them), staged from a 16-byte-aligned column.
    // With a 180-degree flip the image the remap gathers from is the demosaiced frame read backwards: image pixel (y, x) is
    // frame pixel (rows - 1 - y, cols - 1 - x).  The LDS colour image stays in FRAME orientation (the demosaic and the stage
    // code do not change); the tile's rectangle is mirrored into frame coordinates here, and the taps below address it
    // backwards.
    const bool flip = c.flip_angle == 180;
    const int rx0 = flip ? b.cols - d.x0 - d.w : d.x0, ry0 = flip ? b.rows - d.y0 - d.h : d.y0;
    const int X0 = rx0 & ~3, X1 = min((rx0 + d.w + 3) & ~3, b.cols), Y0 = ry0 & ~1, Y1 = min((ry0 + d.h + 1) & ~1, b.rows);
    const int BX0 = max(X0 - 4, 0), BX1 = min(X1 + 4, b.cols), BY0 = max(Y0 - 1, 0), BY1 = min(Y1 + 1, b.rows);
    const unsigned CX0 = (unsigned)BX0 & ~15u;
    const unsigned lp = ((unsigned)BX1 - CX0 + 15u) & ~15u;  // Bayer bytes per staged row
    const unsigned chunks = lp >> 4;
    const unsigned total = d.w > 0 ? chunks * (unsigned)(BY1 - BY0) : 0u;
    const ItemMap cm{(int)chunks, 1.0f / (float)(chunks ? chunks : 1u)};
    unsigned goff[PRE];
#pragma unroll
    for (int j = 0; j < PRE; j++) {
      const unsigned i = (unsigned)tid + (unsigned)j * kRemapTileThreads;
      int r, cc16;
      cm.split((int)i, r, cc16);
      goff[j] = i < total ? __umul24((unsigned)(BY0 + r), step) + CX0 + ((unsigned)cc16 << 4) : 0xFFFFFFF0u;
    }
    // bytes per row of the LDS colour image: FOUR bytes per pixel (b g r x) since round 5 -- a tap row of the gather is then two
    // whole dwords (one aligned 8-byte read, no v_alignbyte) instead of six bytes at any alignment (three dwords); + 8 so that
    // the rows of a wave, whose lanes read dword pairs four dwords apart, alternate between the two halves of every bank quad
    const unsigned bp = (unsigned)(X1 - X0) * 4u + kFusedRowPad;
    unsigned tap_addr[4], wxb[4], wyy[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t w = words[k];
      const bool live = w < kPlanBorder;
      const unsigned relx = w & 0x7ffu, rely = (w >> 11) & 0x7ffu, fx = (w >> 22) & 31u, fy = w >> 27;
      if (!flip) {
        tap_addr[k] = live ? __umul24(rely + (unsigned)(ry0 - Y0), bp) + (relx + (unsigned)(rx0 - X0)) * 4u : 0u;
        wxb[k] = live ? (32u - fx) | (fx << 8) : 0u;
        wyy[k] = (32u - fy) | (fy << 16);
      } else {
        // the tap square {y, y + 1} x {x, x + 1} of the image is the frame square whose top-left pixel is
        // (rows - 2 - y, cols - 2 - x), read upside down and mirrored: the weights change places
        const int sy = b.rows - 2 - (d.y0 + (int)rely), sx = b.cols - 2 - (d.x0 + (int)relx);
        tap_addr[k] = live ? __umul24((unsigned)(sy - Y0), bp) + (unsigned)(sx - X0) * 4u : 0u;
        wxb[k] = live ? fx | ((32u - fx) << 8) : 0u;
        wyy[k] = fy | ((32u - fy) << 16);
      }
    }
    const unsigned dst_off = __umul24((unsigned)yd, (unsigned)b.dst_step) + (unsigned)xd * 3u;
    const int nx = (X1 - X0) >> 2, n_items = d.w > 0 ? nx * ((Y1 - Y0) >> 1) : 0;
    const ItemMap im{nx, 1.0f / (float)(nx ? nx : 1)};
    // frame-invariant part of a demosaic item: LDS offsets of the four window rows at the centre dword, the distances to the
    // left / right dwords (0 at the image's edges: window_offsets' rule), the offset of the item's first output row in the
    // colour image, and its position for the border rule.  The lane's FIRST item is kept in registers across the frames of
    // the visit (all there is for the usual rectangles: <= 256 items); further items of large rectangles are recomputed.
    struct Item {
      unsigned row[4], lr, out;
      int y, x;
    };
    auto make_item = [&](int it) {
      Item e;
      int iy, ix;
      im.split(it, iy, ix);
      e.y = Y0 + 2 * iy;
      e.x = X0 + 4 * ix;
      e.lr = (e.x >= 4 ? 4u : 0u) | ((e.x + 4 < b.cols ? 4u : 0u) << 16);
#pragma unroll
      for (int r = 0; r < 4; r++) e.row[r] = __umul24((unsigned)(clampi(e.y - 1 + r, 0, b.rows - 1) - BY0), lp) + (unsigned)e.x - CX0;
      e.out = __umul24((unsigned)(e.y - Y0), bp) + (unsigned)(e.x - X0) * 4u;
      return e;
    };
    const Item item0 = make_item(tid < n_items ? tid : 0);

    auto issue = [&](int f, int slot) {
      const __amdgpu_buffer_rsrc_t rsrc = frame_rsrc(b.src + (size_t)f * b.src_frame_stride, src_bytes);
      const unsigned stage = lds0 + (unsigned)slot * kStage;
#pragma unroll
      for (int j = 0; j < PRE; j++) lds_dma16(rsrc, goff[j], stage + ((wave_chunk0 + (unsigned)j * kRemapTileThreads) << 4));
    };
    // Bayer stage -> LDS colour image: the fused chain's item (4 px x 2 rows), its window read from LDS instead of HBM
    auto demosaic_item = [&](const uint8_t* bay, uint8_t* img, const FrameWb& w, const Item& e) {
      const unsigned dl = e.lr & 0xffffu, dr = e.lr >> 16;
      Window win;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint8_t* row = bay + e.row[r];
        win.w[r][0] = *reinterpret_cast<const uint32_t*>(row - dl);
        win.w[r][1] = *reinterpret_cast<const uint32_t*>(row);
        win.w[r][2] = *reinterpret_cast<const uint32_t*>(row + dr);
      }
      Planar rowpx[2];
      debayer_tile_any(win, c.bayer_ry, c.bayer_rx, e.y, e.x, b.rows, b.cols, rowpx);
#pragma unroll
      for (int ly = 0; ly < 2; ly++) {
        Planar v = rowpx[ly];
        uint32_t px[4];  // b | g << 8 | r << 16 | (anything) << 24
        if (BITS == 0 && WB == WB_NONE) {
          const uint32_t bg01 = __builtin_amdgcn_perm(v.g, v.b, 0x05010400u), bg23 = __builtin_amdgcn_perm(v.g, v.b, 0x07030602u);  // B0 G0 B1 G1 / B2 G2 B3 G3
          px[0] = __builtin_amdgcn_perm(v.r, bg01, 0x0c040100u);
          px[1] = __builtin_amdgcn_perm(v.r, bg01, 0x0c050302u);
          px[2] = __builtin_amdgcn_perm(v.r, bg23, 0x0c060100u);
          px[3] = __builtin_amdgcn_perm(v.r, bg23, 0x0c070302u);
        } else {
          Pack3 o;
          if (WB == WB_Q8) {
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
          o = pointwise4<BITS, WB == WB_Q8 ? WB_NONE : WB>(c, w, tb, cc, hr, ones, q);
          // twelve interleaved bytes -> four pixels of four bytes; byte 3 of each is whatever follows (the gather never weighs it)
          px[0] = o.a;
          px[1] = __builtin_amdgcn_alignbyte(o.b, o.a, 3);
          px[2] = __builtin_amdgcn_alignbyte(o.c, o.b, 2);
          px[3] = o.c >> 8;
        }
        uint2* out = reinterpret_cast<uint2*>(img + e.out + (ly ? bp : 0u));  // 8-byte aligned: bp and 16 * column are multiples of 8
        out[0] = make_uint2(px[0], px[1]);
        out[1] = make_uint2(px[2], px[3]);
      }
    };
    // Bayer stage -> LDS colour image: the fused chain's item (4 px x 2 rows), its window read from LDS instead of HBM
    auto demosaic = [&](const uint8_t* bay, uint8_t* img, int f) {
      FrameWb w = {};
      if (WB != WB_NONE) w = s_wb[f - f_begin];
      if (tid < n_items) demosaic_item(bay, img, w, item0);
#pragma unroll 1
      for (int it = tid + kRemapTileThreads; it < n_items; it += kRemapTileThreads) demosaic_item(bay, img, w, make_item(it));
    };
    auto gather_store = [&](const uint8_t* img, int f) {
      if (!in_image) return;
      uint32_t t0[4], t1[4], b0[4], b1[4];  // left / right pixel (b g r x) of the top / bottom tap rows
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t* top = reinterpret_cast<const uint32_t*>(img + tap_addr[k]);
        const uint32_t* bot = reinterpret_cast<const uint32_t*>(img + tap_addr[k] + bp);
        t0[k] = top[0];
        t1[k] = top[1];
        b0[k] = bot[0];
        b1[k] = bot[1];
      }
      // non-temporal: the final image, nobody on the device reads it again (round 5: 3.134 -> 3.055 ms per 256 frames at 3840 x 2160;
      // the same bit on the two-kernel remap of config 2, which pulls three times the bytes through the L2s, costs it 14 %)
      // Batches only (c.dst_streaming: >= 8 frames and the chain_nt tunable allows it, set by run_batch as for the chain): a single
      // frame's latency call keeps the default policy
      store12(frame_rsrc(b.dst + (size_t)f * b.dst_frame_stride, dst_bytes), dst_off, blend4_bgrx(t0, t1, b0, b1, wxb, wyy), c.dst_streaming);
    };
    auto wait_landed = [&](int f) {  // the Bayer bytes of frame f are in LDS (this wave's part): frames issued after f may fly
      const int ahead = min(dist - 1, f_end - 1 - f);
      if (ahead >= 2)
        wait_vmcnt<2 * PRE>();
      else if (ahead == 1)
        wait_vmcnt<PRE>();
      else
        wait_vmcnt<0>();
    };

    // every earlier memory operation of this wave is waited for here, so the counted waits below see only this tile's ring
    // loads and stores
    wait_vmcnt<0>();
    // Skewed by one frame, ONE barrier per frame: between two barriers a workgroup demosaics frame f + 1 into one colour
    // image while it gathers frame f from the other.  The barrier of iteration f says: every wave's part of Bayer frame
    // f + 1 has landed; colour image f is complete (every wave waited for its LDS writes); everybody is done gathering frame
    // f - 1, so its colour image may take frame f + 1; and everybody is done demosaicing frame f, so its Bayer stage may be
    // refilled.
    int slot_in = 0, slot_dem = 0;  // ring positions of the next frame to issue / to demosaic
    for (int f = f_begin; f < f_end && f < f_begin + dist; f++) {
      issue(f, slot_in);
      slot_in = slot_in + 1 == nb ? 0 : slot_in + 1;
    }
    wait_landed(f_begin);
    __builtin_amdgcn_s_barrier();
    if (f_begin + dist < f_end) {
      issue(f_begin + dist, slot_in);
      slot_in = slot_in + 1 == nb ? 0 : slot_in + 1;
    }
    demosaic(lds + (unsigned)slot_dem * kStage, bgr, f_begin);
    slot_dem = slot_dem + 1 == nb ? 0 : slot_dem + 1;
    for (int f = f_begin; f < f_end; f++) {
      const bool more = f + 1 < f_end;
      if (more) wait_landed(f + 1);
      lds_write_barrier();
      if (f + 1 + dist < f_end) {
        issue(f + 1 + dist, slot_in);
        slot_in = slot_in + 1 == nb ? 0 : slot_in + 1;
      }
      uint8_t* const img_f = bgr + (((f - f_begin) & 1) ? bgr_bytes : 0u);
      uint8_t* const img_n = bgr + (((f - f_begin) & 1) ? 0u : bgr_bytes);
      if (more) {
        demosaic(lds + (unsigned)slot_dem * kStage, img_n, f + 1);
        slot_dem = slot_dem + 1 == nb ? 0 : slot_dem + 1;
      }
      gather_store(img_f, f);
    }
    lds_write_barrier();  // the next tile's prologue refills the stages and rewrites the colour images
  }
}

// Border pixels of the plan (taps straddling the image edge): cv::remap's per-tap rule with BORDER_CONSTANT 0, every tap
// demosaiced and coloured on the spot from the Bayer frame (debayer_at + the per-pixel stage functions: the same results as
// the packed forms, as every parity test of the generic kernels shows).
__global__ __launch_bounds__(kBlock) void remap_border_bayer_kernel(RemapTiledParams p, ChainParams c) {
  const RemapParams& b = p.base;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.n_border) return;
  const uint32_t packed = p.border_list[i];
  const int yd = (int)(packed >> 16), xd = (int)(packed & 0xffffu);
  const float2 m = reinterpret_cast<const float2*>(b.map_xy)[__umul24((unsigned)yd, (unsigned)b.dcols) + (unsigned)xd];
  const int frame = blockIdx.y;
  const SrcView s{b.src + (size_t)frame * b.src_frame_stride, b.src_step, b.rows, b.cols, SRC_BAYER, c.bayer_ry, c.bayer_rx};
  const GlobalTabs tb{c.tabs};
  FrameWb w = {};
  if (c.wb_mode != WB_NONE) w = c.wb[frame];
  const int sxq = round_map(m.x), syq = round_map(m.y);
  const int sx = clampi(sxq >> 5, -32768, 32767), sy = clampi(syq >> 5, -32768, 32767);
  const int fx = sxq & 31, fy = syq & 31;
  const int wx[2] = {32 - fx, fx}, wy[2] = {32 - fy, fy};
  int acc[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < 2; j++) {
    int rowsum[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int y = sy + j, x = sx + k;
      if (y < 0 || y >= b.rows || x < 0 || x >= b.cols) continue;  // border constant 0
      int pb, pg, pr;
      if (c.flip_angle == 180)
        fetch_src(s, b.rows - 1 - y, b.cols - 1 - x, pb, pg, pr);
      else
        fetch_src(s, y, x, pb, pg, pr);
      pointwise<-1, -1>(c, w, tb, nullptr, nullptr, 1.0f, pb, pg, pr);
      rowsum[0] += mul24(pb, wx[k]);
      rowsum[1] += mul24(pg, wx[k]);
      rowsum[2] += mul24(pr, wx[k]);
    }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) acc[ch] += mul24(rowsum[ch], wy[j]);
  }
  uint8_t* o = b.dst + (size_t)frame * b.dst_frame_stride + (__umul24((unsigned)yd, (unsigned)b.dst_step) + (unsigned)xd * 3u);
  o[0] = (uint8_t)((acc[0] + 512) >> 10);
  o[1] = (uint8_t)((acc[1] + 512) >> 10);
  o[2] = (uint8_t)((acc[2] + 512) >> 10);
}

template <int PRE, int BITS>
void launch_fused_wb(const RemapTiledParams& q, const ChainParams& c, unsigned bgr_off, unsigned bgr_bytes, dim3 grid, unsigned lds, hipStream_t stream) {
  switch (c.wb_mode) {
    case WB_Q8: hipLaunchKernelGGL((remap_bayer_ring_kernel<PRE, BITS, WB_Q8>), grid, dim3(kRemapTileThreads), lds, stream, q, c, bgr_off, bgr_bytes); break;
    case WB_FLOAT: hipLaunchKernelGGL((remap_bayer_ring_kernel<PRE, BITS, WB_FLOAT>), grid, dim3(kRemapTileThreads), lds, stream, q, c, bgr_off, bgr_bytes); break;
    case WB_PCA: hipLaunchKernelGGL((remap_bayer_ring_kernel<PRE, BITS, WB_PCA>), grid, dim3(kRemapTileThreads), lds, stream, q, c, bgr_off, bgr_bytes); break;
    case WB_SIMPLE: hipLaunchKernelGGL((remap_bayer_ring_kernel<PRE, BITS, WB_SIMPLE>), grid, dim3(kRemapTileThreads), lds, stream, q, c, bgr_off, bgr_bytes); break;
    default: hipLaunchKernelGGL((remap_bayer_ring_kernel<PRE, BITS, WB_NONE>), grid, dim3(kRemapTileThreads), lds, stream, q, c, bgr_off, bgr_bytes); break;
  }
}
template <int PRE>
void launch_fused_bits(const RemapTiledParams& q, const ChainParams& c, unsigned bgr_off, unsigned bgr_bytes, dim3 grid, unsigned lds, hipStream_t stream) {
  switch (c.stage_bits & 15) {
    case 0: launch_fused_wb<PRE, 0>(q, c, bgr_off, bgr_bytes, grid, lds, stream); break;
    case ST_CC: launch_fused_wb<PRE, ST_CC>(q, c, bgr_off, bgr_bytes, grid, lds, stream); break;
    case ST_GAMMA: launch_fused_wb<PRE, ST_GAMMA>(q, c, bgr_off, bgr_bytes, grid, lds, stream); break;
    default: launch_fused_wb<PRE, ST_CC | ST_GAMMA>(q, c, bgr_off, bgr_bytes, grid, lds, stream); break;
  }
}

}  // namespace

// p.base.src / src_step / src_frame_stride / rows / cols: the BAYER frames; c: the chain's stage parameters (wb, cc, tabs,
// pattern).  max_rect_w / max_rect_h: the plan's largest source rectangle in pixels.  Returns false -- and launches nothing --
// when the configuration or the geometry does not qualify; the caller then runs the chain and the remap as two kernels.
#if RIP_FP_CONTRACT
#define launch_remap_fused launch_remap_fused_fc1
#endif
bool launch_remap_fused(const RemapTiledParams& p, const ChainParams& c, int max_rect_w, int max_rect_h, const Tunables& tn, hipStream_t stream,
                        bool dry_run) {
#if !RIP_FP_CONTRACT
  if (c.fp_contract == 1) return launch_remap_fused_fc1(p, c, max_rect_w, max_rect_h, tn, stream, dry_run);
#endif
  const RemapParams& b = p.base;
  if (b.n_frames <= 0) return true;
  const bool ok = tn.remap_fused != 0 && tn.remap_ring != 0 && c.src_kind == SRC_BAYER && (c.flip_angle == 0 || c.flip_angle == 180) &&
                  (c.stage_bits & (ST_VIG | ST_HSV)) == 0 && c.tap == nullptr && b.channels == 3 &&
                  bayer_fast_geometry(b.src, b.src_step, b.src_frame_stride, b.rows, b.cols, SRC_BAYER) && b.src_step % 16 == 0 &&
                  b.src_frame_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(b.src) & 15u) == 0 && b.dcols % 4 == 0 && b.dst_step % 4 == 0 &&
                  b.dst_frame_stride % 4 == 0 && aligned4(b.dst) && (reinterpret_cast<uintptr_t>(p.words) & 15u) == 0 &&
                  b.dst_step < (1u << 24) && (unsigned long long)b.dst_step * (unsigned long long)b.drows < (1ull << 32) &&
                  p.tiles_x * kRemapTileW >= b.dcols && p.tiles_y * kRemapTileH >= b.drows && b.drows <= 65535 && b.dcols <= 65535 &&
                  max_rect_w > 0 && max_rect_h > 0;
  if (!ok) return false;
  // upper bounds of the two LDS images over all tiles (a rectangle widened to 4 x 2 items, a dword / a row of halo, the
  // 16-byte alignment of the staged column)
  const unsigned bayer_pitch = ((unsigned)max_rect_w + 3u + 3u + 8u + 15u + 15u) & ~15u;
  const unsigned bayer_rows = (unsigned)max_rect_h + 1u + 2u + 1u;
  const unsigned bayer_chunks = (bayer_pitch >> 4) * bayer_rows;
  if (bayer_chunks > 4u * kRemapTileThreads) return false;
  const int pre = bayer_chunks <= 1u * kRemapTileThreads ? 1 : (bayer_chunks <= 2u * kRemapTileThreads ? 2 : 4);
  const unsigned stage_bytes = (unsigned)pre * kRemapTileThreads * 16u;
  // one colour image (+ the tap reads' overrun), 16-byte granules; two of them: frame f + 1 is demosaiced while frame f is gathered
  const unsigned bgr_bytes = ((((unsigned)max_rect_w + 6u) * 4u + kFusedRowPad) * ((unsigned)max_rect_h + 2u) + 32u + 15u) & ~15u;  // four bytes per pixel + 8 per row (kernel: bp)
  RemapTiledParams q = p;
  q.stages = std::max(2, std::min(4, tn.remap_stages));
  q.deal_run = b.n_frames >= 4 ? remap_deal_run(p.tiles_x, p.tiles_y, tn) : 0;  // batches only (rip_remap.hip)
  const unsigned bgr_off = (unsigned)q.stages * stage_bytes;
  const unsigned lds = ((bgr_off + 2u * bgr_bytes) + 15u) & ~15u;
  if (lds > 60u * 1024u) {
    // the two-kernel path takes over; RIP_DEBUG_OCC says so (the four-byte colour image of round 5 made this limit bind for more
    // geometries: ADVICE round 5)
    if (tn.debug_occupancy && !dry_run) std::fprintf(stderr, "rip: chain inside the remap's tiles refused: %u bytes of LDS per workgroup (limit 61440), two kernels instead\n", lds);
    return false;
  }
  if (dry_run) return true;
  const int ntiles = p.tiles_x * p.tiles_y;
  // persistent workgroups per CU of the grid -- not clamped to what the LDS lets reside (round 5, 28.6 KB per workgroup = five
  // resident: a grid of 6 or 7 per CU runs 2.58-2.60 ms per 256 frames at 3840 x 2160 against 2.61-2.62 with 5)
  const int per_cu = std::max(1, tn.remap_per_cu > 0 ? tn.remap_per_cu : 6);
  int blocks = std::min(256 * per_cu, (ntiles + 7) / 8 * 8);
  blocks = std::max(8, blocks / 8 * 8);
  // frames per tile visit: the per-tile set-up (item geometry, plan words: a quarter of the traffic at 4 frames per visit)
  // is paid once per visit and the Bayer frames of a visit are small, so long visits win -- config5, ms per 256 frames at
  // 2 / 3 / 4 / 6 / 8 / 12 / 16 frames per visit: 5.04 / 4.42 / 4.08 / 3.77 / 3.70 / 3.53 / 3.47
  int frames_per_visit = tn.remap_frames > 0 ? tn.remap_frames : kFusedMaxFrames;
  frames_per_visit = std::min(frames_per_visit, kFusedMaxFrames);
  int groups = std::max((256 * per_cu) / blocks, (b.n_frames + frames_per_visit - 1) / frames_per_visit);
  groups = std::max(1, std::min(b.n_frames, groups));
  const dim3 grid(blocks, groups);
  if (pre == 1)
    launch_fused_bits<1>(q, c, bgr_off, bgr_bytes, grid, lds, stream);
  else if (pre == 2)
    launch_fused_bits<2>(q, c, bgr_off, bgr_bytes, grid, lds, stream);
  else
    launch_fused_bits<4>(q, c, bgr_off, bgr_bytes, grid, lds, stream);
  if (q.n_border > 0) hipLaunchKernelGGL(remap_border_bayer_kernel, dim3((q.n_border + 255) / 256, b.n_frames), dim3(256), 0, stream, q, c);
  return true;
}

}  // namespace rip
