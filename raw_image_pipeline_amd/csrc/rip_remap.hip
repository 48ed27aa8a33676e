// rip_remap.hip -- undistortion remap (cv::remap INTER_LINEAR, BORDER_CONSTANT): LDS-DMA ring over a compiled plan,
// border patch kernel and the direct-gather fallbacks.
// Shared device code and the stage-by-stage reference citations: rip_device.hpp.
#include "rip_device.hpp"
#include "rip_remap_dev.hpp"

namespace rip {
namespace {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 4 destination pixels per thread (CN == 3, dcols % 4 == 0, dword-aligned pitch)
__global__ __launch_bounds__(kBlock) void remap_vec4_kernel(RemapParams p, ItemMap im, int items_per_frame) {
  const int frame = blockIdx.y;
  const RemapSrc s = remap_src(p, frame);
  uint8_t* dst = p.dst + (size_t)frame * p.dst_frame_stride;
  for (int item = blockIdx.x * kBlock + threadIdx.x; item < items_per_frame; item += gridDim.x * kBlock) {
    int yd, grp;
    im.split(item, yd, grp);
    const int xd = grp * 4;
    const float4* m = reinterpret_cast<const float4*>(p.map_xy + ((size_t)(__umul24((unsigned)yd, (unsigned)p.dcols) + (unsigned)xd)) * 2);
    const float4 m0 = m[0], m1 = m[1];
    int q[4][3];
    remap_pixel<3>(s, m0.x, m0.y, q[0]);
    remap_pixel<3>(s, m0.z, m0.w, q[1]);
    remap_pixel<3>(s, m1.x, m1.y, q[2]);
    remap_pixel<3>(s, m1.z, m1.w, q[3]);
    store12(dst + (__umul24((unsigned)yd, (unsigned)p.dst_step) + (unsigned)xd * 3u), pack4(q));
  }
}

// ------------------------------------------------------------------------------------------------
// tiled remap over a compiled plan: one workgroup = one 64x16 destination tile.  The tile's source
// rectangle is copied into LDS with aligned 16-byte loads (every source byte crosses the memory
// pipeline once per tile instead of once per tap), the taps are read back from LDS, and the
// bilinear weights are applied with v_dot4_u32_u8.  The plan word (4 B/px) and the tile descriptor
// are read once per tile and reused for every frame of the batch.
// ------------------------------------------------------------------------------------------------
// Destination pixels whose taps straddle the image border (a few thousand per map) are listed by the
// plan compiler and patched after the tiled kernel by this per-tap kernel, which keeps the heavy
// border logic out of the tiled kernel's register budget.
__global__ __launch_bounds__(kBlock) void remap_border_kernel(RemapTiledParams p) {
  const RemapParams& b = p.base;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.n_border) return;
  const uint32_t packed = p.border_list[i];
  const int yd = (int)(packed >> 16), xd = (int)(packed & 0xffffu);
  const float2 m = reinterpret_cast<const float2*>(b.map_xy)[__umul24((unsigned)yd, (unsigned)b.dcols) + (unsigned)xd];
  const int frame = blockIdx.y;
  const RemapSrc s = remap_src(b, frame);
  if (b.channels == 1 && (p.mono_lut || p.mono_flip180)) {  // uniform: the taps are pixels of the flipped, table-mapped frame
    const int sxq = round_map(m.x), syq = round_map(m.y);
    const int sx = clampi(sxq >> 5, -32768, 32767), sy = clampi(syq >> 5, -32768, 32767);
    const int fx = sxq & 31, fy = syq & 31;
    const int wx[2] = {32 - fx, fx}, wy[2] = {32 - fy, fy};
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int rowsum = 0;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int y = sy + j, x = sx + k;
        if (y < 0 || y >= b.rows || x < 0 || x >= b.cols) continue;  // border constant 0
        const int ys = p.mono_flip180 ? b.rows - 1 - y : y, xs = p.mono_flip180 ? b.cols - 1 - x : x;
        int v = s.frame[__umul24((unsigned)ys, s.step) + (unsigned)xs];
        if (p.mono_lut) v = p.mono_lut[v];
        rowsum += mul24(v, wx[k]);
      }
      acc += mul24(rowsum, wy[j]);
    }
    b.dst[(size_t)frame * b.dst_frame_stride + (__umul24((unsigned)yd, (unsigned)b.dst_step) + (unsigned)xd)] = (uint8_t)((acc + 512) >> 10);
    return;
  }
  if (b.channels == 1) {  // uniform
    int q[1];
    remap_pixel<1>(s, m.x, m.y, q);
    b.dst[(size_t)frame * b.dst_frame_stride + (__umul24((unsigned)yd, (unsigned)b.dst_step) + (unsigned)xd)] = (uint8_t)q[0];
    return;
  }
  int q[3];
  remap_pixel<3>(s, m.x, m.y, q);
  uint8_t* d = b.dst + (size_t)frame * b.dst_frame_stride + (__umul24((unsigned)yd, (unsigned)b.dst_step) + (unsigned)xd * 3u);
  d[0] = (uint8_t)q[0];
  d[1] = (uint8_t)q[1];
  d[2] = (uint8_t)q[2];
}

// PRE: staging slots (16-byte chunks) per lane held in registers while the previous frame is gathered;
// 0 = no software pipeline (one LDS buffer, any rectangle size)
template <int PRE>
__global__ __launch_bounds__(kRemapTileThreads) void remap_tiled_kernel(RemapTiledParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int kPre = PRE > 0 ? PRE : 1;
  const RemapParams& b = p.base;
  const int ntiles = p.tiles_x * p.tiles_y;
  const int per_xcd = (ntiles + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int tid = threadIdx.x;
  const int lrow = tid / kRemapGroupsPerRow, lgrp = tid % kRemapGroupsPerRow;
  const unsigned step = (unsigned)b.src_step;
  const int f_per_group = (b.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(b.n_frames, f_begin + f_per_group);
  if (f_begin >= f_end) return;  // uniform for the workgroup
  uint8_t* const buf0 = lds;
  uint8_t* const buf1 = lds + p.lds_bytes;  // second staging buffer (double_buffer only)
  for (int ti = blockIdx.x >> 3; ti < per_xcd; ti += gridDim.x >> 3) {
    const int tile = xcd * per_xcd + ti;
    if (tile >= ntiles) break;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const RemapTileDesc d = p.tiles[tile];
    const uint4 wd = reinterpret_cast<const uint4*>(p.words + (size_t)tile * kRemapTilePx)[tid];
    const uint32_t words[4] = {wd.x, wd.y, wd.z, wd.w};
    const int yd = ty * kRemapTileH + lrow, xd = tx * kRemapTileW + lgrp * 4;
    const bool in_image = yd < b.drows && xd < b.dcols;
    const unsigned xbyte0 = (unsigned)d.x0 * 3u;
    const unsigned chunk0 = xbyte0 & ~15u, ph = xbyte0 & 15u;
    const unsigned pitch = (ph + (unsigned)d.w * 3u + 15u) & ~15u;  // rip_host.cpp remap_tile_lds_bytes
    const unsigned chunks = pitch >> 4;
    const unsigned total = d.w > 0 ? chunks * (unsigned)d.h : 0u;
    const ItemMap cm{(int)chunks, 1.0f / (float)(chunks ? chunks : 1u)};
    // frame-invariant part of the plan words: LDS address of the top-left tap and the x weights
    unsigned tap_addr[4], wxb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t w = words[k];
      const unsigned relx = w & 0x7ffu, rely = (w >> 11) & 0x7ffu, fx = (w >> 22) & 31u;
      tap_addr[k] = __umul24(rely, pitch) + relx * 3u + ph;
      wxb[k] = (32u - fx) | (fx << 24);
    }
    const unsigned dst_off = __umul24((unsigned)yd, (unsigned)b.dst_step) + (unsigned)xd * 3u;

    auto gather_store = [&](const uint8_t* buf, int f) {
      if (!in_image) return;
      int q[4][3];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t w = words[k];
        if (w >= kPlanBorder) {
          q[k][0] = q[k][1] = q[k][2] = 0;  // outside: border constant; border pixels: patched by remap_border_kernel
          continue;
        }
        uint32_t t0, t1, b0, b1;  // bytes b0 g0 r0 b1 | g1 r1 . .  of the top / bottom tap rows
        lds_load6(buf, tap_addr[k], t0, t1);
        lds_load6(buf, tap_addr[k] + pitch, b0, b1);
        const unsigned fy = w >> 27, wy0 = 32u - fy, wy1 = fy;
        const unsigned wB = wxb[k], wx0 = wB & 0xffu, wx1 = wB >> 24;
        const unsigned wG0 = wx0 << 8, wR0 = wx0 << 16, wR1 = wx1 << 8;
        // every row sum starts at 16: 16 * (32 - fy) + 16 * fy = 512 is the rounding term of
        // ((top*(32-fy) + bot*fy) * 32 + 2^14) >> 15, exact
        const unsigned topB = __builtin_amdgcn_udot4(t0, wB, 16u, false);
        const unsigned topG = __builtin_amdgcn_udot4(t0, wG0, __builtin_amdgcn_udot4(t1, wx1, 16u, false), false);
        const unsigned topR = __builtin_amdgcn_udot4(t0, wR0, __builtin_amdgcn_udot4(t1, wR1, 16u, false), false);
        const unsigned botB = __builtin_amdgcn_udot4(b0, wB, 16u, false);
        const unsigned botG = __builtin_amdgcn_udot4(b0, wG0, __builtin_amdgcn_udot4(b1, wx1, 16u, false), false);
        const unsigned botR = __builtin_amdgcn_udot4(b0, wR0, __builtin_amdgcn_udot4(b1, wR1, 16u, false), false);
        q[k][0] = blend_rows(topB, wy0, botB, wy1);
        q[k][1] = blend_rows(topG, wy0, botG, wy1);
        q[k][2] = blend_rows(topR, wy0, botR, wy1);
      }
      uint8_t* dst = b.dst + (size_t)f * b.dst_frame_stride;
      store12(dst + dst_off, pack4(q));
    };

    if (PRE > 0) {
      // software pipeline: the global loads of frame f+1 are in flight while frame f is gathered out
      // of the other LDS buffer; one barrier per frame
      unsigned goff[kPre], loff[kPre];
#pragma unroll
      for (int j = 0; j < kPre; j++) {
        const unsigned i = (unsigned)tid + (unsigned)j * kRemapTileThreads;
        int r, c;
        cm.split((int)i, r, c);
        goff[j] = i < total ? __umul24((unsigned)(d.y0 + r), step) + chunk0 + ((unsigned)c << 4) : 0xFFFFFFFFu;
        loff[j] = __umul24((unsigned)r, pitch) + ((unsigned)c << 4);
      }
      uint4 pre[kPre];
      auto issue = [&](const RemapSrc& s) {
#pragma unroll
        for (int j = 0; j < kPre; j++) {
          pre[j] = make_uint4(0u, 0u, 0u, 0u);
          if (goff[j] != 0xFFFFFFFFu && goff[j] + 16u <= s.readable) pre[j] = *reinterpret_cast<const uint4*>(s.frame + goff[j]);
        }
      };
      auto commit = [&](uint8_t* buf) {
#pragma unroll
        for (int j = 0; j < kPre; j++)
          if (goff[j] != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(buf + loff[j]) = pre[j];
      };
      issue(remap_src(b, f_begin));
      commit(buf0);
      __syncthreads();
      for (int f = f_begin; f < f_end; f++) {
        uint8_t* cur = ((f - f_begin) & 1) ? buf1 : buf0;
        uint8_t* nxt = ((f - f_begin) & 1) ? buf0 : buf1;
        const bool more = f + 1 < f_end;
        if (more) issue(remap_src(b, f + 1));
        gather_store(cur, f);
        if (more) commit(nxt);
        __syncthreads();
      }
    } else {
      for (int f = f_begin; f < f_end; f++) {
        const RemapSrc s = remap_src(b, f);
        for (unsigned i = tid; i < total; i += kRemapTileThreads) {
          int r, c;
          cm.split((int)i, r, c);
          const unsigned off = __umul24((unsigned)(d.y0 + r), step) + chunk0 + ((unsigned)c << 4);
          uint4 v = make_uint4(0u, 0u, 0u, 0u);
          if (off + 16u <= s.readable) v = *reinterpret_cast<const uint4*>(s.frame + off);
          *reinterpret_cast<uint4*>(buf0 + (__umul24((unsigned)r, pitch) + ((unsigned)c << 4))) = v;
        }
        __syncthreads();
        gather_store(buf0, f);
        __syncthreads();
      }
    }
  }
}

// CN: channels of the image (3: interleaved BGR; 1: mono8 frames, which the reference passes through flip, gamma and
// undistortion unchanged in layout).  The plan -- source rectangles in pixels, 1/32-px tap words -- does not depend on it.
// LUT (CN == 1 only): the taps go through p.mono_lut, i.e. the gather reads the gamma-corrected frame without it ever being
// written; p.mono_flip180 (CN == 1, either LUT): the staged rectangle is the mirrored one of the caller's frame and the four
// taps of a pixel are addressed from its far corner, weights swapped -- the sums are the same integers in another order.
template <int PRE, int CN, bool LUT = false>
__global__ __launch_bounds__(kRemapTileThreads) void remap_ring_kernel(RemapTiledParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  __shared__ uint8_t s_lut[LUT ? 256 : 4];
  static_assert(!LUT || (CN == 1 && kRemapTileThreads == 256), "the table pass is the mono8 chain");
  if constexpr (LUT) {
    s_lut[threadIdx.x] = p.mono_lut[threadIdx.x];
    __syncthreads();
  }
  const bool flip = CN == 1 && p.mono_flip180 != 0;
  constexpr unsigned kStage = (unsigned)PRE * kRemapTileThreads * 16u;  // bytes per stage
  const RemapParams& b = p.base;
  const int ntiles = p.tiles_x * p.tiles_y;
  const TileDeal deal(ntiles, p.deal_run);
  const int xcd = blockIdx.x & 7;
  const int tid = threadIdx.x;
  const int lrow = tid / kRemapGroupsPerRow, lgrp = tid % kRemapGroupsPerRow;
  const unsigned step = (unsigned)b.src_step;
  const int f_per_group = (b.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(b.n_frames, f_begin + f_per_group);
  if (f_begin >= f_end) return;  // uniform for the workgroup
  const int nb = p.stages, dist = nb - 1;  // ring size, prefetch distance (2 or 3)
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);
  const unsigned wave_chunk0 = (unsigned)__builtin_amdgcn_readfirstlane(tid & ~63);
  const unsigned dst_bytes = __umul24((unsigned)(b.drows - 1), (unsigned)b.dst_step) + (unsigned)b.dcols * (unsigned)CN;
#ifdef RIP_EXPERIMENTS
  // timing-only switches (wrong pixels): 1 no per-frame barrier, 2 tile-linear stores, 4 tile-linear source loads,
  // 8 no gather (LDS reads + arithmetic), 16 no stores, 32 no source loads, 64 no plan words (synthesised),
  // 128 the counted waits leave the two youngest stores out (PRE + 2 operations may be in flight), 256 stores as 16-byte lanes
  // (lanes < 192) at tile-linear positions, 512 stores as 16-byte lanes at the tile's row-major positions (12 lanes per row)
  const int ex = __builtin_amdgcn_readfirstlane(p.exp);
#else
  constexpr int ex = 0;
#endif
  for (int ti = blockIdx.x >> 3; ti < deal.per_xcd; ti += gridDim.x >> 3) {
    const int tile = deal.tile(ti, xcd);
    if (tile == -1) continue;
    if (tile < 0) break;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const RemapTileDesc d = p.tiles[tile];
    uint4 wd;
    if (ex & 64) {
      const unsigned r0 = (unsigned)(tid / kRemapGroupsPerRow), c0 = (unsigned)(tid % kRemapGroupsPerRow) * 4u;
      const unsigned w0 = (c0 & 0x7ffu) | (r0 << 11) | (5u << 22) | (9u << 27);
      wd = make_uint4(w0, w0 + 1, w0 + 2, w0 + 3);
    } else {
      wd = reinterpret_cast<const uint4*>(p.words + (size_t)tile * kRemapTilePx)[tid];
    }
    const uint32_t words[4] = {wd.x, wd.y, wd.z, wd.w};
    const int yd = ty * kRemapTileH + lrow, xd = tx * kRemapTileW + lgrp * 4;
    const bool in_image = yd < b.drows && xd < b.dcols;
    // origin of the staged rectangle in the frame the kernel reads (the plan's rectangle, mirrored for flip)
    const int ox = flip ? b.cols - d.x0 - d.w : d.x0, oy = flip ? b.rows - d.y0 - d.h : d.y0;
    const unsigned xbyte0 = (unsigned)ox * (unsigned)CN;
    const unsigned chunk0 = xbyte0 & ~15u, ph = xbyte0 & 15u;
    const unsigned pitch = (ph + (unsigned)d.w * (unsigned)CN + 15u) & ~15u;  // rip_host.cpp remap_tile_lds_bytes (CN == 3)
    const unsigned chunks = pitch >> 4;
    const unsigned total = d.w > 0 ? chunks * (unsigned)d.h : 0u;
    const ItemMap cm{(int)chunks, 1.0f / (float)(chunks ? chunks : 1u)};
    // frame-invariant: source offsets of this lane's chunks, LDS address of the top-left tap, weights.
    // Outside / border pixels gather address 0 with zero weights: 16 * 32 >> 10 == 0 is the border
    // constant, and remap_border_kernel patches the border pixels afterwards -- no branch per pixel.
    unsigned goff[PRE];
#pragma unroll
    for (int j = 0; j < PRE; j++) {
      const unsigned i = (unsigned)tid + (unsigned)j * kRemapTileThreads;
      int r, c;
      cm.split((int)i, r, c);
      goff[j] = i < total ? __umul24((unsigned)(oy + r), step) + chunk0 + ((unsigned)c << 4) : 0xFFFFFFF0u;
      if ((ex & 4) && i < total) goff[j] = ((unsigned)tile * 4608u) % (step * (unsigned)(b.rows - 2)) + (i << 4);
      if (ex & 32) goff[j] = 0xFFFFFFF0u;
    }
    unsigned tap_addr[4], wxb[4], wyy[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t w = words[k];
      const bool live = w < kPlanBorder;
      const unsigned relx = w & 0x7ffu, rely = (w >> 11) & 0x7ffu, fx = (w >> 22) & 31u, fy = w >> 27;
      if (flip) {  // the dword at (h-2-rely, w-2-relx) of the mirrored rectangle holds the taps (x+1, x) of row y+1; the row below it is y
        tap_addr[k] = live ? __umul24((unsigned)d.h - 2u - rely, pitch) + ((unsigned)d.w - 2u - relx) + ph : 0u;
        wxb[k] = live ? fx | ((32u - fx) << 24) : 0u;
        wyy[k] = fy | ((32u - fy) << 16);
      } else {
        tap_addr[k] = live ? __umul24(rely, pitch) + relx * (unsigned)CN + ph : 0u;
        wxb[k] = live ? (32u - fx) | (fx << 24) : 0u;
        wyy[k] = (32u - fy) | (fy << 16);
      }
    }
    unsigned dst_off = __umul24((unsigned)yd, (unsigned)b.dst_step) + (unsigned)xd * (unsigned)CN;
    if (ex & 2) dst_off = ((unsigned)tile * (unsigned)kRemapTilePx + (unsigned)tid * 4u) * (unsigned)CN % (dst_bytes - 64u) & ~3u;

    auto issue = [&](int f, int slot) {
      const RemapSrc s = remap_src(b, f);
      const __amdgpu_buffer_rsrc_t rsrc = frame_rsrc(s.frame, s.readable);
      const unsigned stage = lds0 + (unsigned)slot * kStage;
#pragma unroll
      for (int j = 0; j < PRE; j++) lds_dma16(rsrc, goff[j], stage + ((wave_chunk0 + (unsigned)j * kRemapTileThreads) << 4));
    };
    auto gather_store = [&](const uint8_t* buf, int f) {
      if (!in_image) return;
      if constexpr (CN == 1) {
        // one byte per tap: the two taps of a row are the low bytes of the realigned dword, the x weights bytes 0 and 1 of
        // wxb (byte 3, the colour kernel's place for fx, is moved down once per tile below), so a row sum is one v_dot4
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint32_t tl = lds_load4(buf, tap_addr[k]), bl = lds_load4(buf, tap_addr[k] + pitch);
          if constexpr (LUT) {
            tl = (uint32_t)s_lut[tl & 0xffu] | ((uint32_t)s_lut[(tl >> 8) & 0xffu] << 8);
            bl = (uint32_t)s_lut[bl & 0xffu] | ((uint32_t)s_lut[(bl >> 8) & 0xffu] << 8);
          }
          const unsigned wy0 = wyy[k] & 0xffffu, wy1 = wyy[k] >> 16;
          const unsigned w2 = (wxb[k] & 0xffu) | ((wxb[k] >> 16) & 0xff00u);
          const unsigned top = __builtin_amdgcn_udot4(tl, w2, 16u, false);
          const unsigned bot = __builtin_amdgcn_udot4(bl, w2, 16u, false);
          out |= (uint32_t)blend_rows(top, wy0, bot, wy1) << (8 * k);
        }
        __builtin_amdgcn_raw_buffer_store_b32(out, frame_rsrc(b.dst + (size_t)f * b.dst_frame_stride, dst_bytes), (int)dst_off, 0, 0);
        return;
      }
      uint32_t t0[4], t1[4], b0[4], b1[4];  // bytes b0 g0 r0 b1 | g1 r1 . .  of the top / bottom tap rows
      if (ex & (256 | 512)) {
        if (tid < 192) {
          unsigned o16 = (((unsigned)tile * 3072u) % (dst_bytes - 4096u) & ~15u) + (unsigned)tid * 16u;
          if (ex & 512) o16 = __umul24((unsigned)(ty * kRemapTileH + tid / 12), (unsigned)b.dst_step) + (unsigned)(tx * kRemapTileW) * 3u + (unsigned)(tid % 12) * 16u;
          u32x4 v4 = {tap_addr[0], wxb[1], wyy[2], tap_addr[3]};
          if (!(ex & 8)) {
            uint32_t a0, a1;
            lds_load6(buf, tap_addr[0], a0, a1);
            v4[0] = a0; v4[1] = a1;
          }
          __builtin_amdgcn_raw_buffer_store_b128(v4, frame_rsrc(b.dst + (size_t)f * b.dst_frame_stride, dst_bytes), (int)o16, 0, 0);
        }
        return;
      }
      if (ex & 8) {
        if (!(ex & 16)) store12(frame_rsrc(b.dst + (size_t)f * b.dst_frame_stride, dst_bytes), dst_off, Pack3{tap_addr[0], wxb[1], wyy[2]});
        return;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        lds_load6(buf, tap_addr[k], t0[k], t1[k]);
        lds_load6(buf, tap_addr[k] + pitch, b0[k], b1[k]);
      }
      const Pack3 px = blend4_bgr(t0, t1, b0, b1, wxb, wyy);
      if (ex & 16) {
        if (px.a == 0x12345678u && px.b == 0x9abcdef0u && px.c == 0x0fedcba9u) store12(frame_rsrc(b.dst + (size_t)f * b.dst_frame_stride, dst_bytes), dst_off, px);
        return;
      }
#ifdef RIP_EXPERIMENTS
      if (ex & (2048 | 4096)) {  // cache-policy bits of the output stores under the round-robin deal: 2048 nt, 4096 sc1, both: nt sc1 (correct pixels)
        const u32x3 u = {px.a, px.b, px.c};
        const __amdgpu_buffer_rsrc_t rs = frame_rsrc(b.dst + (size_t)f * b.dst_frame_stride, dst_bytes);
        if ((ex & (2048 | 4096)) == 2048)
          __builtin_amdgcn_raw_buffer_store_b96(u, rs, (int)dst_off, 0, 2);
        else if ((ex & (2048 | 4096)) == 4096)
          __builtin_amdgcn_raw_buffer_store_b96(u, rs, (int)dst_off, 0, 16);
        else
          __builtin_amdgcn_raw_buffer_store_b96(u, rs, (int)dst_off, 0, 18);
        return;
      }
#endif
      store12(frame_rsrc(b.dst + (size_t)f * b.dst_frame_stride, dst_bytes), dst_off, px);
    };

    // every earlier memory operation of this wave (plan words, tile descriptor, previous stores) is
    // waited for here, so the counted waits below see only this tile's ring loads and stores
    wait_vmcnt<0>();
    int slot_in = 0, slot_out = 0;  // ring positions of the next frame to issue / to gather
    for (int f = f_begin; f < f_end && f < f_begin + dist; f++) {
      issue(f, slot_in);
      slot_in = slot_in + 1 == nb ? 0 : slot_in + 1;
    }
    for (int f = f_begin; f < f_end; f++) {
      const int ahead = min(dist - 1, f_end - 1 - f);  // frames issued after f and still allowed in flight
      if (ex & 128) {
        if (ahead >= 2)
          wait_vmcnt<2 * PRE + 2>();
        else if (ahead == 1)
          wait_vmcnt<PRE + 2>();
        else
          wait_vmcnt<(PRE + 2 < 2 ? PRE + 2 : 2)>();
      } else if (ahead >= 2)
        wait_vmcnt<2 * PRE>();
      else if (ahead == 1)
        wait_vmcnt<PRE>();
      else
        wait_vmcnt<0>();
      if (!(ex & 1)) __builtin_amdgcn_s_barrier();
      if (f + dist < f_end) {
        issue(f + dist, slot_in);
        slot_in = slot_in + 1 == nb ? 0 : slot_in + 1;
      }
      gather_store(lds + (unsigned)slot_out * kStage, f);
      slot_out = slot_out + 1 == nb ? 0 : slot_out + 1;
    }
    __builtin_amdgcn_s_barrier();  // the next tile's prologue refills stages other waves may still be reading
  }
}

template <int CN>
__global__ __launch_bounds__(kBlock) void remap_generic_kernel(RemapParams p) {
  const int frame = blockIdx.y;
  const RemapSrc s = remap_src(p, frame);
  uint8_t* dst = p.dst + (size_t)frame * p.dst_frame_stride;
  const long long npix = (long long)p.drows * p.dcols;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < npix; i += (long long)gridDim.x * kBlock) {
    int yd = (int)(i / p.dcols), xd = (int)(i - (long long)yd * p.dcols);
    const float* m = p.map_xy + (size_t)i * 2;
    int o[CN];
    remap_pixel<CN>(s, m[0], m[1], o);
    uint8_t* d = dst + (size_t)yd * p.dst_step + (size_t)xd * CN;
#pragma unroll
    for (int c = 0; c < CN; c++) d[c] = (uint8_t)o[c];
  }
}

}  // namespace

int remap_deal_run(int tiles_x, int tiles_y, const Tunables& tn) {
  if (tn.remap_deal <= 0 || tiles_x <= 0 || tiles_y <= 0) return 0;
  // runs of about remap_deal tile rows, sized so that every XCD gets the same number of runs (a ragged last run at most)
  const int ntiles = tiles_x * tiles_y;
  const int runs_per_xcd = (tiles_y + 8 * tn.remap_deal - 1) / (8 * tn.remap_deal);
  return (ntiles + 8 * runs_per_xcd - 1) / (8 * runs_per_xcd);
}

bool launch_remap_tiled(const RemapTiledParams& p, const Tunables& tn, hipStream_t stream, bool dry_run) {
  const RemapParams& b = p.base;
  if (b.n_frames <= 0) return true;
  if ((p.mono_lut || p.mono_flip180) && b.channels != 1) return false;
  const bool ok = (b.channels == 3 || b.channels == 1) && b.dcols % 4 == 0 && b.dst_step % 4 == 0 && b.dst_frame_stride % 4 == 0 && aligned4(b.dst) &&
                  b.src_step % 16 == 0 && b.src_frame_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(b.src) & 15u) == 0 &&
                  (reinterpret_cast<uintptr_t>(p.words) & 15u) == 0 && b.src_step < (1u << 24) && b.rows < (1 << 23) &&
                  (unsigned long long)b.src_step * (unsigned long long)b.rows < (1ull << 32) && b.dst_step < (1u << 24) &&
                  (unsigned long long)b.dst_step * (unsigned long long)b.drows < (1ull << 32) && p.lds_bytes <= 64u * 1024u &&
                  p.tiles_x * kRemapTileW >= b.dcols && p.tiles_y * kRemapTileH >= b.drows && b.drows <= 65535 && b.dcols <= 65535;
  if (!ok) return false;
  const int ntiles = p.tiles_x * p.tiles_y;
  // persistent workgroups, a multiple of 8 (one share of the tile range per XCD); LDS bounds residency
  RemapTiledParams q = p;
  q.lds_bytes = (std::max(p.lds_bytes, 16u) + 15u) & ~15u;
  const unsigned chunks = q.lds_bytes / 16u;  // upper bound of the 16-byte chunks of any tile
  const int ring_env = tn.remap_ring;
  // measured on config2 (sweeps in DESIGN.md): 3 stages (two frames ahead) with 4 workgroups per CU; more resident
  // workgroups fetch more (the source rectangles of neighbouring tiles stop meeting in L2) and run slower
  const int stages_env = tn.remap_stages;
  // one-channel frames run the ring kernel only (PRE chosen from the three-channel footprint of the plan: an upper bound of
  // the one-channel one, the surplus lanes load from an out-of-range offset, i.e. nothing)
  if (b.channels == 1 && !(ring_env && chunks <= 4u * kRemapTileThreads)) return false;
  if (dry_run) return true;
  if (ring_env && chunks <= 4u * kRemapTileThreads) {
    // LDS-DMA ring: PRE chunks per lane and frame, `stages` buffers of PRE * 4 KiB
    const int pre = chunks <= 1u * kRemapTileThreads ? 1 : (chunks <= 2u * kRemapTileThreads ? 2 : 4);
    const unsigned stage_bytes = (unsigned)pre * kRemapTileThreads * 16u;
    q.stages = std::max(2, std::min(4, stages_env));
    q.exp = tn.remap_exp;
    // batches only: a single frame (latency calls: 63.5 against 64.9 us for the reference's example configuration) gains nothing
    // from one band across the chip and loses the neighbouring rectangles' meeting in one L2
    q.deal_run = b.n_frames >= 4 ? remap_deal_run(p.tiles_x, p.tiles_y, tn) : 0;
    const unsigned lds = (unsigned)q.stages * stage_bytes + 16u;  // the three-dword tap reads run up to 11 B past a row
    const int per_cu = std::max(1, std::min(tn.remap_per_cu > 0 ? tn.remap_per_cu : 6, (int)((160u * 1024u) / (lds + 256u))));
    int blocks = std::min(256 * per_cu, (ntiles + 7) / 8 * 8);
    blocks = std::max(8, blocks / 8 * 8);
    // blockIdx.y splits the batch into groups of frames: enough groups to fill the chip when there are few tiles,
    // and a bounded number of frames per tile visit (RIP_REMAP_FRAMES) -- the workgroups in flight then work on the
    // same few source frames (L2 / Infinity Cache resident, so the overlapping rectangles of neighbouring tiles are
    // fetched once) and the dispatcher balances many more, smaller units.  Measured on config2, 64 frames: 64 frames per
    // visit 0.61 ms, 8: 0.57, 4: 0.54, 2: 0.58, 1: 0.71.  Round 3, 256 frames, ms per launch at 3 / 4 / 6 / 8 / 12 / 16 frames
    // per visit: 3840x2160 3.35 / 3.22 / 3.17 / 3.16 / 3.21 / 3.31; 2448x2048 2.05 / 2.02 / 2.08 / 2.19 / 2.31 / 2.39; 1920x1200
    // 1.00 / 0.96 / 0.90 / 0.88 / 0.87 / 0.88; 1440x1080 0.74 / 0.71 / 0.67 / 0.65 / 0.64 / 0.64 -- the best count keeps the
    // source frames of a visit near 64 MB, so that is the default rule (4 .. 12 frames).
    int frames_per_visit = tn.remap_frames;
    if (frames_per_visit <= 0) {
      const unsigned long long frame_bytes = (unsigned long long)b.src_step * (unsigned long long)b.rows;
      // round 6, with the round-robin deal (one tile row per run) longer visits paid at first (2448x2048 at 4 / 6 / 8 / 12 / 16 frames: 2.091 /
      // 1.955 / 1.935 / 1.998 / 2.090); with runs of four tile rows, inside the config 2 step, the optimum is back at the 64 MB rule:
      // 3 / 4 / 5 / 6 / 8 / 12 frames 1.920 / 1.860 / 1.874 / 1.902 / 1.999 / 2.135 (another box: . / . / 1.744 / 1.771 / 1.863 / .); 1440x1080 at
      // 4 / 6 / 8 / 12: 0.569 / 0.533 / 0.526 / 0.533; 1920x1200 0.804 / 0.756 / 0.738 / 0.736; 3840x2160 3.077 / 3.054 / 3.066 / 3.149 -- the
      // fewer frames the grid writes at once, the better for the write stream (tools/probes/write_pattern_probe.hip), the more the
      // plan words and the per-visit start-up cost the read side
      frames_per_visit = (int)std::min<unsigned long long>(12, std::max<unsigned long long>(4, ((64ull << 20) + frame_bytes / 2) / std::max<unsigned long long>(1, frame_bytes)));
    }
    int groups = std::max((256 * per_cu) / blocks, (b.n_frames + frames_per_visit - 1) / frames_per_visit);
    groups = std::max(1, std::min(b.n_frames, groups));
    const dim3 grid(blocks, groups);
    if (b.channels == 1 && q.mono_lut) {
      if (pre == 1)
        hipLaunchKernelGGL((remap_ring_kernel<1, 1, true>), grid, dim3(kRemapTileThreads), lds, stream, q);
      else if (pre == 2)
        hipLaunchKernelGGL((remap_ring_kernel<2, 1, true>), grid, dim3(kRemapTileThreads), lds, stream, q);
      else
        hipLaunchKernelGGL((remap_ring_kernel<4, 1, true>), grid, dim3(kRemapTileThreads), lds, stream, q);
    } else if (b.channels == 1) {
      if (pre == 1)
        hipLaunchKernelGGL((remap_ring_kernel<1, 1>), grid, dim3(kRemapTileThreads), lds, stream, q);
      else if (pre == 2)
        hipLaunchKernelGGL((remap_ring_kernel<2, 1>), grid, dim3(kRemapTileThreads), lds, stream, q);
      else
        hipLaunchKernelGGL((remap_ring_kernel<4, 1>), grid, dim3(kRemapTileThreads), lds, stream, q);
    } else if (pre == 1)
      hipLaunchKernelGGL((remap_ring_kernel<1, 3>), grid, dim3(kRemapTileThreads), lds, stream, q);
    else if (pre == 2)
      hipLaunchKernelGGL((remap_ring_kernel<2, 3>), grid, dim3(kRemapTileThreads), lds, stream, q);
    else
      hipLaunchKernelGGL((remap_ring_kernel<4, 3>), grid, dim3(kRemapTileThreads), lds, stream, q);
  } else {
    // rectangles larger than 4 * kRemapTileThreads chunks (strong local magnification) or RIP_REMAP_RING=0 (A/B runs)
    int pre = !ring_env && b.n_frames >= 2 && chunks <= 2u * kRemapTileThreads ? 2 : 0;
    q.double_buffer = pre > 0 ? 1 : 0;
    const unsigned lds = (q.double_buffer ? 2u * q.lds_bytes : q.lds_bytes) + 16u;
    const int per_cu = std::max(1, std::min(tn.remap_per_cu > 0 ? tn.remap_per_cu : 8, (int)((160u * 1024u) / (lds + 256u))));
    int blocks = std::min(256 * per_cu, (ntiles + 7) / 8 * 8);
    blocks = std::max(8, blocks / 8 * 8);
    const int groups = std::max(1, std::min(b.n_frames, (256 * per_cu) / blocks));  // few tiles: split the batch too
    const dim3 grid(blocks, groups);
    if (pre == 2)
      hipLaunchKernelGGL(remap_tiled_kernel<2>, grid, dim3(kRemapTileThreads), lds, stream, q);
    else
      hipLaunchKernelGGL(remap_tiled_kernel<0>, grid, dim3(kRemapTileThreads), lds, stream, q);
  }
  if (q.n_border > 0)
    hipLaunchKernelGGL(remap_border_kernel, dim3((q.n_border + 255) / 256, b.n_frames), dim3(256), 0, stream, q);
  return true;
}

bool launch_remap(const RemapParams& p, hipStream_t stream) {
  if (p.n_frames <= 0) return true;
  // remap_pixel addresses a source frame with 32-bit offsets and 24-bit multiplies
  if (p.src_step >= (1u << 24) || p.rows >= (1 << 23) || (unsigned long long)p.src_step * (unsigned long long)p.rows >= (1ull << 32) ||
      p.dst_step >= (1u << 24) || (unsigned long long)p.dst_step * (unsigned long long)p.drows >= (1ull << 32))
    return false;  // the API layer turns this into RIP_ERR_INVALID_ARGUMENT (it rejects such pitches up front)
  const bool vec = p.channels == 3 && p.dcols % 4 == 0 && p.dst_step % 4 == 0 && p.dst_frame_stride % 4 == 0 &&
                   aligned4(p.dst) && (reinterpret_cast<uintptr_t>(p.map_xy) & 15u) == 0;
  if (vec) {
    ItemMap im{p.dcols / 4, 1.0f / (float)(p.dcols / 4)};
    const int items = p.drows * (p.dcols / 4);
    int per_frame = grid_blocks_for(items, std::max(8, 8192 / std::max(1, std::min(p.n_frames, 16))));
    hipLaunchKernelGGL(remap_vec4_kernel, dim3(per_frame, p.n_frames), dim3(kBlock), 0, stream, p, im, items);
    return true;
  }
  long long npix = (long long)p.drows * p.dcols;
  dim3 grid(grid_blocks_for(npix, 4096), p.n_frames);
  if (p.channels == 3)
    hipLaunchKernelGGL(remap_generic_kernel<3>, grid, dim3(kBlock), 0, stream, p);
  else
    hipLaunchKernelGGL(remap_generic_kernel<1>, grid, dim3(kBlock), 0, stream, p);
  return true;
}

}  // namespace rip
