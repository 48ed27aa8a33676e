// rip_remap_dev.hpp -- device helpers shared by the remap kernels (rip_remap.hip) and the kernel that demosaics the source
// rectangle on the fly (rip_fused.hip): cv::remap's fixed-point tap arithmetic, LDS tap reads, the LDS-DMA load and the
// counted waits of the ring.
#pragma once
#include "rip_device.hpp"

namespace rip {
namespace {

// ------------------------------------------------------------------------------------------------
// remap: cv::remap(INTER_LINEAR, BORDER_CONSTANT 0), undistortion.cpp:240-245
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int round_map(float v) {
  float s = v * 32.f;
  if (!(s > -2147483648.f && s < 2147483648.f)) return INT_MIN;  // cvRound of NaN/inf/out of range
  return (int)__builtin_rintf(s);
}

// Six consecutive source bytes starting at byte offset `off` of the frame (any alignment), fetched
// as three aligned dwords (one global_load_dwordx3) and realigned with v_alignbyte_b32.
// Requires off + 12 <= readable bytes (checked by the caller).
__device__ __forceinline__ void load6(const uint8_t* frame, unsigned off, uint32_t& lo, uint32_t& hi) {
  const uint3 v = *reinterpret_cast<const uint3*>(frame + (off & ~3u));
  lo = __builtin_amdgcn_alignbyte(v.y, v.x, off & 3u);
  hi = __builtin_amdgcn_alignbyte(v.z, v.y, off & 3u);
}

struct RemapSrc {
  const uint8_t* frame;
  unsigned step;      // bytes per row (< 2^24)
  unsigned readable;  // bytes that may be read starting at `frame` (to the end of the batch buffer)
  int rows, cols;
  bool wide_ok;       // frame base is dword aligned: load6 may be used
};

template <int CN>
__device__ __forceinline__ void remap_pixel(const RemapSrc& s, float mx, float my, int (&out)[CN]) {
  const int sxq = round_map(mx), syq = round_map(my);
  const int sx = clampi(sxq >> 5, -32768, 32767), sy = clampi(syq >> 5, -32768, 32767);
  const int fx = sxq & 31, fy = syq & 31;
  // cv::remap's Q15 bilinear weights 32(32-fx)(32-fy)...; separable form, exact in integers:
  // ((top*(32-fy) + bot*fy) * 32 + 2^14) >> 15 == (top*(32-fy) + bot*fy + 512) >> 10
  const int wx1 = fx, wx0 = 32 - fx, wy1 = fy, wy0 = 32 - fy;
  if ((unsigned)sx < (unsigned)(s.cols - 1) && (unsigned)sy < (unsigned)(s.rows - 1)) {
    const unsigned off0 = __umul24((unsigned)sy, s.step) + (unsigned)sx * CN;
    const unsigned off1 = off0 + s.step;
    int p0[2 * CN], p1[2 * CN];
    if (CN == 3 && s.wide_ok && off1 + 12u <= s.readable) {
      uint32_t l0, h0, l1, h1;
      load6(s.frame, off0, l0, h0);
      load6(s.frame, off1, l1, h1);
      p0[0] = l0 & 0xff; p0[1] = (l0 >> 8) & 0xff; p0[2] = (l0 >> 16) & 0xff; p0[3] = l0 >> 24; p0[4] = h0 & 0xff; p0[5] = (h0 >> 8) & 0xff;
      p1[0] = l1 & 0xff; p1[1] = (l1 >> 8) & 0xff; p1[2] = (l1 >> 16) & 0xff; p1[3] = l1 >> 24; p1[4] = h1 & 0xff; p1[5] = (h1 >> 8) & 0xff;
    } else {
#pragma unroll
      for (int k = 0; k < 2 * CN; k++) {
        p0[k] = s.frame[off0 + k];
        p1[k] = s.frame[off1 + k];
      }
    }
#pragma unroll
    for (int c = 0; c < CN; c++) {
      const int top = mul24(p0[c], wx0) + mul24(p0[CN + c], wx1);
      const int bot = mul24(p1[c], wx0) + mul24(p1[CN + c], wx1);
      out[c] = (mul24(top, wy0) + mul24(bot, wy1) + 512) >> 10;  // <= 255: convex combination
    }
    return;
  }
  if (sx >= s.cols || sx + 1 < 0 || sy >= s.rows || sy + 1 < 0) {
#pragma unroll
    for (int c = 0; c < CN; c++) out[c] = 0;
    return;
  }
  // partially outside: taps beyond the image contribute the border constant 0
  const bool x0 = sx >= 0 && sx < s.cols, x1 = sx + 1 >= 0 && sx + 1 < s.cols;
  const bool y0 = sy >= 0 && sy < s.rows, y1 = sy + 1 >= 0 && sy + 1 < s.rows;
#pragma unroll
  for (int c = 0; c < CN; c++) {
    const int p00 = (x0 && y0) ? s.frame[(size_t)sy * s.step + (size_t)sx * CN + c] : 0;
    const int p01 = (x1 && y0) ? s.frame[(size_t)sy * s.step + (size_t)(sx + 1) * CN + c] : 0;
    const int p10 = (x0 && y1) ? s.frame[(size_t)(sy + 1) * s.step + (size_t)sx * CN + c] : 0;
    const int p11 = (x1 && y1) ? s.frame[(size_t)(sy + 1) * s.step + (size_t)(sx + 1) * CN + c] : 0;
    out[c] = (mul24(mul24(p00, wx0) + mul24(p01, wx1), wy0) + mul24(mul24(p10, wx0) + mul24(p11, wx1), wy1) + 512) >> 10;
  }
}

__device__ __forceinline__ RemapSrc remap_src(const RemapParams& p, int frame) {
  RemapSrc s;
  s.frame = p.src + (size_t)frame * p.src_frame_stride;
  s.step = (unsigned)p.src_step;
  const unsigned long long rest = (unsigned long long)(p.n_frames - frame) * p.src_frame_stride;
  s.readable = rest > 0xffffffffull ? 0xffffffffu : (unsigned)rest;
  s.rows = p.rows;
  s.cols = p.cols;
  s.wide_ok = (reinterpret_cast<uintptr_t>(s.frame) & 3u) == 0;
  return s;
}

// (top * wy0 + bot * wy1) >> 10 with both products on the 24-bit multiplier: one v_mul_u32_u24 + one v_mad_u32_u24.
// Written out because hipcc proves the operands small, turns the builtins into plain multiplies and then selects the
// quarter-rate v_mul_lo_u32 for a quarter of them (6 of the 24 per lane and frame).
// `top` / `bot` come straight out of v_dot4_u32_u8.  On gfx90a and later a dot instruction's result needs THREE wait states
// before a different VALU instruction may read it (LLVM GCNHazardRecognizer: DotWriteDifferentVALURead); the compiler pads
// that for the instructions it emits, but it does not look inside inline assembly -- hence the s_nop 2 in front.  (Found in
// round 4: the one-channel gather, where nothing else sat between the dot and the multiply, read stale registers; the
// colour gathers had exactly three instructions in between by luck of the schedule.)
__device__ __forceinline__ int blend_rows(unsigned top, unsigned wy0, unsigned bot, unsigned wy1) {
  unsigned acc;
  asm("s_nop 2\n\tv_mul_u32_u24 %0, %1, %2\n\tv_mad_u32_u24 %0, %3, %4, %0" : "=&v"(acc) : "v"(top), "v"(wy0), "v"(bot), "v"(wy1));
  return (int)(acc >> 10);
}
__device__ __forceinline__ void lds_load6(const uint8_t* lds, unsigned a, uint32_t& lo, uint32_t& hi) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(lds + (a & ~3u));
  const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
  lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
  hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
}

// four bytes starting at LDS byte address a (any alignment): two aligned dwords realigned
__device__ __forceinline__ uint32_t lds_load4(const uint8_t* lds, unsigned a) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(lds + (a & ~3u));
  return __builtin_amdgcn_alignbyte(w[1], w[0], a & 3u);
}

// ------------------------------------------------------------------------------------------------
// Ring version of the tiled remap: the source rectangle of frame f+D is copied global -> LDS by the
// LDS-DMA path (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass) while frame f is
// gathered, D = stages - 1 frames ahead, so the HBM/TLB latency of the 15 MB frame-to-frame stride is
// covered by D gathers instead of one.  hipcc drains vmcnt to 0 at every barrier when it knows about
// an LDS-DMA in flight, so the loads are inline asm and counted here: every wave issues exactly PRE
// loads per frame (lanes past the rectangle load from an out-of-range offset, which the buffer
// resource turns into zeros), loads of one wave land in order, and "loads of frame f have landed" is
// s_waitcnt vmcnt((frames issued after f) * PRE) -- stores in flight only make that wait longer.
// One barrier per frame: after it every wave's part of frame f is in LDS and every wave is done
// with frame f-1, whose stage is the one refilled next.
// The staged image is chunk-linear (chunk i of the rectangle at byte 16*i: the row pitch is a whole
// number of chunks), which is exactly the order LDS-DMA writes (M0 base + lane * 16).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voffset), "s"(rsrc), "s"(lds_wave_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

}  // namespace
}  // namespace rip
