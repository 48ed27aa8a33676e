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
// The same sum left unshifted, for row weights the caller has multiplied by 64 (wy0 = 64 (32 - fy), wy1 = 64 fy: still 24-bit
// operands, the sum stays below 2^24): the result byte (sum >> 10 of the unscaled weights) then IS byte 2 of the accumulator,
// and four of them are merged into an output dword with two v_perm_b32 and one v_or_b32 (pack_byte2) instead of a shift, a
// mask and a shift-or per byte.
__device__ __forceinline__ uint32_t blend_rows_b2(unsigned top, unsigned wy0_64, unsigned bot, unsigned wy1_64) {
  unsigned acc;
  asm("s_nop 2\n\tv_mul_u32_u24 %0, %1, %2\n\tv_mad_u32_u24 %0, %3, %4, %0" : "=&v"(acc) : "v"(top), "v"(wy0_64), "v"(bot), "v"(wy1_64));
  return acc;
}
// byte 2 of a, b, c, d -> one dword (a in byte 0)
__device__ __forceinline__ uint32_t pack_byte2(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  return __builtin_amdgcn_perm(b, a, 0x0c0c0602u) | __builtin_amdgcn_perm(d, c, 0x06020c0cu);
}
// Four BGR pixels from their tap rows (t0 / t1: bytes b0 g0 r0 b1 | g1 r1 . . of the top row, b0 / b1 of the bottom row), the x
// weights wxb = wx0 | wx1 << 24 and the y weights wyy = wy0 | wy1 << 16 (wx0 + wx1 = wy0 + wy1 = 32): cv::remap's
// ((top (32 - fy) + bot fy) 32 + 2^14) >> 15 per channel, exact in integers; 12 interleaved output bytes.
// Round 5: six quarter-rate instructions per tap row instead of seven, and the output bytes merged by v_perm_b32.  The two
// taps of B are bytes 0 and 3 of the realigned dword (one v_dot4, weights wx0 . . wx1); G and R used to take two v_dot4 each
// (their second tap lies in the next dword) -- one v_perm_b32 with a constant selector gathers g0 r0 g1 r1 into ONE dword, and
// G and R are a v_dot4 each on it with the weights (wx0 . wx1 .) and (. wx0 . wx1): the same integer sums.  Every row sum
// starts at 16: 16 (32 - fy) + 16 fy = 512 is the rounding term.
__device__ __forceinline__ Pack3 blend4_bgr(const uint32_t (&t0)[4], const uint32_t (&t1)[4], const uint32_t (&b0)[4], const uint32_t (&b1)[4],
                                            const unsigned (&wxb)[4], const unsigned (&wyy)[4]) {
  uint32_t q[4][3];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned wy0 = (wyy[k] & 0xffffu) << 6, wy1 = (wyy[k] >> 16) << 6;        // x 64: blend_rows_b2 (frame-invariant)
    const unsigned wB = wxb[k];                                                    // wx0 . . wx1
    const unsigned wG = (wB & 0xffu) | ((wB >> 8) & 0xff0000u), wR = wG << 8;      // wx0 . wx1 .  /  . wx0 . wx1 (frame-invariant)
    const uint32_t mt = __builtin_amdgcn_perm(t1[k], t0[k], 0x05040201u), mb = __builtin_amdgcn_perm(b1[k], b0[k], 0x05040201u);
    const unsigned topB = __builtin_amdgcn_udot4(t0[k], wB, 16u, false);
    const unsigned topG = __builtin_amdgcn_udot4(mt, wG, 16u, false);
    const unsigned topR = __builtin_amdgcn_udot4(mt, wR, 16u, false);
    const unsigned botB = __builtin_amdgcn_udot4(b0[k], wB, 16u, false);
    const unsigned botG = __builtin_amdgcn_udot4(mb, wG, 16u, false);
    const unsigned botR = __builtin_amdgcn_udot4(mb, wR, 16u, false);
    q[k][0] = blend_rows_b2(topB, wy0, botB, wy1);
    q[k][1] = blend_rows_b2(topG, wy0, botG, wy1);
    q[k][2] = blend_rows_b2(topR, wy0, botR, wy1);
  }
  return Pack3{pack_byte2(q[0][0], q[0][1], q[0][2], q[1][0]), pack_byte2(q[1][1], q[1][2], q[2][0], q[2][1]),
               pack_byte2(q[2][2], q[3][0], q[3][1], q[3][2])};
}
// The same for taps read from an image of FOUR bytes per pixel (b g r x: the colour image the chain-inside-remap kernel keeps in
// LDS, rip_fused.hip): a tap row is two whole dwords (d0 = left pixel, d1 = right pixel) -- one aligned 8-byte LDS read, no
// realignment -- and two v_perm_b32 transpose them into b0 b1 g0 g1 and r0 r1 . ., so that every channel is one v_dot4 with the
// weights w01 = wx0 | wx1 << 8 (B, R) or w01 << 16 (G): five quarter-rate instructions per tap row.  The x bytes never meet a
// non-zero weight.
__device__ __forceinline__ Pack3 blend4_bgrx(const uint32_t (&t0)[4], const uint32_t (&t1)[4], const uint32_t (&b0)[4], const uint32_t (&b1)[4],
                                             const unsigned (&w01)[4], const unsigned (&wyy)[4]) {
  uint32_t q[4][3];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned wy0 = (wyy[k] & 0xffffu) << 6, wy1 = (wyy[k] >> 16) << 6;  // x 64: blend_rows_b2 (frame-invariant)
    const unsigned wL = w01[k], wH = w01[k] << 16;                            // wx0 wx1 . .  /  . . wx0 wx1 (frame-invariant)
    const uint32_t tbg = __builtin_amdgcn_perm(t1[k], t0[k], 0x05010400u), tr = __builtin_amdgcn_perm(t1[k], t0[k], 0x0c0c0602u);
    const uint32_t bbg = __builtin_amdgcn_perm(b1[k], b0[k], 0x05010400u), br = __builtin_amdgcn_perm(b1[k], b0[k], 0x0c0c0602u);
    const unsigned topB = __builtin_amdgcn_udot4(tbg, wL, 16u, false);
    const unsigned topG = __builtin_amdgcn_udot4(tbg, wH, 16u, false);
    const unsigned topR = __builtin_amdgcn_udot4(tr, wL, 16u, false);
    const unsigned botB = __builtin_amdgcn_udot4(bbg, wL, 16u, false);
    const unsigned botG = __builtin_amdgcn_udot4(bbg, wH, 16u, false);
    const unsigned botR = __builtin_amdgcn_udot4(br, wL, 16u, false);
    q[k][0] = blend_rows_b2(topB, wy0, botB, wy1);
    q[k][1] = blend_rows_b2(topG, wy0, botG, wy1);
    q[k][2] = blend_rows_b2(topR, wy0, botR, wy1);
  }
  return Pack3{pack_byte2(q[0][0], q[0][1], q[0][2], q[1][0]), pack_byte2(q[1][1], q[1][2], q[2][0], q[2][1]),
               pack_byte2(q[2][2], q[3][0], q[3][1], q[3][2])};
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
// How the persistent workgroups of the ring kernels find their tiles.  Workgroup b runs on XCD b % 8 (observed dispatch order;
// speed only) and takes the positions ti = b / 8, b / 8 + gridDim.x / 8, ... of that XCD's share.  Rounds 1-5 gave every XCD one
// contiguous range of tiles (eight bands of the image in flight at once, each marching down on its own).  Round 6: runs of
// `run` consecutive tiles (about one tile row) are dealt round-robin, so the tiles in flight on the whole chip form ONE band of
// consecutive tile rows, read and written by all XCDs together -- the same requests, served 5 % faster at 4 frames per visit
// and 12 % faster with 8 (which the contiguous deal could not use: its XCDs drift apart).  EXPERIMENTS.md "the deal".
struct TileDeal {
  int run, per_xcd, ntiles;
  __device__ __forceinline__ TileDeal(int ntiles_, int run_) : run(run_), ntiles(ntiles_) {
    per_xcd = run > 0 ? ((ntiles + run - 1) / run + 7) / 8 * run : (ntiles + 7) / 8;
  }
  // tile at position ti of XCD xcd; -1: none here (a ragged last run), keep going; -2: past the end of a contiguous range
  __device__ __forceinline__ int tile(int ti, int xcd) const {
    if (run > 0) {
      const int r = ti / run;
      const int t = (r * 8 + xcd) * run + (ti - r * run);
      return t < ntiles ? t : -1;
    }
    const int t = xcd * per_xcd + ti;
    return t < ntiles ? t : -2;
  }
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

}  // namespace
}  // namespace rip
