// rip_device.hpp -- device-side building blocks shared by the kernel translation units
// (rip_chain.hip, rip_stats.hip, rip_ccc.hip, rip_remap.hip): source views, LDS / global table
// accessors, the per-pixel stages, the Bayer window and its SWAR demosaic, packing and buffer-resource
// addressing helpers, and the small host-side launch helpers.  Everything lives in an anonymous
// namespace: each translation unit gets its own copy and hipcc inlines it.
//
// Stage semantics follow the reference's CPU/OpenCV path (file:line relative to the reference tree):
//   debayer   raw_image_pipeline/src/raw_image_pipeline/modules/debayer.cpp:45-79
//   flip      .../modules/flip.cpp:37-58
//   wb        .../modules/white_balance.cpp:59-64 (grey world), :73-136 (pca), :52-57 (simple),
//             raw_image_pipeline_white_balance/src/.../convolutional_color_constancy.cpp:91-113 (ccc)
//   colour    .../modules/color_calibration.cpp:91-104
//   gamma     .../modules/gamma_correction.cpp:35-60
//   vignette  .../modules/vignetting_correction.cpp:32-93
//   hsv       .../modules/color_enhancer.cpp:38-47
//   remap     .../modules/undistortion.cpp:240-245
// Layout: everything is uint8 interleaved BGR in HBM.  Compile with -ffp-contract=off: the float
// stages reproduce OpenCV's separate mul/add.
#pragma once
#include "rip_kernels.hpp"

#include <algorithm>
#include <climits>
#include <cstddef>
#include <cstdlib>

// Floating-point contraction model of the float stages (PARITY.md): 0 (default) = every product and every sum rounded --
// OpenCV built for baseline x86-64; 1 = the fused forms GCC / Clang emit wherever the target has FMA (every aarch64 build: the
// reference's Jetson deployment), oracle/rip_oracle.c mode 1.  The translation units that hold these stages (rip_chain.hip,
// rip_fused.hip) are compiled once per model (build.py); rip_set_fp_contraction() picks the set of kernels at launch time.
#ifndef RIP_FP_CONTRACT
#define RIP_FP_CONTRACT 0
#endif

namespace rip {
namespace {

constexpr int kBlock = 256;
constexpr bool kFpContract = RIP_FP_CONTRACT != 0;
// a * b + c: one fma under the contracted model, multiply then add otherwise (the build runs with -ffp-contract=off, so the
// second form stays two instructions)
__device__ __forceinline__ float mul_add(float a, float b, float c) {
  if constexpr (kFpContract) return __builtin_fmaf(a, b, c);
  return a * b + c;
}

// ------------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sat_round_u8(float v) {
  // saturate_cast<uchar>(float) = round half to even, clamp to [0,255], NaN -> 0: exactly
  // v_cvt_pk_u8_f32 (checked on gfx950 by tools/probes/cvt_pk_u8_probe.hip)
  return (int)__builtin_amdgcn_cvt_pk_u8_f32(v, 0, 0u);
}
// 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate).  Operands must fit 24 bits.
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
// a*b + c as ONE v_mad_i32_i24 (hipcc otherwise splits multiply-add chains into mul, mul, mad, add3)
__device__ __forceinline__ int mad24(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// the same with the multiplier in a scalar register (a compile-time constant the loop keeps there)
__device__ __forceinline__ int mad24_ks(int a, int k, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
  return d;
}
// ... and with the addend there
__device__ __forceinline__ int mad24_cs(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}
__device__ __forceinline__ unsigned umulhi24(unsigned a, unsigned b) {
  return (unsigned)(((unsigned long long)(a & 0xffffffu) * (unsigned long long)(b & 0xffffffu)) >> 32);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
// Put inside a wave-uniform `if` body: the empty volatile asm cannot be speculated, so hipcc keeps the
// scalar branch instead of computing both sides and selecting per lane with v_cndmask.
__device__ __forceinline__ void keep_branch() { asm volatile(""); }

struct SrcView {
  const uint8_t* base;
  size_t step;
  int rows, cols, kind, ry, rx;
};

// Bilinear demosaic at one position with OpenCV's border rule (the interior formula evaluated
// at the position clamped to [1, n-2]).
__device__ __forceinline__ void debayer_at(const SrcView& s, int y, int x, int& b, int& g, int& r) {
  int yc = clampi(y, 1, s.rows - 2), xc = clampi(x, 1, s.cols - 2);
  const uint8_t* p = s.base + (size_t)yc * s.step + xc;
  const ptrdiff_t st = (ptrdiff_t)s.step;
  int dy = (yc - s.ry) & 1, dx = (xc - s.rx) & 1;  // (0,0): R site, (1,1): B site
  int c = p[0];
  if (dy != dx) {
    int h = (p[-1] + p[1] + 1) >> 1;
    int v = (p[-st] + p[st] + 1) >> 1;
    g = c;
    if (dy == 0) {
      r = h;
      b = v;
    } else {
      b = h;
      r = v;
    }
  } else {
    int x4 = (p[-1] + p[1] + p[-st] + p[st] + 2) >> 2;
    int d4 = (p[-st - 1] + p[-st + 1] + p[st - 1] + p[st + 1] + 2) >> 2;
    g = x4;
    if (dy == 0) {
      r = c;
      b = d4;
    } else {
      b = c;
      r = d4;
    }
  }
}

// Colour of the (pre-flip) source image at (y,x) for any supported input kind.
__device__ __forceinline__ void fetch_src(const SrcView& s, int y, int x, int& b, int& g, int& r) {
  if (s.kind == SRC_BAYER) {
    debayer_at(s, y, x, b, g, r);
  } else if (s.kind == SRC_MONO) {
    b = g = r = s.base[(size_t)y * s.step + x];
  } else {
    const uint8_t* p = s.base + (size_t)y * s.step + (size_t)x * 3;
    int c0 = p[0], c1 = p[1], c2 = p[2];
    g = c1;
    if (s.kind == SRC_RGB) {  // cvtColor(RGB2BGR), debayer.cpp:72-73
      b = c2;
      r = c0;
    } else {
      b = c0;
      r = c2;
    }
  }
}

// destination (post-flip) -> source coordinates, flip.cpp:37-58
__device__ __forceinline__ void unflip(int angle, int rows, int cols, int yd, int xd, int& ys, int& xs) {
  if (angle == 180) {
    ys = rows - 1 - yd;
    xs = cols - 1 - xd;
  } else if (angle == 90) {
    ys = rows - 1 - xd;
    xs = yd;
  } else if (angle == 270) {
    ys = xd;
    xs = cols - 1 - yd;
  } else {
    ys = yd;
    xs = xd;
  }
}

// ------------------------------------------------------------------------------------------------
// table accessors: global (generic kernel) or LDS (fast kernel)
// ------------------------------------------------------------------------------------------------
struct GlobalTabs {
  const DevTables* t;
  __device__ __forceinline__ int gamma(int i) const { return t->gamma_lut[i]; }

  __device__ __forceinline__ unsigned yf(int i) const { return t->yf_tab[i]; }
  __device__ __forceinline__ int invg(int i) const { return t->inv_gamma[i]; }
  __device__ __forceinline__ int sdiv(int i) const { return t->sdiv[i]; }
  __device__ __forceinline__ int hdiv(int i) const { return t->hdiv[i]; }
  __device__ __forceinline__ float linf(int i) const { return (float)t->lin_tab[i]; }
  __device__ __forceinline__ float cbrtf(unsigned i) const { return (float)t->cbrt_tab[i]; }
};

template <bool ON, typename T, int N>
struct LdsArr {
  T v[N];
};
template <typename T, int N>
struct LdsArr<false, T, N> {
  T v[1];
};

// ------------------------------------------------------------------------------------------------
// per-pixel stages
// ------------------------------------------------------------------------------------------------
// white_balance.cpp: grey-world applyChannelGains (Q8, truncating), ccc cv::multiply (float,
// round-half-even), pca quadratic on B and R.
__device__ __forceinline__ void apply_wb(int mode, const FrameWb& w, int& b, int& g, int& r) {
  if (mode == WB_Q8) {
    b = (b * w.q8[0]) >> 8;
    g = (g * w.q8[1]) >> 8;
    r = (r * w.q8[2]) >> 8;
  } else if (mode == WB_FLOAT) {
    b = sat_round_u8((float)b * w.fg[0]);
    g = sat_round_u8((float)g * w.fg[1]);
    r = sat_round_u8((float)r * w.fg[2]);
  } else if (mode == WB_SIMPLE) {
    // SimpleWB's stretch: convertTo(8U, alpha, beta) = saturate(float(x) * alpha + beta), no FMA
    b = sat_round_u8((float)b * w.fg[0] + w.pca[0]);
    g = sat_round_u8((float)g * w.fg[1] + w.pca[1]);
    r = sat_round_u8((float)r * w.fg[2] + w.pca[2]);
  } else if (mode == WB_PCA) {
    float fb = (float)b, fr = (float)r;
    float b2 = fb * fb, r2 = fr * fr;
    // cv::addWeighted = v_fma(A, a, v_fma(B, b, 0)): under the contracted model only the second product is rounded on its own
    float bp = mul_add(b2, w.pca[0], fb * w.pca[1]);
    float rp = mul_add(r2, w.pca[2], fr * w.pca[3]);
    bp = bp > 255.f ? 255.f : bp;  // THRESH_TRUNC
    rp = rp > 255.f ? 255.f : rp;
    b = sat_round_u8(bp);
    r = sat_round_u8(rp);
  }
}

// The colour matrix held in VGPRs: as SGPR operands (kernel arguments) every one of the nine multiplies would issue
// at 4.3 cycles instead of 2.45.  The empty asm keeps hipcc from re-materialising the values from SGPRs.
struct CcRegs {
  float m[9];
  __device__ __forceinline__ void load(const ChainParams& p) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      m[i] = p.cc_m[i];
      asm volatile("" : "+v"(m[i]));
    }
  }
};
struct HsvRegs {
  float g[3];     // gains on H, S, V (VGPR copies, same reason)
  unsigned unit;  // bit c: gain c is exactly 1 (wave-uniform)
  __device__ __forceinline__ void load(const ChainParams& p) {
    unit = (p.hsv_gain[0] == 1.f ? 1u : 0u) | (p.hsv_gain[1] == 1.f ? 2u : 0u) | (p.hsv_gain[2] == 1.f ? 4u : 0u);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      g[i] = p.hsv_gain[i];
      asm volatile("" : "+v"(g[i]));
    }
  }
};
// color_calibration.cpp:93-103: ((m0*B + m1*G) + m2*R) + bias in float32, no FMA
// BIAS: 1 = test the (wave-uniform) bias per call; 0 = the caller knows it is all zero (t + 0.0f == t up to the sign of
// zero, which the saturating conversion drops) -- the fast kernel tests once per launch and runs a body without the branch
template <int BIAS = 1>
__device__ __forceinline__ void apply_cc_f(const ChainParams& p, const CcRegs& cc, int b, int g, int r, float (&o)[3]) {
  float fb = (float)b, fg = (float)g, fr = (float)r;
  // contracted model: (a0*b0 + a1*b1) + a2*b2 -> fma(a2, b2, fma(a0, b0, a1*b1)) (leftmost multiply fused first); the bias is a
  // separate cv::add over the whole Mat and never fuses
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = mul_add(fr, cc.m[c * 3 + 2], mul_add(fb, cc.m[c * 3], fg * cc.m[c * 3 + 1]));
  if constexpr (BIAS != 0) {
    if (p.cc_bias[0] != 0.f || p.cc_bias[1] != 0.f || p.cc_bias[2] != 0.f) {
      keep_branch();
#pragma unroll
      for (int c = 0; c < 3; c++) o[c] = o[c] + p.cc_bias[c];
    }
  }
}
template <int BIAS = 1>
__device__ __forceinline__ void apply_cc(const ChainParams& p, const CcRegs& cc, int& b, int& g, int& r) {
  float o[3];
  apply_cc_f<BIAS>(p, cc, b, g, r, o);
  b = sat_round_u8(o[0]);
  g = sat_round_u8(o[1]);
  r = sat_round_u8(o[2]);
}
__device__ __forceinline__ void apply_cc(const ChainParams& p, int& b, int& g, int& r) {
  float fb = (float)b, fg = (float)g, fr = (float)r;
  float o[3];
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = mul_add(fr, p.cc_m[c * 3 + 2], mul_add(fb, p.cc_m[c * 3], fg * p.cc_m[c * 3 + 1]));
  // t + 0.0f == t (up to the sign of zero, which the saturating conversion drops): the usual all-zero
  // bias costs nothing; the test is wave-uniform (kernel arguments)
  if (p.cc_bias[0] != 0.f || p.cc_bias[1] != 0.f || p.cc_bias[2] != 0.f) {
    keep_branch();
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = o[c] + p.cc_bias[c];
  }
  b = sat_round_u8(o[0]);
  g = sat_round_u8(o[1]);
  r = sat_round_u8(o[2]);
}

// abToXZ_b[i - minABvalue] (OpenCV color_lab.cpp initLabTabs), evaluated arithmetically.
// i > 3390: the cube i*i/BASE*i/BASE;  i <= 3390 (L* below ~8, dark pixels only): the linear segment
// i*108/841 - 290 with C truncation.
__device__ __forceinline__ int ab_to_xz_cube(int i) {
  return mul24(mul24(i, i) >> 14, i) >> 14;  // i in (3390, 28719]: both products < 2^31
}
__device__ __forceinline__ int ab_to_xz_linear(int i) {
  // n = i*108 is in [-879660, 366120]; trunc(n/841) = floor((n + (n<0 ? 840 : 0)) / 841); the floor
  // division is one v_mul_hi_u32_u24 by ceil(2^32/841) after biasing by 841*1100 (exact below 11.9e6);
  // BASE*16/116*108/841 == 290
  int n = mul24(i, 108);
  n += (n >> 31) & 840;
  const unsigned q = umulhi24((unsigned)(n + 841 * 1100), 5106977u);
  return (int)q - (1100 + 290);
}
// Both lookups of one pixel.  The linear segment is rare, so it sits behind a wave-uniform branch:
// hipcc otherwise predicates both sides and issues all of it for every pixel.
__device__ __forceinline__ void ab_to_xz_pair(int ix, int iz, int& x, int& z) {
  x = ab_to_xz_cube(ix);
  z = ab_to_xz_cube(iz);
  const bool dark = ix <= 3390 || iz <= 3390;
  if (__builtin_amdgcn_ballot_w64(dark) != 0ull) {
    if (ix <= 3390) x = ab_to_xz_linear(ix);
    if (iz <= 3390) z = ab_to_xz_linear(iz);
  }
}

// BGR -> 8-bit Lab -> L * mask -> BGR (vignetting_correction.cpp:68-93; RGB2Lab_b /
// Lab2RGBinteger).  The table `linf` already folds the gamma LUT when the gamma stage is on.
//
// The forward half runs on v_fma_f32 (2 cycles per wave64 on gfx950, against 4 for the 24-bit
// integer multiply-add): every operand is a small integer held exactly in fp32 and every sum stays
// below 2^23, so the products and sums are exact.  floor(T / 2^n) of an integer T is taken as
// RN((T + 0.5) / 2^n - 0.5) -- never a tie -- by adding the magic constant 1.5 * 2^23 inside the
// FMA, which leaves the integer in the low mantissa bits.
typedef short i16x2 __attribute__((ext_vector_type(2)));
template <typename Tabs>
__device__ __forceinline__ void apply_vignette(const ChainParams& p, const Tabs& tb, const float* fwd, const int* inv,
                                               float mask, int& b, int& g, int& r) {
  constexpr float kMagic = 12582912.0f;        // 1.5 * 2^23: ulp 1 in [2^23, 2^24)
  constexpr unsigned kMagicBits = 0x4B400000u;  // bit pattern of kMagic
  const float v0 = tb.linf(b), v1 = tb.linf(g), v2 = tb.linf(r);
  // (C . v + 2048) >> 12 == RN((C . v + 0.5) / 4096): sums <= 2040 * 4096 + 0.5 < 2^23
  const float sx = __builtin_fmaf(v2, fwd[2], __builtin_fmaf(v1, fwd[1], __builtin_fmaf(v0, fwd[0], 0.5f)));
  const float sy = __builtin_fmaf(v2, fwd[5], __builtin_fmaf(v1, fwd[4], __builtin_fmaf(v0, fwd[3], 0.5f)));
  const float sz = __builtin_fmaf(v2, fwd[8], __builtin_fmaf(v1, fwd[7], __builtin_fmaf(v0, fwd[6], 0.5f)));
  const unsigned ix = __float_as_uint(__builtin_fmaf(sx, 1.0f / 4096.0f, kMagic)) - kMagicBits;
  const unsigned iy = __float_as_uint(__builtin_fmaf(sy, 1.0f / 4096.0f, kMagic)) - kMagicBits;
  const unsigned iz = __float_as_uint(__builtin_fmaf(sz, 1.0f / 4096.0f, kMagic)) - kMagicBits;
  const float fX = tb.cbrtf(ix), fY = tb.cbrtf(iy), fZ = tb.cbrtf(iz);
  // L = (296 fY - 1336935 + 16384) >> 15, in [0, 255] by construction
  const float tl = __builtin_fmaf(fY, 296.0f, -1336934.5f);  // T + 0.5 - 16384, T = 296 fY - 1320551 >= 0
  const float Lf = __builtin_fmaf(tl, 1.0f / 32768.0f, kMagic) - kMagic;
  const int L = sat_round_u8(Lf * mask);  // convertTo(32F), multiply, convertTo(8U)
  // a = clamp((500 (fX - fY) + 128 * 2^15 + 2^14) >> 15, 0, 255); outside [0, 2^23) the FMA may round,
  // but those values clamp to the same end of the range anyway
  const float ta = __builtin_fmaf(fX - fY, 500.0f, 4194304.5f);
  const float tb2 = __builtin_fmaf(fY - fZ, 200.0f, 4194304.5f);
  // OpenCV saturates a and b to [0, 255]; over all 2^24 inputs they stay inside [42, 226] and [20, 223]
  // (exhaustive check: tests/test_oracle_known_answers.py::test_lab_ab_never_saturate), so the clamp is dead.
  // abits = kMagicBits + a: the 24-bit multiply reads 0x400000 + a, the constant takes 0x400000 * K back
  // (mod 2^32), leaving a * K + rounding in one v_mad_u32_u24.
  const unsigned abits = __float_as_uint(__builtin_fmaf(ta, 1.0f / 32768.0f, kMagic));
  const unsigned bbits = __float_as_uint(__builtin_fmaf(tb2, 1.0f / 32768.0f, kMagic));
  constexpr unsigned kA = 5u * 53687u, kB = 41943u;
  const unsigned yf = tb.yf(L);
  const int y = (int)(yf & 0xffffu), ify = (int)(yf >> 16);
  const int adiv = (int)((__umul24(abits, kA) + ((1u << 7) - 0x400000u * kA)) >> 13) - 128 * 16384 / 500;
  const int bdiv = (int)((__umul24(bbits, kB) + ((1u << 4) - 0x400000u * kB)) >> 9) - 128 * 16384 / 200 + 1;
  int x, z;
  ab_to_xz_pair(ify + adiv, ify - bdiv, x, z);
  // x in [-652, 28028] (a in [42, 226], L <= 255) and y in [0, 16384] fit 16 bits, the coefficients too: the
  // x and y terms of a row are one v_dot2_i32_i16; z reaches 59.9k and stays on the 24-bit multiply-add.
  // No intermediate leaves 31 bits.  inv = DevTables::lab_inv_pk.
  const i16x2 xy = {(short)x, (short)y};
  const int bo = __builtin_amdgcn_sdot2(xy, __builtin_bit_cast(i16x2, inv[0]), mad24(inv[1], z, 1 << 13), false) >> 14;
  const int go = __builtin_amdgcn_sdot2(xy, __builtin_bit_cast(i16x2, inv[2]), mad24(inv[3], z, 1 << 13), false) >> 14;
  const int ro = __builtin_amdgcn_sdot2(xy, __builtin_bit_cast(i16x2, inv[4]), mad24(inv[5], z, 1 << 13), false) >> 14;
  b = tb.invg(clampi(bo, 0, 4095));
  g = tb.invg(clampi(go, 0, 4095));
  r = tb.invg(clampi(ro, 0, 4095));
}

// ------------------------------------------------------------------------------------------------
// Lab round trip of the fast kernel (same arithmetic as apply_vignette above, re-tabulated so that the
// per-pixel work shrinks to what cannot be tabulated):
//  * forward 3x3 as nine v_fma_f32 whose coefficients are 32-bit literals (kLabFwd: OpenCV's fixed sRGB/D65
//    matrix; the host refuses to start if its tables ever disagree): exact in fp32 (small integers scaled by
//    2^-12, every partial sum below 2^24 ulps).  Literal and VGPR sources issue at the full rate (2.45 cycles per
//    wave64), an SGPR source makes the same opcode cost 4.3 -- profiles/r02_valu_issue_rates.txt.  The same
//    product as three v_mfma_f32_4x4x1_16B_f32 per pixel (each lane's own scalar times the four coefficients its
//    4-lane block holds) was measured too: bit-identical, but 8.4 cycles per MFMA and no overlap with the VALU
//    issue of the other waves, i.e. slower than the FMAs it replaces (that variant, the packed-fp32 one and the extra-VALU probe are in the history at 84c88d7; they are not part of the library).
//  * LabCbrtTab_b re-tabulated as three float tables whose entries already carry the factors and
//    tie-breaking offsets of the L / a / b formulas: X -> 25 f + 1/8, Y -> {L, 25 f}, Z -> 25 f - 1/8 (or the X table
//    again and + 1/4 after the subtraction: 8 KB less LDS), so
//    a = RN((X - Y) * 5 / 8192) + 128 and b = RN((Y - Z) / 4096) + 128 exactly (the +-1/8 turns CV_DESCALE's
//    round-half-up into a never-tying round-to-nearest; every table index stays <= 2040 because each
//    forward row sums to 4096).
//  * LabToYF_b re-tabulated per L as {ify, cy_b * y + r, cy_g * y + r, cy_r * y + r}: the y terms of the
//    inverse matrix and its rounding constant become the accumulator of one v_dot2_i32_i16 per output
//    channel, whose two products are the x and z terms (z is biased by kZoff to fit 16 bits; the bias is
//    pre-multiplied into r).  Ranges (all 2^24 inputs x every L'): tests/test_oracle_known_answers.py.
// ------------------------------------------------------------------------------------------------
// RGB2Lab_b coefficients (B, G, R columns of the X, Y, Z rows) and Lab2RGBinteger coefficients (X, Y, Z columns of
// the B, G, R rows), lab_shift = 12: compile-time so that they reach the instructions as literals.
// rip_api.cpp checks them against the host-built tables (make_color_tables) when a handle is created.
constexpr int kLabFwd[9] = {778, 1541, 1777, 296, 2929, 871, 3575, 448, 73};
constexpr int kLabInv[9] = {217, -836, 4715, -3773, 7684, 185, 12615, -6296, -2223};
constexpr int kVigCbrtN = 2048;  // LabCbrtTab_b indices reachable from 8-bit input: 0 .. 2040
constexpr int kZoff = 27500;     // z in [-999, 59828] -> z - kZoff fits int16
constexpr int kVigGroup = 4;     // pixels of a row taken through the round trip together (ILP against live registers)
// The descaled sums of Lab2RGBinteger stay inside [-7253, 16753] for every (L', a, b) (tests/test_oracle_known_answers.py::
// test_ranges_the_fast_lab_kernel_relies_on), so sRGBInvGammaTab_b is held in LDS extended by its two saturated ends and
// read without a clamp (three v_med3_i32 per pixel less); the offset is a multiple of 4 so that the middle of the table is
// a dword copy, and it rides in the accumulators of the inverse matrix.
constexpr int kInvgOff = 7256;
constexpr int kInvgExtN = kInvgOff + 16753 + 3;  // 24012 bytes, a multiple of 4
// adiv / bdiv of Lab2RGBinteger folded into ONE multiply-add each on top of a per-L' table word (VigTabs::yf[].x):
//   fx = ify + ((5 a 53687 + 128) >> 13) - 4194            = (ify 2^13 + a kA + 128 - 4194 2^13) >> 13
//   fz = ify - (((b 41943 + 16) >> 9) - 10484)             = (ify 2^13 + 10484 2^13 + 8191 - 256 - b kB16) >> 13
// (kB16 = 16 * 41943; -floor(u / n) = floor((n - 1 - u) / n)).  The 24-bit multiplies read the low mantissa bits of the
// floats that hold a and b, i.e. 0x400000 + a + kBiasA and 0x400000 + b + kBiasB; the biases are chosen (tools/
// lab_fold_search.py) so that both formulas share the table word ify 2^13 + kFoldC, each within the slack its floor leaves
// (x: [-31, +25], z: [-95, +128]; taken: -2 and +18).  tests/test_oracle_known_answers.py checks every (L', a, b).
constexpr int kBiasA = -38465, kBiasB = -39212;
constexpr unsigned kFoldA = 5u * 53687u, kFoldB16 = 16u * 41943u;
constexpr unsigned kFoldC = 128u - 4194u * 8192u - 2u - kFoldA * (0x400000u + (unsigned)kBiasA);
struct VigTabs {
  float lin[256];
  float cbx[kVigCbrtN];
  float2 cby[kVigCbrtN];
  int4 yf[256];
  uint8_t invg[kInvgExtN];
  // the LDS address of a __shared__ object is the low half of its flat address
  __device__ __forceinline__ unsigned invg_lds_address() const { return (unsigned)reinterpret_cast<uintptr_t>(&invg[0]); }
  // A ready-made image of this struct (built once per table change by vig_image_kernel with invg_base = 0) copied with
  // 16-byte loads, all of them in flight at once; the per-L' records get this kernel's LDS address of the table added to
  // their three accumulators on the way.  Building the 54 KB from DevTables in every workgroup cost ~5 us of its ~90.
  template <int NT>
  __device__ __forceinline__ void load_image(const uint32_t* image) {
    static_assert(sizeof(VigTabs) % 16 == 0, "copied as uint4");
    constexpr int kVec = (int)(sizeof(VigTabs) / 16), kIter = (kVec + NT - 1) / NT;
    const unsigned add = invg_lds_address() << 14;
    const int yf0 = (int)(offsetof(VigTabs, yf) / 16);
    const uint4* src = reinterpret_cast<const uint4*>(image);
    uint4* dst = reinterpret_cast<uint4*>(this);
    uint4 v[kIter];
#pragma unroll
    for (int k = 0; k < kIter; k++) {
      const int i = (int)threadIdx.x + k * NT;
      if (i < kVec) v[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < kIter; k++) {
      const int i = (int)threadIdx.x + k * NT;
      if (i >= kVec) continue;
      if (i >= yf0 && i < yf0 + 256) {
        v[k].y += add;
        v[k].z += add;
        v[k].w += add;
      }
      dst[i] = v[k];
    }
  }
  template <int NT>
  __device__ __forceinline__ void load(const DevTables* t) { load<NT>(t, invg_lds_address()); }
  template <int NT>
  __device__ __forceinline__ void load(const DevTables* t, const unsigned invg_base) {
    for (int i = threadIdx.x; i < 256; i += NT) {
      lin[i] = (float)t->lin_tab[i];
      const uint32_t e = t->yf_tab[i];
      const int y = (int)(e & 0xffffu);
      int4 o;
      o.x = (int)(((e >> 16) << 13) + kFoldC);
      // the accumulators of the inverse matrix carry the rounding constant, the z bias and the absolute LDS address of
      // invg[kInvgOff], so that (sum >> 14) addresses the table entry directly
      // (sums modulo 2^32, read back as unsigned)
#define RIP_YACC(c) (int)((unsigned)(kLabInv[(c) * 3 + 1] * y + (1 << 13) + kZoff * kLabInv[(c) * 3 + 2]) + ((invg_base + kInvgOff) << 14))
      o.y = RIP_YACC(0);
      o.z = RIP_YACC(1);
      o.w = RIP_YACC(2);
#undef RIP_YACC
      yf[i] = o;
    }
    for (int i = threadIdx.x; i < kVigCbrtN; i += NT) {
      const int f = (int)t->cbrt_tab[i];
      const float f25 = (float)(25 * f);  // <= 25 * 32768: exact
      // L = CV_DESCALE(Lscale * fY + Lshift, lab_shift2), saturate_cast<uchar> (RGB2Lab_b)
      const int L = clampi((296 * f - 1336934 + (1 << 14)) >> 15, 0, 255);
      cbx[i] = f25 + 0.125f;
      cby[i] = make_float2((float)L, f25);
    }
    uint32_t* d = reinterpret_cast<uint32_t*>(invg);
    const uint32_t* s = reinterpret_cast<const uint32_t*>(t->inv_gamma);
    const uint32_t below = 0x01010101u * t->inv_gamma[0], above = 0x01010101u * t->inv_gamma[4095];
    for (int i = threadIdx.x; i < kInvgExtN / 4; i += NT) {
      const int j = i - kInvgOff / 4;
      d[i] = j < 0 ? below : j >= 1024 ? above : s[j];
    }
  }
};
// Four pixels of one row through BGR -> Lab -> L * mask -> the three lookups that end Lab -> BGR
// (vignetting_correction.cpp:68-93).  lin_off: byte offsets into VigTabs::lin (4 * value); out: absolute LDS byte addresses of the three VigTabs::invg entries.
// Every table address leaves the arithmetic already scaled: shifts to the left are quarter-rate-class instructions on
// gfx950 (4 cycles per wave64), right shifts, AND and the fp32 add are full rate (2).
template <int N>
__device__ __forceinline__ void vignette_n(const VigTabs& tb, const float* mask, const unsigned (*lin_off)[3], unsigned (*out)[3]) {
  constexpr float kMagic = 12582912.0f;  // 1.5 * 2^23: ulp 1 in [2^23, 2^24); its bit pattern 0x4B400000 has 22 low zero bits
  const char* lin_b = reinterpret_cast<const char*>(tb.lin);
  const char* cbx_b = reinterpret_cast<const char*>(tb.cbx);
  const char* cby_b = reinterpret_cast<const char*>(tb.cby);
  const char* yf_b = reinterpret_cast<const char*>(tb.yf);
  unsigned ax[N], ay[N], az[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    const float v0 = *reinterpret_cast<const float*>(lin_b + lin_off[k][0]), v1 = *reinterpret_cast<const float*>(lin_b + lin_off[k][1]),
                v2 = *reinterpret_cast<const float*>(lin_b + lin_off[k][2]);
    // index i_r = (C_r . v + 2048) >> 12 = floor(s_r + 1/2), s_r = C_r . v / 4096 <= 2040.  The X and Z rows accumulate
    // t = 4 s + 2 + (-1/2 + 2^-11) -- exact: multiples of 2^-11 below 2^13 -- and RN(t + magic) = floor(4 s + 2) (never a
    // tie), whose bits above the low two are 4 i: the byte offset into the float table.  The Y row the same with 8 (float2).
    float acc[3] = {1.5f + 1.0f / 2048.0f, 3.5f + 1.0f / 1024.0f, 1.5f + 1.0f / 2048.0f};
    constexpr float sc[3] = {1.0f / 1024.0f, 1.0f / 512.0f, 1.0f / 1024.0f};
#pragma unroll
    for (int r = 0; r < 3; r++)
      acc[r] = __builtin_fmaf(v2, (float)kLabFwd[r * 3 + 2] * sc[r],
                              __builtin_fmaf(v1, (float)kLabFwd[r * 3 + 1] * sc[r], __builtin_fmaf(v0, (float)kLabFwd[r * 3] * sc[r], acc[r])));
    ax[k] = __float_as_uint(acc[0] + kMagic) & 0x1FFCu;
    ay[k] = __float_as_uint(acc[1] + kMagic) & 0x3FF8u;
    az[k] = __float_as_uint(acc[2] + kMagic) & 0x1FFCu;
  }
  int fx[N], fz[N], x[N], z[N];
  int4 e[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    const float X = *reinterpret_cast<const float*>(cbx_b + ax[k]), Zt = *reinterpret_cast<const float*>(cbx_b + az[k]);
    const float2 LY = *reinterpret_cast<const float2*>(cby_b + ay[k]);
    // L' = saturate_cast<uchar>(L * mask) (convertTo(32F), multiply, convertTo(8U)), placed in byte 1 of the dword:
    // (L' << 8) >> 4 is the byte offset of its int4
    const unsigned L16 = __builtin_amdgcn_cvt_pk_u8_f32(LY.x * mask[k], 1, 0u) >> 4;
    // a, b never leave [0, 255] (exhaustive test), so saturate_cast is dead; the floats hold 0x400000 + a + kBiasA (b: kBiasB)
    // in their low 24 bits
    const unsigned abits = __float_as_uint(__builtin_fmaf(X - LY.y, 5.0f / 8192.0f, kMagic + (float)(128 + kBiasA)));
    // 25 fY - (25 fZ + 1/8) + 1/4 = 25 (fY - fZ) + 1/8, exact (multiples of 1/8 below 2^20)
    const unsigned bbits = __float_as_uint(__builtin_fmaf((LY.y - Zt) + 0.25f, 1.0f / 4096.0f, kMagic + (float)(128 + kBiasB)));
    e[k] = *reinterpret_cast<const int4*>(yf_b + L16);
    fx[k] = (int)(__umul24(abits, kFoldA) + (unsigned)e[k].x) >> 13;
    fz[k] = mad24_ks((int)bbits, -(int)kFoldB16, e[k].x) >> 13;
    x[k] = ab_to_xz_cube(fx[k]);
    // z - kZoff: the subtrahend rides in the second multiply
    z[k] = mad24_cs(mul24(fz[k], fz[k]) >> 14, fz[k], -(kZoff << 14)) >> 14;
  }
  // abToXZ_b's linear segment (i <= 3390: L* below ~8) is rare: one wave-uniform test for the group
  int lo;
  if constexpr (N == 4) {
    // eight values in three v_min3_i32 and one v_min_i32 (hipcc makes three v_min + two v_min3 of the linear chain)
    const int m0 = min(min(fx[0], fz[0]), fx[1]), m1 = min(min(fz[1], fx[2]), fz[2]);
    lo = min(min(min(fx[3], fz[3]), m0), m1);
  } else {
    lo = min(fx[0], fz[0]);
#pragma unroll
    for (int k = 1; k < N; k++) lo = min(lo, min(fx[k], fz[k]));
  }
  if (__builtin_amdgcn_ballot_w64(lo <= 3390) != 0ull) {
#pragma unroll
    for (int k = 0; k < N; k++) {
      if (fx[k] <= 3390) x[k] = ab_to_xz_linear(fx[k]);
      if (fz[k] <= 3390) z[k] = ab_to_xz_linear(fz[k]) - kZoff;
    }
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    // {x, z - kZoff} as int16 pair; x in [-361, 28027]
    const i16x2 xz = __builtin_bit_cast(i16x2, __builtin_amdgcn_perm((uint32_t)z[k], (uint32_t)x[k], 0x05040100u));
    constexpr i16x2 cb = {(short)kLabInv[0], (short)kLabInv[2]}, cg = {(short)kLabInv[3], (short)kLabInv[5]},
                    cr = {(short)kLabInv[6], (short)kLabInv[8]};
    // (the 4 KB sRGBInvGammaTab_b served by buffer_load_ubyte gathers from L1 instead of LDS: 2.43 -> 3.07 ms per 256
    // frames, round 3 -- a wave's 64 scattered bytes cost the texture path more than the LDS bank conflicts they avoid)
    out[k][0] = (unsigned)__builtin_amdgcn_sdot2(xz, cb, e[k].y, false) >> 14;
    out[k][1] = (unsigned)__builtin_amdgcn_sdot2(xz, cg, e[k].z, false) >> 14;
    out[k][2] = (unsigned)__builtin_amdgcn_sdot2(xz, cr, e[k].w, false) >> 14;
  }
}

// The sRGBInvGammaTab_b lookups that end the round trip.  `addr` are absolute LDS byte addresses (vignette_n).  On gfx950
// (SRAM-ECC on) the d16 LDS loads do not preserve the other half of their destination, they zero it
// (tools/probes/d16_probe.hip): ds_read_u8_d16_hi returns byte << 16.  Bytes 0 and 1 of an output dword are read with
// ds_read_u8, bytes 2 and 3 with ds_read_u8_d16_hi, and the dword is ((C | D') << 8) | (A | B'): two full-rate ORs and one
// v_lshl_or_b32 instead of two shifts, a shift-or and a three-way or.  hipcc does not emit the d16 forms (it merges byte
// reads with v_perm_b32), hence the inline assembly; LDS data returns in order, so the one s_waitcnt that closes the group
// covers every read, and it ties the registers so that no consumer is scheduled above it.
__device__ __forceinline__ void lds_u8(uint32_t& d, unsigned addr) { asm volatile("ds_read_u8 %0, %1" : "=v"(d) : "v"(addr)); }
__device__ __forceinline__ void lds_u8_hi(uint32_t& d, unsigned addr) { asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(d) : "v"(addr)); }
struct Pack3 {
  uint32_t a, b, c;
};
__device__ __forceinline__ uint32_t lshl8_or(uint32_t a, uint32_t b) {  // (a << 8) | b
  uint32_t d;
  asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ Pack3 invg_pack4(const unsigned (&addr)[4][3]) {
  const unsigned* f = &addr[0][0];  // output byte j comes from f[j]
  uint32_t v[12];
#pragma unroll
  for (int j = 0; j < 12; j++) {
    if ((j & 3) < 2)
      lds_u8(v[j], f[j]);
    else
      lds_u8_hi(v[j], f[j]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                 "+v"(v[10]), "+v"(v[11]));
  // ((B | D') << 8) | (A | C') as v_or, v_or, v_lshl_or_b32: hipcc picks v_or, v_lshlrev, v_or3 (two quarter-rate-class
  // instructions instead of one)
  Pack3 o;
  o.a = lshl8_or(v[1] | v[3], v[0] | v[2]);
  o.b = lshl8_or(v[5] | v[7], v[4] | v[6]);
  o.c = lshl8_or(v[9] | v[11], v[8] | v[10]);
  return o;
}
// the same lookups as separate values (a stage follows: the colour enhancer)
__device__ __forceinline__ void invg_values4(const unsigned (&addr)[4][3], int (&q)[4][3]) {
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) asm volatile("ds_read_u8 %0, %1" : "=v"(q[k][c]) : "v"(addr[k][c]));
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[2][0]), "+v"(q[2][1]),
                 "+v"(q[2][2]), "+v"(q[3][0]), "+v"(q[3][1]), "+v"(q[3][2]));
}

// max / min of two floats with the VOP3 output clamp to [0, 1]: one instruction (hipcc turns __saturatef into two compares
// and two selects)
__device__ __forceinline__ float max_sat(float a, float b) {
  float d;
  asm("v_max_f32_e64 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float min_sat(float a, float b) {
  float d;
  asm("v_min_f32_e64 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// color_enhancer.cpp:38-47: RGB2HSV_b (H in [0,180)), float gain with u8 saturation,
// HSV2RGB_b (float)
// UNIT: compile-time set of channels whose gain is exactly 1 (bit c: channel c of H, S, V)
// apply_hsv_f: the three results before their saturate_cast<uchar> (the fast kernel converts them straight into their place
// in the packed output: v_cvt_pk_u8_f32 inserts its byte into a given dword)
template <unsigned UNIT = 0u, typename Tabs>
__device__ __forceinline__ void apply_hsv_f(const float (&hg)[3], const Tabs& tb, int b, int g, int r, float (&out)[3]) {
  int v = max(b, max(g, r)), vmin = min(b, min(g, r));
  int diff = v - vmin;
  int s = (mul24(diff, tb.sdiv(v)) + (1 << 11)) >> 12;
  // (both candidates computed and selected with v_cndmask instead of the branches hipcc makes of this: 0.896 against 0.888 ms)
  int h;
  if (v == r)
    h = g - b;
  else if (v == g)
    h = b - r + 2 * diff;
  else
    h = r - g + 4 * diff;
  h = (mul24(h, tb.hdiv(diff)) + (1 << 11)) >> 12;
  h += h < 0 ? 180 : 0;
  // saturate_cast<uchar>(h) of RGB2HSV_b is dead: the numerator is within [-diff, 5 diff], so the scaled value is within
  // [-30, 150] and the wrapped one within [0, 179] (tests/test_oracle_known_answers.py::test_hsv_hue_needs_no_saturation)
  // cv::multiply(hsv, Scalar(gains)): saturate_cast<uchar>(float(x) * gain) per channel, then back to float for HSV2RGB_f.
  // A gain of exactly 1 leaves the 8-bit value as it is (h, s, v are already in [0, 255]), so the multiply and the two
  // conversions of such a channel can be skipped: UNIT names those channels at compile time (the caller branches once per
  // row of pixels on the wave-uniform gain pattern; a branch per pixel and channel cost what the skipped work saved).
  float fH = (float)h, fS = (float)s, fV = (float)v;
  if constexpr (!(UNIT & 1u)) fH = (float)sat_round_u8(fH * hg[0]);
  if constexpr (!(UNIT & 2u)) fS = (float)sat_round_u8(fS * hg[1]);
  if constexpr (!(UNIT & 4u)) fV = (float)sat_round_u8(fV * hg[2]);
  float fh = fH, fs = fS * (1.f / 255.f), fv = fV * (1.f / 255.f);
  // HSV2RGB_f (color_hsv.cpp): tab = {v, v(1-s), v(1-s*f), v(1-s*(1-f))}, (b, g, r) = tab[sector_data[sector][..]], i.e.
  // every output channel is v * (1 - s * w) with w in {0, 1, f, 1 - f} chosen by the sector (w = 0 and w = 1 reproduce tab[0]
  // and tab[1] exactly).  As a function of h = sector + f in [0, 6) the three weights are piecewise linear:
  //   w_b = sat(max(3 - h, h - 5))   w_g = sat(max(1 - h, h - 3))   w_r = sat(min(h - 1, 5 - h))
  // and each linear piece is bit-identical to OpenCV's f = h - sector or 1.f - f: h - k is exact for the integer k = sector,
  // and e.g. 5 - h and 1 - (h - 4) are single roundings of the same real number.  No sector, no per-lane select: three
  // subtract pairs, three min/max with the free [0, 1] output clamp.  With s == 0 every channel is v * 1 = v exactly, which
  // is OpenCV's early-out.
  fh = fh * (6.f / 180.f);
  // fmod(h, 6): h <= 255/30 < 12, and h >= 0, so sector is always in [0, 5].  With a hue gain of exactly 1 the 8-bit hue is
  // below 180 (above) and 179 * (6.f / 180.f) < 6: nothing to wrap
  if constexpr (!(UNIT & 1u)) fh = fh >= 6.f ? fh - 6.f : fh;
  const float wb = max_sat(3.f - fh, fh - 5.f);
  const float wg = max_sat(1.f - fh, fh - 3.f);
  const float wr = min_sat(fh - 1.f, 5.f - fh);
  // contracted model: 1 - s * w is one fnma (w = 0 and w = 1 still give tab[0] = v and tab[1] = v (1 - s) exactly)
  const float ob = fv * mul_add(-fs, wb, 1.f);
  const float og = fv * mul_add(-fs, wg, 1.f);
  const float orr = fv * mul_add(-fs, wr, 1.f);
  out[0] = ob * 255.f;
  out[1] = og * 255.f;
  out[2] = orr * 255.f;
}
// One entry (index i = the 8-bit h / s / v BEFORE its gain) of the per-launch tables of the fast kernels (rip_chain_dev.hpp HsvTab):
// exactly the operations apply_hsv_f<0> applies to a pixel whose channel has the value i.
__device__ __forceinline__ void hsv_tables_entry(const float* hg, int i, float& hs, float& hv, float4& hw) {
  const float fi = (float)i;
  hs = (float)sat_round_u8(fi * hg[1]) * (1.f / 255.f);
  hv = (float)sat_round_u8(fi * hg[2]) * (1.f / 255.f);
  float fh = (float)sat_round_u8(fi * hg[0]) * (6.f / 180.f);
  fh = fh >= 6.f ? fh - 6.f : fh;
  hw = make_float4(max_sat(3.f - fh, fh - 5.f), max_sat(1.f - fh, fh - 3.f), min_sat(fh - 1.f, 5.f - fh), 0.f);
}
template <unsigned UNIT = 0u, typename Tabs>
__device__ __forceinline__ void apply_hsv(const float (&hg)[3], const Tabs& tb, int& b, int& g, int& r) {
  float o[3];
  apply_hsv_f<UNIT>(hg, tb, b, g, r, o);
  b = sat_round_u8(o[0]);
  g = sat_round_u8(o[1]);
  r = sat_round_u8(o[2]);
}
// saturate_cast<uchar> of twelve floats (four pixels x B, G, R) straight into 12 interleaved bytes: v_cvt_pk_u8_f32 writes
// its result into byte `sel` of the dword it is given, so no shift / or merges the bytes afterwards
__device__ __forceinline__ Pack3 pack4_from_floats(const float (&f)[4][3]) {
  uint32_t d[3] = {0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 12; j++) d[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(f[j / 3][j % 3], j & 3, d[j >> 2]);
  return Pack3{d[0], d[1], d[2]};
}

// The pointwise chain after flip.  BITS >= 0: compile-time stage set; BITS < 0: runtime.
template <int BITS, int WB, typename Tabs>
__device__ __forceinline__ void pointwise(const ChainParams& p, const FrameWb& w, const Tabs& tb, const float* fwd,
                                          const int* inv, float mask, int& b, int& g, int& r) {
  const int bits = BITS >= 0 ? BITS : p.stage_bits;
  apply_wb(WB >= 0 ? WB : p.wb_mode, w, b, g, r);
  if (bits & ST_CC) apply_cc(p, b, g, r);
  if (bits & ST_VIG) {
    // gamma folded into lin_tab by the host
    apply_vignette(p, tb, fwd, inv, mask, b, g, r);
  } else if (bits & ST_GAMMA) {
    b = tb.gamma(b);
    g = tb.gamma(g);
    r = tb.gamma(r);
  }
  if (bits & ST_HSV) apply_hsv(p.hsv_gain, tb, b, g, r);
}

// ------------------------------------------------------------------------------------------------
// fast path: Bayer input, flip 0/180, cols % 4 == 0, rows % 2 == 0, dword-aligned pitches.
// One work item = 4 px x 2 rows (two Bayer quads): a 6x4 sample window held in 12 registers
// (three aligned dwords per row), 24 output bytes stored as two dwordx3.
// ------------------------------------------------------------------------------------------------
struct Window {
  uint32_t w[4][3];  // rows y0-1 .. y0+2; dwords at x0-4, x0, x0+4
  // sample at window row r, column offset i in [-1, 4] relative to x0
  __device__ __forceinline__ int at(int r, int i) const {
    const int idx = 4 + i;
    return (int)((w[r][idx >> 2] >> (8 * (idx & 3))) & 0xffu);
  }
};

// Frames are addressed through buffer resources (uniform base in SGPRs, 32-bit per-lane byte offset
// in one VGPR): the frame base changes per iteration of the frames loop on the scalar unit, and no
// 64-bit per-lane address arithmetic is issued on the VALU.  A frame is < 4 GiB (checked by the
// launchers); reads past `bytes` return 0.
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// byte offsets of the window dwords inside a frame: rows y0-1 .. y0+2 (clamped), left dword and
// centre dword (the right dword is centre + 4 except at the right image edge, where it is the centre)
struct WindowOffsets {
  unsigned left[4], centre[4];
  unsigned right_delta;  // 4, or 0 at the right edge
};
__device__ __forceinline__ WindowOffsets window_offsets(unsigned step, int rows, int cols, int y0, int x0) {
  WindowOffsets o;
  const int xl = x0 >= 4 ? x0 - 4 : x0;
  o.right_delta = x0 + 4 < cols ? 4u : 0u;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int y = clampi(y0 - 1 + r, 0, rows - 1);
    const unsigned row = __umul24((unsigned)y, step);  // 32-bit offsets: a frame is < 4 GiB, a row < 16 MiB
    o.left[r] = row + (unsigned)xl;
    o.centre[r] = row + (unsigned)x0;
  }
  return o;
}
__device__ __forceinline__ void load_window(__amdgpu_buffer_rsrc_t frame, const WindowOffsets& o, Window& win) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    win.w[r][0] = __builtin_amdgcn_raw_buffer_load_b32(frame, (int)o.left[r], 0, 0);
    win.w[r][1] = __builtin_amdgcn_raw_buffer_load_b32(frame, (int)o.centre[r], 0, 0);
    win.w[r][2] = __builtin_amdgcn_raw_buffer_load_b32(frame, (int)(o.centre[r] + o.right_delta), 0, 0);
  }
}

// Four pixels of one image row as planar byte vectors: byte j of .b/.g/.r = pixel j.
struct Planar {
  uint32_t b, g, r;
};
__device__ __forceinline__ uint32_t bfi32(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

// Bilinear demosaic of the 4x2 tile, four pixels per instruction, entirely on v_lerp_u8
// (D.byte = (S0.byte + S1.byte + (S2.byte & 1)) >> 1, all four bytes of a dword at once):
//   two taps   (a + b + 1) >> 1                      = lerp(a, b, 1)
//   four taps  (a + b + c + d + 2) >> 2              = lerp(lerp(a, b, 1), lerp(c, d, 1), ~((a ^ b) | (c ^ d)))
// The second line is exact: with ca = (a + b + 1) >> 1, ra = (a ^ b) & 1 (likewise cb, rb) the sum is
// 2 (ca + cb) + 2 - ra - rb, whose quarter is (ca + cb + 1) >> 1 when ra = rb = 0 and (ca + cb) >> 1 otherwise
// (tests/test_oracle_known_answers.py checks the identity over all byte quadruples of a generating set).
// No even/odd byte split, no widening to 16-bit lanes, and the pair averages are shared: the horizontal pair of a
// row serves the two-tap H of that row and the diagonal four-tap of the rows above and below it.
// One window row prepared for the demosaic: the centre dword, the two shifted views, their two-tap average and
// their parity word.
struct RowPrep {
  uint32_t c, wm, wp;  // columns x0 .. x0+3, x0-1 .. x0+2, x0+1 .. x0+4
  uint32_t h;          // (left + right + 1) >> 1 of every pixel
  uint32_t hx;         // left ^ right
};
__device__ __forceinline__ RowPrep prep_row(uint32_t left, uint32_t centre, uint32_t right) {
  RowPrep r;
  r.c = centre;
  r.wm = __builtin_amdgcn_alignbyte(centre, left, 3);
  r.wp = __builtin_amdgcn_alignbyte(right, centre, 1);
  r.h = __builtin_amdgcn_lerp(r.wm, r.wp, 0x01010101u);
  r.hx = r.wm ^ r.wp;
  return r;
}

// One output row (four pixels) from the prepared rows above / at / below it.
// RED_ROW: the row holds R samples; RX: column parity of the R samples.
template <bool RED_ROW, int RX>
__device__ __forceinline__ Planar debayer_row(const RowPrep& up, const RowPrep& at, const RowPrep& dn) {
  constexpr uint32_t kEven = RX == 0 ? 0x00FF00FFu : 0xFF00FF00u;  // byte lanes with dx == 0
  const uint32_t H = at.h;
  const uint32_t V = __builtin_amdgcn_lerp(up.c, dn.c, 0x01010101u);
  const uint32_t X4 = __builtin_amdgcn_lerp(H, V, ~(at.hx | (up.c ^ dn.c)));     // left, right, up, down
  const uint32_t D4 = __builtin_amdgcn_lerp(up.h, dn.h, ~(up.hx | dn.hx));       // the four diagonal neighbours
  const uint32_t C = at.c;
  Planar o;
  if (RED_ROW) {
    // red row: dx == 0 -> R site (B = diag, G = cross, R = centre); dx == 1 -> G site (B = vert, R = horiz)
    o.b = bfi32(kEven, D4, V);
    o.g = bfi32(kEven, X4, C);
    o.r = bfi32(kEven, C, H);
  } else {
    // blue row: dx == 0 -> G site (B = horiz, R = vert); dx == 1 -> B site (B = centre, G = cross, R = diag)
    o.b = bfi32(kEven, H, C);
    o.g = bfi32(kEven, C, X4);
    o.r = bfi32(kEven, V, D4);
  }
  return o;
}

// RY/RX: position of the R sample in the 2x2 cell.  out[ly] = image row y0 + ly; r[k] = row y0 - 1 + k.
template <int RY, int RX>
__device__ __forceinline__ void debayer_rows(const RowPrep& r0, const RowPrep& r1, const RowPrep& r2, const RowPrep& r3,
                                             Planar (&out)[2]) {
  out[0] = debayer_row<RY == 0, RX>(r0, r1, r2);
  out[1] = debayer_row<RY == 1, RX>(r1, r2, r3);
}
template <int RY, int RX>
__device__ __forceinline__ void debayer_swar(const Window& win, Planar (&out)[2]) {
  RowPrep r[4];
#pragma unroll
  for (int k = 0; k < 4; k++) r[k] = prep_row(win.w[k][0], win.w[k][1], win.w[k][2]);
  debayer_rows<RY, RX>(r[0], r[1], r[2], r[3], out);
}

// One output row (four pixels) from the prepared rows above / at / below it, for any Bayer pattern without a per-pattern
// code path.  `even`: the byte lanes whose column holds the row's first-listed site kind -- for a red row the R sites.
// Returns p (diagonal four-tap at those lanes, vertical two-tap at the others), g (cross four-tap / centre) and q (centre /
// horizontal two-tap): on a red row (b, g, r) = (p, g, q); a blue row is the same computation with the complementary lanes
// and (b, g, r) = (q, g, p).
struct RowPgq {
  uint32_t p, g, q;
};
__device__ __forceinline__ RowPgq debayer_row_pgq(const RowPrep& up, const RowPrep& at, const RowPrep& dn, uint32_t even) {
  const uint32_t H = at.h;
  const uint32_t V = __builtin_amdgcn_lerp(up.c, dn.c, 0x01010101u);
  const uint32_t X4 = __builtin_amdgcn_lerp(H, V, ~(at.hx | (up.c ^ dn.c)));     // left, right, up, down
  const uint32_t D4 = __builtin_amdgcn_lerp(up.h, dn.h, ~(up.hx | dn.hx));       // the four diagonal neighbours
  const uint32_t C = at.c;
  return RowPgq{bfi32(even, D4, V), bfi32(even, X4, C), bfi32(even, C, H)};
}

// ry / rx: position of the R sample in the 2x2 cell (wave-uniform).  out[ly] = image row y0 + ly; r[k] = row y0 - 1 + k.
// Exactly one of the two rows is a red row: with ry == 0 it is the first, the lanes of its R sites are the columns of
// parity rx, and the blue row below uses the complementary lanes; with ry == 1 the same code runs with the complementary
// masks and every (b, r) pair comes out exchanged -- one uniform branch with two register swaps puts them right.  (In the
// statistics kernel a switch over the four patterns made hipcc evaluate the byte merges of two patterns per tile; the
// fused chain keeps the switch -- there the templated merges with literal selectors are 2 % faster.)
__device__ __forceinline__ void debayer_rows_any(const RowPrep& r0, const RowPrep& r1, const RowPrep& r2, const RowPrep& r3, int ry,
                                                 int rx, Planar (&out)[2]) {
  const uint32_t lanes_rx = rx == 0 ? 0x00FF00FFu : 0xFF00FF00u;  // columns of parity rx
  const uint32_t m0 = ry == 0 ? lanes_rx : ~lanes_rx;
  const RowPgq a = debayer_row_pgq(r0, r1, r2, m0);
  const RowPgq c = debayer_row_pgq(r1, r2, r3, ~m0);
  out[0].b = a.p;
  out[0].g = a.g;
  out[0].r = a.q;
  out[1].b = c.q;
  out[1].g = c.g;
  out[1].r = c.p;
  if (ry != 0) {
    keep_branch();
    uint32_t t = out[0].b;
    out[0].b = out[0].r;
    out[0].r = t;
    t = out[1].b;
    out[1].b = out[1].r;
    out[1].r = t;
  }
}

// OpenCV's border replication on a demosaiced 4x2 tile: column 0 := column 1, column W-1 := W-2,
// then row 0 := row 1, row H-1 := H-2
// reversed: the planar bytes are in mirrored order (byte 3 = column x0: the 180-degree flip folded into the demosaic's byte
// merges, debayer_tile_sel), so the two column fix-ups trade places
// the wave-level test of debayer_fix_edges: lanes whose tile touches the image border.  It depends on the position only, so a
// kernel that walks the frames of a batch innermost takes it once per item (hipcc re-evaluates a ballot inside the loop with a
// v_cndmask / v_cmp pair per trip)
__device__ __forceinline__ unsigned long long debayer_edge_lanes(int y0, int x0, int rows, int cols) {
  return __builtin_amdgcn_ballot_w64(x0 == 0 || x0 + 4 == cols || y0 == 0 || y0 + 2 == rows);
}
__device__ __forceinline__ void debayer_fix_edges(int y0, int x0, int rows, int cols, Planar (&out)[2], bool reversed, unsigned long long edge_lanes) {
  // interior items (all but a 1 / (rows / 2) + 1 / (cols / 4) fraction) skip all of it behind one wave-uniform test
  if (edge_lanes == 0ull) return;
  const bool first_col = x0 == 0, last_col = x0 + 4 == cols;
  if (reversed ? last_col : first_col) {
#pragma unroll
    for (int ly = 0; ly < 2; ly++) {
      out[ly].b = (out[ly].b & 0xFFFFFF00u) | ((out[ly].b >> 8) & 0xFFu);
      out[ly].g = (out[ly].g & 0xFFFFFF00u) | ((out[ly].g >> 8) & 0xFFu);
      out[ly].r = (out[ly].r & 0xFFFFFF00u) | ((out[ly].r >> 8) & 0xFFu);
    }
  }
  if (reversed ? first_col : last_col) {
#pragma unroll
    for (int ly = 0; ly < 2; ly++) {
      out[ly].b = (out[ly].b & 0x00FFFFFFu) | ((out[ly].b << 8) & 0xFF000000u);
      out[ly].g = (out[ly].g & 0x00FFFFFFu) | ((out[ly].g << 8) & 0xFF000000u);
      out[ly].r = (out[ly].r & 0x00FFFFFFu) | ((out[ly].r << 8) & 0xFF000000u);
    }
  }
  if (y0 == 0) out[0] = out[1];
  if (y0 + 2 == rows) out[1] = out[0];
}
__device__ __forceinline__ void debayer_fix_edges(int y0, int x0, int rows, int cols, Planar (&out)[2], bool reversed = false) {
  debayer_fix_edges(y0, x0, rows, cols, out, reversed, debayer_edge_lanes(y0, x0, rows, cols));
}
// demosaic of the 4x2 tile at (y0, x0) including OpenCV's border replication
__device__ __forceinline__ void debayer_tile_any(const Window& win, int ry, int rx, int y0, int x0, int rows, int cols,
                                                 Planar (&out)[2]) {
  switch (ry * 2 + rx) {
    case 0: debayer_swar<0, 0>(win, out); break;
    case 1: debayer_swar<0, 1>(win, out); break;
    case 2: debayer_swar<1, 0>(win, out); break;
    default: debayer_swar<1, 1>(win, out); break;
  }
  debayer_fix_edges(y0, x0, rows, cols, out);
}

// The same tile with the column parity of the R samples AND the 180-degree flip folded into the selectors of the six byte
// merges (v_perm_b32 with the selector in an SGPR): one code path for the four patterns, no byte reversal afterwards.
// Selector of "byte lanes of parity `par` from the first operand, the others from the second", output mirrored or not:
// output byte k takes source byte j = mirrored ? 3 - k : k; v_perm_b32 numbers the first operand's bytes 4..7, the second's 0..3.
struct DemosaicSel {
  uint32_t first, second;  // red-site lanes of image row y0 / of row y0 + 1 (the complementary lanes)
  int ry;                  // 1: row y0 is the blue row -- every (b, r) pair comes out exchanged
  int mirrored;
};
__device__ __forceinline__ uint32_t merge_selector(int par, int mirrored) {
  return mirrored ? (par == 0 ? 0x04010603u : 0x00050207u) : (par == 0 ? 0x03060104u : 0x07020500u);
}
__device__ __forceinline__ DemosaicSel demosaic_selectors(int ry, int rx, int mirrored) {
  const int par = (rx ^ ry) & 1;  // parity of the columns whose byte lanes take the red-row roles in image row y0
  return DemosaicSel{merge_selector(par, mirrored), merge_selector(par ^ 1, mirrored), ry, mirrored};
}
__device__ __forceinline__ RowPgq debayer_row_sel(const RowPrep& up, const RowPrep& at, const RowPrep& dn, uint32_t sel) {
  const uint32_t H = at.h;
  const uint32_t V = __builtin_amdgcn_lerp(up.c, dn.c, 0x01010101u);
  const uint32_t X4 = __builtin_amdgcn_lerp(H, V, ~(at.hx | (up.c ^ dn.c)));  // left, right, up, down
  const uint32_t D4 = __builtin_amdgcn_lerp(up.h, dn.h, ~(up.hx | dn.hx));    // the four diagonal neighbours
  const uint32_t C = at.c;
  return RowPgq{__builtin_amdgcn_perm(D4, V, sel), __builtin_amdgcn_perm(X4, C, sel), __builtin_amdgcn_perm(C, H, sel)};
}
__device__ __forceinline__ void swap_regs(uint32_t& a, uint32_t& b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// the two rows of a tile from four prepared window rows (r0 = image row y0 - 1), any pattern, mirrored or not; no border rule
__device__ __forceinline__ void debayer_rows_sel(const RowPrep& r0, const RowPrep& r1, const RowPrep& r2, const RowPrep& r3, const DemosaicSel& ds,
                                                 Planar (&out)[2]) {
  const RowPgq a = debayer_row_sel(r0, r1, r2, ds.first);
  const RowPgq c = debayer_row_sel(r1, r2, r3, ds.second);
  out[0].b = a.p;
  out[0].g = a.g;
  out[0].r = a.q;
  out[1].b = c.q;
  out[1].g = c.g;
  out[1].r = c.p;
  if (ds.ry != 0) {  // scalar branch; the exchange happens in place (no copies on either side)
    swap_regs(out[0].b, out[0].r);
    swap_regs(out[1].b, out[1].r);
  }
}
__device__ __forceinline__ void debayer_tile_sel(const Window& win, const DemosaicSel& ds, int y0, int x0, int rows, int cols,
                                                 unsigned long long edge_lanes, Planar (&out)[2]) {
  RowPrep r[4];
#pragma unroll
  for (int k = 0; k < 4; k++) r[k] = prep_row(win.w[k][0], win.w[k][1], win.w[k][2]);
  debayer_rows_sel(r[0], r[1], r[2], r[3], ds, out);
  debayer_fix_edges(y0, x0, rows, cols, out, ds.mirrored != 0, edge_lanes);
}

// planar -> interleaved BGR (12 bytes) with six v_perm_b32
__device__ __forceinline__ void interleave4(const Planar& v, uint32_t& d0, uint32_t& d1, uint32_t& d2) {
  const uint32_t bg01 = __builtin_amdgcn_perm(v.g, v.b, 0x05010400u);  // B0 G0 B1 G1
  const uint32_t bg23 = __builtin_amdgcn_perm(v.g, v.b, 0x07030602u);  // B2 G2 B3 G3
  d0 = __builtin_amdgcn_perm(v.r, bg01, 0x02040100u);                  // B0 G0 R0 B1
  const uint32_t g1r1 = __builtin_amdgcn_perm(v.r, bg01, 0x00000503u); // G1 R1 . .
  d1 = __builtin_amdgcn_perm(bg23, g1r1, 0x05040100u);                 // G1 R1 B2 G2
  d2 = __builtin_amdgcn_perm(v.r, bg23, 0x07030206u);                  // R2 B3 G3 R3
}

// packs four BGR pixels (in the given order) into 12 bytes
__device__ __forceinline__ Pack3 pack4(const int (&q)[4][3]) {
  Pack3 o;
  o.a = (uint32_t)q[0][0] | ((uint32_t)q[0][1] << 8) | ((uint32_t)q[0][2] << 16) | ((uint32_t)q[1][0] << 24);
  o.b = (uint32_t)q[1][1] | ((uint32_t)q[1][2] << 8) | ((uint32_t)q[2][0] << 16) | ((uint32_t)q[2][1] << 24);
  o.c = (uint32_t)q[2][2] | ((uint32_t)q[3][0] << 8) | ((uint32_t)q[3][1] << 16) | ((uint32_t)q[3][2] << 24);
  return o;
}
__device__ __forceinline__ void store12(uint8_t* ptr, const Pack3& v) {
  uint3 u;
  u.x = v.a;
  u.y = v.b;
  u.z = v.c;
  *reinterpret_cast<uint3*>(ptr) = u;
}

__device__ __forceinline__ void store12(__amdgpu_buffer_rsrc_t frame, unsigned off, const Pack3& v) {
  u32x3 u = {v.a, v.b, v.c};
  __builtin_amdgcn_raw_buffer_store_b96(u, frame, (int)off, 0, 0);
}
// `nt` (wave-uniform): non-temporal store for an image no kernel of this batch reads again (aux bit 1 = nt on gfx950)
__device__ __forceinline__ void store12(__amdgpu_buffer_rsrc_t frame, unsigned off, const Pack3& v, bool nt) {
  u32x3 u = {v.a, v.b, v.c};
  if (nt) {
    keep_branch();
    __builtin_amdgcn_raw_buffer_store_b96(u, frame, (int)off, 0, 2);
  } else {
    __builtin_amdgcn_raw_buffer_store_b96(u, frame, (int)off, 0, 0);
  }
}

// the same with the flag as the kernel argument itself (an SGPR compared on the scalar unit: a bool computed outside a loop
// comes back through v_cndmask / v_cmp pairs in every trip)
__device__ __forceinline__ void store12(__amdgpu_buffer_rsrc_t frame, unsigned off, const Pack3& v, int nt) {
  u32x3 u = {v.a, v.b, v.c};
  asm volatile("" : "+s"(nt));  // keeps the comparison here, on the scalar unit (hipcc otherwise hoists it as a lane mask)
  if (nt != 0) {
    keep_branch();
    __builtin_amdgcn_raw_buffer_store_b96(u, frame, (int)off, 0, 2);
  } else {
    __builtin_amdgcn_raw_buffer_store_b96(u, frame, (int)off, 0, 0);
  }
}

// Grey-world applyChannelGains (x * q) >> 8 on four packed bytes.  q <= 256 (the gains are normalised
// by the largest one), so the products of the even and of the odd bytes stay inside their 16-bit
// lanes and one 24-bit multiply serves two pixels.
__device__ __forceinline__ uint32_t mul_u24_raw(uint32_t a, uint32_t b) {
  // v_mul_u32_u24 reads bits 23:0 of both sources whatever the rest holds; through the builtin hipcc masks an operand it
  // cannot prove small (the per-frame gain: three v_and_b32 per item)
  uint32_t d;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t gains_q8_swar(uint32_t v, unsigned q) {
  const uint32_t pe = mul_u24_raw(v & 0x00FF00FFu, q);
  const uint32_t po = mul_u24_raw((v >> 8) & 0x00FF00FFu, q);
  return bfi32(0xFF00FF00u, po, pe >> 8);
}

// item index -> (row pair, 4-px group) without an integer division per item
struct ItemMap {
  int groups_per_row;
  float inv_groups;
  __device__ __forceinline__ void split(int item, int& pair, int& grp) const {
    int q = (int)((float)item * inv_groups);
    int rem = item - q * groups_per_row;
    if (rem < 0) {
      q--;
      rem += groups_per_row;
    } else if (rem >= groups_per_row) {
      q++;
      rem -= groups_per_row;
    }
    pair = q;
    grp = rem;
  }
};

// four interleaved pixels (12 bytes) -> ints; rgb8 input is swapped to BGR here (debayer.cpp:72-73)
__device__ __forceinline__ void unpack12(const uint3& v, bool rgb, int (&q)[4][3]) {
  const uint32_t w[3] = {v.x, v.y, v.z};
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int byte = k * 3 + c;
      q[k][c] = (int)((w[byte >> 2] >> (8 * (byte & 3))) & 0xFFu);
    }
  if (rgb) {  // cvtColor(RGB2BGR), debayer.cpp:72-73
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int t = q[k][0];
      q[k][0] = q[k][2];
      q[k][2] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
int grid_blocks_for(long long work_items, int max_blocks) {
  long long b = (work_items + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

bool bayer_fast_geometry(const uint8_t* src, size_t step, size_t frame_stride, int rows, int cols, int kind) {
  return kind == SRC_BAYER && cols % 4 == 0 && rows % 2 == 0 && rows >= 4 && cols >= 4 && step % 4 == 0 &&
         frame_stride % 4 == 0 && aligned4(src) && step < (1u << 24) && rows < (1 << 23) &&
         (unsigned long long)step * (unsigned long long)rows < (1ull << 32);
}

// persistent grids are a multiple of 8 workgroups and at least 8: the kernels give every XCD its own share and stride by gridDim.x / 8
int grid_multiple_of_8(int blocks) { return std::max(8, blocks / 8 * 8); }

}  // namespace
}  // namespace rip
