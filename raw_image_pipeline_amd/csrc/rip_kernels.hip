// rip_kernels.hip -- gfx950 (MI355X) kernels of the RAW image chain.
//
// Stage semantics follow the reference's CPU/OpenCV path (file:line below are relative to the
// reference tree):
//   debayer   raw_image_pipeline/src/raw_image_pipeline/modules/debayer.cpp:45-79
//   flip      .../modules/flip.cpp:37-58
//   wb        .../modules/white_balance.cpp:59-64 (grey world), :73-136 (pca),
//             raw_image_pipeline_white_balance/src/.../convolutional_color_constancy.cpp:91-113 (ccc)
//   colour    .../modules/color_calibration.cpp:91-104
//   gamma     .../modules/gamma_correction.cpp:35-60
//   vignette  .../modules/vignetting_correction.cpp:32-93
//   hsv       .../modules/color_enhancer.cpp:38-47
//   remap     .../modules/undistortion.cpp:240-245
//
// Layout: everything is uint8 interleaved BGR in HBM.  The whole per-pixel chain
// (debayer -> flip -> wb gains -> 3x3 -> gamma -> vignette -> hsv) is ONE kernel: 1 B/px read,
// 3 B/px written; tables live in LDS; no intermediate image exists between those stages.
// Compile with -ffp-contract=off: the float stages reproduce OpenCV's separate mul/add.
#include "rip_kernels.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>

namespace rip {
namespace {

constexpr int kBlock = 256;

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

template <int BITS>
struct LdsTabs {
  static constexpr bool kVig = (BITS & ST_VIG) != 0;
  static constexpr bool kHsv = (BITS & ST_HSV) != 0;
  // gamma bytes are only needed when the gamma result itself is consumed (not folded into lin_tab)
  static constexpr bool kGamma = (BITS & ST_GAMMA) != 0 && !kVig;
  LdsArr<kGamma, uint8_t, 256> gamma_;
  LdsArr<kVig, float, 256> lin_;    // exact small integers held as float: the Lab forward sums run
  LdsArr<kVig, float, 3072> cbrt_;  // on v_fma_f32 (2 cycles) instead of v_mad_i32_i24 (4 cycles)
  LdsArr<kVig, uint32_t, 256> yf_;
  LdsArr<kVig, uint8_t, 4096> invg_;
  LdsArr<kHsv, int32_t, 256> sdiv_;
  LdsArr<kHsv, int32_t, 256> hdiv_;
  __device__ __forceinline__ int gamma(int i) const { return gamma_.v[i]; }
  __device__ __forceinline__ float linf(int i) const { return lin_.v[i]; }
  __device__ __forceinline__ float cbrtf(unsigned i) const { return cbrt_.v[i]; }
  __device__ __forceinline__ unsigned yf(int i) const { return yf_.v[i]; }
  __device__ __forceinline__ int invg(int i) const { return invg_.v[i]; }
  __device__ __forceinline__ int sdiv(int i) const { return sdiv_.v[i]; }
  __device__ __forceinline__ int hdiv(int i) const { return hdiv_.v[i]; }

  template <typename T, int N>
  static __device__ __forceinline__ void copy(T (&dst)[N], const T* src) {
    static_assert((N * sizeof(T)) % 4 == 0, "table size");
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    for (int i = threadIdx.x; i < (int)(N * sizeof(T) / 4); i += kBlock) d[i] = s[i];
  }
  __device__ __forceinline__ void load(const DevTables* t) {
    if constexpr (kGamma) copy(gamma_.v, t->gamma_lut);
    if constexpr (kVig) {
      for (int i = threadIdx.x; i < 256; i += kBlock) lin_.v[i] = (float)t->lin_tab[i];
      for (int i = threadIdx.x; i < 3072; i += kBlock) cbrt_.v[i] = (float)t->cbrt_tab[i];
      copy(yf_.v, t->yf_tab);
      copy(invg_.v, t->inv_gamma);
    }
    if constexpr (kHsv) {
      copy(sdiv_.v, t->sdiv);
      copy(hdiv_.v, t->hdiv);
    }
  }
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
    float bp = b2 * w.pca[0] + fb * w.pca[1];
    float rp = r2 * w.pca[2] + fr * w.pca[3];
    bp = bp > 255.f ? 255.f : bp;  // THRESH_TRUNC
    rp = rp > 255.f ? 255.f : rp;
    b = sat_round_u8(bp);
    r = sat_round_u8(rp);
  }
}

// color_calibration.cpp:93-103: ((m0*B + m1*G) + m2*R) + bias in float32, no FMA
__device__ __forceinline__ void apply_cc(const ChainParams& p, int& b, int& g, int& r) {
  float fb = (float)b, fg = (float)g, fr = (float)r;
  float o[3];
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = fb * p.cc_m[c * 3] + fg * p.cc_m[c * 3 + 1] + fr * p.cc_m[c * 3 + 2];
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

// vignetting mask value at destination pixel (row, col): vignetting_correction.cpp:32-63
__device__ __forceinline__ float vignette_mask(const ChainParams& p, int row, int col) {
  int dx2 = 2 * col - p.dcols, dy2 = 2 * row - p.drows;
  double s = (double)(mul24(dx2, dx2) + mul24(dy2, dy2)) * 0.25;  // exact (|dx2|, |dy2| < 2^23)
  double s2 = s * s;
  double k = s * p.vig_a2 + s2 * p.vig_a4;
  float m = (float)k;
  if (p.vig_has_max) m = m * p.vig_inv_max;
  m = m * p.vig_scale;
  m = m + 1.0f;
  return m;
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

// color_enhancer.cpp:38-47: RGB2HSV_b (H in [0,180)), float gain with u8 saturation,
// HSV2RGB_b (float)
template <typename Tabs>
__device__ __forceinline__ void apply_hsv(const ChainParams& p, const Tabs& tb, int& b, int& g, int& r) {
  int v = max(b, max(g, r)), vmin = min(b, min(g, r));
  int diff = v - vmin;
  int s = (mul24(diff, tb.sdiv(v)) + (1 << 11)) >> 12;
  int h;
  if (v == r)
    h = g - b;
  else if (v == g)
    h = b - r + 2 * diff;
  else
    h = r - g + 4 * diff;
  h = (mul24(h, tb.hdiv(diff)) + (1 << 11)) >> 12;
  h += h < 0 ? 180 : 0;
  h = clampi(h, 0, 255);
  int H = sat_round_u8((float)h * p.hsv_gain[0]);
  int S = sat_round_u8((float)s * p.hsv_gain[1]);
  int V = sat_round_u8((float)v * p.hsv_gain[2]);
  float fh = (float)H, fs = (float)S * (1.f / 255.f), fv = (float)V * (1.f / 255.f);
  float ob, og, orr;
  if (fs == 0.f) {
    ob = og = orr = fv;
  } else {
    fh = fh * (6.f / 180.f);
    if (fh >= 6.f) fh = fh - 6.f;  // fmod(h, 6): h <= 255/30 < 12
    int sector = (int)fh;          // floor, h >= 0
    fh = fh - (float)sector;
    if ((unsigned)sector >= 6u) {
      sector = 0;
      fh = 0.f;
    }
    float t0 = fv;
    float t1 = fv * (1.f - fs);
    float t2 = fv * (1.f - fs * fh);
    float t3 = fv * (1.f - fs * (1.f - fh));
    switch (sector) {
      case 0: ob = t1; og = t3; orr = t0; break;
      case 1: ob = t1; og = t0; orr = t2; break;
      case 2: ob = t3; og = t0; orr = t1; break;
      case 3: ob = t0; og = t2; orr = t1; break;
      case 4: ob = t0; og = t1; orr = t3; break;
      default: ob = t2; og = t1; orr = t0; break;
    }
  }
  b = sat_round_u8(ob * 255.f);
  g = sat_round_u8(og * 255.f);
  r = sat_round_u8(orr * 255.f);
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
  if (bits & ST_HSV) apply_hsv(p, tb, b, g, r);
}

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
    pointwise<-1, -1>(p, w, tb, fwdf, p.tabs->lab_inv_pk, (p.stage_bits & ST_VIG) ? vignette_mask(p, yd, xd) : 1.0f, b, g, r);
    uint8_t* o = dst + (size_t)yd * p.dst_step + (size_t)xd * 3;
    o[0] = (uint8_t)b;
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)r;
  }
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

// Bilinear demosaic of the 4x2 tile, four pixels per instruction (SWAR): every row of the window is
// split into its even and odd bytes, widened to 16-bit lanes inside a dword, so one v_add_u32 adds
// two taps of two pixels and the sums (<= 4*255 + 2) cannot carry across lanes.
// RY/RX: position of the R sample in the 2x2 cell.  out[ly] = image row y0 + ly.
// One window row prepared for the SWAR sums: the centre dword, the two shifted views and their
// even / odd bytes widened to 16-bit lanes.
struct RowPrep {
  uint32_t c, wm, wp;        // columns x0 .. x0+3, x0-1 .. x0+2, x0+1 .. x0+4
  uint32_t hs_lo, hs_hi;     // left + right neighbour of pixels (0, 2) and (1, 3)
  uint32_t c_lo, c_hi;       // centre bytes of pixels (0, 2) and (1, 3)
};
__device__ __forceinline__ RowPrep prep_row(uint32_t left, uint32_t centre, uint32_t right) {
  constexpr uint32_t M8 = 0x00FF00FFu;
  RowPrep r;
  r.c = centre;
  r.wm = __builtin_amdgcn_alignbyte(centre, left, 3);
  r.wp = __builtin_amdgcn_alignbyte(right, centre, 1);
  r.hs_lo = (r.wm & M8) + (r.wp & M8);
  r.hs_hi = ((r.wm >> 8) & M8) + ((r.wp >> 8) & M8);
  r.c_lo = centre & M8;
  r.c_hi = (centre >> 8) & M8;
  return r;
}

// One output row (four pixels) from the prepared rows above / at / below it.
// RED_ROW: the row holds R samples; RX: column parity of the R samples.
template <bool RED_ROW, int RX>
__device__ __forceinline__ Planar debayer_row(const RowPrep& up, const RowPrep& at, const RowPrep& dn) {
  constexpr uint32_t M8 = 0x00FF00FFu;
  constexpr uint32_t kEven = RX == 0 ? 0x00FF00FFu : 0xFF00FF00u;  // byte lanes with dx == 0
  // two-tap averages (a + b + 1) >> 1 of all four byte lanes in one v_lerp_u8
  const uint32_t H = __builtin_amdgcn_lerp(at.wm, at.wp, 0x01010101u);
  const uint32_t V = __builtin_amdgcn_lerp(up.c, dn.c, 0x01010101u);
  const uint32_t X4 = (((at.hs_lo + up.c_lo + dn.c_lo + 0x00020002u) >> 2) & M8) |
                      ((((at.hs_hi + up.c_hi + dn.c_hi + 0x00020002u) >> 2) & M8) << 8);
  const uint32_t D4 = (((up.hs_lo + dn.hs_lo + 0x00020002u) >> 2) & M8) | ((((up.hs_hi + dn.hs_hi + 0x00020002u) >> 2) & M8) << 8);
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

// Bilinear demosaic of the 4x2 tile, four pixels per instruction (SWAR): every row of the window is
// split into its even and odd bytes, widened to 16-bit lanes inside a dword, so one v_add_u32 adds
// two taps of two pixels and the sums (<= 4*255 + 2) cannot carry across lanes.
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

// OpenCV's border replication on a demosaiced 4x2 tile: column 0 := column 1, column W-1 := W-2,
// then row 0 := row 1, row H-1 := H-2
__device__ __forceinline__ void debayer_fix_edges(int y0, int x0, int rows, int cols, Planar (&out)[2]) {
  if (x0 == 0) {
#pragma unroll
    for (int ly = 0; ly < 2; ly++) {
      out[ly].b = (out[ly].b & 0xFFFFFF00u) | ((out[ly].b >> 8) & 0xFFu);
      out[ly].g = (out[ly].g & 0xFFFFFF00u) | ((out[ly].g >> 8) & 0xFFu);
      out[ly].r = (out[ly].r & 0xFFFFFF00u) | ((out[ly].r >> 8) & 0xFFu);
    }
  }
  if (x0 + 4 == cols) {
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
__device__ __forceinline__ void debayer_rows_any(const RowPrep& r0, const RowPrep& r1, const RowPrep& r2, const RowPrep& r3, int ry,
                                                 int rx, Planar (&out)[2]) {
  switch (ry * 2 + rx) {
    case 0: debayer_rows<0, 0>(r0, r1, r2, r3, out); break;
    case 1: debayer_rows<0, 1>(r0, r1, r2, r3, out); break;
    case 2: debayer_rows<1, 0>(r0, r1, r2, r3, out); break;
    default: debayer_rows<1, 1>(r0, r1, r2, r3, out); break;
  }
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

// planar -> interleaved BGR (12 bytes) with six v_perm_b32
__device__ __forceinline__ void interleave4(const Planar& v, uint32_t& d0, uint32_t& d1, uint32_t& d2) {
  const uint32_t bg01 = __builtin_amdgcn_perm(v.g, v.b, 0x05010400u);  // B0 G0 B1 G1
  const uint32_t bg23 = __builtin_amdgcn_perm(v.g, v.b, 0x07030602u);  // B2 G2 B3 G3
  d0 = __builtin_amdgcn_perm(v.r, bg01, 0x02040100u);                  // B0 G0 R0 B1
  const uint32_t g1r1 = __builtin_amdgcn_perm(v.r, bg01, 0x00000503u); // G1 R1 . .
  d1 = __builtin_amdgcn_perm(bg23, g1r1, 0x05040100u);                 // G1 R1 B2 G2
  d2 = __builtin_amdgcn_perm(v.r, bg23, 0x07030206u);                  // R2 B3 G3 R3
}

struct Pack3 {
  uint32_t a, b, c;
};
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

// Grey-world applyChannelGains (x * q) >> 8 on four packed bytes.  q <= 256 (the gains are normalised
// by the largest one), so the products of the even and of the odd bytes stay inside their 16-bit
// lanes and one 24-bit multiply serves two pixels.
__device__ __forceinline__ uint32_t gains_q8_swar(uint32_t v, unsigned q) {
  const uint32_t pe = __umul24(v & 0x00FF00FFu, q);
  const uint32_t po = __umul24((v >> 8) & 0x00FF00FFu, q);
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

template <int BITS, int WB>
__global__ __launch_bounds__(kBlock) void chain_fast_kernel(ChainParams p, ItemMap im, int items_per_frame) {
  __shared__ LdsTabs<BITS> tb;
  __shared__ float s_fwd[9];
  __shared__ int s_inv[6];
  tb.load(p.tabs);
  if (threadIdx.x < 9) {
    s_fwd[threadIdx.x] = (float)p.tabs->lab_fwd[threadIdx.x];
    if (threadIdx.x < 6) s_inv[threadIdx.x] = p.tabs->lab_inv_pk[threadIdx.x];
  }
  __syncthreads();
  // Persistent workgroups: the LDS tables are loaded once and amortised over many chunks of
  // kBlock items.  Block b runs on XCD b % 8 (observed dispatch order; speed only), so each
  // XCD walks its own contiguous range of chunks and vertically adjacent row pairs -- which
  // share two halo rows -- hit the same L2.  Frames are the innermost loop: everything that
  // depends only on the position (item split, vignetting mask in FP64, addresses) is computed
  // once per item and reused for every frame of the batch.
  const int chunks_per_frame = (items_per_frame + kBlock - 1) / kBlock;
  // small frames do not fill the chip with one frame's chunks: blockIdx.y splits the batch
  const int f_per_group = (p.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(p.n_frames, f_begin + f_per_group);
  const int per_xcd = (chunks_per_frame + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const bool flip180 = p.flip_angle == 180;
  for (int ci = blockIdx.x >> 3; ci < per_xcd; ci += gridDim.x >> 3) {
    const int chunk = xcd * per_xcd + ci;
    if (chunk >= chunks_per_frame) break;
    const int item = chunk * kBlock + threadIdx.x;
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
#pragma unroll
      for (int k = 0; k < 4; k++) mask[ly][k] = (BITS & ST_VIG) ? vignette_mask(p, yd, xbase + k) : 1.0f;
    }
    const WindowOffsets wo = window_offsets((unsigned)p.src_step, p.rows, p.cols, y0, x0);
    const unsigned src_bytes = __umul24((unsigned)(p.rows - 1), (unsigned)p.src_step) + (unsigned)p.cols;
    const unsigned dst_bytes = __umul24((unsigned)(p.drows - 1), (unsigned)p.dst_step) + (unsigned)p.dcols * 3u;
    const unsigned tap_bytes = __umul24((unsigned)p.drows, (unsigned)p.dcols) * 3u;
    for (int frame = f_begin; frame < f_end; frame++) {
      const __amdgpu_buffer_rsrc_t src = frame_rsrc(p.src + (size_t)frame * p.src_frame_stride, src_bytes);
      const __amdgpu_buffer_rsrc_t dst = frame_rsrc(p.dst + (size_t)frame * p.dst_frame_stride, dst_bytes);
      const bool has_tap = p.tap != nullptr;
      const __amdgpu_buffer_rsrc_t tap = frame_rsrc(has_tap ? p.tap + (size_t)frame * p.tap_frame_stride : nullptr, has_tap ? tap_bytes : 0u);
      FrameWb w;
      if (WB != WB_NONE) w = p.wb[frame];
      Window win;
      load_window(src, wo, win);
      Planar rowpx[2];
      debayer_tile_any(win, p.bayer_ry, p.bayer_rx, y0, x0, p.rows, p.cols, rowpx);
#pragma unroll
      for (int ly = 0; ly < 2; ly++) {
        Planar v = rowpx[ly];
        if (flip180) {  // the group is written mirrored: reverse the four pixels
          keep_branch();
          v.b = __builtin_bswap32(v.b);
          v.g = __builtin_bswap32(v.g);
          v.r = __builtin_bswap32(v.r);
        }
        Pack3 raw;
        const bool need_raw = has_tap || (BITS == 0 && WB == WB_NONE);
        if (need_raw) interleave4(v, raw.a, raw.b, raw.c);
        if (has_tap) store12(tap, tap_off[ly], raw);
        if (BITS == 0 && WB == WB_NONE) {
          store12(dst, dst_off[ly], raw);  // pure demosaic: no per-pixel stage
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
          pointwise<BITS, WB == WB_Q8 ? WB_NONE : WB>(p, w, tb, s_fwd, s_inv, mask[ly][k], q[k][0], q[k][1], q[k][2]);
        }
        store12(dst, dst_off[ly], pack4(q));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// colour-input path (bgr8 / rgb8 frames, e.g. the reference's Python demo): 4 px per lane, 12-byte
// loads and stores, flip 0/180, stage set decided at run time (wave-uniform branches), all tables in LDS
// ------------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(kBlock) void chain_color_kernel(ChainParams p, ItemMap im, int items_per_frame) {
  __shared__ LdsTabs<ST_CC | ST_GAMMA | ST_VIG | ST_HSV> tb;
  __shared__ uint8_t s_gamma[256];  // LdsTabs<...VIG> folds gamma into lin_tab; the plain LUT is needed too
  __shared__ float s_fwd[9];
  __shared__ int s_inv[6];
  tb.load(p.tabs);
  s_gamma[threadIdx.x] = p.tabs->gamma_lut[threadIdx.x];
  if (threadIdx.x < 9) {
    s_fwd[threadIdx.x] = (float)p.tabs->lab_fwd[threadIdx.x];
    if (threadIdx.x < 6) s_inv[threadIdx.x] = p.tabs->lab_inv_pk[threadIdx.x];
  }
  __syncthreads();
  const int chunks_per_frame = (items_per_frame + kBlock - 1) / kBlock;
  const int f_per_group = (p.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(p.n_frames, f_begin + f_per_group);
  const bool flip180 = p.flip_angle == 180;
  const bool rgb = p.src_kind == SRC_RGB;
  const bool vig = (p.stage_bits & ST_VIG) != 0, gam = (p.stage_bits & ST_GAMMA) != 0;
  for (int chunk = blockIdx.x; chunk < chunks_per_frame; chunk += gridDim.x) {
    const int item = chunk * kBlock + threadIdx.x;
    if (item >= items_per_frame) continue;
    int ys, grp;
    im.split(item, ys, grp);
    const int x0 = grp * 4;
    const int yd = flip180 ? p.rows - 1 - ys : ys;
    const int xbase = flip180 ? p.cols - 4 - x0 : x0;
    float mask[4];
#pragma unroll
    for (int k = 0; k < 4; k++) mask[k] = vig ? vignette_mask(p, yd, xbase + k) : 1.0f;
    const unsigned src_off = __umul24((unsigned)ys, (unsigned)p.src_step) + (unsigned)x0 * 3u;
    const unsigned dst_off = __umul24((unsigned)yd, (unsigned)p.dst_step) + (unsigned)xbase * 3u;
    const unsigned tap_off = (__umul24((unsigned)yd, (unsigned)p.dcols) + (unsigned)xbase) * 3u;
    for (int frame = f_begin; frame < f_end; frame++) {
      const uint3 in = *reinterpret_cast<const uint3*>(p.src + (size_t)frame * p.src_frame_stride + src_off);
      FrameWb w;
      if (p.wb_mode != WB_NONE) w = p.wb[frame];
      int s[4][3], q[4][3];
      unpack12(in, rgb, s);
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) q[k][c] = flip180 ? s[3 - k][c] : s[k][c];
      if (p.tap) store12(p.tap + (size_t)frame * p.tap_frame_stride + tap_off, pack4(q));
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int b = q[k][0], g = q[k][1], r = q[k][2];
        apply_wb(p.wb_mode, w, b, g, r);
        if (p.stage_bits & ST_CC) apply_cc(p, b, g, r);
        if (vig) {
          apply_vignette(p, tb, s_fwd, s_inv, mask[k], b, g, r);
        } else if (gam) {
          b = s_gamma[b];
          g = s_gamma[g];
          r = s_gamma[r];
        }
        if (p.stage_bits & ST_HSV) apply_hsv(p, tb, b, g, r);
        q[k][0] = b;
        q[k][1] = g;
        q[k][2] = r;
      }
      store12(p.dst + (size_t)frame * p.dst_frame_stride + dst_off, pack4(q));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Bayer input with a 90 / 270 degree flip (flip.cpp:45-60: transpose + flip == cv::rotate).  Same
// window / SWAR demosaic / per-pixel stages as chain_fast_kernel, stage set decided at run time.  A
// 4x2 item lands as four 2-pixel (6-byte) pieces in four output rows, so the lanes of a workgroup are
// laid out 4 column groups x 64 row pairs: for one output row the 16 row pairs a wave holds write 96
// contiguous bytes (and the four waves of the workgroup 384), while each source row is still read in
// 16..24-byte runs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store6(__amdgpu_buffer_rsrc_t frame, unsigned off, uint32_t first, uint32_t second) {
  // two packed pixels (b | g << 8 | r << 16) as three 16-bit stores: the address is only 2-byte aligned
  const uint32_t lo = first | (second << 24), hi = second >> 8;
  __builtin_amdgcn_raw_buffer_store_b16((short)(lo & 0xffffu), frame, (int)off, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b16((short)(lo >> 16), frame, (int)off + 2, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b16((short)(hi & 0xffffu), frame, (int)off + 4, 0, 0);
}

__global__ __launch_bounds__(kBlock) void chain_rot_kernel(ChainParams p, int tiles_x, int tiles_per_frame) {
  __shared__ LdsTabs<ST_CC | ST_GAMMA | ST_VIG | ST_HSV> tb;
  __shared__ uint8_t s_gamma[256];  // LdsTabs<...VIG> folds gamma into lin_tab; the plain LUT is needed too
  __shared__ float s_fwd[9];
  __shared__ int s_inv[6];
  tb.load(p.tabs);
  s_gamma[threadIdx.x] = p.tabs->gamma_lut[threadIdx.x];
  if (threadIdx.x < 9) {
    s_fwd[threadIdx.x] = (float)p.tabs->lab_fwd[threadIdx.x];
    if (threadIdx.x < 6) s_inv[threadIdx.x] = p.tabs->lab_inv_pk[threadIdx.x];
  }
  __syncthreads();
  const int f_per_group = (p.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(p.n_frames, f_begin + f_per_group);
  const bool rot90 = p.flip_angle == 90;
  const bool vig = (p.stage_bits & ST_VIG) != 0, gam = (p.stage_bits & ST_GAMMA) != 0;
  const int groups = p.cols >> 2, pairs = p.rows >> 1;
  const unsigned src_bytes = __umul24((unsigned)(p.rows - 1), (unsigned)p.src_step) + (unsigned)p.cols;
  const unsigned dst_bytes = __umul24((unsigned)(p.drows - 1), (unsigned)p.dst_step) + (unsigned)p.dcols * 3u;
  const unsigned tap_bytes = __umul24((unsigned)p.drows, (unsigned)p.dcols) * 3u;
  const bool has_tap = p.tap != nullptr;
  for (int tile = blockIdx.x; tile < tiles_per_frame; tile += gridDim.x) {
    const int tpy = tile / tiles_x, tgx = tile - tpy * tiles_x;
    const int grp = tgx * 4 + (int)(threadIdx.x & 3u), pair = tpy * 64 + (int)(threadIdx.x >> 2);
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
      for (int ly = 0; ly < 2; ly++) mask[ly][k] = vig ? vignette_mask(p, row_d, rot90 ? col_d + 1 - ly : col_d + ly) : 1.0f;
    }
    const WindowOffsets wo = window_offsets((unsigned)p.src_step, p.rows, p.cols, y0, x0);
    for (int frame = f_begin; frame < f_end; frame++) {
      const __amdgpu_buffer_rsrc_t src = frame_rsrc(p.src + (size_t)frame * p.src_frame_stride, src_bytes);
      const __amdgpu_buffer_rsrc_t dst = frame_rsrc(p.dst + (size_t)frame * p.dst_frame_stride, dst_bytes);
      const __amdgpu_buffer_rsrc_t tap = frame_rsrc(has_tap ? p.tap + (size_t)frame * p.tap_frame_stride : nullptr, has_tap ? tap_bytes : 0u);
      FrameWb w;
      if (p.wb_mode != WB_NONE) w = p.wb[frame];
      Window win;
      load_window(src, wo, win);
      Planar rowpx[2];
      debayer_tile_any(win, p.bayer_ry, p.bayer_rx, y0, x0, p.rows, p.cols, rowpx);
      uint32_t raw[2][4], pix[2][4];  // b | g << 8 | r << 16
#pragma unroll
      for (int ly = 0; ly < 2; ly++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          int b = (int)((rowpx[ly].b >> (8 * k)) & 0xFFu), g = (int)((rowpx[ly].g >> (8 * k)) & 0xFFu),
              r = (int)((rowpx[ly].r >> (8 * k)) & 0xFFu);
          raw[ly][k] = (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16);
          apply_wb(p.wb_mode, w, b, g, r);
          if (p.stage_bits & ST_CC) apply_cc(p, b, g, r);
          if (vig) {
            apply_vignette(p, tb, s_fwd, s_inv, mask[ly][k], b, g, r);
          } else if (gam) {
            b = s_gamma[b];
            g = s_gamma[g];
            r = s_gamma[r];
          }
          if (p.stage_bits & ST_HSV) apply_hsv(p, tb, b, g, r);
          pix[ly][k] = (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16);
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

// ------------------------------------------------------------------------------------------------
// statistics kernels (grey-world sums, pca sums/maxima): integer reductions, wave64 shuffles
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ unsigned wave_max(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_down(v, off, 64));
  return v;
}

struct StatAcc {
  unsigned s[5];
  unsigned m[3];
};

__device__ __forceinline__ void stat_add(const StatsParams& p, int b, int g, int r, StatAcc& a, unsigned* s_hist) {
  if (p.mode == WB_SIMPLE) {
    // SimpleWB: per-channel 256-bin histograms, privatised in LDS
    atomicAdd(&s_hist[b], 1u);
    atomicAdd(&s_hist[256 + g], 1u);
    atomicAdd(&s_hist[512 + r], 1u);
  } else if (p.mode == WB_Q8) {
    // GrayworldWB calculateChannelSums: skip when (max-min)*255 > thresh255*max
    unsigned mn = (unsigned)min(b, min(g, r)), mx = (unsigned)max(b, max(g, r));
    if ((mx - mn) * 255u > p.thresh255 * mx) return;
    a.s[0] += b;
    a.s[1] += g;
    a.s[2] += r;
  } else {
    a.s[0] += b;
    a.s[1] += b * b;
    a.s[2] += r;
    a.s[3] += r * r;
    a.s[4] += g;
    a.m[0] = max(a.m[0], (unsigned)b);
    a.m[1] = max(a.m[1], (unsigned)r);
    a.m[2] = max(a.m[2], (unsigned)g);
  }
}

__device__ __forceinline__ void stat_hist_init(const StatsParams& p, unsigned* s_hist) {
  if (p.mode != WB_SIMPLE) return;
  for (int i = threadIdx.x; i < 768; i += kBlock) s_hist[i] = 0u;
  __syncthreads();
}

__device__ __forceinline__ void stat_flush(const StatsParams& p, StatAcc& a, FrameStats* out, unsigned* s_hist, int frame) {
  if (p.mode == WB_SIMPLE) {
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += kBlock)
      if (s_hist[i]) atomicAdd(&p.hist3[(size_t)frame * 768 + i], s_hist[i]);
    return;
  }
  __shared__ unsigned sh[8][kBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    unsigned v = wave_sum(a.s[k]);
    if (lane == 0) sh[k][wid] = v;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    unsigned v = wave_max(a.m[k]);
    if (lane == 0) sh[5 + k][wid] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    unsigned long long t = 0;
    for (int i = 0; i < kBlock / 64; i++) t += sh[threadIdx.x][i];
    if (t) atomicAdd(&out->sum[threadIdx.x], t);
  } else if (threadIdx.x < 8 && p.mode == WB_PCA) {
    unsigned t = 0;
    for (int i = 0; i < kBlock / 64; i++) t = max(t, sh[threadIdx.x][i]);
    atomicMax(&out->mx[threadIdx.x - 5], t);
  }
}

// Grey-world statistics of four planar pixels, two pixels per instruction: the bytes are widened to
// 16-bit lanes; max/min with v_pk_max/min_u16; both sides of the saturation test
// (max - min) * 255 > thresh255 * max fit 16 bits (thresh255 <= 255 after the clamp below, which does
// not change the outcome: for thresh255 >= 255 no pixel is ever skipped); the masked channel sums are
// one v_dot2_u32_u16 per channel with the 0/1 keep flags as weights.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ void grayworld_add_swar(const Planar& v, unsigned thresh255, StatAcc& a) {
  constexpr uint32_t M8 = 0x00FF00FFu;
  const u16x2 t2 = as_u16x2(thresh255 * 0x00010001u), one2 = as_u16x2(0x00010001u);
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const uint32_t bw = (half ? v.b >> 8 : v.b) & M8, gw = (half ? v.g >> 8 : v.g) & M8, rw = (half ? v.r >> 8 : v.r) & M8;
    const u16x2 b2 = as_u16x2(bw), g2 = as_u16x2(gw), r2 = as_u16x2(rw);
    const u16x2 mx = __builtin_elementwise_max(__builtin_elementwise_max(b2, g2), r2);
    const u16x2 mn = __builtin_elementwise_min(__builtin_elementwise_min(b2, g2), r2);
    const uint32_t d = __builtin_bit_cast(uint32_t, mx) - __builtin_bit_cast(uint32_t, mn);  // lane-wise: max >= min
    const u16x2 lhs = as_u16x2((d << 8) - d);                                                // * 255, <= 65025 per lane
    const u16x2 rhs = mx * t2;
    const u16x2 skip = __builtin_elementwise_min(__builtin_elementwise_sub_sat(lhs, rhs), one2);  // 1 where lhs > rhs
    const u16x2 keep = one2 - skip;
    a.s[0] = __builtin_amdgcn_udot2(b2, keep, a.s[0], false);
    a.s[1] = __builtin_amdgcn_udot2(g2, keep, a.s[1], false);
    a.s[2] = __builtin_amdgcn_udot2(r2, keep, a.s[2], false);
  }
}

// Statistics of a Bayer frame.  A wave owns a strip 64 groups (256 px) wide and walks down
// `pairs_per_task` row pairs of it: the row index is wave-uniform, so a row's byte offset lives in an
// SGPR (the buffer instruction's soffset) and the three per-lane column offsets never change -- no
// per-item address arithmetic -- and two of the four window rows (with their SWAR preparation) carry
// over from one row pair to the next.
template <int MODE>
__global__ __launch_bounds__(kBlock) void stats_fast_kernel(StatsParams p, int col_waves, int pairs_per_task, int n_tasks) {
  __shared__ unsigned s_hist[MODE == WB_SIMPLE ? 768 : 1];
  p.mode = MODE;  // the per-pixel switch in stat_add folds away
  stat_hist_init(p, s_hist);
  const int frame = blockIdx.y;
  const unsigned step = (unsigned)p.src_step;
  const unsigned src_bytes = __umul24((unsigned)(p.rows - 1), step) + (unsigned)p.cols;
  const __amdgpu_buffer_rsrc_t src = frame_rsrc(p.src + (size_t)frame * p.src_frame_stride, src_bytes);
  const unsigned thresh255 = min(p.thresh255, 255u);
  const int lane = threadIdx.x & 63;
  const int task = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)));
  StatAcc a = {};
  if (task < n_tasks) {
    const int cw = task % col_waves, range = task / col_waves;
    const int grp = cw * 64 + lane;
    const bool active = grp * 4 < p.cols;
    const int x0 = active ? grp * 4 : 0;
    const int off_c = x0, off_l = x0 >= 4 ? x0 - 4 : x0, off_r = x0 + 4 < p.cols ? x0 + 4 : x0;
    const int n_pairs = p.rows >> 1;
    const int pair_begin = range * pairs_per_task, pair_end = min(n_pairs, pair_begin + pairs_per_task);
    struct RawRow {
      uint32_t l, c, r;
    };
    auto fetch_row = [&](int y) {
      const int row = (int)__umul24((unsigned)clampi(y, 0, p.rows - 1), step);  // wave-uniform: scalar
      return RawRow{__builtin_amdgcn_raw_buffer_load_b32(src, off_l, row, 0), __builtin_amdgcn_raw_buffer_load_b32(src, off_c, row, 0),
                    __builtin_amdgcn_raw_buffer_load_b32(src, off_r, row, 0)};
    };
    auto prep = [&](const RawRow& w) { return prep_row(w.l, w.c, w.r); };
    auto consume = [&](const RowPrep& r0, const RowPrep& r1, const RowPrep& r2, const RowPrep& r3, int y0) {
      Planar rowpx[2];
      debayer_rows_any(r0, r1, r2, r3, p.bayer_ry, p.bayer_rx, rowpx);
      debayer_fix_edges(y0, x0, p.rows, p.cols, rowpx);
      if (!active) return;
#pragma unroll
      for (int ly = 0; ly < 2; ly++) {
        if (MODE == WB_Q8) {
          grayworld_add_swar(rowpx[ly], thresh255, a);
          continue;
        }
#pragma unroll
        for (int lx = 0; lx < 4; lx++)
          stat_add(p, (int)((rowpx[ly].b >> (8 * lx)) & 0xFFu), (int)((rowpx[ly].g >> (8 * lx)) & 0xFFu),
                   (int)((rowpx[ly].r >> (8 * lx)) & 0xFFu), a, s_hist);
      }
    };
    // two row pairs per iteration so the carried rows change roles without register moves; the two
    // rows of the next pair are in flight while the current pair is reduced
    int y0 = pair_begin * 2;
    RowPrep ra = prep(fetch_row(y0 - 1)), rb = prep(fetch_row(y0));
    RawRow n0 = fetch_row(y0 + 1), n1 = fetch_row(y0 + 2);
    for (int pair = pair_begin; pair < pair_end; pair += 2, y0 += 4) {
      const RowPrep rc = prep(n0), rd = prep(n1);
      n0 = fetch_row(y0 + 3);
      n1 = fetch_row(y0 + 4);
      consume(ra, rb, rc, rd, y0);
      if (pair + 1 >= pair_end) break;
      ra = prep(n0);
      rb = prep(n1);
      n0 = fetch_row(y0 + 5);
      n1 = fetch_row(y0 + 6);
      consume(rc, rd, ra, rb, y0 + 2);
    }
  }
  stat_flush(p, a, p.stats + frame, s_hist, frame);
}

// colour input (bgr8 / rgb8), 4 px per lane
__global__ __launch_bounds__(kBlock) void stats_color_kernel(StatsParams p, ItemMap im, int items_per_frame) {
  __shared__ unsigned s_hist[768];
  stat_hist_init(p, s_hist);
  const int frame = blockIdx.y;
  const uint8_t* src = p.src + (size_t)frame * p.src_frame_stride;
  const bool rgb = p.src_kind == SRC_RGB;
  StatAcc a = {};
  for (int item = blockIdx.x * kBlock + threadIdx.x; item < items_per_frame; item += gridDim.x * kBlock) {
    int y, grp;
    im.split(item, y, grp);
    const uint3 in = *reinterpret_cast<const uint3*>(src + (__umul24((unsigned)y, (unsigned)p.src_step) + (unsigned)grp * 12u));
    int q[4][3];
    unpack12(in, rgb, q);
#pragma unroll
    for (int k = 0; k < 4; k++) stat_add(p, q[k][0], q[k][1], q[k][2], a, s_hist);
  }
  stat_flush(p, a, p.stats + frame, s_hist, frame);
}

__global__ __launch_bounds__(kBlock) void stats_generic_kernel(StatsParams p) {
  __shared__ unsigned s_hist[768];
  stat_hist_init(p, s_hist);
  const int frame = blockIdx.y;
  SrcView s{p.src + (size_t)frame * p.src_frame_stride, p.src_step, p.rows, p.cols, p.src_kind, p.bayer_ry, p.bayer_rx};
  const long long npix = (long long)p.rows * p.cols;
  StatAcc a = {};
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < npix; i += (long long)gridDim.x * kBlock) {
    int y = (int)(i / p.cols), x = (int)(i - (long long)y * p.cols);
    int b, g, r;
    fetch_src(s, y, x, b, g, r);
    stat_add(p, b, g, r, a, s_hist);
  }
  stat_flush(p, a, p.stats + frame, s_hist, frame);
}

// ------------------------------------------------------------------------------------------------
// white-balance finalisation: statistics -> per-frame gains, on the device (no host round trip)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void solve2(float m00, float m01, float m10, float m11, float g0, float g1, float& o0, float& o1) {
  // Eigen::Matrix2f::inverse() * vec (white_balance.cpp:104-115)
  float det = m00 * m11 - m01 * m10;
  float invdet = 1.0f / det;
  float i00 = m11 * invdet, i01 = -m01 * invdet, i10 = -m10 * invdet, i11 = m00 * invdet;
  o0 = i00 * g0 + i01 * g1;
  o1 = i10 * g0 + i11 * g1;
}

__global__ void wb_finalize_kernel(int mode, const FrameStats* stats, const int* ccc_argmax, CccState* st,
                                   const DevTables* tabs, FrameWb* out, int n_frames, const unsigned* simple_hist,
                                   float simple_p, int simple_total) {
  if (mode == WB_FLOAT) {
    // ccc: temporal filter is sequential over the frames of the stream
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    CccState s = *st;
    for (int f = 0; f < n_frames; f++) {
      s.uv_x = ccc_argmax[2 * f];
      s.uv_y = ccc_argmax[2 * f + 1];
      FrameWb w = {};
      w.uv_raw[0] = s.uv_x;
      w.uv_raw[1] = s.uv_y;
      if (s.temporal) {
        if (s.first_frame) {
          s.first_frame = 0;
          s.st_x = (float)s.uv_x;
          s.st_y = (float)s.uv_y;
        } else {
          // cv::KalmanFilter(2,2,0) predict + correct with A = I, Q = I, H = h I, R = r I
          float xs[2] = {s.st_x, s.st_y}, ps[2] = {s.p_x, s.p_y};
          int z[2] = {s.uv_x, s.uv_y}, o[2];
          for (int a = 0; a < 2; a++) {
            float x_pre = xs[a];
            float p_pre = ps[a] + 1.0f;
            float t2 = s.kf_h * p_pre;
            float t3 = t2 * s.kf_h + s.kf_r;
            float k = t2 / t3;
            float innov = (float)z[a] - s.kf_h * x_pre;
            xs[a] = x_pre + k * innov;
            ps[a] = p_pre - k * t2;
            o[a] = (int)xs[a];
          }
          s.st_x = xs[0];
          s.st_y = xs[1];
          s.p_x = ps[0];
          s.p_y = ps[1];
          s.uv_x = o[0];
          s.uv_y = o[1];
        }
      }
      // computeGains (:342-381) with exp(-L) taken from the host-built table
      int ux = clampi(s.uv_x, 0, 255), uy = clampi(s.uv_y, 0, 255);
      float gain_r = 1.0f / tabs->exp_neg_tab[ux];
      float gain_g = 1.0f;
      float gain_b = 1.0f / tabs->exp_neg_tab[uy];
      float factor = fminf(fminf(gain_r, gain_g), gain_b);
      gain_r /= factor;
      gain_g /= factor;
      gain_b /= factor;
      w.fg[0] = gain_b;
      w.fg[1] = gain_g;
      w.fg[2] = gain_r;
      w.uv[0] = s.uv_x;
      w.uv[1] = s.uv_y;
      out[f] = w;
    }
    *st = s;
    return;
  }
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  FrameWb w = {};
  if (mode == WB_SIMPLE) {
    // cv::xphoto::SimpleWB (simple_color_balance.cpp balanceWhiteSimple<uchar>), restated literally:
    // two-level tree of 16-bin histograms whose second level of bin 0 aliases the first level
    const unsigned* h3 = simple_hist + (size_t)f * 768;
    for (int c = 0; c < 3; c++) {
      const unsigned* hist256 = h3 + c * 256;
      int hist[256];
      for (int i = 0; i < 256; i++) hist[i] = 0;
      for (int v = 0; v < 256; v++) {
        const int cnt = (int)hist256[v];
        if (!cnt) continue;
        int pos = 0;
        float minValue = 0.f - 0.5f;
        float interval = (255.5f - minValue) / 16;
        for (int j = 0; j < 2; ++j) {
          const int currentBin = (int)(((float)v - minValue + 1e-4f) / interval);
          hist[pos + currentBin] += cnt;
          pos = (pos + currentBin) * 16;
          minValue = minValue + currentBin * interval;
          interval /= 16;
        }
      }
      const float s1 = simple_p, s2 = simple_p;
      int p1 = 0, p2 = 15, n1 = 0, n2 = simple_total;
      float minValue = 0.f - 0.5f, maxValue = 255.f + 0.5f;
      float interval = (maxValue - minValue) / 16.0f;
      for (int j = 0; j < 2; ++j) {
        while (p1 < 255 && (float)(n1 + hist[p1]) < s1 * (float)simple_total / 100.0f) {
          n1 += hist[p1++];
          minValue += interval;
        }
        p1 *= 16;
        while (p2 > 0 && (float)(n2 - hist[p2]) > (100.0f - s2) * (float)simple_total / 100.0f) {
          n2 -= hist[p2--];
          maxValue -= interval;
        }
        p2 = (p2 + 1) * 16 - 1;
        interval /= 16;
        if (p1 > 255) p1 = 255;
        if (p2 > 255) p2 = 255;
      }
      const double d = (double)(maxValue - minValue);
      const double inv = 1.0 / d;
      w.fg[c] = (float)((1.0 * 255.0) * inv);
      w.pca[c] = (float)(((-(double)minValue) * 255.0) * inv + 0.0);
    }
    out[f] = w;
    return;
  }
  const FrameStats& fs = stats[f];
  if (mode == WB_Q8) {
    // GrayworldWBImpl::balanceWhite + applyChannelGains
    double sb = (double)fs.sum[0], sg = (double)fs.sum[1], sr = (double)fs.sum[2];
    double max_sum = fmax(sb, fmax(sr, sg));
    float gb = sb < 0.1 ? 0.f : (float)(max_sum / sb);
    float gg = sg < 0.1 ? 0.f : (float)(max_sum / sg);
    float gr = sr < 0.1 ? 0.f : (float)(max_sum / sr);
    float gmax = fmaxf(gb, fmaxf(gg, gr));
    if (gmax > 0) {
      gb /= gmax;
      gg /= gmax;
      gr /= gmax;
    }
    w.q8[0] = (int)__builtin_rintf(gb * 256.f);
    w.q8[1] = (int)__builtin_rintf(gg * 256.f);
    w.q8[2] = (int)__builtin_rintf(gr * 256.f);
    w.fg[0] = gb;
    w.fg[1] = gg;
    w.fg[2] = gr;
  } else if (mode == WB_PCA) {
    double s_b = (double)fs.sum[0], s_b2 = (double)fs.sum[1], s_r = (double)fs.sum[2], s_r2 = (double)fs.sum[3],
           s_g = (double)fs.sum[4];
    float mb = (float)fs.mx[0], mr = (float)fs.mx[1], mg = (float)fs.mx[2];
    float mb2 = mb * mb, mr2 = mr * mr;
    solve2((float)s_b2, (float)s_b, mb2, mb, (float)s_g, mg, w.pca[0], w.pca[1]);
    solve2((float)s_r2, (float)s_r, mr2, mr, (float)s_g, mg, w.pca[2], w.pca[3]);
  }
  out[f] = w;
}

// ------------------------------------------------------------------------------------------------
// ccc estimator: resize-sample -> log-chroma histogram -> FFT convolution -> argmax
// ------------------------------------------------------------------------------------------------
// colour of the post-flip image at (yd, xd)
__device__ __forceinline__ void fetch_flipped(const SrcView& s, int angle, int yd, int xd, int& b, int& g, int& r) {
  int ys, xs;
  unflip(angle, s.rows, s.cols, yd, xd, ys, xs);
  fetch_src(s, ys, xs, b, g, r);
}

__global__ __launch_bounds__(kBlock) void ccc_hist_kernel(CccParams p) {
  const int frame = blockIdx.y;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= 360 * 270) return;
  const int dy = i / 360, dx = i - dy * 360;
  SrcView s{p.src + (size_t)frame * p.src_frame_stride, p.src_step, p.rows, p.cols, p.src_kind, p.bayer_ry, p.bayer_rx};
  int sm[3];
  if (p.geom.area_fast) {
    int acc[3] = {2, 2, 2};
#pragma unroll
    for (int t = 0; t < 4; t++) {
      int b, g, r;
      fetch_flipped(s, p.flip_angle, 2 * dy + (t >> 1), 2 * dx + (t & 1), b, g, r);
      acc[0] += b;
      acc[1] += g;
      acc[2] += r;
    }
    sm[0] = acc[0] >> 2;
    sm[1] = acc[1] >> 2;
    sm[2] = acc[2] >> 2;
  } else {
    // cv::resize INTER_LINEAR, 8U: Q11 coefficients, two-pass integer arithmetic
    const int sx = p.geom.xofs[dx];
    const int sx1 = sx + 1 < p.dcols ? sx + 1 : sx;
    const int a0 = p.geom.ialpha[dx * 2], a1 = p.geom.ialpha[dx * 2 + 1];
    const int y0 = p.geom.yofs[dy * 2], y1 = p.geom.yofs[dy * 2 + 1];
    const int b0 = p.geom.ibeta[dy * 2], b1 = p.geom.ibeta[dy * 2 + 1];
    int p00[3], p01[3], p10[3], p11[3];
    fetch_flipped(s, p.flip_angle, y0, sx, p00[0], p00[1], p00[2]);
    fetch_flipped(s, p.flip_angle, y0, sx1, p01[0], p01[1], p01[2]);
    fetch_flipped(s, p.flip_angle, y1, sx, p10[0], p10[1], p10[2]);
    fetch_flipped(s, p.flip_angle, y1, sx1, p11[0], p11[1], p11[2]);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      int r0 = p00[c] * a0 + p01[c] * a1;
      int r1 = p10[c] * a0 + p11[c] * a1;
      sm[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    }
  }
  // calculateHistogramFeature (:210-271)
  float fb = (float)sm[0], fg = (float)sm[1], fr = (float)sm[2];
  float gray = fb * 0.114f + fg * 0.587f + fr * 0.299f;
  bool ok = !(gray > p.upper) && (gray > p.lower);
  if (sm[0] == 0 || sm[1] == 0 || sm[2] == 0) ok = false;  // log(0) = -inf is skipped
  if (!ok) return;
  const float bin_size = 1.0f / 64.0f, uv0 = -1.421875f;
  float lb = p.tabs->log_tab[sm[0]], lg = p.tabs->log_tab[sm[1]], lr = p.tabs->log_tab[sm[2]];
  int u = (int)roundf((lg - lr - uv0) / bin_size);
  int v = (int)roundf((lg - lb - uv0) / bin_size);
  u = clampi(u, 0, 255);
  v = clampi(v, 0, 255);
  atomicAdd(&p.hist_counts[(size_t)frame * 65536 + u * 256 + v], 1u);
}

// 256-point radix-2 DIT FFT in LDS, 128 threads, same butterfly order as the host reference
// (rip_host.cpp host_fft256).  re/im hold bit-reversed input on entry.
__device__ __forceinline__ unsigned bitrev8(unsigned x) { return __brev(x) >> 24; }

__device__ __forceinline__ void fft256_lds(float* re, float* im, const float* twr, const float* twi, bool inverse) {
  const int b = threadIdx.x;  // butterfly index 0..127
  for (int len = 2; len <= 256; len <<= 1) {
    const int half = len >> 1, tstep = 256 / len;
    const int k = b & (half - 1), lo = (b / half) * len + k, hi = lo + half;
    const float wr = twr[k * tstep], wi = inverse ? -twi[k * tstep] : twi[k * tstep];
    const float xr = re[hi], xi = im[hi];
    const float tr = wr * xr - wi * xi;
    const float ti = wr * xi + wi * xr;
    const float ur = re[lo], ui = im[lo];
    __syncthreads();
    re[lo] = ur + tr;
    im[lo] = ui + ti;
    re[hi] = ur - tr;
    im[hi] = ui - ti;
    __syncthreads();
  }
}

// forward FFT of the histogram rows (counts -> float via the sequential-accumulation table)
__global__ __launch_bounds__(128) void ccc_fft_rows_kernel(CccParams p) {
  __shared__ float re[256], im[256], twr[128], twi[128];
  const int row = blockIdx.x, frame = blockIdx.y, t = threadIdx.x;
  twr[t] = p.tabs->tw_re[t];
  twi[t] = p.tabs->tw_im[t];
  const unsigned int* h = p.hist_counts + (size_t)frame * 65536 + row * 256;
  for (int i = t; i < 256; i += 128) {
    unsigned j = bitrev8((unsigned)i);
    re[j] = p.accum_tab[h[i]];
    im[j] = 0.f;
  }
  __syncthreads();
  fft256_lds(re, im, twr, twi, false);
  float2* out = reinterpret_cast<float2*>(p.work) + (size_t)frame * 65536 + row * 256;
  for (int i = t; i < 256; i += 128) out[i] = make_float2(re[i], im[i]);
}

// forward FFT of a column, spectrum product + bias, inverse FFT of the column
__global__ __launch_bounds__(128) void ccc_fft_cols_kernel(CccParams p) {
  __shared__ float re[256], im[256], twr[128], twi[128], tr[256], ti[256];
  const int col = blockIdx.x, frame = blockIdx.y, t = threadIdx.x;
  twr[t] = p.tabs->tw_re[t];
  twi[t] = p.tabs->tw_im[t];
  float2* data = reinterpret_cast<float2*>(p.work) + (size_t)frame * 65536 + col;
  for (int i = t; i < 256; i += 128) {
    float2 v = data[(size_t)i * 256];
    unsigned j = bitrev8((unsigned)i);
    re[j] = v.x;
    im[j] = v.y;
  }
  __syncthreads();
  fft256_lds(re, im, twr, twi, false);
  const float2* F = reinterpret_cast<const float2*>(p.filter_fft) + col;
  const float2* B = reinterpret_cast<const float2*>(p.bias_fft) + col;
  for (int i = t; i < 256; i += 128) {
    float2 f = F[(size_t)i * 256], bb = B[(size_t)i * 256];
    float ar = f.x, ai = f.y, br = re[i], bi = im[i];
    float pr = ar * br - ai * bi;  // mulSpectrums, no conjugation
    float pi = ar * bi + ai * br;
    unsigned j = bitrev8((unsigned)i);
    tr[j] = pr + bb.x;
    ti[j] = pi + bb.y;
  }
  __syncthreads();
  fft256_lds(tr, ti, twr, twi, true);
  for (int i = t; i < 256; i += 128) data[(size_t)i * 256] = make_float2(tr[i], ti[i]);
}

// inverse FFT of the rows; per-row first maximum of the real part
__global__ __launch_bounds__(128) void ccc_ifft_rows_kernel(CccParams p) {
  __shared__ float re[256], im[256], twr[128], twi[128];
  __shared__ float bv[128];
  __shared__ int bi[128];
  const int row = blockIdx.x, frame = blockIdx.y, t = threadIdx.x;
  twr[t] = p.tabs->tw_re[t];
  twi[t] = p.tabs->tw_im[t];
  const float2* in = reinterpret_cast<const float2*>(p.work) + (size_t)frame * 65536 + row * 256;
  for (int i = t; i < 256; i += 128) {
    float2 v = in[i];
    unsigned j = bitrev8((unsigned)i);
    re[j] = v.x;
    im[j] = v.y;
  }
  __syncthreads();
  fft256_lds(re, im, twr, twi, true);
  float v0 = re[t], v1 = re[t + 128];
  bv[t] = v1 > v0 ? v1 : v0;
  bi[t] = v1 > v0 ? t + 128 : t;
  __syncthreads();
  for (int off = 64; off > 0; off >>= 1) {
    if (t < off) {
      float ov = bv[t + off];
      int oi = bi[t + off];
      if (ov > bv[t] || (ov == bv[t] && oi < bi[t])) {
        bv[t] = ov;
        bi[t] = oi;
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    p.row_best[((size_t)frame * 256 + row) * 2] = bv[0];
    p.row_best[((size_t)frame * 256 + row) * 2 + 1] = (float)bi[0];
  }
}

// cv::minMaxLoc: first maximum in row-major order -> Point(x = column, y = row)
__global__ __launch_bounds__(256) void ccc_argmax_kernel(CccParams p) {
  __shared__ float bv[256];
  __shared__ int br[256];
  const int frame = blockIdx.x, t = threadIdx.x;
  bv[t] = p.row_best[((size_t)frame * 256 + t) * 2];
  br[t] = t;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) {
      float ov = bv[t + off];
      int orow = br[t + off];
      if (ov > bv[t] || (ov == bv[t] && orow < br[t])) {
        bv[t] = ov;
        br[t] = orow;
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    int row = br[0];
    p.argmax[2 * frame] = (int)p.row_best[((size_t)frame * 256 + row) * 2 + 1];
    p.argmax[2 * frame + 1] = row;
  }
}

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
__device__ __forceinline__ void lds_load6(const uint8_t* lds, unsigned a, uint32_t& lo, uint32_t& hi) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(lds + (a & ~3u));
  const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
  lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
  hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
}

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
__global__ __launch_bounds__(kBlock) void remap_tiled_kernel(RemapTiledParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int kPre = PRE > 0 ? PRE : 1;
  const RemapParams& b = p.base;
  const int ntiles = p.tiles_x * p.tiles_y;
  const int per_xcd = (ntiles + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int tid = threadIdx.x;
  const int lrow = tid >> 4, lgrp = tid & 15;
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
    const uint4 wd = reinterpret_cast<const uint4*>(p.words + (size_t)tile * 1024)[tid];
    const uint32_t words[4] = {wd.x, wd.y, wd.z, wd.w};
    const int yd = ty * 16 + lrow, xd = tx * 64 + lgrp * 4;
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
        q[k][0] = (int)((__umul24(topB, wy0) + __umul24(botB, wy1)) >> 10);
        q[k][1] = (int)((__umul24(topG, wy0) + __umul24(botG, wy1)) >> 10);
        q[k][2] = (int)((__umul24(topR, wy0) + __umul24(botR, wy1)) >> 10);
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
        const unsigned i = (unsigned)tid + (unsigned)j * kBlock;
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
        for (unsigned i = tid; i < total; i += kBlock) {
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

template <int PRE>
__global__ __launch_bounds__(kBlock) void remap_ring_kernel(RemapTiledParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr unsigned kStage = (unsigned)PRE * kBlock * 16u;  // bytes per stage
  const RemapParams& b = p.base;
  const int ntiles = p.tiles_x * p.tiles_y;
  const int per_xcd = (ntiles + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int tid = threadIdx.x;
  const int lrow = tid >> 4, lgrp = tid & 15;
  const unsigned step = (unsigned)b.src_step;
  const int f_per_group = (b.n_frames + (int)gridDim.y - 1) / (int)gridDim.y;
  const int f_begin = (int)blockIdx.y * f_per_group, f_end = min(b.n_frames, f_begin + f_per_group);
  if (f_begin >= f_end) return;  // uniform for the workgroup
  const int nb = p.stages, dist = nb - 1;  // ring size, prefetch distance (2 or 3)
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);
  const unsigned wave_chunk0 = (unsigned)__builtin_amdgcn_readfirstlane(tid & ~63);
  const unsigned dst_bytes = __umul24((unsigned)(b.drows - 1), (unsigned)b.dst_step) + (unsigned)b.dcols * 3u;
  for (int ti = blockIdx.x >> 3; ti < per_xcd; ti += gridDim.x >> 3) {
    const int tile = xcd * per_xcd + ti;
    if (tile >= ntiles) break;
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const RemapTileDesc d = p.tiles[tile];
    const uint4 wd = reinterpret_cast<const uint4*>(p.words + (size_t)tile * 1024)[tid];
    const uint32_t words[4] = {wd.x, wd.y, wd.z, wd.w};
    const int yd = ty * 16 + lrow, xd = tx * 64 + lgrp * 4;
    const bool in_image = yd < b.drows && xd < b.dcols;
    const unsigned xbyte0 = (unsigned)d.x0 * 3u;
    const unsigned chunk0 = xbyte0 & ~15u, ph = xbyte0 & 15u;
    const unsigned pitch = (ph + (unsigned)d.w * 3u + 15u) & ~15u;  // rip_host.cpp remap_tile_lds_bytes
    const unsigned chunks = pitch >> 4;
    const unsigned total = d.w > 0 ? chunks * (unsigned)d.h : 0u;
    const ItemMap cm{(int)chunks, 1.0f / (float)(chunks ? chunks : 1u)};
    // frame-invariant: source offsets of this lane's chunks, LDS address of the top-left tap, weights.
    // Outside / border pixels gather address 0 with zero weights: 16 * 32 >> 10 == 0 is the border
    // constant, and remap_border_kernel patches the border pixels afterwards -- no branch per pixel.
    unsigned goff[PRE];
#pragma unroll
    for (int j = 0; j < PRE; j++) {
      const unsigned i = (unsigned)tid + (unsigned)j * kBlock;
      int r, c;
      cm.split((int)i, r, c);
      goff[j] = i < total ? __umul24((unsigned)(d.y0 + r), step) + chunk0 + ((unsigned)c << 4) : 0xFFFFFFF0u;
    }
    unsigned tap_addr[4], wxb[4], wyy[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t w = words[k];
      const bool live = w < kPlanBorder;
      const unsigned relx = w & 0x7ffu, rely = (w >> 11) & 0x7ffu, fx = (w >> 22) & 31u, fy = w >> 27;
      tap_addr[k] = live ? __umul24(rely, pitch) + relx * 3u + ph : 0u;
      wxb[k] = live ? (32u - fx) | (fx << 24) : 0u;
      wyy[k] = (32u - fy) | (fy << 16);
    }
    const unsigned dst_off = __umul24((unsigned)yd, (unsigned)b.dst_step) + (unsigned)xd * 3u;

    auto issue = [&](int f, int slot) {
      const RemapSrc s = remap_src(b, f);
      const __amdgpu_buffer_rsrc_t rsrc = frame_rsrc(s.frame, s.readable);
      const unsigned stage = lds0 + (unsigned)slot * kStage;
#pragma unroll
      for (int j = 0; j < PRE; j++) lds_dma16(rsrc, goff[j], stage + ((wave_chunk0 + (unsigned)j * kBlock) << 4));
    };
    auto gather_store = [&](const uint8_t* buf, int f) {
      if (!in_image) return;
      uint32_t t0[4], t1[4], b0[4], b1[4];  // bytes b0 g0 r0 b1 | g1 r1 . .  of the top / bottom tap rows
#pragma unroll
      for (int k = 0; k < 4; k++) {
        lds_load6(buf, tap_addr[k], t0[k], t1[k]);
        lds_load6(buf, tap_addr[k] + pitch, b0[k], b1[k]);
      }
      int q[4][3];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const unsigned wy0 = wyy[k] & 0xffffu, wy1 = wyy[k] >> 16;
        const unsigned wB = wxb[k], wx0 = wB & 0xffu, wx1 = wB >> 24;
        const unsigned wG0 = wx0 << 8, wR0 = wx0 << 16, wR1 = wx1 << 8;
        // every row sum starts at 16: 16 * (32 - fy) + 16 * fy = 512 is the rounding term of
        // ((top*(32-fy) + bot*fy) * 32 + 2^14) >> 15, exact
        const unsigned topB = __builtin_amdgcn_udot4(t0[k], wB, 16u, false);
        const unsigned topG = __builtin_amdgcn_udot4(t0[k], wG0, __builtin_amdgcn_udot4(t1[k], wx1, 16u, false), false);
        const unsigned topR = __builtin_amdgcn_udot4(t0[k], wR0, __builtin_amdgcn_udot4(t1[k], wR1, 16u, false), false);
        const unsigned botB = __builtin_amdgcn_udot4(b0[k], wB, 16u, false);
        const unsigned botG = __builtin_amdgcn_udot4(b0[k], wG0, __builtin_amdgcn_udot4(b1[k], wx1, 16u, false), false);
        const unsigned botR = __builtin_amdgcn_udot4(b0[k], wR0, __builtin_amdgcn_udot4(b1[k], wR1, 16u, false), false);
        q[k][0] = (int)((__umul24(topB, wy0) + __umul24(botB, wy1)) >> 10);
        q[k][1] = (int)((__umul24(topG, wy0) + __umul24(botG, wy1)) >> 10);
        q[k][2] = (int)((__umul24(topR, wy0) + __umul24(botR, wy1)) >> 10);
      }
      store12(frame_rsrc(b.dst + (size_t)f * b.dst_frame_stride, dst_bytes), dst_off, pack4(q));
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
      if (ahead >= 2)
        wait_vmcnt<2 * PRE>();
      else if (ahead == 1)
        wait_vmcnt<PRE>();
      else
        wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
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

// grid-size tunables (persistent workgroups per launch), overridable from the environment for experiments
int tune_env(const char* name, int dflt) {
  const char* e = std::getenv(name);
  if (!e || !*e) return dflt;
  const int v = std::atoi(e);
  return v >= 8 ? v / 8 * 8 : (v > 0 ? v : dflt);
}

template <int BITS, int WB>
void launch_fast(const ChainParams& p, const ItemMap& im, int items, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL((chain_fast_kernel<BITS, WB>), grid, dim3(kBlock), 0, stream, p, im, items);
}

template <int BITS>
void launch_fast_wb(const ChainParams& p, const ItemMap& im, int items, dim3 grid, hipStream_t stream) {
  switch (p.wb_mode) {
    case WB_Q8: launch_fast<BITS, WB_Q8>(p, im, items, grid, stream); break;
    case WB_FLOAT: launch_fast<BITS, WB_FLOAT>(p, im, items, grid, stream); break;
    case WB_PCA: launch_fast<BITS, WB_PCA>(p, im, items, grid, stream); break;
    case WB_SIMPLE: launch_fast<BITS, WB_SIMPLE>(p, im, items, grid, stream); break;
    default: launch_fast<BITS, WB_NONE>(p, im, items, grid, stream); break;
  }
}

}  // namespace

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

bool chain_uses_rot_path(const ChainParams& p) {
  return bayer_fast_geometry(p.src, p.src_step, p.src_frame_stride, p.rows, p.cols, p.src_kind) &&
         (p.flip_angle == 90 || p.flip_angle == 270) && p.channels == 3 && p.drows == p.cols && p.dcols == p.rows &&
         p.dst_step % 2 == 0 && p.dst_step < (1u << 24) && p.cols < (1 << 23) &&
         (unsigned long long)p.dst_step * (unsigned long long)p.drows < (1ull << 32) && p.dst_frame_stride % 2 == 0 &&
         (reinterpret_cast<uintptr_t>(p.dst) & 1u) == 0 &&
         (!p.tap || ((reinterpret_cast<uintptr_t>(p.tap) & 1u) == 0 && p.tap_frame_stride % 2 == 0));
}

void launch_chain(const ChainParams& p, hipStream_t stream) {
  if (p.n_frames <= 0) return;
  if (chain_uses_rot_path(p)) {
    const int tiles_x = (p.cols / 4 + 3) / 4, tiles_y = (p.rows / 2 + 63) / 64;
    const int tiles = tiles_x * tiles_y;
    const int cap = tune_env("RIP_CHAIN_BLOCKS", 2048);
    const int blocks = std::min(cap, tiles);
    const int groups = std::max(1, std::min(p.n_frames, cap / blocks));
    hipLaunchKernelGGL(chain_rot_kernel, dim3(blocks, groups), dim3(kBlock), 0, stream, p, tiles_x, tiles);
    return;
  }
  if (chain_uses_fast_path(p)) {
    ItemMap im{p.cols / 4, 1.0f / (float)(p.cols / 4)};
    const int items = (p.rows / 2) * (p.cols / 4);
    const long long chunks = (long long)((items + kBlock - 1) / kBlock);
    // persistent grid: at most 256 CUs x 8 workgroups, a multiple of 8 (one share per XCD)
    const int cap = tune_env("RIP_CHAIN_BLOCKS", 2048);
    int blocks = (int)std::min<long long>(cap, (chunks + 7) / 8 * 8);
    const int groups = std::max(1, std::min(p.n_frames, cap / blocks));
    dim3 grid(blocks, groups);
    switch (p.stage_bits & 15) {
#define RIP_CASE(B) case B: launch_fast_wb<B>(p, im, items, grid, stream); break;
      RIP_CASE(0) RIP_CASE(1) RIP_CASE(2) RIP_CASE(3) RIP_CASE(4) RIP_CASE(5) RIP_CASE(6) RIP_CASE(7)
      RIP_CASE(8) RIP_CASE(9) RIP_CASE(10) RIP_CASE(11) RIP_CASE(12) RIP_CASE(13) RIP_CASE(14) RIP_CASE(15)
#undef RIP_CASE
    }
    return;
  }
  if (chain_uses_color_path(p)) {
    ItemMap im{p.cols / 4, 1.0f / (float)(p.cols / 4)};
    const int items = p.rows * (p.cols / 4);
    const int chunks = (items + kBlock - 1) / kBlock;
    const int blocks = std::min(2048, chunks);
    const int groups = std::max(1, std::min(p.n_frames, 2048 / blocks));
    hipLaunchKernelGGL(chain_color_kernel, dim3(blocks, groups), dim3(kBlock), 0, stream, p, im, items);
    return;
  }
  long long npix = (long long)p.drows * p.dcols;
  dim3 grid(grid_blocks_for(npix, 2048), p.n_frames);
  hipLaunchKernelGGL(chain_generic_kernel, grid, dim3(kBlock), 0, stream, p);
}

void launch_stats(const StatsParams& p, hipStream_t stream) {
  if (p.n_frames <= 0) return;
  if (bayer_fast_geometry(p.src, p.src_step, p.src_frame_stride, p.rows, p.cols, p.src_kind)) {
    ItemMap im{p.cols / 4, 1.0f / (float)(p.cols / 4)};
    const int items = (p.rows / 2) * (p.cols / 4);
    // keep >= 1 block per 2^20 items so the 32-bit per-thread partial sums cannot overflow
    // wave tasks: 64 groups wide x pairs_per_task row pairs.  The wave total of the largest statistic
    // (pca: sum of squares) is 64 lanes * 8 px * 255^2 * pairs_per_task: 128 pairs keep it below 2^32
    const int groups = p.cols / 4, n_pairs = p.rows / 2;
    const int col_waves = (groups + 63) / 64;
    // per frame: 512 wave tasks when the batch fills the chip anyway, at most 1024 for a single frame (more
    // tasks only queue up on the three 64-bit atomics every workgroup ends with: 22 -> 12.6 us for one frame)
    const int budget = tune_env("RIP_STATS_BLOCKS", 2048) * 4;
    const int target_tasks = std::max(8, std::min(budget / 8, budget / std::max(1, std::min(p.n_frames, 16))));
    int pairs_per_task = std::max(2, (int)(((long long)col_waves * n_pairs + target_tasks - 1) / target_tasks));
    pairs_per_task = std::min((pairs_per_task + 1) & ~1, 128);  // even: the kernel consumes two pairs per iteration
    const int n_tasks = col_waves * ((n_pairs + pairs_per_task - 1) / pairs_per_task);
    const dim3 grid((n_tasks + kBlock / 64 - 1) / (kBlock / 64), p.n_frames);
    (void)items;
    (void)im;
    if (p.mode == WB_Q8)
      hipLaunchKernelGGL(stats_fast_kernel<WB_Q8>, grid, dim3(kBlock), 0, stream, p, col_waves, pairs_per_task, n_tasks);
    else if (p.mode == WB_SIMPLE)
      hipLaunchKernelGGL(stats_fast_kernel<WB_SIMPLE>, grid, dim3(kBlock), 0, stream, p, col_waves, pairs_per_task, n_tasks);
    else
      hipLaunchKernelGGL(stats_fast_kernel<WB_PCA>, grid, dim3(kBlock), 0, stream, p, col_waves, pairs_per_task, n_tasks);
    return;
  }
  if (color_fast_geometry(p.src, p.src_step, p.src_frame_stride, p.rows, p.cols, p.src_kind)) {
    ItemMap im{p.cols / 4, 1.0f / (float)(p.cols / 4)};
    const int items = p.rows * (p.cols / 4);
    int per_frame = grid_blocks_for(items, std::max(8, 2048 / std::max(1, std::min(p.n_frames, 16))));
    per_frame = std::max(per_frame, (int)((items + (1 << 20) - 1) >> 20));
    hipLaunchKernelGGL(stats_color_kernel, dim3(per_frame, p.n_frames), dim3(kBlock), 0, stream, p, im, items);
    return;
  }
  long long npix = (long long)p.rows * p.cols;
  int blocks = std::max(grid_blocks_for(npix, 1024), (int)((npix + (1 << 22) - 1) >> 22));
  hipLaunchKernelGGL(stats_generic_kernel, dim3(blocks, p.n_frames), dim3(kBlock), 0, stream, p);
}

void launch_ccc_estimate(const CccParams& p, hipStream_t stream) {
  if (p.n_frames <= 0) return;
  hipLaunchKernelGGL(ccc_hist_kernel, dim3((360 * 270 + kBlock - 1) / kBlock, p.n_frames), dim3(kBlock), 0, stream, p);
  hipLaunchKernelGGL(ccc_fft_rows_kernel, dim3(256, p.n_frames), dim3(128), 0, stream, p);
  hipLaunchKernelGGL(ccc_fft_cols_kernel, dim3(256, p.n_frames), dim3(128), 0, stream, p);
  hipLaunchKernelGGL(ccc_ifft_rows_kernel, dim3(256, p.n_frames), dim3(128), 0, stream, p);
  hipLaunchKernelGGL(ccc_argmax_kernel, dim3(p.n_frames), dim3(256), 0, stream, p);
}

void launch_wb_finalize(int mode, const FrameStats* stats, const int* ccc_argmax, CccState* ccc_state,
                        const DevTables* tabs, FrameWb* out, int n_frames, hipStream_t stream, const unsigned* simple_hist,
                        float simple_p, int simple_total) {
  if (n_frames <= 0) return;
  if (mode == WB_FLOAT) {
    hipLaunchKernelGGL(wb_finalize_kernel, dim3(1), dim3(64), 0, stream, mode, stats, ccc_argmax, ccc_state, tabs, out, n_frames,
                       simple_hist, simple_p, simple_total);
  } else {
    hipLaunchKernelGGL(wb_finalize_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, stream, mode, stats, ccc_argmax,
                       ccc_state, tabs, out, n_frames, simple_hist, simple_p, simple_total);
  }
}

bool launch_remap_tiled(const RemapTiledParams& p, hipStream_t stream) {
  const RemapParams& b = p.base;
  if (b.n_frames <= 0) return true;
  const bool ok = b.channels == 3 && b.dcols % 4 == 0 && b.dst_step % 4 == 0 && b.dst_frame_stride % 4 == 0 && aligned4(b.dst) &&
                  b.src_step % 16 == 0 && b.src_frame_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(b.src) & 15u) == 0 &&
                  (reinterpret_cast<uintptr_t>(p.words) & 15u) == 0 && b.src_step < (1u << 24) && b.rows < (1 << 23) &&
                  (unsigned long long)b.src_step * (unsigned long long)b.rows < (1ull << 32) && b.dst_step < (1u << 24) &&
                  (unsigned long long)b.dst_step * (unsigned long long)b.drows < (1ull << 32) && p.lds_bytes <= 64u * 1024u &&
                  p.tiles_x * 64 >= b.dcols && p.tiles_y * 16 >= b.drows && b.drows <= 65535 && b.dcols <= 65535;
  if (!ok) return false;
  const int ntiles = p.tiles_x * p.tiles_y;
  // persistent workgroups, a multiple of 8 (one share of the tile range per XCD); LDS bounds residency
  RemapTiledParams q = p;
  q.lds_bytes = (std::max(p.lds_bytes, 16u) + 15u) & ~15u;
  const unsigned chunks = q.lds_bytes / 16u;  // upper bound of the 16-byte chunks of any tile
  const int ring_env = std::getenv("RIP_REMAP_RING") ? std::atoi(std::getenv("RIP_REMAP_RING")) : 1;
  // measured on config2 (sweeps in DESIGN.md): 3 stages (two frames ahead) with 4 workgroups per CU; more resident
  // workgroups fetch more (the source rectangles of neighbouring tiles stop meeting in L2) and run slower
  const int stages_env = tune_env("RIP_REMAP_STAGES", 3);
  if (ring_env && chunks <= 4u * kBlock) {
    // LDS-DMA ring: PRE chunks per lane and frame, `stages` buffers of PRE * 4 KiB
    const int pre = chunks <= 1u * kBlock ? 1 : (chunks <= 2u * kBlock ? 2 : 4);
    const unsigned stage_bytes = (unsigned)pre * kBlock * 16u;
    q.stages = std::max(2, std::min(4, stages_env));
    const unsigned lds = (unsigned)q.stages * stage_bytes + 16u;  // the three-dword tap reads run up to 11 B past a row
    const int per_cu = std::max(1, std::min(tune_env("RIP_REMAP_PER_CU", 4), (int)((160u * 1024u) / (lds + 256u))));
    int blocks = std::min(256 * per_cu, (ntiles + 7) / 8 * 8);
    blocks = std::max(8, blocks / 8 * 8);
    const int groups = std::max(1, std::min(b.n_frames, (256 * per_cu) / blocks));  // few tiles: split the batch too
    const dim3 grid(blocks, groups);
    if (pre == 1)
      hipLaunchKernelGGL(remap_ring_kernel<1>, grid, dim3(kBlock), lds, stream, q);
    else if (pre == 2)
      hipLaunchKernelGGL(remap_ring_kernel<2>, grid, dim3(kBlock), lds, stream, q);
    else
      hipLaunchKernelGGL(remap_ring_kernel<4>, grid, dim3(kBlock), lds, stream, q);
  } else {
    // rectangles larger than 4 * kBlock chunks (strong local magnification) or RIP_REMAP_RING=0 (A/B runs)
    int pre = !ring_env && b.n_frames >= 2 && chunks <= 2u * kBlock ? 2 : 0;
    q.double_buffer = pre > 0 ? 1 : 0;
    const unsigned lds = (q.double_buffer ? 2u * q.lds_bytes : q.lds_bytes) + 16u;
    const int per_cu = std::max(1, std::min(tune_env("RIP_REMAP_PER_CU", 8), (int)((160u * 1024u) / (lds + 256u))));
    int blocks = std::min(256 * per_cu, (ntiles + 7) / 8 * 8);
    blocks = std::max(8, blocks / 8 * 8);
    const int groups = std::max(1, std::min(b.n_frames, (256 * per_cu) / blocks));  // few tiles: split the batch too
    const dim3 grid(blocks, groups);
    if (pre == 2)
      hipLaunchKernelGGL(remap_tiled_kernel<2>, grid, dim3(kBlock), lds, stream, q);
    else
      hipLaunchKernelGGL(remap_tiled_kernel<0>, grid, dim3(kBlock), lds, stream, q);
  }
  if (q.n_border > 0)
    hipLaunchKernelGGL(remap_border_kernel, dim3((q.n_border + kBlock - 1) / kBlock, b.n_frames), dim3(kBlock), 0, stream, q);
  return true;
}

void launch_remap(const RemapParams& p, hipStream_t stream) {
  if (p.n_frames <= 0) return;
  // remap_pixel addresses a source frame with 32-bit offsets and 24-bit multiplies
  if (p.src_step >= (1u << 24) || p.rows >= (1 << 23) || (unsigned long long)p.src_step * (unsigned long long)p.rows >= (1ull << 32) ||
      p.dst_step >= (1u << 24) || (unsigned long long)p.dst_step * (unsigned long long)p.drows >= (1ull << 32))
    return;  // rejected by the API layer before we get here (see rip_api.cpp make_plan)
  const bool vec = p.channels == 3 && p.dcols % 4 == 0 && p.dst_step % 4 == 0 && p.dst_frame_stride % 4 == 0 &&
                   aligned4(p.dst) && (reinterpret_cast<uintptr_t>(p.map_xy) & 15u) == 0;
  if (vec) {
    ItemMap im{p.dcols / 4, 1.0f / (float)(p.dcols / 4)};
    const int items = p.drows * (p.dcols / 4);
    int per_frame = grid_blocks_for(items, std::max(8, 8192 / std::max(1, std::min(p.n_frames, 16))));
    hipLaunchKernelGGL(remap_vec4_kernel, dim3(per_frame, p.n_frames), dim3(kBlock), 0, stream, p, im, items);
    return;
  }
  long long npix = (long long)p.drows * p.dcols;
  dim3 grid(grid_blocks_for(npix, 4096), p.n_frames);
  if (p.channels == 3)
    hipLaunchKernelGGL(remap_generic_kernel<3>, grid, dim3(kBlock), 0, stream, p);
  else
    hipLaunchKernelGGL(remap_generic_kernel<1>, grid, dim3(kBlock), 0, stream, p);
}

}  // namespace rip
