// rip_chain_dev.hpp -- device code the fused per-pixel chain shares between its kernels (rip_chain.hip) and the kernel that
// runs the chain inside the remap's tiles (rip_fused.hip): the LDS tables of a compile-time stage set and the per-pixel
// stages of four pixels of one row.  Stage functions and reference citations: rip_device.hpp.
#pragma once
#include "rip_device.hpp"

namespace rip {
namespace {

// Tables of the fast kernel in LDS: only what the compile-time stage set reads.
template <bool ON, typename T>
struct OptTab {
  T v;
};
template <typename T>
struct OptTab<false, T> {};
// 256-byte aligned in LDS: v_cvt_pk_u8_f32 writes its byte into byte 0 of a dword that already holds the table's address, so the
// conversion that ends the colour matrix is also the address of the look-up (pointwise4)
#ifndef RIP_GAMMA_FOLD
#define RIP_GAMMA_FOLD 0  // rip_chain.hip sets it; inside the remap's tiles (rip_fused.hip) the plain look-ups stay
#endif
struct alignas(RIP_GAMMA_FOLD ? 256 : 4) GammaTab {
  uint8_t lut[256];
  __device__ __forceinline__ unsigned lds_address() const { return (unsigned)reinterpret_cast<uintptr_t>(&lut[0]); }
};
// Round 6: everything of cv::multiply(hsv, gains) + HSV2RGB_f that depends on ONE 8-bit channel alone is tabulated per launch
// (the gains are launch constants): hs[s] = float(sat_u8(float(s) gain_s)) (1 / 255), hv[v] likewise, hw[h] = the three sector
// weights w_b, w_g, w_r of the hue after its gain and wrap -- the same float operations apply_hsv_f spells out per pixel, done
// once per table entry, so every pixel's bytes are unchanged (tests: test_color_enhancer*, the fuzz).  Per pixel that replaces
// 3 int -> float conversions, up to 3 gain multiplies + saturating conversions + conversions back, 3 scale multiplies, 6
// subtractions and 3 clamped min / max (18-24 VALU instructions of 66) by three LDS reads.
struct HsvTab {
  int32_t sdiv[256], hdiv[256];
  float hs[256], hv[256];
  float4 hw[256];
};
template <int BITS>
struct FastTabs {
  static constexpr bool kVig = (BITS & ST_VIG) != 0;
  static constexpr bool kHsv = (BITS & ST_HSV) != 0;
  static constexpr bool kGamma = (BITS & ST_GAMMA) != 0 && !kVig;  // with vignetting the LUT is folded into VigTabs::lin
  OptTab<kVig, VigTabs> vig;
  OptTab<kGamma, GammaTab> gam;
  OptTab<kHsv, HsvTab> hsv;
  __device__ __forceinline__ int sdiv(int i) const {
    if constexpr (kHsv) return hsv.v.sdiv[i];
    return 0;
  }
  __device__ __forceinline__ int hdiv(int i) const {
    if constexpr (kHsv) return hsv.v.hdiv[i];
    return 0;
  }
  template <int NT>
  __device__ __forceinline__ void load(const DevTables* t, const uint32_t* vig_image, const float* hsv_gain = nullptr) {
    if constexpr (kVig) {
      if (vig_image)
        vig.v.template load_image<NT>(vig_image);
      else
        vig.v.template load<NT>(t);
    }
    if constexpr (kGamma)
      for (int i = threadIdx.x; i < 64; i += NT) reinterpret_cast<uint32_t*>(gam.v.lut)[i] = reinterpret_cast<const uint32_t*>(t->gamma_lut)[i];
    if constexpr (kHsv)
      for (int i = threadIdx.x; i < 256; i += NT) {
        hsv.v.sdiv[i] = t->sdiv[i];
        hsv.v.hdiv[i] = t->hdiv[i];
        if (hsv_gain) hsv_tables_entry(hsv_gain, i, hsv.v.hs[i], hsv.v.hv[i], hsv.v.hw[i]);
      }
  }
};
// workgroup size of the fast kernel: the Lab tables are 33 KB, so the vignetting variants share them among
// 8 waves (3 workgroups = 24 waves per CU); the others keep 256 threads
constexpr int kVigThreads = 512;
constexpr int kVigWavesPerSimd = 6;  // waves per SIMD the register allocation must allow: 3 workgroups of 512 threads per CU
template <int BITS>
constexpr int fast_threads() {
  return (BITS & ST_VIG) ? kVigThreads : 256;
}
template <int BITS>
constexpr int fast_waves_per_simd() {
  // vignetting + enhancer: 56 KB of tables, two workgroups per CU
  return (BITS & ST_VIG) ? ((BITS & ST_HSV) ? 4 : kVigWavesPerSimd) : 1;
}

// RGB2HSV_b (integer, as apply_hsv_f) + the tabulated rest of the colour enhancer (HsvTab)
__device__ __forceinline__ void apply_hsv_tab(const HsvTab& t, int b, int g, int r, float (&out)[3]) {
  const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
  const int diff = v - vmin;
  const int s = (mul24(diff, t.sdiv[v]) + (1 << 11)) >> 12;
  int h;
  if (v == r)
    h = g - b;
  else if (v == g)
    h = b - r + 2 * diff;
  else
    h = r - g + 4 * diff;
  h = (mul24(h, t.hdiv[diff]) + (1 << 11)) >> 12;
  h += h < 0 ? 180 : 0;  // [0, 179]: apply_hsv_f
  const float fs = t.hs[s], fv = t.hv[v];
  const float4 w = t.hw[h];
  // contracted model: 1 - s * w is one fnma (apply_hsv_f)
  out[0] = (fv * mul_add(-fs, w.x, 1.f)) * 255.f;
  out[1] = (fv * mul_add(-fs, w.y, 1.f)) * 255.f;
  out[2] = (fv * mul_add(-fs, w.z, 1.f)) * 255.f;
}

// The per-pixel stages after the demosaic for the four pixels of one row; returns the 12 interleaved output bytes.
// BIAS: see apply_cc_f
template <int BITS, int WB, int BIAS = 1>
__device__ __forceinline__ Pack3 pointwise4(const ChainParams& p, const FrameWb& w, const FastTabs<BITS>& tb, const CcRegs& cc,
                                            const HsvRegs& hr, const float (&mask)[4], int (&q)[4][3]) {
#pragma unroll
  for (int k = 0; k < 4; k++) apply_wb(WB, w, q[k][0], q[k][1], q[k][2]);
  if constexpr ((BITS & ST_VIG) != 0) {
    // byte offsets into VigTabs::lin (gamma folded into that table by the host)
    unsigned lin_off[4][3], out_idx[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if constexpr ((BITS & ST_CC) != 0) {
        float o[3];
        apply_cc_f<BIAS>(p, cc, q[k][0], q[k][1], q[k][2], o);
        // saturate_cast<uchar> into byte 1 of the dword: (v << 8) >> 6 = 4 v, a full-rate right shift
#pragma unroll
        for (int c = 0; c < 3; c++) lin_off[k][c] = __builtin_amdgcn_cvt_pk_u8_f32(o[c], 1, 0u) >> 6;
      } else {
#pragma unroll
        for (int c = 0; c < 3; c++) lin_off[k][c] = (unsigned)q[k][c] << 2;
      }
    }
#pragma unroll
    for (int k0 = 0; k0 < 4; k0 += kVigGroup) vignette_n<kVigGroup>(tb.vig.v, mask + k0, lin_off + k0, out_idx + k0);
    if constexpr ((BITS & ST_HSV) == 0) return invg_pack4(out_idx);
    invg_values4(out_idx, q);
  } else {
    if constexpr ((BITS & ST_CC) != 0 && (BITS & (ST_GAMMA | ST_HSV)) == 0) {
      // the colour matrix is the last stage: its results are converted straight into their place in the packed output
      float of[4][3];
#pragma unroll
      for (int k = 0; k < 4; k++) apply_cc_f<BIAS>(p, cc, q[k][0], q[k][1], q[k][2], of[k]);
      return pack4_from_floats(of);
    }
    if constexpr (RIP_GAMMA_FOLD != 0 && (BITS & ST_CC) != 0 && (BITS & ST_GAMMA) != 0 && (BITS & ST_HSV) == 0) {
      // colour matrix -> gamma table -> output (the reference's default stage set): saturate_cast<uchar> lands in byte 0 of the
      // table's LDS address, the twelve look-ups are merged like the Lab round trip's (ds_read_u8 / ds_read_u8_d16_hi pairs:
      // two ORs and one v_lshl_or_b32 per output dword instead of two shifts, a shift-or and a three-way or)
      // the folded look-up needs byte 0 of the table's LDS address to be zero (v_cvt_pk_u8_f32 writes the index there)
      static_assert(alignof(GammaTab) == 256 && alignof(FastTabs<BITS>) >= 256, "gamma fold: the table must sit on a 256-byte boundary");
      const unsigned base = tb.gam.v.lds_address();
      unsigned addr[4][3];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float o[3];
        apply_cc_f<BIAS>(p, cc, q[k][0], q[k][1], q[k][2], o);
#pragma unroll
        for (int c = 0; c < 3; c++) addr[k][c] = __builtin_amdgcn_cvt_pk_u8_f32(o[c], 0, base);
      }
      return invg_pack4(addr);
    }
    if constexpr ((BITS & ST_CC) != 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) apply_cc<BIAS>(p, cc, q[k][0], q[k][1], q[k][2]);
    }
    if constexpr ((BITS & ST_GAMMA) != 0) {
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) q[k][c] = tb.gam.v.lut[q[k][c]];
    }
  }
  if constexpr ((BITS & ST_HSV) != 0) {
    float of[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) apply_hsv_tab(tb.hsv.v, q[k][0], q[k][1], q[k][2], of[k]);
    return pack4_from_floats(of);
  }
  return pack4(q);
}

}  // namespace
}  // namespace rip
