/*
 * rip_oracle.c -- CPU restatement of the raw_image_pipeline per-frame chain.
 * TEST INFRASTRUCTURE ONLY (see rip_oracle.h).  PARITY UNPINNED: no OpenCV here and the
 * reference has no golden vectors; each function cites the reference call site it
 * restates and the OpenCV 4.2 routine whose published algorithm it follows.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (see Makefile).  FP contraction
 * must stay off: the float stages mirror OpenCV's separate multiply/add sequence.
 */
#include "rip_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * Floating-point contraction model.  The float stages below restate C expressions of
 * OpenCV 4.2 (and of the reference's own translation units) that a compiler may or may not
 * contract into fused multiply-adds: GCC and Clang do so by default wherever the target has
 * FMA -- every aarch64 build (the reference's Jetson target), x86-64 only in the AVX2 / FMA3
 * dispatch variants.  Which rounding a given OpenCV binary performs is therefore a property
 * of its build, not of its source.  Mode 0 (default, what the HIP kernels implement) is the
 * uncontracted sequence: every product and every sum rounded.  Mode 1 contracts each
 * expression the way GCC's and Clang's passes do (a multiply feeding an add becomes one fma,
 * leftmost multiply first).  Mode 2: like 1, but the first sum of the 3-term dot product
 * takes the SECOND product as the fused one (the other legal association).  The tests bound
 * the difference between the modes per stage (tests/test_fp_contraction.py).
 * ---------------------------------------------------------------------------------- */
static int g_fp_contract = 0;
void ripo_set_fp_contraction(int mode) { g_fp_contract = (mode == 1 || mode == 2) ? mode : 0; }
int ripo_get_fp_contraction(void) { return g_fp_contract; }

/* ------------------------------------------------------------------------------------
 * OpenCV scalar helpers (core/fast_math.hpp, core/saturate.hpp)
 * ---------------------------------------------------------------------------------- */
static inline int cv_round_d(double v) { return (int)lrint(v); }   /* half to even */
static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* half to even */
static inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}
static inline uint8_t sat_u8_i(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline uint8_t sat_u8_f(float v) { return sat_u8_i(cv_round_f(v)); }
static inline uint8_t sat_u8_d(double v) { return sat_u8_i(cv_round_d(v)); }
static inline int16_t sat_s16_i(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
#define CV_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------
 * Debayer -- debayer.cpp:45-79.
 * cv::demosaicing(src, out, COLOR_BayerXX2BGR) (imgproc/demosaicing.cpp, Bayer2RGB_Invoker,
 * bilinear) followed by cvtColor(RGB2BGR).  OpenCV's two-letter Bayer code names the
 * pixels at (1,1),(1,2); the reference maps ROS "bayer_rggb8" -> BayerRG etc., i.e. it
 * demosaics with R/B exchanged and the swap exchanges them back: the net result is the
 * true-colour BGR bilinear demosaic of the ROS-named pattern.
 *   interior (1<=y<=H-2, 1<=x<=W-2):
 *     at an R/B site : own colour = centre, G = (N+S+W+E+2)>>2, other = (NW+NE+SW+SE+2)>>2
 *     at a  G  site  : colour of the left/right neighbours = (W+E+1)>>1,
 *                      colour of the up/down  neighbours   = (N+S+1)>>1
 *   borders: col 0 := col 1, col W-1 := col W-2 (per interior row), then
 *            row 0 := row 1, row H-1 := row H-2 (whole rows).
 * ---------------------------------------------------------------------------------- */
int ripo_bayer_pattern(const char* e) {
  if (!strcmp(e, "bayer_rggb8")) return RIPO_BAYER_RGGB;
  if (!strcmp(e, "bayer_grbg8")) return RIPO_BAYER_GRBG;
  if (!strcmp(e, "bayer_gbrg8")) return RIPO_BAYER_GBRG;
  if (!strcmp(e, "bayer_bggr8")) return RIPO_BAYER_BGGR;
  return -1;
}

/* colour (0=B,1=G,2=R) of the sensor sample at (y,x) for a ROS pattern */
static inline int bayer_color(int pattern, int y, int x) {
  static const int tab[4][2][2] = {
      /* RGGB */ {{2, 1}, {1, 0}},
      /* GRBG */ {{1, 2}, {0, 1}},
      /* GBRG */ {{1, 0}, {2, 1}},
      /* BGGR */ {{0, 1}, {1, 2}}};
  return tab[pattern][y & 1][x & 1];
}

void ripo_debayer_bilinear(const uint8_t* s, int rows, int cols, int pattern, uint8_t* d) {
  const int W = cols, H = rows;
  for (int y = 1; y < H - 1; y++) {
    for (int x = 1; x < W - 1; x++) {
      const uint8_t* p = s + (size_t)y * W + x;
      uint8_t* o = d + ((size_t)y * W + x) * 3;
      int c = bayer_color(pattern, y, x);
      if (c == 1) {
        int h = (p[-1] + p[1] + 1) >> 1;
        int v = (p[-W] + p[W] + 1) >> 1;
        int ch = bayer_color(pattern, y, x + 1); /* colour of horizontal neighbours */
        o[1] = p[0];
        o[ch] = (uint8_t)h;
        o[2 - ch] = (uint8_t)v;
      } else {
        int g = (p[-1] + p[1] + p[-W] + p[W] + 2) >> 2;
        int q = (p[-W - 1] + p[-W + 1] + p[W - 1] + p[W + 1] + 2) >> 2;
        o[c] = p[0];
        o[1] = (uint8_t)g;
        o[2 - c] = (uint8_t)q;
      }
    }
    /* first and last pixel of the row */
    uint8_t* r = d + (size_t)y * W * 3;
    memcpy(r, r + 3, 3);
    memcpy(r + (size_t)(W - 1) * 3, r + (size_t)(W - 2) * 3, 3);
  }
  /* first and last rows */
  memcpy(d, d + (size_t)W * 3, (size_t)W * 3);
  memcpy(d + (size_t)(H - 1) * W * 3, d + (size_t)(H - 2) * W * 3, (size_t)W * 3);
}

/* 16-bit Bayer (extension beyond the reference, which rejects bayer_*16 at debayer.cpp:76-78): cv::demosaicing's
 * Bayer2RGB_Invoker<ushort> runs the same two-tap / four-tap rounding averages and border replication on 16-bit
 * samples; restated on uint16_t, output BGR interleaved. */
void ripo_debayer_bilinear16(const uint16_t* s, int rows, int cols, int pattern, uint16_t* d) {
  const int W = cols, H = rows;
  for (int y = 1; y < H - 1; y++) {
    for (int x = 1; x < W - 1; x++) {
      const uint16_t* p = s + (size_t)y * W + x;
      uint16_t* o = d + ((size_t)y * W + x) * 3;
      int c = bayer_color(pattern, y, x);
      if (c == 1) {
        int h = ((int)p[-1] + p[1] + 1) >> 1;
        int v = ((int)p[-W] + p[W] + 1) >> 1;
        int ch = bayer_color(pattern, y, x + 1);
        o[1] = p[0];
        o[ch] = (uint16_t)h;
        o[2 - ch] = (uint16_t)v;
      } else {
        int g = ((int)p[-1] + p[1] + p[-W] + p[W] + 2) >> 2;
        int q = ((int)p[-W - 1] + p[-W + 1] + p[W - 1] + p[W + 1] + 2) >> 2;
        o[c] = p[0];
        o[1] = (uint16_t)g;
        o[2 - c] = (uint16_t)q;
      }
    }
    uint16_t* r = d + (size_t)y * W * 3;
    memcpy(r, r + 3, 6);
    memcpy(r + (size_t)(W - 1) * 3, r + (size_t)(W - 2) * 3, 6);
  }
  memcpy(d, d + (size_t)W * 3, (size_t)W * 6);
  memcpy(d + (size_t)(H - 1) * W * 3, d + (size_t)(H - 2) * W * 3, (size_t)W * 6);
}

void ripo_swap_rb(const uint8_t* src, size_t npix, uint8_t* dst) {
  for (size_t i = 0; i < npix; i++) {
    uint8_t b = src[i * 3], g = src[i * 3 + 1], r = src[i * 3 + 2];
    dst[i * 3] = r;
    dst[i * 3 + 1] = g;
    dst[i * 3 + 2] = b;
  }
}

/* ------------------------------------------------------------------------------------
 * Flip -- flip.cpp:37-58.  90: transpose then flip(...,1) (mirror x) = clockwise;
 * 180: flip(-1); 270: transpose then flip(...,0) (mirror y).
 * ---------------------------------------------------------------------------------- */
void ripo_flip(const uint8_t* src, int rows, int cols, int cn, int angle, uint8_t* dst, int* orows,
               int* ocols) {
  if (angle == 180) {
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++)
        memcpy(dst + ((size_t)y * cols + x) * cn,
               src + ((size_t)(rows - 1 - y) * cols + (cols - 1 - x)) * cn, (size_t)cn);
    *orows = rows;
    *ocols = cols;
  } else if (angle == 90) {
    /* t(y',x') = src(x',y');  out(y',x') = t(y', R-1-x') = src(R-1-x', y'), R = rows */
    int R = rows, C = cols; /* out is C rows x R cols */
    for (int y = 0; y < C; y++)
      for (int x = 0; x < R; x++)
        memcpy(dst + ((size_t)y * R + x) * cn, src + ((size_t)(R - 1 - x) * C + y) * cn, (size_t)cn);
    *orows = C;
    *ocols = R;
  } else if (angle == 270) {
    /* out(y',x') = t(C-1-y', x') = src(x', C-1-y') */
    int R = rows, C = cols;
    for (int y = 0; y < C; y++)
      for (int x = 0; x < R; x++)
        memcpy(dst + ((size_t)y * R + x) * cn, src + ((size_t)x * C + (C - 1 - y)) * cn, (size_t)cn);
    *orows = C;
    *ocols = R;
  } else {
    memcpy(dst, src, (size_t)rows * cols * cn);
    *orows = rows;
    *ocols = cols;
  }
}

/* ------------------------------------------------------------------------------------
 * Grey-world WB -- white_balance.cpp:59-64 -> cv::xphoto::GrayworldWB
 * (xphoto/src/grayworld_white_balance.cpp: calculateChannelSums + applyChannelGains).
 * OpenCV accumulates in uint32 (no overflow below ~16.8 Mpx); uint64 here.
 * ---------------------------------------------------------------------------------- */
void ripo_wb_grayworld(uint8_t* d, size_t npix, double thr, uint64_t sums_out[3], int ig_out[3]) {
  unsigned thresh255 = (unsigned)(uint16_t)cv_round_f((float)thr * 255);
  uint64_t sb = 0, sg = 0, sr = 0;
  for (size_t i = 0; i < npix; i++) {
    unsigned b = d[i * 3], g = d[i * 3 + 1], r = d[i * 3 + 2];
    unsigned mn = b < g ? b : g;
    mn = mn < r ? mn : r;
    unsigned mx = b > g ? b : g;
    mx = mx > r ? mx : r;
    if ((mx - mn) * 255 > thresh255 * mx) continue;
    sb += b;
    sg += g;
    sr += r;
  }
  double dsb = (double)sb, dsg = (double)sg, dsr = (double)sr;
  double max_sum = fmax(dsb, fmax(dsr, dsg));
  const double eps = 0.1;
  float gb = dsb < eps ? 0.f : (float)(max_sum / dsb);
  float gg = dsg < eps ? 0.f : (float)(max_sum / dsg);
  float gr = dsr < eps ? 0.f : (float)(max_sum / dsr);
  /* applyChannelGains, CV_8UC3 branch */
  float gmax = fmaxf(gb, fmaxf(gg, gr));
  if (gmax > 0) {
    gb /= gmax;
    gg /= gmax;
    gr /= gmax;
  }
  int ib = cv_round_f(gb * (1 << 8)), ig = cv_round_f(gg * (1 << 8)), ir = cv_round_f(gr * (1 << 8));
  for (size_t i = 0; i < npix; i++) {
    d[i * 3] = (uint8_t)((d[i * 3] * ib) >> 8);
    d[i * 3 + 1] = (uint8_t)((d[i * 3 + 1] * ig) >> 8);
    d[i * 3 + 2] = (uint8_t)((d[i * 3 + 2] * ir) >> 8);
  }
  if (sums_out) {
    sums_out[0] = sb;
    sums_out[1] = sg;
    sums_out[2] = sr;
  }
  if (ig_out) {
    ig_out[0] = ib;
    ig_out[1] = ig;
    ig_out[2] = ir;
  }
}

/* ------------------------------------------------------------------------------------
 * "simple" WB -- white_balance.cpp:52-57 -> cv::xphoto::SimpleWB (xphoto/src/
 * simple_color_balance.cpp, balanceWhiteSimple<uchar>): per channel a two-level tree of 16-bin
 * histograms over [-0.5, 255.5]; low / high cut where the cumulative count crosses p% / (100-p)%;
 * then the affine stretch 255 * (x - lo) / (hi - lo) evaluated as one convertTo(alpha, beta) in
 * float.  Restated literally, including the tree layout in which the second level of bin 0 shares
 * its storage with the first level (hist[pos + currentBin] with pos = 0 at both).
 * ---------------------------------------------------------------------------------- */
void ripo_simple_wb_stretch(const uint32_t hist256[256], int total, float p, float* alpha_out, float* beta_out) {
  const int bins = 16, depth = 2;
  int hist[256];
  memset(hist, 0, sizeof(hist));
  for (int v = 0; v < 256; v++) {
    int c = (int)hist256[v];
    if (!c) continue;
    /* histogram filling for one value (all c pixels of value v take the same path) */
    int pos = 0;
    float minValue = 0.f - 0.5f, maxValue = 255.f + 0.5f;
    float interval = (float)(maxValue - minValue) / bins;
    for (int j = 0; j < depth; ++j) {
      int currentBin = (int)(((float)v - minValue + 1e-4f) / interval);
      hist[pos + currentBin] += c;
      pos = (pos + currentBin) * bins;
      minValue = minValue + currentBin * interval;
      maxValue = minValue + interval;
      interval /= bins;
    }
    (void)maxValue;
  }
  const float s1 = p, s2 = p;
  int p1 = 0, p2 = bins - 1;
  int n1 = 0, n2 = total;
  float minValue = 0.f - 0.5f, maxValue = 255.f + 0.5f;
  float interval = (maxValue - minValue) / (float)bins;
  for (int j = 0; j < depth; ++j) {
    while (p1 < 255 && n1 + hist[p1] < s1 * total / 100.0f) {
      n1 += hist[p1++];
      minValue += interval;
    }
    p1 *= bins;
    while (p2 > 0 && n2 - hist[p2] > (100.0f - s2) * total / 100.0f) {
      n2 -= hist[p2--];
      maxValue -= interval;
    }
    p2 = (p2 + 1) * bins - 1;
    interval /= bins;
    if (p1 > 255) p1 = 255; /* the reference would read out of bounds here; unreachable for sane p */
    if (p2 > 255) p2 = 255;
  }
  /* src = (outputMax - outputMin) * (src - minValue) / (maxValue - minValue) + outputMin as a MatExpr:
   * alpha = 255 * (1/d), beta = (-minValue * 255) * (1/d) + 0 in double, narrowed to float by convertTo */
  double d = (double)(float)(maxValue - minValue);
  double inv = 1.0 / d;
  double alpha = (1.0 * (double)(255.f - 0.f)) * inv;
  double beta = ((-(double)minValue) * (double)(255.f - 0.f)) * inv + (double)0.f;
  *alpha_out = (float)alpha;
  *beta_out = (float)beta;
}

void ripo_wb_simple(uint8_t* d, size_t npix, double percentile, float ab_out[6]) {
  for (int c = 0; c < 3; c++) {
    uint32_t h[256];
    memset(h, 0, sizeof(h));
    for (size_t i = 0; i < npix; i++) h[d[i * 3 + c]]++;
    float a, b;
    ripo_simple_wb_stretch(h, (int)npix, (float)percentile, &a, &b);
    for (size_t i = 0; i < npix; i++) d[i * 3 + c] = sat_u8_f((float)d[i * 3 + c] * a + b);
    if (ab_out) {
      ab_out[c * 2] = a;
      ab_out[c * 2 + 1] = b;
    }
  }
}

/* ------------------------------------------------------------------------------------
 * "pca" WB -- white_balance.cpp:73-136, literal.  cv::sum on 32F accumulates in double
 * (all terms are integers < 2^53 so the sum is exact in any order); minMaxLoc maxima;
 * Eigen::Matrix2f filled from doubles (narrowing), inverse() = adjugate / determinant in
 * float; b' = c0*b^2 + c1*b as float addWeighted; THRESH_TRUNC at 255; convertTo(8U).
 * ---------------------------------------------------------------------------------- */
static void solve2f(float m00, float m01, float m10, float m11, float g0, float g1, float out[2]) {
  float det = m00 * m11 - m01 * m10;
  float invdet = 1.0f / det;
  float i00 = m11 * invdet, i01 = -m01 * invdet, i10 = -m10 * invdet, i11 = m00 * invdet;
  out[0] = i00 * g0 + i01 * g1;
  out[1] = i10 * g0 + i11 * g1;
}

void ripo_wb_pca(uint8_t* d, size_t npix, float coeffs_out[4]) {
  double s_b = 0, s_b2 = 0, s_r = 0, s_r2 = 0, s_g = 0;
  int mx_b = 0, mx_r = 0, mx_g = 0;
  for (size_t i = 0; i < npix; i++) {
    int b = d[i * 3], g = d[i * 3 + 1], r = d[i * 3 + 2];
    s_b += b;
    s_b2 += (double)((float)b * (float)b);
    s_r += r;
    s_r2 += (double)((float)r * (float)r);
    s_g += g;
    mx_b = imax(mx_b, b);
    mx_r = imax(mx_r, r);
    mx_g = imax(mx_g, g);
  }
  double mx_b2 = (double)((float)mx_b * (float)mx_b), mx_r2 = (double)((float)mx_r * (float)mx_r);
  float cb[2], cr[2];
  solve2f((float)s_b2, (float)s_b, (float)mx_b2, (float)mx_b, (float)s_g, (float)mx_g, cb);
  solve2f((float)s_r2, (float)s_r, (float)mx_r2, (float)mx_r, (float)s_g, (float)mx_g, cr);
  for (size_t i = 0; i < npix; i++) {
    float b = d[i * 3], r = d[i * 3 + 2];
    float b2 = b * b, r2 = r * r;
    float bp = b2 * cb[0] + b * cb[1];
    float rp = r2 * cr[0] + r * cr[1];
    if (g_fp_contract) {
      /* cv::addWeighted (the MatExpr a*A + b*B): op_add_weighted is v_fma(A, a, v_fma(B, b, 0)) -- a true fma in
       * NEON / FMA3 builds, so only the second product is rounded on its own */
      bp = fmaf(b2, cb[0], b * cb[1]);
      rp = fmaf(r2, cr[0], r * cr[1]);
    }
    bp = bp > 255.f ? 255.f : bp;
    rp = rp > 255.f ? 255.f : rp;
    d[i * 3] = sat_u8_f(bp);
    d[i * 3 + 2] = sat_u8_f(rp);
  }
  if (coeffs_out) {
    coeffs_out[0] = cb[0];
    coeffs_out[1] = cb[1];
    coeffs_out[2] = cr[0];
    coeffs_out[3] = cr[1];
  }
}

/* ------------------------------------------------------------------------------------
 * cv::resize(8U, INTER_LINEAR) (imgproc/resize.cpp, non-IPP path): coefficients Q11,
 * horizontal pass to int, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2.
 * When both scale factors are exactly 2, INTER_LINEAR is silently replaced by INTER_AREA
 * (2x2 mean, (a+b+c+d+2)>>2).
 * ---------------------------------------------------------------------------------- */
void ripo_resize_linear_8u(const uint8_t* src, int rows, int cols, int cn, uint8_t* dst, int drows,
                           int dcols) {
  double scale_x = (double)cols / dcols, scale_y = (double)rows / drows;
  int iscale_x = cv_round_d(scale_x), iscale_y = cv_round_d(scale_y);
  int area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
  if (area_fast && iscale_x == 2 && iscale_y == 2) {
    for (int y = 0; y < drows; y++)
      for (int x = 0; x < dcols; x++)
        for (int c = 0; c < cn; c++) {
          const uint8_t* p = src + ((size_t)(2 * y) * cols + 2 * x) * cn + c;
          dst[((size_t)y * dcols + x) * cn + c] =
              (uint8_t)((p[0] + p[cn] + p[(size_t)cols * cn] + p[(size_t)cols * cn + cn] + 2) >> 2);
        }
    return;
  }
  int* xofs = (int*)malloc(sizeof(int) * dcols);
  short* ialpha = (short*)malloc(sizeof(short) * dcols * 2);
  for (int dx = 0; dx < dcols; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx >= cols - 1) {
      fx = 0;
      sx = cols - 1;
    }
    xofs[dx] = sx;
    ialpha[dx * 2] = sat_s16_i(cv_round_f((1.f - fx) * 2048));
    ialpha[dx * 2 + 1] = sat_s16_i(cv_round_f(fx * 2048));
  }
  int* row0 = (int*)malloc(sizeof(int) * (size_t)dcols * cn);
  int* row1 = (int*)malloc(sizeof(int) * (size_t)dcols * cn);
  for (int dy = 0; dy < drows; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    short b0 = sat_s16_i(cv_round_f((1.f - fy) * 2048)), b1 = sat_s16_i(cv_round_f(fy * 2048));
    int sy0 = sy < 0 ? 0 : (sy < rows ? sy : rows - 1);
    int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < rows ? sy + 1 : rows - 1);
    const uint8_t* S0 = src + (size_t)sy0 * cols * cn;
    const uint8_t* S1 = src + (size_t)sy1 * cols * cn;
    for (int dx = 0; dx < dcols; dx++) {
      int sx = xofs[dx];
      int sx1 = sx + 1 < cols ? sx + 1 : sx; /* weight is 0 whenever sx is the last column */
      int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
      for (int c = 0; c < cn; c++) {
        row0[dx * cn + c] = S0[sx * cn + c] * a0 + S0[sx1 * cn + c] * a1;
        row1[dx * cn + c] = S1[sx * cn + c] * a0 + S1[sx1 * cn + c] * a1;
      }
    }
    uint8_t* D = dst + (size_t)dy * dcols * cn;
    for (int i = 0; i < dcols * cn; i++)
      D[i] = (uint8_t)((((b0 * (row0[i] >> 4)) >> 16) + ((b1 * (row1[i] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs);
  free(ialpha);
  free(row0);
  free(row1);
}

/* ------------------------------------------------------------------------------------
 * Convolutional colour constancy -- convolutional_color_constancy.cpp.
 * cv::dft is restated as a float32 radix-2 (decimation in time) FFT with a host-built
 * twiddle table; forward and inverse are both unscaled as in the reference (:283,:292);
 * only the argmax of the response is consumed (:295).  The product's HIP FFT uses the same
 * butterfly order and table so the response is bit-identical; ripo_ccc_response_direct()
 * checks the argmax against an O(N^2 * nnz) double-precision circular convolution.
 * ---------------------------------------------------------------------------------- */
#define CCC_N 256
struct ripo_ccc {
  int w, h;
  float* filter_t; /* transposed model (loadModel :131-132) */
  float* bias_t;
  float* filter_fft_re;
  float* filter_fft_im;
  float* bias_fft_re;
  float* bias_fft_im;
  float tw_re[CCC_N / 2], tw_im[CCC_N / 2];
  float log_tab[256];
  float bright_thr, dark_thr;
  int temporal;
  float kf_h, kf_r;
  /* state */
  int first_frame;
  int uv_x, uv_y;
  float st_x, st_y; /* statePost */
  float p_x, p_y;   /* errorCovPost diagonal */
};

void ripo_fft256_twiddles(float* re, float* im) {
  for (int k = 0; k < CCC_N / 2; k++) {
    double a = -2.0 * 3.14159265358979323846 * k / CCC_N;
    re[k] = (float)cos(a);
    im[k] = (float)sin(a);
  }
}

static unsigned bitrev8(unsigned x) {
  x = ((x & 0xF0) >> 4) | ((x & 0x0F) << 4);
  x = ((x & 0xCC) >> 2) | ((x & 0x33) << 2);
  x = ((x & 0xAA) >> 1) | ((x & 0x55) << 1);
  return x;
}

/* In-place 256-point FFT over a strided complex vector.  inverse: conjugated twiddles. */
static void fft256(float* re, float* im, int stride, const float* twr, const float* twi, int inverse) {
  float ar[CCC_N], ai[CCC_N];
  for (int i = 0; i < CCC_N; i++) {
    unsigned j = bitrev8((unsigned)i);
    ar[j] = re[(size_t)i * stride];
    ai[j] = im[(size_t)i * stride];
  }
  for (int s = 1; s <= 8; s++) {
    int m = 1 << s, half = m >> 1, tstep = CCC_N / m;
    for (int b = 0; b < CCC_N / 2; b++) { /* butterfly index */
      int grp = b / half, k = b % half;
      int i0 = grp * m + k, i1 = i0 + half;
      float wr = twr[k * tstep], wi = inverse ? -twi[k * tstep] : twi[k * tstep];
      float xr = ar[i1], xi = ai[i1];
      float tr = wr * xr - wi * xi;
      float ti = wr * xi + wi * xr;
      float ur = ar[i0], ui = ai[i0];
      ar[i0] = ur + tr;
      ai[i0] = ui + ti;
      ar[i1] = ur - tr;
      ai[i1] = ui - ti;
    }
  }
  for (int i = 0; i < CCC_N; i++) {
    re[(size_t)i * stride] = ar[i];
    im[(size_t)i * stride] = ai[i];
  }
}

/* forward: rows then columns; inverse: columns then rows (the order the product's kernels use, so
 * that the float response is bit-identical; cv::dft's own pass order is not observable through the
 * argmax the reference consumes) */
static void fft2d(float* re, float* im, const float* twr, const float* twi, int inverse) {
  if (!inverse)
    for (int r = 0; r < CCC_N; r++) fft256(re + (size_t)r * CCC_N, im + (size_t)r * CCC_N, 1, twr, twi, 0);
  for (int c = 0; c < CCC_N; c++) fft256(re + c, im + c, CCC_N, twr, twi, inverse);
  if (inverse)
    for (int r = 0; r < CCC_N; r++) fft256(re + (size_t)r * CCC_N, im + (size_t)r * CCC_N, 1, twr, twi, 1);
}

ripo_ccc* ripo_ccc_create(int width, int height, const float* filter, const float* bias) {
  if (width != CCC_N || height != CCC_N) return NULL;
  ripo_ccc* c = (ripo_ccc*)calloc(1, sizeof(ripo_ccc));
  size_t n = (size_t)CCC_N * CCC_N;
  c->w = width;
  c->h = height;
  c->filter_t = (float*)malloc(n * 4);
  c->bias_t = (float*)malloc(n * 4);
  c->filter_fft_re = (float*)malloc(n * 4);
  c->filter_fft_im = (float*)calloc(n, 4);
  c->bias_fft_re = (float*)malloc(n * 4);
  c->bias_fft_im = (float*)calloc(n, 4);
  for (int y = 0; y < CCC_N; y++)
    for (int x = 0; x < CCC_N; x++) {
      c->filter_t[(size_t)y * CCC_N + x] = filter[(size_t)x * CCC_N + y];
      c->bias_t[(size_t)y * CCC_N + x] = bias[(size_t)x * CCC_N + y];
    }
  ripo_fft256_twiddles(c->tw_re, c->tw_im);
  memcpy(c->filter_fft_re, c->filter_t, n * 4);
  memcpy(c->bias_fft_re, c->bias_t, n * 4);
  fft2d(c->filter_fft_re, c->filter_fft_im, c->tw_re, c->tw_im, 0);
  fft2d(c->bias_fft_re, c->bias_fft_im, c->tw_re, c->tw_im, 0);
  /* cv::log on integer-valued floats 0..255 (:228); log(0) = -inf is skipped at :243 */
  for (int i = 0; i < 256; i++) c->log_tab[i] = i == 0 ? -INFINITY : logf((float)i);
  c->bright_thr = 0.9f; /* ctor defaults :24-25 */
  c->dark_thr = 0.1f;
  c->temporal = 0;
  c->kf_h = 0.f; /* default cv::KalmanFilter: measurementMatrix = 0, measurementNoiseCov = I */
  c->kf_r = 1.f;
  ripo_ccc_reset(c);
  c->uv_x = CCC_N / 2; /* :178 */
  c->uv_y = CCC_N / 2;
  c->st_x = (float)c->uv_x;
  c->st_y = (float)c->uv_y;
  return c;
}

void ripo_ccc_destroy(ripo_ccc* c) {
  if (!c) return;
  free(c->filter_t);
  free(c->bias_t);
  free(c->filter_fft_re);
  free(c->filter_fft_im);
  free(c->bias_fft_re);
  free(c->bias_fft_im);
  free(c);
}
void ripo_ccc_set_thresholds(ripo_ccc* c, float b, float d) {
  c->bright_thr = b;
  c->dark_thr = d;
}
void ripo_ccc_set_temporal_consistency(ripo_ccc* c, int e) { c->temporal = e; }
void ripo_ccc_set_kalman_model(ripo_ccc* c, float h, float r) {
  c->kf_h = h;
  c->kf_r = r;
}
/* resetTemporalConsistency :433-435 only re-arms first_frame_; the filter covariance is kept */
void ripo_ccc_reset(ripo_ccc* c) { c->first_frame = 1; }

/* calculateHistogramFeature :210-271 */
void ripo_ccc_histogram(const ripo_ccc* c, const uint8_t* small, int rows, int cols, float* hist) {
  const float bin_size = 1.0f / 64.0f, uv0 = -1.421875f;
  memset(hist, 0, sizeof(float) * CCC_N * CCC_N);
  float num_pixels = (float)(rows * cols);
  float pixel_weight = 1.0f / num_pixels;
  float upper = 255 * c->bright_thr, lower = 255 * c->dark_thr;
  for (int i = 0; i < rows * cols; i++) {
    float b = small[i * 3], g = small[i * 3 + 1], r = small[i * 3 + 2];
    /* cvtColor(BGR2GRAY) on CV_32F: b*0.114f + g*0.587f + r*0.299f */
    float gray = b * 0.114f + g * 0.587f + r * 0.299f;
    int upper_ok = !(gray > upper); /* THRESH_BINARY_INV */
    int lower_ok = gray > lower;    /* THRESH_BINARY */
    float lb = c->log_tab[small[i * 3]], lg = c->log_tab[small[i * 3 + 1]], lr = c->log_tab[small[i * 3 + 2]];
    if (!isfinite(lr) || !isfinite(lg) || !isfinite(lb)) continue;
    if (!(upper_ok && lower_ok)) continue;
    int u = (int)roundf((lg - lr - uv0) / bin_size);
    int v = (int)roundf((lg - lb - uv0) / bin_size);
    u = imax(imin(u, 255), 0);
    v = imax(imin(v, 255), 0);
    hist[(size_t)u * CCC_N + v] += pixel_weight;
  }
}

/* computeResponse :273-298 (dft, mulSpectrums, add, inverse dft) */
void ripo_ccc_response(const ripo_ccc* c, const float* hist, float* response) {
  size_t n = (size_t)CCC_N * CCC_N;
  float* re = (float*)malloc(n * 4);
  float* im = (float*)calloc(n, 4);
  memcpy(re, hist, n * 4);
  fft2d(re, im, c->tw_re, c->tw_im, 0);
  for (size_t i = 0; i < n; i++) {
    float ar = c->filter_fft_re[i], ai = c->filter_fft_im[i], br = re[i], bi = im[i];
    float pr = ar * br - ai * bi;
    float pi = ar * bi + ai * br;
    re[i] = pr + c->bias_fft_re[i];
    im[i] = pi + c->bias_fft_im[i];
  }
  fft2d(re, im, c->tw_re, c->tw_im, 1);
  memcpy(response, re, n * 4);
  free(re);
  free(im);
}

void ripo_ccc_response_direct(const ripo_ccc* c, const float* hist, double* response) {
  size_t n = (size_t)CCC_N * CCC_N;
  for (size_t i = 0; i < n; i++) response[i] = (double)c->bias_t[i];
  for (int hy = 0; hy < CCC_N; hy++)
    for (int hx = 0; hx < CCC_N; hx++) {
      double hv = hist[(size_t)hy * CCC_N + hx];
      if (hv == 0) continue;
      for (int fy = 0; fy < CCC_N; fy++)
        for (int fx = 0; fx < CCC_N; fx++)
          response[(size_t)((hy + fy) & 255) * CCC_N + ((hx + fx) & 255)] +=
              hv * (double)c->filter_t[(size_t)fy * CCC_N + fx];
    }
}

/* computeGains :342-381 */
void ripo_ccc_gains_from_uv(int ux, int uy, float gains_bgr[3]) {
  const float bin_size = 1.0f / 64.0f, uv0 = -1.421875f;
  float Lu = ux * bin_size + uv0;
  float Lv = uy * bin_size + uv0;
  float z = 1.0f;
  float gain_r = z / expf(-Lu);
  float gain_g = z;
  float gain_b = z / expf(-Lv);
  float factor = fminf(fminf(gain_r, gain_g), gain_b);
  gain_r /= factor;
  gain_g /= factor;
  gain_b /= factor;
  gains_bgr[0] = gain_b;
  gains_bgr[1] = gain_g;
  gains_bgr[2] = gain_r;
}

/* kalmanFiltering :300-340 with cv::KalmanFilter(2,2,0) reduced to its diagonal form:
 * A = I, Q = I, H = h I, R = r I, P0 = 0. */
static void ccc_kalman(ripo_ccc* c) {
  if (c->first_frame) {
    c->first_frame = 0;
    c->st_x = (float)c->uv_x;
    c->st_y = (float)c->uv_y;
    return;
  }
  float h = c->kf_h, r = c->kf_r;
  float* st[2] = {&c->st_x, &c->st_y};
  float* pp[2] = {&c->p_x, &c->p_y};
  int z[2] = {c->uv_x, c->uv_y};
  int* out[2] = {&c->uv_x, &c->uv_y};
  for (int a = 0; a < 2; a++) {
    float x_pre = *st[a];           /* statePre = A * statePost */
    float p_pre = *pp[a] + 1.0f;    /* errorCovPre = A P A' + Q */
    float t2 = h * p_pre;           /* temp2 = H * errorCovPre */
    float t3 = t2 * h + r;          /* temp3 = temp2 * H' + R */
    float k = t2 / t3;              /* gain = (temp3^-1 * temp2)' */
    float innov = (float)z[a] - h * x_pre;
    float x_post = x_pre + k * innov;
    float p_post = p_pre - k * t2;
    *st[a] = x_post;
    *pp[a] = p_post;
    *out[a] = (int)x_post; /* truncation :336-337 */
  }
}

static void argmax_first(const float* r, int* x, int* y) {
  float best = r[0];
  size_t bi = 0;
  for (size_t i = 1; i < (size_t)CCC_N * CCC_N; i++)
    if (r[i] > best) {
      best = r[i];
      bi = i;
    }
  *x = (int)(bi % CCC_N);
  *y = (int)(bi / CCC_N);
}

void ripo_ccc_balance(ripo_ccc* c, uint8_t* bgr, int rows, int cols, int info_out[4], float gains_out[3]) {
  const int SW = 360, SH = 270; /* small_size_ :22 */
  uint8_t* small = (uint8_t*)malloc((size_t)SW * SH * 3);
  ripo_resize_linear_8u(bgr, rows, cols, 3, small, SH, SW);
  float* hist = (float*)malloc(sizeof(float) * CCC_N * CCC_N);
  float* resp = (float*)malloc(sizeof(float) * CCC_N * CCC_N);
  ripo_ccc_histogram(c, small, SH, SW, hist);
  ripo_ccc_response(c, hist, resp);
  argmax_first(resp, &c->uv_x, &c->uv_y);
  if (info_out) {
    info_out[0] = c->uv_x;
    info_out[1] = c->uv_y;
  }
  if (c->temporal) ccc_kalman(c);
  if (info_out) {
    info_out[2] = c->uv_x;
    info_out[3] = c->uv_y;
  }
  float g[3];
  ripo_ccc_gains_from_uv(c->uv_x, c->uv_y, g);
  /* applyGains :383-386: cv::multiply(8UC3, Scalar) = sat_u8(float(x) * float(gain)) */
  size_t npix = (size_t)rows * cols;
  for (size_t i = 0; i < npix; i++) {
    bgr[i * 3] = sat_u8_f((float)bgr[i * 3] * g[0]);
    bgr[i * 3 + 1] = sat_u8_f((float)bgr[i * 3 + 1] * g[1]);
    bgr[i * 3 + 2] = sat_u8_f((float)bgr[i * 3 + 2] * g[2]);
  }
  if (gains_out) memcpy(gains_out, g, sizeof(g));
  free(small);
  free(hist);
  free(resp);
}

/* ------------------------------------------------------------------------------------
 * Colour calibration -- color_calibration.cpp:91-104.  cv::gemm small-matrix case:
 * t = a0*b0 + a1*b1 + a2*b2 in float32 (left to right, no FMA), + float(bias),
 * convertTo(8U) = round-half-even + clamp.
 * ---------------------------------------------------------------------------------- */
void ripo_color_matrix(uint8_t* d, size_t npix, const double m[9], const double bias[3]) {
  float M[9], B[3];
  for (int i = 0; i < 9; i++) M[i] = (float)m[i];
  for (int i = 0; i < 3; i++) B[i] = (float)bias[i];
  for (size_t i = 0; i < npix; i++) {
    float a0 = d[i * 3], a1 = d[i * 3 + 1], a2 = d[i * 3 + 2];
    for (int c = 0; c < 3; c++) {
      float t = a0 * M[c * 3] + a1 * M[c * 3 + 1] + a2 * M[c * 3 + 2];
      if (g_fp_contract == 1) /* (a0*b0 + a1*b1) + a2*b2: fma(a0, b0, a1*b1), then fma(a2, b2, .) */
        t = fmaf(a2, M[c * 3 + 2], fmaf(a0, M[c * 3], a1 * M[c * 3 + 1]));
      else if (g_fp_contract == 2)
        t = fmaf(a2, M[c * 3 + 2], fmaf(a1, M[c * 3 + 1], a0 * M[c * 3]));
      t = t + B[c]; /* a separate cv::add over the whole Mat: never fused with the product */
      d[i * 3 + c] = sat_u8_f(t);
    }
  }
}

/* ------------------------------------------------------------------------------------
 * Gamma -- gamma_correction.cpp:35-43 literal; cv::LUT :54-60.
 * ---------------------------------------------------------------------------------- */
void ripo_gamma_lut(double k, uint8_t lut[256]) {
  for (int i = 0; i < 256; i++) {
    float f = (float)(i / 255.0);
    f = (float)pow((double)f, k);
    lut[i] = sat_u8_d((double)f * 255.0);
  }
}
void ripo_apply_lut(uint8_t* d, size_t n, const uint8_t lut[256]) {
  for (size_t i = 0; i < n; i++) d[i] = lut[d[i]];
}

/* ------------------------------------------------------------------------------------
 * 8-bit Lab -- cvtColor(BGR2Lab / Lab2BGR) (imgproc/color_lab.cpp: RGB2Lab_b and
 * Lab2RGBinteger, the default bit-exact path).  Tables follow initLabTabs(); OpenCV builds
 * them with softfloat (IEEE single, round-to-nearest-even), restated with plain float
 * arithmetic.  softfloat's cbrt is the cvCbrt rational approximation (restated below);
 * its pow is not reproducible from the published description -> double pow() narrowed to
 * float (a handful of table entries may differ by one unit; flagged lowest-confidence).
 * ---------------------------------------------------------------------------------- */
enum { LAB_SHIFT = 12, GAMMA_SHIFT = 3, LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT };
enum { LAB_CBRT_TAB_SIZE_B = 256 * 3 / 2 * (1 << GAMMA_SHIFT) };
enum { INV_GAMMA_TAB_SIZE = 4096, LAB_BASE = 1 << 14, MIN_AB_VALUE = -8145 };

static uint16_t g_srgb_gamma_b[256];
static uint16_t g_cbrt_b[LAB_CBRT_TAB_SIZE_B];
static uint16_t g_lab_to_yf_b[512];
static uint16_t g_inv_gamma_b[INV_GAMMA_TAB_SIZE];
static int g_fwd_coeffs[9], g_inv_coeffs[9];
static int g_sdiv[256], g_hdiv180[256];
static int g_tabs_ready = 0;

/* core/mathfuncs_core: cubeRoot() -- quartic rational approximation, error < 2^-24 */
static float cv_cbrt(float value) {
  union {
    float f;
    int32_t i;
    uint32_t u;
  } v, m;
  v.f = value;
  int ix = v.i & 0x7fffffff;
  uint32_t s = v.u & 0x80000000u;
  int ex = (ix >> 23) - 127;
  int shx = ex % 3;
  shx -= shx >= 0 ? 3 : 0;
  ex = (ex - shx) / 3;
  v.i = (ix & ((1 << 23) - 1)) | ((shx + 127) << 23);
  double fr = v.f;
  fr = (float)(((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr +
                  119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr +
                0.1636161226585754240958355063) /
               ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr +
                  168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr + 1.0));
  m.f = value;
  v.f = (float)fr;
  v.u = (v.u + ((uint32_t)ex << 23) + s) & ((m.u * 2u) != 0 ? 0xffffffffu : 0u);
  return v.f;
}

static float apply_gamma_f(float x) {
  const float thr = 809.f / 20000.f, low = 323.f / 25.f, power = 12.f / 5.f, xshift = 11.f / 200.f;
  if (x <= thr) return x / low;
  float base = (x + xshift) / (1.0f + xshift);
  return (float)pow((double)base, (double)power);
}
static float apply_inv_gamma_f(float x) {
  const float thr = 7827.f / 2500000.f, low = 323.f / 25.f, power = 12.f / 5.f, xshift = 11.f / 200.f;
  if (x <= thr) return x * low;
  float e = 1.0f / power;
  float p = (float)pow((double)x, (double)e);
  return p * (1.0f + xshift) - xshift;
}

static void init_tables(void) {
  if (g_tabs_ready) return;
  /* sRGBGammaTab_b / sRGBInvGammaTab_b */
  for (int i = 0; i < 256; i++) {
    float x = (float)i / 255.f;
    g_srgb_gamma_b[i] = (uint16_t)cv_round_f((255.f * (1 << GAMMA_SHIFT)) * apply_gamma_f(x));
  }
  for (int i = 0; i < INV_GAMMA_TAB_SIZE; i++) {
    float x = (1.0f / INV_GAMMA_TAB_SIZE) * (float)i;
    g_inv_gamma_b[i] = (uint16_t)cv_round_f(255.f * apply_inv_gamma_f(x));
  }
  /* LabCbrtTab_b */
  {
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f;
    const float tscale = 1.0f / (255.f * (1 << GAMMA_SHIFT));
    for (int i = 0; i < LAB_CBRT_TAB_SIZE_B; i++) {
      float x = tscale * (float)i;
      float f = x < lthresh ? fmaf(x, lscale, lbias) : cv_cbrt(x);
      g_cbrt_b[i] = (uint16_t)cv_round_f((float)(1 << LAB_SHIFT2) * f);
    }
  }
  /* LabToYF_b */
  for (int i = 0; i < 256; i++) {
    int y, ify;
    const int BASE = LAB_BASE;
    if (i <= 20) {
      y = cv_round_f((float)(i * BASE * 20 * 9) / (float)(17 * 29 * 29 * 29));
      ify = cv_round_f((float)BASE * (16.f / 116.f + (float)(i * 5) / (float)(3 * 17 * 29)));
    } else {
      float fy = (float)(i * 100 * BASE) / (float)(255 * 116) + (float)(16 * BASE) / 116.f;
      ify = cv_round_f(fy);
      y = cv_round_f(fy * fy * fy / (float)(BASE * BASE));
    }
    g_lab_to_yf_b[i * 2] = (uint16_t)y;
    g_lab_to_yf_b[i * 2 + 1] = (uint16_t)ify;
  }
  /* coefficient matrices (sRGB, D65), softdouble in OpenCV == IEEE double */
  {
    static const double sRGB2XYZ[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160,
                                       0.072169, 0.019334, 0.119193, 0.950227};
    static const double XYZ2sRGB[9] = {3.240479, -1.53715,  -0.498535, -0.969256, 1.875991,
                                       0.041556, 0.055648, -0.204043, 1.057311};
    static const double D65[3] = {0.950456, 1., 1.088754};
    const double lshift = (double)(1 << LAB_SHIFT);
    /* forward, blueIdx = 0: coefficient order in memory is (B,G,R) */
    for (int i = 0; i < 3; i++) {
      g_fwd_coeffs[i * 3 + 2] = cv_round_d(lshift * sRGB2XYZ[i * 3 + 0] / D65[i]);
      g_fwd_coeffs[i * 3 + 1] = cv_round_d(lshift * sRGB2XYZ[i * 3 + 1] / D65[i]);
      g_fwd_coeffs[i * 3 + 0] = cv_round_d(lshift * sRGB2XYZ[i * 3 + 2] / D65[i]);
    }
    /* inverse: row 0 = B, row 1 = G, row 2 = R; column i multiplies X_i */
    for (int i = 0; i < 3; i++) {
      g_inv_coeffs[i + 2 * 3] = cv_round_d(lshift * XYZ2sRGB[i + 0 * 3] * D65[i]); /* R row */
      g_inv_coeffs[i + 1 * 3] = cv_round_d(lshift * XYZ2sRGB[i + 1 * 3] * D65[i]); /* G row */
      g_inv_coeffs[i + 0 * 3] = cv_round_d(lshift * XYZ2sRGB[i + 2 * 3] * D65[i]); /* B row */
    }
  }
  /* HSV tables (color_hsv.cpp RGB2HSV_b) */
  g_sdiv[0] = g_hdiv180[0] = 0;
  for (int i = 1; i < 256; i++) {
    g_sdiv[i] = cv_round_d((255 << 12) / (1. * i));
    g_hdiv180[i] = cv_round_d((180 << 12) / (6. * i));
  }
  g_tabs_ready = 1;
}

int ripo_ab_to_xz(int i) {
  /* abToXZ_b[i - minABvalue] */
  const int BASE = LAB_BASE;
  if (i <= 3390) return i * 108 / 841 - BASE * 16 / 116 * 108 / 841;
  return i * i / BASE * i / BASE;
}

int ripo_table(int which, int32_t* out, int cap) {
  init_tables();
  int n = 0;
#define COPY(tab, cnt)                                   \
  n = (cnt);                                             \
  for (int i = 0; i < n && i < cap; i++) out[i] = (int32_t)(tab)[i];
  switch (which) {
    case 0: COPY(g_srgb_gamma_b, 256); break;
    case 1: COPY(g_cbrt_b, LAB_CBRT_TAB_SIZE_B); break;
    case 2: COPY(g_lab_to_yf_b, 512); break;
    case 3: COPY(g_inv_gamma_b, INV_GAMMA_TAB_SIZE); break;
    case 4: COPY(g_fwd_coeffs, 9); break;
    case 5: COPY(g_inv_coeffs, 9); break;
    case 6: COPY(g_sdiv, 256); break;
    case 7: COPY(g_hdiv180, 256); break;
    default: return -1;
  }
#undef COPY
  return n;
}

static inline void bgr2lab_px(const uint8_t* s, uint8_t* d) {
  const int Lscale = (116 * 255 + 50) / 100;
  const int Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
  const int* C = g_fwd_coeffs;
  int v0 = g_srgb_gamma_b[s[0]], v1 = g_srgb_gamma_b[s[1]], v2 = g_srgb_gamma_b[s[2]];
  int fX = g_cbrt_b[CV_DESCALE(v0 * C[0] + v1 * C[1] + v2 * C[2], LAB_SHIFT)];
  int fY = g_cbrt_b[CV_DESCALE(v0 * C[3] + v1 * C[4] + v2 * C[5], LAB_SHIFT)];
  int fZ = g_cbrt_b[CV_DESCALE(v0 * C[6] + v1 * C[7] + v2 * C[8], LAB_SHIFT)];
  int L = CV_DESCALE(Lscale * fY + Lshift, LAB_SHIFT2);
  int a = CV_DESCALE(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
  int b = CV_DESCALE(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
  d[0] = sat_u8_i(L);
  d[1] = sat_u8_i(a);
  d[2] = sat_u8_i(b);
}

static inline void lab2bgr_px(const uint8_t* s, uint8_t* d) {
  const int BASE = LAB_BASE;
  const int shift = LAB_SHIFT + (14 - 12); /* lab_shift + (base_shift - inv_gamma_shift) */
  int LL = s[0], aa = s[1], bb = s[2];
  int y = g_lab_to_yf_b[LL * 2];
  int ify = g_lab_to_yf_b[LL * 2 + 1];
  int adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
  int bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
  int x = ripo_ab_to_xz(ify + adiv);
  int z = ripo_ab_to_xz(ify - bdiv);
  const int* C = g_inv_coeffs;
  int bo = CV_DESCALE(C[0] * x + C[1] * y + C[2] * z, shift);
  int go = CV_DESCALE(C[3] * x + C[4] * y + C[5] * z, shift);
  int ro = CV_DESCALE(C[6] * x + C[7] * y + C[8] * z, shift);
  bo = imax(0, imin(INV_GAMMA_TAB_SIZE - 1, bo));
  go = imax(0, imin(INV_GAMMA_TAB_SIZE - 1, go));
  ro = imax(0, imin(INV_GAMMA_TAB_SIZE - 1, ro));
  d[0] = sat_u8_i(g_inv_gamma_b[bo]);
  d[1] = sat_u8_i(g_inv_gamma_b[go]);
  d[2] = sat_u8_i(g_inv_gamma_b[ro]);
}

void ripo_bgr2lab(const uint8_t* bgr, size_t npix, uint8_t* lab) {
  init_tables();
  for (size_t i = 0; i < npix; i++) bgr2lab_px(bgr + i * 3, lab + i * 3);
}
void ripo_lab2bgr(const uint8_t* lab, size_t npix, uint8_t* bgr) {
  init_tables();
  for (size_t i = 0; i < npix; i++) lab2bgr_px(lab + i * 3, bgr + i * 3);
}

/* ------------------------------------------------------------------------------------
 * Vignetting -- vignetting_correction.cpp:32-63 (mask) and :68-93 (apply).
 * correct() passes (cols, rows) into (height, width); the mask is created (width, height)
 * and written at(x, y): the double swap cancels and mask(row, col) has
 * r^2 = (col - cols/2.0)^2 + (row - rows/2.0)^2.  Then mask = mask/max (x (float)(1/max)),
 * x (float)scale, + 1.0f -- three float roundings as cv::Mat arithmetic performs them.
 * ---------------------------------------------------------------------------------- */
void ripo_vignetting_mask(int rows, int cols, double scale, double a2, double a4, float* mask) {
  int height = cols, width = rows; /* as received by precomputeVignettingMask */
  double cx = width / 2.0, cy = height / 2.0;
  float mx = -FLT_MAX;
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      double r = sqrt(pow(y - cy, 2) + pow(x - cx, 2));
      double k = pow(r, 2) * a2 + pow(r, 4) * a4;
      if (g_fp_contract) { /* the reference's own TU: pow(., 2) is a multiply, both sums of products contract */
        double dy = y - cy, dx = x - cx;
        r = sqrt(fma(dy, dy, dx * dx));
        k = fma(pow(r, 2), a2, pow(r, 4) * a4);
      }
      float kf = (float)k;
      mask[(size_t)x * height + y] = kf; /* at<float>(x, y): row x, col y */
      if (kf > mx) mx = kf;
    }
  double maxv = (double)mx;
  size_t n = (size_t)rows * cols;
  if (maxv > 0) {
    float inv = (float)(1.0 / maxv);
    for (size_t i = 0; i < n; i++) mask[i] = mask[i] * inv;
  }
  float sc = (float)scale;
  for (size_t i = 0; i < n; i++) mask[i] = mask[i] * sc;
  for (size_t i = 0; i < n; i++) mask[i] = mask[i] + 1.0f;
}

void ripo_vignetting(uint8_t* bgr, int rows, int cols, const float* mask) {
  init_tables();
  size_t n = (size_t)rows * cols;
  for (size_t i = 0; i < n; i++) {
    uint8_t lab[3];
    bgr2lab_px(bgr + i * 3, lab);
    lab[0] = sat_u8_f((float)lab[0] * mask[i]);
    lab2bgr_px(lab, bgr + i * 3);
  }
}

/* ------------------------------------------------------------------------------------
 * Colour enhancer -- color_enhancer.cpp:38-47.  cvtColor(BGR2HSV) 8-bit, H in [0,180)
 * (imgproc/color_hsv.cpp RGB2HSV_b), cv::multiply(8UC3, Scalar) in float with u8
 * saturation (H is not wrapped), cvtColor(HSV2BGR) (HSV2RGB_b -> float HSV2RGB_native).
 * ---------------------------------------------------------------------------------- */
static inline void bgr2hsv_px(const uint8_t* s, uint8_t* d) {
  const int hsv_shift = 12;
  int b = s[0], g = s[1], r = s[2];
  int h, sat, v = b, vmin = b;
  v = imax(v, g);
  v = imax(v, r);
  vmin = imin(vmin, g);
  vmin = imin(vmin, r);
  int diff = v - vmin;
  int vr = v == r ? -1 : 0;
  int vg = v == g ? -1 : 0;
  sat = (diff * g_sdiv[v] + (1 << (hsv_shift - 1))) >> hsv_shift;
  h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  h = (h * g_hdiv180[diff] + (1 << (hsv_shift - 1))) >> hsv_shift;
  h += h < 0 ? 180 : 0;
  d[0] = sat_u8_i(h);
  d[1] = (uint8_t)sat;
  d[2] = (uint8_t)v;
}

static inline void hsv2bgr_px(const uint8_t* s, uint8_t* d) {
  static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
  const float hscale = 6.f / 180.f;
  float h = s[0], sa = s[1] * (1.f / 255.f), v = s[2] * (1.f / 255.f);
  float b, g, r;
  if (sa == 0)
    b = g = r = v;
  else {
    float tab[4];
    h *= hscale;
    h = fmodf(h, 6.f);
    int sector = cv_floor_f(h);
    h -= sector;
    if ((unsigned)sector >= 6u) {
      sector = 0;
      h = 0.f;
    }
    tab[0] = v;
    tab[1] = v * (1.f - sa);
    tab[2] = v * (1.f - sa * h);
    tab[3] = v * (1.f - sa * (1.f - h));
    if (g_fp_contract) { /* 1 - s*h and 1 - s*(1 - h) each become one fnma */
      tab[2] = v * fmaf(-sa, h, 1.f);
      tab[3] = v * fmaf(-sa, 1.f - h, 1.f);
    }
    b = tab[sector_data[sector][0]];
    g = tab[sector_data[sector][1]];
    r = tab[sector_data[sector][2]];
  }
  d[0] = sat_u8_f(b * 255.f);
  d[1] = sat_u8_f(g * 255.f);
  d[2] = sat_u8_f(r * 255.f);
}

void ripo_bgr2hsv(const uint8_t* bgr, size_t npix, uint8_t* hsv) {
  init_tables();
  for (size_t i = 0; i < npix; i++) bgr2hsv_px(bgr + i * 3, hsv + i * 3);
}
void ripo_hsv2bgr(const uint8_t* hsv, size_t npix, uint8_t* bgr) {
  for (size_t i = 0; i < npix; i++) hsv2bgr_px(hsv + i * 3, bgr + i * 3);
}
void ripo_color_enhance(uint8_t* bgr, size_t npix, double hg, double sg, double vg) {
  init_tables();
  float fh = (float)hg, fs = (float)sg, fv = (float)vg;
  for (size_t i = 0; i < npix; i++) {
    uint8_t hsv[3];
    bgr2hsv_px(bgr + i * 3, hsv);
    hsv[0] = sat_u8_f((float)hsv[0] * fh);
    hsv[1] = sat_u8_f((float)hsv[1] * fs);
    hsv[2] = sat_u8_f((float)hsv[2] * fv);
    hsv2bgr_px(hsv, bgr + i * 3);
  }
}

/* ------------------------------------------------------------------------------------
 * Fisheye maps -- undistortion.cpp:197-220 -> calib3d/src/fisheye.cpp (4.2.0), all double.
 * ---------------------------------------------------------------------------------- */
static void mat3_mul(const double a[9], const double b[9], double c[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
/* (PP*RR).inv(DECOMP_SVD): OpenCV inverts through a Jacobi SVD; restated as the adjugate
 * inverse (agrees to a few double ulps; map values are narrowed to float afterwards). */
static void mat3_inv(const double m[9], double o[9]) {
  double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  double id = 1.0 / det;
  o[0] = c00 * id;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

static void fisheye_undistort_point(const double K[9], const double D[4], const double R[9], double px,
                                    double py, double out[2]) {
  double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  double pwx = (px - cx) / fx, pwy = (py - cy) / fy;
  double scale = 1.0;
  double theta_d = sqrt(pwx * pwx + pwy * pwy);
  const double PI = 3.1415926535897932384626433832795;
  theta_d = fmin(fmax(-PI / 2., theta_d), PI / 2.);
  if (theta_d > 1e-8) {
    double theta = theta_d;
    const double EPS = 1e-8;
    for (int j = 0; j < 10; j++) {
      double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
      double k0 = D[0] * t2, k1 = D[1] * t4, k2 = D[2] * t6, k3 = D[3] * t8;
      double fix = (theta * (1 + k0 + k1 + k2 + k3) - theta_d) / (1 + 3 * k0 + 5 * k1 + 7 * k2 + 9 * k3);
      theta = theta - fix;
      if (fabs(fix) < EPS) break;
    }
    scale = tan(theta) / theta_d;
  }
  double pux = pwx * scale, puy = pwy * scale;
  double prx = R[0] * pux + R[1] * puy + R[2];
  double pry = R[3] * pux + R[4] * puy + R[5];
  double prz = R[6] * pux + R[7] * puy + R[8];
  out[0] = prx / prz;
  out[1] = pry / prz;
}

/* estimateNewCameraMatrixForUndistortRectify, 4.2.0 (including its `cn[0] *= aspect_ratio`,
 * which later releases changed to cn[1]). */
void ripo_fisheye_new_camera_matrix(const double K[9], const double D[4], int w, int h, const double R[9],
                                    double balance, int new_w, int new_h, double fov_scale, double newK[9]) {
  balance = fmin(fmax(balance, 0.0), 1.0);
  double pts[4][2] = {{(double)(w / 2), 0}, {(double)w, (double)(h / 2)}, {(double)(w / 2), (double)h}, {0, (double)(h / 2)}};
  for (int i = 0; i < 4; i++) fisheye_undistort_point(K, D, R, pts[i][0], pts[i][1], pts[i]);
  double cn[2] = {0, 0};
  for (int i = 0; i < 4; i++) {
    cn[0] += pts[i][0];
    cn[1] += pts[i][1];
  }
  cn[0] /= 4; /* cv::mean */
  cn[1] /= 4;
  double aspect = K[0] / K[4];
  cn[0] *= aspect;
  for (int i = 0; i < 4; i++) pts[i][1] *= aspect;
  double minx = DBL_MAX, miny = DBL_MAX, maxx = -DBL_MAX, maxy = -DBL_MAX;
  for (int i = 0; i < 4; i++) {
    miny = fmin(miny, pts[i][1]);
    maxy = fmax(maxy, pts[i][1]);
    minx = fmin(minx, pts[i][0]);
    maxx = fmax(maxx, pts[i][0]);
  }
  double f1 = w * 0.5 / (cn[0] - minx);
  double f2 = w * 0.5 / (maxx - cn[0]);
  double f3 = h * 0.5 * aspect / (cn[1] - miny);
  double f4 = h * 0.5 * aspect / (maxy - cn[1]);
  double fmn = fmin(f1, fmin(f2, fmin(f3, f4)));
  double fmx = fmax(f1, fmax(f2, fmax(f3, f4)));
  double f = balance * fmn + (1.0 - balance) * fmx;
  f *= fov_scale > 0 ? 1.0 / fov_scale : 1.0;
  double nf[2] = {f, f};
  double nc[2] = {-cn[0] * f + w * 0.5, -cn[1] * f + (h * aspect) * 0.5};
  nf[1] /= aspect;
  nc[1] /= aspect;
  if (new_w > 0 && new_h > 0) {
    double rx = new_w / (double)w, ry = new_h / (double)h;
    nf[0] *= rx;
    nf[1] *= ry;
    nc[0] *= rx;
    nc[1] *= ry;
  }
  double o[9] = {nf[0], 0, nc[0], 0, nf[1], nc[1], 0, 0, 1};
  memcpy(newK, o, sizeof(o));
}

/* initUndistortRectifyMap(K, D, R, P, size, CV_32F): incremental _x += iR(0,0) per pixel */
void ripo_fisheye_maps(const double K[9], const double D[4], const double R[9], const double P[9], int w,
                       int h, float* map_x, float* map_y) {
  double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  double PR[9], iR[9];
  mat3_mul(P, R, PR);
  mat3_inv(PR, iR);
  for (int i = 0; i < h; i++) {
    double _x = i * iR[1] + iR[2], _y = i * iR[4] + iR[5], _w = i * iR[7] + iR[8];
    for (int j = 0; j < w; j++) {
      double x = _x / _w, y = _y / _w;
      double r = sqrt(x * x + y * y);
      double theta = atan(r);
      double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      double theta_d = theta * (1 + D[0] * t2 + D[1] * t4 + D[2] * t6 + D[3] * t8);
      double scale = (r == 0) ? 1.0 : theta_d / r;
      double u = fx * x * scale + cx;
      double v = fy * y * scale + cy;
      map_x[(size_t)i * w + j] = (float)u;
      map_y[(size_t)i * w + j] = (float)v;
      _x += iR[0];
      _y += iR[3];
      _w += iR[6];
    }
  }
}

/* ------------------------------------------------------------------------------------
 * Remap -- undistortion.cpp:240-245 -> cv::remap(INTER_LINEAR, BORDER_CONSTANT, 0)
 * (imgproc/imgwarp.cpp): coordinates quantised to 1/32 px (cvRound(map*32)), integer part
 * saturated to int16, Q15 weights 32(32-fx)(32-fy)..., (sum + 2^14) >> 15.
 * ---------------------------------------------------------------------------------- */
static inline int cv_round_map(float v) {
  /* cvRound on x86 returns INT_MIN for NaN/inf/out-of-range */
  float s = v * 32.f;
  if (!(s > -2147483648.f && s < 2147483648.f)) return INT_MIN;
  return (int)lrintf(s);
}

void ripo_remap_linear(const uint8_t* src, int rows, int cols, int cn, const float* map_x, const float* map_y,
                       int drows, int dcols, uint8_t* dst) {
  for (int dy = 0; dy < drows; dy++)
    for (int dx = 0; dx < dcols; dx++) {
      size_t di = (size_t)dy * dcols + dx;
      int sxq = cv_round_map(map_x[di]), syq = cv_round_map(map_y[di]);
      int sx = sat_s16_i(sxq >> 5), sy = sat_s16_i(syq >> 5);
      int fx = sxq & 31, fy = syq & 31;
      int w00 = 32 * (32 - fx) * (32 - fy), w01 = 32 * fx * (32 - fy), w10 = 32 * (32 - fx) * fy, w11 = 32 * fx * fy;
      uint8_t* D = dst + di * cn;
      if (sx >= cols || sx + 1 < 0 || sy >= rows || sy + 1 < 0) {
        memset(D, 0, (size_t)cn);
        continue;
      }
      int x0 = sx >= 0 && sx < cols, x1 = sx + 1 >= 0 && sx + 1 < cols;
      int y0 = sy >= 0 && sy < rows, y1 = sy + 1 >= 0 && sy + 1 < rows;
      for (int c = 0; c < cn; c++) {
        int p00 = (x0 && y0) ? src[((size_t)sy * cols + sx) * cn + c] : 0;
        int p01 = (x1 && y0) ? src[((size_t)sy * cols + sx + 1) * cn + c] : 0;
        int p10 = (x0 && y1) ? src[((size_t)(sy + 1) * cols + sx) * cn + c] : 0;
        int p11 = (x1 && y1) ? src[((size_t)(sy + 1) * cols + sx + 1) * cn + c] : 0;
        D[c] = sat_u8_i((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
      }
    }
}

/* ------------------------------------------------------------------------------------
 * Whole chain -- raw_image_pipeline.hpp:143-172.
 * ---------------------------------------------------------------------------------- */
int ripo_pipeline(const ripo_params* p, ripo_ccc* ccc, const uint8_t* in, int rows, int cols, int cn,
                  const char* encoding, uint8_t* out, int* out_rows, int* out_cols, int* out_cn,
                  char enc_out[32], uint8_t* tap_debayered, uint8_t* tap_color) {
  int pattern = ripo_bayer_pattern(encoding);
  int is16 = !strcmp(encoding, "bayer_bggr16") || !strcmp(encoding, "bayer_gbrg16") ||
             !strcmp(encoding, "bayer_grbg16") || !strcmp(encoding, "bayer_rggb16");
  if (is16) return -1; /* debayer.cpp:76-78 (and quirk Q3 fixed: all four 16-bit names) */
  size_t npix = (size_t)rows * cols;
  uint8_t* img = (uint8_t*)malloc(npix * 3 > npix * (size_t)cn ? npix * 3 : npix * (size_t)cn);
  uint8_t* tmp = (uint8_t*)malloc(npix * 3 > npix * (size_t)cn ? npix * 3 : npix * (size_t)cn);
  uint8_t* keep = p->reference_schedule ? (uint8_t*)malloc(npix * 3 > npix * (size_t)cn ? npix * 3 : npix * (size_t)cn) : NULL;
  int ch = cn;
  strncpy(enc_out, encoding, 31);
  enc_out[31] = 0;
  /* 1. debayer (debayer.hpp:36-47) */
  if (pattern >= 0 && cn == 1) {
    if (p->reference_schedule) {
      /* literally as debayer.cpp:48-70: demosaic with R/B exchanged (the reference's
       * BayerXX code names the opposite pattern), then cvtColor(RGB2BGR) swaps them back */
      ripo_debayer_bilinear(in, rows, cols, 3 - pattern, tmp);
      ripo_swap_rb(tmp, npix, img);
    } else {
      ripo_debayer_bilinear(in, rows, cols, pattern, img);
    }
    ch = 3;
    strcpy(enc_out, "bgr8");
  } else if (!strcmp(encoding, "rgb8") && cn == 3) {
    ripo_swap_rb(in, npix, img); /* encoding string stays "rgb8" on the CPU path (:72-73) */
  } else {
    memcpy(img, in, npix * (size_t)cn);
  }
  if (keep) memcpy(keep, img, npix * (size_t)ch); /* saveDebayeredImage :81-83 */
  int r = rows, c = cols;
  /* 2. flip (flip.hpp:36-45) */
  if (p->flip_enabled && (p->flip_angle == 90 || p->flip_angle == 180 || p->flip_angle == 270)) {
    ripo_flip(img, r, c, ch, p->flip_angle, tmp, &r, &c);
    uint8_t* t = img;
    img = tmp;
    tmp = t;
  }
  if (tap_debayered) memcpy(tap_debayered, img, npix * (size_t)ch); /* saveFlippedImage :60-62 */
  else if (keep) memcpy(keep, img, npix * (size_t)ch);
  /* 3. white balance (white_balance.hpp:45-86) */
  if (p->wb_enabled && ch == 3) {
    if (p->wb_method == 1)
      ripo_wb_grayworld(img, npix, p->wb_bright_thr, NULL, NULL);
    else if (p->wb_method == 4)
      ripo_wb_pca(img, npix, NULL);
    else if (p->wb_method == 0)
      ripo_wb_simple(img, npix, p->wb_percentile, NULL);
    else if (p->wb_method == 3) {
      if (!ccc) {
        free(img); free(tmp); free(keep);
        return -3;
      }
      ripo_ccc_set_thresholds(ccc, (float)p->wb_bright_thr, (float)p->wb_dark_thr);
      ripo_ccc_set_temporal_consistency(ccc, p->wb_temporal_consistency);
      ripo_ccc_balance(ccc, img, r, c, NULL, NULL);
    } else {
      free(img); free(tmp); free(keep);
      return -4; /* learned: needs the model compiled into opencv_contrib */
    }
  }
  /* 4. colour calibration (color_calibration.hpp:42-56) */
  if (p->cc_enabled && ch == 3 && p->cc_available) ripo_color_matrix(img, npix, p->cc_matrix, p->cc_bias);
  /* 5. gamma (gamma_correction.hpp:32-43): "default" == "custom" on the CPU path */
  if (p->gamma_enabled) {
    uint8_t lut[256];
    ripo_gamma_lut(p->gamma_k, lut);
    ripo_apply_lut(img, npix * (size_t)ch, lut);
  }
  /* 6. vignetting (vignetting_correction.hpp:26-33) */
  if (p->vig_enabled) {
    if (ch != 3) {
      free(img); free(tmp); free(keep);
      return -2; /* cvtColor(BGR2Lab) asserts on non-3-channel input */
    }
    if (p->vig_mask && !p->reference_schedule) {
      ripo_vignetting(img, r, c, p->vig_mask);
    } else {
      /* the reference rebuilds the mask every frame whenever W != H (Q6) */
      float* mask = (float*)malloc(npix * sizeof(float));
      ripo_vignetting_mask(r, c, p->vig_scale, p->vig_a2, p->vig_a4, mask);
      ripo_vignetting(img, r, c, mask);
      free(mask);
    }
  }
  /* 7. colour enhancer (color_enhancer.hpp:33-43) */
  if (p->ce_enabled && ch == 3) ripo_color_enhance(img, npix, p->ce_h_gain, p->ce_s_gain, p->ce_v_gain);
  /* 8. undistortion (undistortion.hpp:66-80): copy kept first, always */
  if (tap_color) memcpy(tap_color, img, npix * (size_t)ch);
  else if (keep) memcpy(keep, img, npix * (size_t)ch);
  if (p->und_enabled && p->map_x && p->map_y) {
    ripo_remap_linear(img, r, c, ch, p->map_x, p->map_y, p->map_rows, p->map_cols, tmp);
    r = p->map_rows;
    c = p->map_cols;
    uint8_t* t = img;
    img = tmp;
    tmp = t;
  }
  /* 9. saveOutput (raw_image_pipeline.hpp:174-177) */
  if (keep) memcpy(keep, img, (size_t)r * c * ch);
  memcpy(out, img, (size_t)r * c * ch);
  *out_rows = r;
  *out_cols = c;
  *out_cn = ch;
  free(img);
  free(tmp);
  free(keep);
  return 0;
}
