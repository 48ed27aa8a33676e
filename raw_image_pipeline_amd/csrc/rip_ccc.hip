// rip_ccc.hip -- convolutional colour constancy estimator: log-chroma histogram, 256 x 256 FFT convolution with the
// model, arg-max (convolutional_color_constancy.cpp:91-340).
// Shared device code and the stage-by-stage reference citations: rip_device.hpp.
#include "rip_device.hpp"
#include <atomic>

namespace rip {
namespace {

// ------------------------------------------------------------------------------------------------
// ccc estimator: resize-sample -> log-chroma histogram -> FFT convolution -> argmax
// ------------------------------------------------------------------------------------------------
// colour of the post-flip image at (yd, xd)
__device__ __forceinline__ void fetch_flipped(const SrcView& s, int angle, int yd, int xd, int& b, int& g, int& r) {
  int ys, xs;
  unflip(angle, s.rows, s.cols, yd, xd, ys, xs);
  fetch_src(s, ys, xs, b, g, r);
}

// Demosaic of the 2x2 block whose top-left pixel is (y, x), all four pixels interior (no border rule):
// the 4x4 raw window around it is read with two aligned dwords per row instead of nine byte loads per
// pixel.  Every 2x2 block holds one site of each kind; out[i * 2 + j] = pixel (y + i, x + j) as b, g, r.
// lo / hi: the two aligned dwords that hold bytes x - 1 .. x + 2 of rows y - 1 .. y + 2; sh: byte offset of x - 1 in lo
__device__ __forceinline__ void debayer_block2x2_win(const SrcView& s, int y, int x, const uint32_t (&lo)[4], const uint32_t (&hi)[4], unsigned sh,
                                                     int (&out)[4][3]) {
  int v[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t w = __builtin_amdgcn_alignbyte(hi[r], lo[r], sh);
#pragma unroll
    for (int c = 0; c < 4; c++) v[r][c] = (int)((w >> (8 * c)) & 0xFFu);
  }
  const int py = (y - s.ry) & 1, px = (x - s.rx) & 1;  // parity of the block's top-left pixel: (0,0) = R site
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int cy = 1 + i, cx = 1 + j, dy = py ^ i, dx = px ^ j;
      const int c = v[cy][cx];
      const int h = (v[cy][cx - 1] + v[cy][cx + 1] + 1) >> 1, vv = (v[cy - 1][cx] + v[cy + 1][cx] + 1) >> 1;
      const int x4 = (v[cy][cx - 1] + v[cy][cx + 1] + v[cy - 1][cx] + v[cy + 1][cx] + 2) >> 2;
      const int d4 = (v[cy - 1][cx - 1] + v[cy - 1][cx + 1] + v[cy + 1][cx - 1] + v[cy + 1][cx + 1] + 2) >> 2;
      int b, g, r;
      if (dy != dx) {  // green site
        g = c;
        r = dy == 0 ? h : vv;
        b = dy == 0 ? vv : h;
      } else {
        g = x4;
        r = dy == 0 ? c : d4;
        b = dy == 0 ? d4 : c;
      }
      out[i * 2 + j][0] = b;
      out[i * 2 + j][1] = g;
      out[i * 2 + j][2] = r;
    }
}

__device__ __forceinline__ void debayer_block2x2(const SrcView& s, int y, int x, int (&out)[4][3]) {
  uint32_t lo[4], hi[4];
  unsigned sh = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const unsigned off = __umul24((unsigned)(y - 1 + r), (unsigned)s.step) + (unsigned)(x - 1);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(s.base + (off & ~3u));
    lo[r] = q[0];
    hi[r] = q[1];
    sh = off & 3u;  // step % 4 == 0 on this path: the same for the four rows
  }
  debayer_block2x2_win(s, y, x, lo, hi, sh, out);
}

// The same 2x2 block in packed form.  The 4x4 window is four dwords (bytes = columns x - 1 .. x + 2); every candidate value
// of the two block columns -- centre, horizontal / vertical two-tap, cross and diagonal four-tap averages -- is evaluated for
// both columns at once with v_lerp_u8 (the identities of the fused chain's demosaic, rip_device.hpp), and the site kinds,
// which differ from lane to lane with the block's parity, pick bytes with v_cndmask / v_perm.
// pair[i][c]: channel c (b, g, r) of block row i as two 16-bit fields, left pixel | right pixel << 16.
__device__ __forceinline__ void debayer_block2x2_pairs(const SrcView& s, int y, int x, const uint32_t (&lo)[4], const uint32_t (&hi)[4],
                                                       unsigned sh, uint32_t (&pair)[2][3]) {
  constexpr uint32_t kOnes = 0x01010101u;
  uint32_t W[4], H[4], NXH[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    W[r] = __builtin_amdgcn_alignbyte(hi[r], lo[r], sh);
    const uint32_t left = W[r] << 8, right = W[r] >> 8;  // byte j: column j - 1 / j + 1 of the window
    H[r] = __builtin_amdgcn_lerp(left, right, kOnes);
    NXH[r] = ~(left ^ right);
  }
  const bool py = ((y - s.ry) & 1) != 0, px = ((x - s.rx) & 1) != 0;  // parity of the block's top-left pixel: (0, 0) = R site
  // bytes 1 and 2 of the sources go to the low halves of the two 16-bit fields; with px the columns swap site kinds
  const uint32_t sel = px ? 0x0c020c05u : 0x0c060c01u;  // perm(S0 = odd-column kind, S1 = even-column kind)
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const uint32_t C = W[i + 1];
    const uint32_t V = __builtin_amdgcn_lerp(W[i], W[i + 2], kOnes);
    const uint32_t X4 = __builtin_amdgcn_lerp(H[i + 1], V, NXH[i + 1] & ~(W[i] ^ W[i + 2]));
    const uint32_t D4 = __builtin_amdgcn_lerp(H[i], H[i + 2], NXH[i] & NXH[i + 2]);
    const bool dy = py != (i != 0);  // red row: dy == 0
    // value at an even-parity column (dx == 0) / odd-parity column (dx == 1) of this row
    const uint32_t r0 = dy ? V : C, r1 = dy ? D4 : H[i + 1];
    const uint32_t b0 = dy ? H[i + 1] : D4, b1 = dy ? C : V;
    const uint32_t g0 = dy ? C : X4, g1 = dy ? X4 : C;
    pair[i][0] = __builtin_amdgcn_perm(b1, b0, sel);
    pair[i][1] = __builtin_amdgcn_perm(g1, g0, sel);
    pair[i][2] = __builtin_amdgcn_perm(r1, r0, sel);
  }
}

// the 2x2 block of post-flip pixels at (y, x): fast window path for unflipped Bayer frames, else per pixel
__device__ __forceinline__ void fetch_block2x2(const SrcView& s, int angle, int y, int x, int y1, int x1, int (&out)[4][3]) {
  const bool fast = s.kind == SRC_BAYER && angle == 0 && y1 == y + 1 && x1 == x + 1 && y >= 1 && y1 <= s.rows - 2 && x >= 1 &&
                    x1 <= s.cols - 2 && ((unsigned)(x + 6) < (unsigned)s.step || y + 2 < s.rows - 1) && (reinterpret_cast<uintptr_t>(s.base) & 3u) == 0 &&
                    (s.step & 3u) == 0 && (unsigned long long)s.step * (unsigned long long)s.rows < (1ull << 32) && s.step < (1u << 24);
  if (fast) {
    debayer_block2x2(s, y, x, out);
    return;
  }
  fetch_flipped(s, angle, y, x, out[0][0], out[0][1], out[0][2]);
  fetch_flipped(s, angle, y, x1, out[1][0], out[1][1], out[1][2]);
  fetch_flipped(s, angle, y1, x, out[2][0], out[2][1], out[2][2]);
  fetch_flipped(s, angle, y1, x1, out[3][0], out[3][1], out[3][2]);
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// Samples per frame: 360 x 270 (small_size_, convolutional_color_constancy.cpp:22).  The resize geometry and the log
// table live in LDS (ten table reads per sample otherwise go to L2).
struct CccSampleTabs {
  int xofs[360], yofs[540];
  short ialpha[720], ibeta[540];
  float logt[256];
  template <int NT>
  __device__ __forceinline__ void load(const CccParams& p) {
    for (int i = threadIdx.x; i < 360; i += NT) xofs[i] = p.geom.xofs[i];
    for (int i = threadIdx.x; i < 540; i += NT) {
      yofs[i] = p.geom.yofs[i];
      ibeta[i] = p.geom.ibeta[i];
    }
    for (int i = threadIdx.x; i < 720; i += NT) ialpha[i] = p.geom.ialpha[i];
    for (int i = threadIdx.x; i < 256; i += NT) logt[i] = p.tabs->log_tab[i];
  }
};

// Grey mask and log-chroma bin of one resized sample (b, g, r); -1 when the sample is masked out
// (calculateHistogramFeature :210-271).  Branch-free: straight-line callers keep several samples in flight.
__device__ __forceinline__ int ccc_bin_of(const CccParams& p, const CccSampleTabs& tb, const int (&sm)[3]) {
  float fb = (float)sm[0], fg = (float)sm[1], fr = (float)sm[2];
  float gray = fb * 0.114f + fg * 0.587f + fr * 0.299f;
  bool ok = !(gray > p.upper) && (gray > p.lower);
  if (sm[0] == 0 || sm[1] == 0 || sm[2] == 0) ok = false;  // log(0) = -inf is skipped
  const float bin_size = 1.0f / 64.0f, uv0 = -1.421875f;
  float lb = tb.logt[sm[0]], lg = tb.logt[sm[1]], lr = tb.logt[sm[2]];
  int u = (int)roundf((lg - lr - uv0) / bin_size);
  int v = (int)roundf((lg - lb - uv0) / bin_size);
  u = clampi(u, 0, 255);
  v = clampi(v, 0, 255);
  return ok ? u * 256 + v : -1;
}

// Sample i of the 360 x 270 grid: cv::resize tap arithmetic, grey mask, log-chroma bin.  Returns the bin index
// u * 256 + v (hist.at(u, v), :260), or -1 when the sample is masked out (calculateHistogramFeature :210-271).
__device__ __forceinline__ int ccc_sample_bin(const CccParams& p, const CccSampleTabs& tb, const SrcView& s, int i) {
  const int dy = i / 360, dx = i - dy * 360;
  int sm[3];
  int t[4][3];  // taps (y0,x0) (y0,x1) (y1,x0) (y1,x1)
  if (p.geom.area_fast) {
    fetch_block2x2(s, p.flip_angle, 2 * dy, 2 * dx, 2 * dy + 1, 2 * dx + 1, t);
#pragma unroll
    for (int c = 0; c < 3; c++) sm[c] = (t[0][c] + t[1][c] + t[2][c] + t[3][c] + 2) >> 2;
  } else {
    // cv::resize INTER_LINEAR, 8U: Q11 coefficients, two-pass integer arithmetic
    const int sx = tb.xofs[dx];
    const int sx1 = sx + 1 < p.dcols ? sx + 1 : sx;
    const int a0 = tb.ialpha[dx * 2], a1 = tb.ialpha[dx * 2 + 1];
    const int y0 = tb.yofs[dy * 2], y1 = tb.yofs[dy * 2 + 1];
    const int b0 = tb.ibeta[dy * 2], b1 = tb.ibeta[dy * 2 + 1];
    fetch_block2x2(s, p.flip_angle, y0, sx, y1, sx1, t);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      int r0 = t[0][c] * a0 + t[1][c] * a1;
      int r1 = t[2][c] * a0 + t[3][c] * a1;
      sm[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    }
  }
  return ccc_bin_of(p, tb, sm);
}

// K samples of one lane in straight-line code.  ccc_sample_bin branches per sample (resize mode, window path, mask), so the
// compiler cannot start a sample's loads before the previous sample is finished and every sample pays its own chain of
// latencies (table lookups -> eight window loads -> log table -> histogram atomic: the LDS-histogram kernel sat in s_waitcnt
// for 64 % of its wave cycles).  Here the geometry of all K samples is looked up first and checked once for the wave; when
// every lane's K blocks are interior 2x2 blocks of an unflipped, dword-aligned Bayer frame (every sample of the usual
// geometries) the 8 K loads are issued together and the arithmetic follows -- same operations, same results.
struct CccSampleGeom {
  int y, x, b0, b1;   // top-left pixel of the 2x2 block; Q11 row weights (unused by the exact-2x area mode)
  uint32_t alpha2;    // Q11 column weights, left | right << 16
  bool straight;             // interior 2x2 block with y1 == y + 1, x1 == x + 1
};
__device__ __forceinline__ CccSampleGeom ccc_sample_geom(const CccParams& p, const CccSampleTabs& tb, const SrcView& s, int i) {
  const int dy = i / 360, dx = i - dy * 360;
  CccSampleGeom g;
  int y1, x1;
  if (p.geom.area_fast) {  // uniform
    g.y = 2 * dy;
    g.x = 2 * dx;
    y1 = g.y + 1;
    x1 = g.x + 1;
    g.b0 = g.b1 = 0;
    g.alpha2 = 0;
  } else {
    g.x = tb.xofs[dx];
    x1 = g.x + 1 < p.dcols ? g.x + 1 : g.x;
    g.alpha2 = reinterpret_cast<const uint32_t*>(tb.ialpha)[dx];  // ialpha[2 dx], ialpha[2 dx + 1]
    g.y = tb.yofs[dy * 2];
    y1 = tb.yofs[dy * 2 + 1];
    g.b0 = tb.ibeta[dy * 2];
    g.b1 = tb.ibeta[dy * 2 + 1];
  }
  const bool block = y1 == g.y + 1 && x1 == g.x + 1;
  if (p.flip_angle == 180) {  // uniform
    // The estimator samples the image AFTER the flip (white_balance follows flip in pipeline<T>(), raw_image_pipeline.hpp:
    // 143-177).  Post-flip pixel (y, x) is source pixel (rows - 1 - y, cols - 1 - x), so the 2x2 block {y, y + 1} x {x, x + 1}
    // is the source block whose top-left pixel is (rows - 2 - y, cols - 2 - x), read upside down and mirrored: the row weights
    // and the two column weights change places, the demosaic runs on source coordinates (its parities are the frame's).
    g.y = s.rows - 2 - g.y;
    g.x = s.cols - 2 - g.x;
    const int t = g.b0;
    g.b0 = g.b1;
    g.b1 = t;
    g.alpha2 = (g.alpha2 >> 16) | (g.alpha2 << 16);
  }
  // the two dwords of a window row end at most at byte x + 6 of that row: past the row's end that is the next row of the
  // frame, except in the frame's last row (y + 2 == rows - 1), where it could leave the buffer
  g.straight = block && g.y >= 1 && g.y + 1 <= s.rows - 2 && g.x >= 1 && g.x + 1 <= s.cols - 2 &&
               ((unsigned)(g.x + 6) < (unsigned)s.step || g.y + 2 < s.rows - 1);
  return g;
}
// frame-level part of fetch_block2x2's window-path condition (wave-uniform)
__device__ __forceinline__ bool ccc_frame_straight(const CccParams& p, const SrcView& s) {
  return s.kind == SRC_BAYER && (p.flip_angle == 0 || p.flip_angle == 180) && (reinterpret_cast<uintptr_t>(s.base) & 3u) == 0 && (s.step & 3u) == 0 &&
         (unsigned long long)s.step * (unsigned long long)s.rows < (1ull << 32) && s.step < (1u << 24);
}

template <int K>
__device__ __forceinline__ void ccc_sample_bins(const CccParams& p, const CccSampleTabs& tb, const SrcView& s, bool frame_straight,
                                                const int (&idx)[K], const bool (&live)[K], int (&bin)[K]) {
  CccSampleGeom g[K];
  bool all = frame_straight;
  if (frame_straight) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      g[k] = ccc_sample_geom(p, tb, s, idx[k]);
      all = all && g[k].straight;
    }
    all = __builtin_amdgcn_ballot_w64(!all) == 0ull;  // every active lane
  }
  if (!all) {
#pragma unroll
    for (int k = 0; k < K; k++) bin[k] = live[k] ? ccc_sample_bin(p, tb, s, idx[k]) : -1;
    return;
  }
  uint32_t lo[K][4], hi[K][4];
  unsigned sh[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const unsigned off = __umul24((unsigned)(g[k].y - 1 + r), (unsigned)s.step) + (unsigned)(g[k].x - 1);
      const uint32_t* q = reinterpret_cast<const uint32_t*>(s.base + (off & ~3u));
      lo[k][r] = q[0];
      hi[k][r] = q[1];
      if (r == 0) sh[k] = off & 3u;  // step % 4 == 0: the same for the four rows
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    uint32_t pr[2][3];
    debayer_block2x2_pairs(s, g[k].y, g[k].x, lo[k], hi[k], sh[k], pr);
    int sm[3];
    if (p.geom.area_fast) {
      const u16x2 one2 = as_u16x2(0x00010001u);
#pragma unroll
      for (int c = 0; c < 3; c++)
        sm[c] = (int)(__builtin_amdgcn_udot2(as_u16x2(pr[1][c]), one2, __builtin_amdgcn_udot2(as_u16x2(pr[0][c]), one2, 2u, false), false) >> 2);
    } else {
      const u16x2 alpha = as_u16x2(g[k].alpha2);  // Q11 taps of the two columns, non-negative
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const unsigned r0 = __builtin_amdgcn_udot2(as_u16x2(pr[0][c]), alpha, 0u, false);
        const unsigned r1 = __builtin_amdgcn_udot2(as_u16x2(pr[1][c]), alpha, 0u, false);
        sm[c] = (int)(((__umul24((unsigned)g[k].b0, r0 >> 4) >> 16) + (__umul24((unsigned)g[k].b1, r1 >> 4) >> 16) + 2u) >> 2);
      }
    }
    const int b = ccc_bin_of(p, tb, sm);
    bin[k] = live[k] ? b : -1;
  }
}

// Small batches: kHistBlocks workgroups per frame, one global atomic per sample into a zeroed histogram (a frame's
// samples spread over the chip: lowest latency for a single frame; one resident 1440 x 1080 frame, kernel run time with
// 24 / 96 / 190 workgroups: 17.7 / 11.1 / 9.8 us -- 190 is one trip per lane, but the whole call is no shorter than with 96).
#ifndef RIP_CCC_HIST_BLOCKS
#define RIP_CCC_HIST_BLOCKS 96
#endif
constexpr int kHistBlocks = RIP_CCC_HIST_BLOCKS;
__global__ __launch_bounds__(kBlock) void ccc_hist_kernel(CccParams p) {
  __shared__ CccSampleTabs tb;
  tb.load<kBlock>(p);
  __syncthreads();
  const int frame = blockIdx.y;
  SrcView s{p.src + (size_t)frame * p.src_frame_stride, p.src_step, p.rows, p.cols, p.src_kind, p.bayer_ry, p.bayer_rx};
  // two samples per thread and trip through the straight-line sampler (window loads of both in flight together, packed
  // demosaic; unflipped and 180-degree frames), like the LDS-histogram kernel below
  constexpr int kUnroll = 2;
  const bool frame_straight = ccc_frame_straight(p, s);
  for (int i0 = blockIdx.x * kBlock + threadIdx.x; i0 < 360 * 270; i0 += kUnroll * kHistBlocks * kBlock) {
    int bin[kUnroll], idx[kUnroll];
    bool live[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      const int i = i0 + k * kHistBlocks * kBlock;
      live[k] = i < 360 * 270;
      idx[k] = live[k] ? i : 360 * 270 - 1;
    }
    ccc_sample_bins<kUnroll>(p, tb, s, frame_straight, idx, live, bin);
#pragma unroll
    for (int k = 0; k < kUnroll; k++)
      if (bin[k] >= 0) atomicAdd(&p.hist_counts[(size_t)frame * 65536 + bin[k]], 1u);
  }
}

// Batches: no global atomics and no memset.  ONE 1024-thread workgroup per frame holds the whole 256 x 256 histogram in
// LDS as 16-bit counters, two bins per dword (128 KB), walks the frame's 97 200 samples once with LDS atomics and then
// writes every bin -- zeros included -- to HBM as u32 with coalesced stores.  A 16-bit counter can wrap: a bin that
// receives more than 65 535 of the 97 200 samples (a flat frame).  At most one bin per frame can do that, and only once;
// the returning atomic shows the wrap (old half == 0xFFFF), the carry into the neighbouring bin is taken back and the
// bin is remembered, so the written count is exact.  Counts are integers: the result does not depend on the order.
#ifndef RIP_CCC_UNROLL
#define RIP_CCC_UNROLL 2
#endif
constexpr int kHistLdsThreads = 1024, kHistWords = 32768;
__global__ __launch_bounds__(kHistLdsThreads) void ccc_hist_lds_kernel(CccParams p) {
  extern __shared__ __align__(16) unsigned char ccc_smem[];
  unsigned* words = reinterpret_cast<unsigned*>(ccc_smem);
  CccSampleTabs& tb = *reinterpret_cast<CccSampleTabs*>(ccc_smem + kHistWords * sizeof(unsigned));
  __shared__ int wrapped_bin;
  tb.load<kHistLdsThreads>(p);
  uint4* z = reinterpret_cast<uint4*>(words);
  for (int i = threadIdx.x; i < kHistWords / 4; i += kHistLdsThreads) z[i] = make_uint4(0u, 0u, 0u, 0u);
  if (threadIdx.x == 0) wrapped_bin = -1;
  __syncthreads();
  // p.hist_split workgroups per frame, each with its own share of the samples and its own partial histogram (summed by the
  // row transform that reads them): with one workgroup per frame a batch of 64 frames kept 64 of 256 CUs busy
  const int frame = (int)blockIdx.x / p.hist_split, part = (int)blockIdx.x - frame * p.hist_split;
  SrcView s{p.src + (size_t)frame * p.src_frame_stride, p.src_step, p.rows, p.cols, p.src_kind, p.bayer_ry, p.bayer_rx};
  // RIP_CCC_UNROLL independent samples per thread and trip, their window loads in flight together (ccc_sample_bins)
  constexpr int kUnroll = RIP_CCC_UNROLL;
  const bool frame_straight = ccc_frame_straight(p, s);
  for (int i0 = threadIdx.x + part * kUnroll * kHistLdsThreads; i0 < 360 * 270; i0 += p.hist_split * kUnroll * kHistLdsThreads) {
    int bin[kUnroll], idx[kUnroll];
    bool live[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      const int i = i0 + k * kHistLdsThreads;
      live[k] = i < 360 * 270;
      idx[k] = live[k] ? i : 360 * 270 - 1;
    }
    ccc_sample_bins<kUnroll>(p, tb, s, frame_straight, idx, live, bin);
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      if (bin[k] < 0) continue;
      const unsigned hi = (unsigned)bin[k] & 1u;
      const unsigned old = atomicAdd(&words[bin[k] >> 1], hi ? 0x10000u : 1u);
      if (((old >> (16 * hi)) & 0xFFFFu) == 0xFFFFu) {  // this add wrapped the counter
        if (!hi) atomicSub(&words[bin[k] >> 1], 0x10000u);  // the carry went into the odd neighbour: take it back
        wrapped_bin = bin[k];
      }
    }
  }
  __syncthreads();
  const int wb = wrapped_bin;
  uint4* out = reinterpret_cast<uint4*>(p.hist_counts + (size_t)blockIdx.x * 65536);
  for (int i = threadIdx.x; i < kHistWords / 2; i += kHistLdsThreads) {  // two words = four bins per store
    const unsigned w0 = words[2 * i], w1 = words[2 * i + 1];
    uint4 o = make_uint4(w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16);
    if ((wb >> 2) == i) {
      const int k = wb & 3;
      if (k == 0) o.x += 65536u;
      if (k == 1) o.y += 65536u;
      if (k == 2) o.z += 65536u;
      if (k == 3) o.w += 65536u;
    }
    out[i] = o;
  }
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

// Forward FFT of the columns, spectrum product + bias, inverse FFT of the columns.  A workgroup takes
// kFftCols = 16 adjacent columns: every row of its slab is 16 complex numbers = one 128-byte line, so the
// loads of the work buffer, of the filter / bias spectra and the stores move whole lines (one column per
// workgroup touched 8 bytes of every line: 1.9 GB of traffic for 0.27 GB of data, PMC).  The butterflies of a
// column are evaluated exactly as fft256_lds does, so the response is unchanged bit for bit.
#ifndef RIP_CCC_FFT_FEW
#define RIP_CCC_FFT_FEW 4
#endif
constexpr int kFftFewFrames = RIP_CCC_FFT_FEW;  // batches up to this size run the transforms four columns per one-wave workgroup
constexpr int kFftCols = 16, kFftPitch = 257;  // odd pitch: the 16 columns of one element fall into 16 banks

// 256-point FFTs of kFftCols columns held as re/im[c * kFftPitch + i] (bit-reversed input on entry, after a barrier;
// natural-order output, followed by a barrier); 256 threads: column t & 15, sixteen elements each.
//
// The eight radix-2 stages are the butterflies of fft256_lds / host_fft256 -- same operands, same twiddle index, same
// operations, so every value is bit-identical -- but scheduled in three register rounds instead of eight LDS passes:
// stages 2-4-8 on eight contiguous elements, stages 16-32-64 on eight elements 8 apart, stages 128-256 on four elements
// 64 apart.  A thread loads its elements, runs the stages of the round in registers and stores them back: 6 LDS accesses
// per element and 3 barriers per transform instead of 16 and 16 (the transforms were LDS-bound).
__device__ __forceinline__ void fft_bfly(float& ur, float& ui, float& xr, float& xi, float wr, float wi) {
  const float tr = wr * xr - wi * xi;
  const float ti = wr * xi + wi * xr;
  const float lr = ur + tr, li = ui + ti, hr = ur - tr, hi = ui - ti;
  ur = lr;
  ui = li;
  xr = hr;
  xi = hi;
}

// COLS columns on 16 * COLS threads: 16 for batches (a row of the slab is one 128-byte line), 4 -- one wave per workgroup,
// 64 workgroups per frame -- for the few frames of a latency-bound call; the butterflies and their order are the same.
template <int COLS>
__device__ __forceinline__ void fft256_columns_lds(float* re, float* im, const float* twr, const float* twi, bool inverse) {
  const int c = threadIdx.x & (COLS - 1), g = threadIdx.x / COLS;
  float* cre = re + c * kFftPitch;
  float* cim = im + c * kFftPitch;
  auto tw = [&](int idx, float& wr, float& wi) {
    wr = twr[idx];
    wi = inverse ? -twi[idx] : twi[idx];
  };
  // round A: stages len = 2, 4, 8 on elements B .. B + 7
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int B = 8 * (2 * g + h);
    float xr[8], xi[8], wr, wi;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      xr[j] = cre[B + j];
      xi[j] = cim[B + j];
    }
    tw(0, wr, wi);
#pragma unroll
    for (int j = 0; j < 8; j += 2) fft_bfly(xr[j], xi[j], xr[j + 1], xi[j + 1], wr, wi);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      tw(j * 64, wr, wi);
      fft_bfly(xr[j], xi[j], xr[j + 2], xi[j + 2], wr, wi);
      fft_bfly(xr[j + 4], xi[j + 4], xr[j + 6], xi[j + 6], wr, wi);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      tw(j * 32, wr, wi);
      fft_bfly(xr[j], xi[j], xr[j + 4], xi[j + 4], wr, wi);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {  // a thread's own elements: no barrier needed before the store
      cre[B + j] = xr[j];
      cim[B + j] = xi[j];
    }
  }
  __syncthreads();
  // round B: stages len = 16, 32, 64 on elements B64 + r + 8 m
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int q = 2 * g + h, base = (q >> 3) * 64 + (q & 7), r = q & 7;
    float xr[8], xi[8], wr, wi;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      xr[m] = cre[base + 8 * m];
      xi[m] = cim[base + 8 * m];
    }
    tw(r * 16, wr, wi);
#pragma unroll
    for (int m = 0; m < 8; m += 2) fft_bfly(xr[m], xi[m], xr[m + 1], xi[m + 1], wr, wi);
#pragma unroll
    for (int m = 0; m < 2; m++) {
      tw((r + 8 * m) * 8, wr, wi);
      fft_bfly(xr[m], xi[m], xr[m + 2], xi[m + 2], wr, wi);
      fft_bfly(xr[m + 4], xi[m + 4], xr[m + 6], xi[m + 6], wr, wi);
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      tw((r + 8 * m) * 4, wr, wi);
      fft_bfly(xr[m], xi[m], xr[m + 4], xi[m + 4], wr, wi);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
      cre[base + 8 * m] = xr[m];
      cim[base + 8 * m] = xi[m];
    }
  }
  __syncthreads();
  // round C: stages len = 128, 256 on elements r + 64 m; the four row groups of a wave sit 16 apart (two lanes per bank)
#pragma unroll
  for (int h = 0; h < 4; h++) {
    const int r = 16 * (g & 3) + (g >> 2) + 4 * h;
    float xr[4], xi[4], wr, wi;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      xr[m] = cre[r + 64 * m];
      xi[m] = cim[r + 64 * m];
    }
    tw(r * 2, wr, wi);
    fft_bfly(xr[0], xi[0], xr[1], xi[1], wr, wi);
    fft_bfly(xr[2], xi[2], xr[3], xi[3], wr, wi);
#pragma unroll
    for (int m = 0; m < 2; m++) {
      tw(r + 64 * m, wr, wi);
      fft_bfly(xr[m], xi[m], xr[m + 2], xi[m + 2], wr, wi);
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      cre[r + 64 * m] = xr[m];
      cim[r + 64 * m] = xi[m];
    }
  }
  __syncthreads();
}

// Row transforms, kFftCols rows per 256-thread workgroup on the same slab code (a row is 1 KB of contiguous floats, so
// the loads and stores are whole lines as well).  One row per 128-thread workgroup (eight barriers per transform, two waves)
// was latency-bound: 0.20 ms of the estimator's 0.77 ms per 256 frames.
template <int COLS>
__global__ __launch_bounds__(16 * COLS) void ccc_fft_rows16_kernel(CccParams p) {
  __shared__ float re[COLS * kFftPitch], im[COLS * kFftPitch], twr[128], twi[128];
  const int row0 = blockIdx.x * COLS, frame = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < 128; i += 16 * COLS) {
    twr[i] = p.tabs->tw_re[i];
    twi[i] = p.tabs->tw_im[i];
  }
  unsigned int* h = p.hist_counts + (size_t)frame * p.hist_split * 65536 + (size_t)row0 * 256;
  for (int e = t; e < COLS * 256; e += 16 * COLS) {  // e = r * 256 + i: consecutive lanes read consecutive counters
    const int r = e >> 8, i = e & 255;
    unsigned count = h[e];
    // the global-atomic kernel accumulates into zeroed counters: the only reader hands them back zeroed, so the next
    // frame needs no memset (one dependent operation less per frame)
    if (p.hist_zero_after) h[e] = 0u;  // every counter, coalesced: a store only where the count is non-zero runs 1 us longer
    for (int s = 1; s < p.hist_split; s++) count += h[(size_t)s * 65536 + e];  // partial histograms of the frame (counts: exact)
    re[r * kFftPitch + bitrev8((unsigned)i)] = p.accum_tab[count];
    im[r * kFftPitch + bitrev8((unsigned)i)] = 0.f;
  }
  __syncthreads();
  fft256_columns_lds<COLS>(re, im, twr, twi, false);
  float2* out = reinterpret_cast<float2*>(p.work) + (size_t)frame * 65536 + (size_t)row0 * 256;
  for (int e = t; e < COLS * 256; e += 16 * COLS) {
    const int r = e >> 8, i = e & 255;
    out[e] = make_float2(re[r * kFftPitch + i], im[r * kFftPitch + i]);
  }
}

// inverse FFT of kFftCols rows; per-row first maximum of the real part
template <int COLS>
__global__ __launch_bounds__(16 * COLS) void ccc_ifft_rows16_kernel(CccParams p) {
  __shared__ float re[COLS * kFftPitch], im[COLS * kFftPitch], twr[128], twi[128];
  const int row0 = blockIdx.x * COLS, frame = blockIdx.y, t = threadIdx.x;
  for (int i = t; i < 128; i += 16 * COLS) {
    twr[i] = p.tabs->tw_re[i];
    twi[i] = p.tabs->tw_im[i];
  }
  const float2* in = reinterpret_cast<const float2*>(p.work) + (size_t)frame * 65536 + (size_t)row0 * 256;
  for (int e = t; e < COLS * 256; e += 16 * COLS) {
    const int r = e >> 8, i = e & 255;
    const float2 v = in[e];
    re[r * kFftPitch + bitrev8((unsigned)i)] = v.x;
    im[r * kFftPitch + bitrev8((unsigned)i)] = v.y;
  }
  __syncthreads();
  fft256_columns_lds<COLS>(re, im, twr, twi, true);
  // first maximum of each row: 16 lanes per row, each scans 16 consecutive columns in order, then a 16-lane min-index
  // reduction with the same tie rule (larger value, else smaller column)
  const int r = t >> 4, l = t & 15;
  float bv = re[r * kFftPitch + l * 16];
  int bi = l * 16;
  for (int k = 1; k < 16; k++) {
    const float v = re[r * kFftPitch + l * 16 + k];
    if (v > bv) {
      bv = v;
      bi = l * 16 + k;
    }
  }
  for (int off = 8; off > 0; off >>= 1) {
    const float ov = __shfl_down(bv, off, 16);
    const int oi = __shfl_down(bi, off, 16);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if (l == 0) {
    p.row_best[((size_t)frame * 256 + row0 + r) * 2] = bv;
    p.row_best[((size_t)frame * 256 + row0 + r) * 2 + 1] = (float)bi;
  }
}

template <int COLS>
__global__ __launch_bounds__(16 * COLS) void ccc_fft_cols_kernel(CccParams p) {
  __shared__ float re[COLS * kFftPitch], im[COLS * kFftPitch], twr[128], twi[128];
  const int col0 = blockIdx.x * COLS, frame = blockIdx.y, t = threadIdx.x;
  const int c = t & (COLS - 1), r0 = t / COLS;  // this thread moves rows r0, r0 + 16, ... of column col0 + c
  for (int i = t; i < 128; i += 16 * COLS) {
    twr[i] = p.tabs->tw_re[i];
    twi[i] = p.tabs->tw_im[i];
  }
  float2* data = reinterpret_cast<float2*>(p.work) + (size_t)frame * 65536 + col0 + c;
#pragma unroll 4
  for (int i = r0; i < 256; i += 16) {
    const float2 v = data[(size_t)i * 256];
    const unsigned j = bitrev8((unsigned)i);
    re[c * kFftPitch + j] = v.x;
    im[c * kFftPitch + j] = v.y;
  }
  __syncthreads();
  fft256_columns_lds<COLS>(re, im, twr, twi, false);
  const float2* F = reinterpret_cast<const float2*>(p.filter_fft) + col0 + c;
  const float2* B = reinterpret_cast<const float2*>(p.bias_fft) + col0 + c;
  float qr[16], qi[16];
#pragma unroll
  for (int m = 0; m < 16; m++) {
    const int i = r0 + 16 * m;
    const float2 f = F[(size_t)i * 256], bb = B[(size_t)i * 256];
    const float ar = f.x, ai = f.y, br = re[c * kFftPitch + i], bi = im[c * kFftPitch + i];
    const float pr = ar * br - ai * bi;  // mulSpectrums, no conjugation
    const float pi = ar * bi + ai * br;
    qr[m] = pr + bb.x;
    qi[m] = pi + bb.y;
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 16; m++) {
    const unsigned j = bitrev8((unsigned)(r0 + 16 * m));
    re[c * kFftPitch + j] = qr[m];
    im[c * kFftPitch + j] = qi[m];
  }
  __syncthreads();
  fft256_columns_lds<COLS>(re, im, twr, twi, true);
#pragma unroll 4
  for (int i = r0; i < 256; i += 16) data[(size_t)i * 256] = make_float2(re[c * kFftPitch + i], im[c * kFftPitch + i]);
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

}  // namespace

bool launch_ccc_estimate(const CccParams& p, const Tunables& tn, hipStream_t stream, int* hist_left_clean) {
  if (hist_left_clean) *hist_left_clean = 0;
  if (p.n_frames <= 0) return true;
  // Batches of at least tn.ccc_lds_hist_min frames take the LDS histogram (one workgroup = one CU per frame).  It needs
  // ~135 KB of dynamic LDS, an opt-in above 64 KB that is per device (a process that drives several GPUs needs it on each);
  // where the runtime refuses it the global-atomic kernel -- which runs on any device -- takes over.
  bool lds_hist = p.n_frames >= tn.ccc_lds_hist_min;
  constexpr unsigned lds = kHistWords * sizeof(unsigned) + sizeof(CccSampleTabs);
  if (lds_hist) {
    static std::atomic<unsigned long long> opted_in{0}, refused{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (refused.load(std::memory_order_relaxed) & bit) {
      lds_hist = false;
    } else if (!(opted_in.load(std::memory_order_relaxed) & bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(ccc_hist_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
        opted_in.fetch_or(bit, std::memory_order_relaxed);
      } else {
        (void)hipGetLastError();  // the refusal is handled here; it must not surface as the batch's launch error
        refused.fetch_or(bit, std::memory_order_relaxed);
        lds_hist = false;
      }
    }
  }
  CccParams q = p;
  q.hist_split = lds_hist ? std::max(1, p.hist_split) : 1;  // the caller's split: its buffer is sized for it
  q.hist_zero_after = lds_hist ? 0 : 1;
  if (lds_hist) {
    hipLaunchKernelGGL(ccc_hist_lds_kernel, dim3(p.n_frames * q.hist_split), dim3(kHistLdsThreads), lds, stream, q);
  } else {
    if (!p.hist_is_clean && hipMemsetAsync(p.hist_counts, 0, (size_t)p.n_frames * 65536 * sizeof(unsigned), stream) != hipSuccess) return false;
    hipLaunchKernelGGL(ccc_hist_kernel, dim3(kHistBlocks, p.n_frames), dim3(kBlock), 0, stream, q);
  }
  // a histogram that was not launched leaves stale counts behind: stop before anything consumes them (the caller must
  // not advance the Kalman state either)
  if (hipGetLastError() != hipSuccess) return false;
  if (p.n_frames <= kFftFewFrames) {  // a few frames: 64 one-wave workgroups per frame instead of 16 four-wave ones
    hipLaunchKernelGGL(ccc_fft_rows16_kernel<4>, dim3(64, p.n_frames), dim3(64), 0, stream, q);
    hipLaunchKernelGGL(ccc_fft_cols_kernel<4>, dim3(64, p.n_frames), dim3(64), 0, stream, p);
    hipLaunchKernelGGL(ccc_ifft_rows16_kernel<4>, dim3(64, p.n_frames), dim3(64), 0, stream, p);
  } else {
    hipLaunchKernelGGL(ccc_fft_rows16_kernel<kFftCols>, dim3(256 / kFftCols, p.n_frames), dim3(16 * kFftCols), 0, stream, q);
    hipLaunchKernelGGL(ccc_fft_cols_kernel<kFftCols>, dim3(256 / kFftCols, p.n_frames), dim3(16 * kFftCols), 0, stream, p);
    hipLaunchKernelGGL(ccc_ifft_rows16_kernel<kFftCols>, dim3(256 / kFftCols, p.n_frames), dim3(16 * kFftCols), 0, stream, p);
  }
  if (!ccc_argmax_in_finalize(p.n_frames)) hipLaunchKernelGGL(ccc_argmax_kernel, dim3(p.n_frames), dim3(256), 0, stream, p);
  const bool ok = hipGetLastError() == hipSuccess;
  if (ok && !lds_hist && hist_left_clean) *hist_left_clean = 1;
  return ok;
}

}  // namespace rip
