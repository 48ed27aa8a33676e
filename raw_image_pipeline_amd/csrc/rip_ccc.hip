// rip_ccc.hip -- convolutional colour constancy estimator: log-chroma histogram, 256 x 256 FFT convolution with the
// model, arg-max (convolutional_color_constancy.cpp:91-340).
// Shared device code and the stage-by-stage reference citations: rip_device.hpp.
#include "rip_device.hpp"

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
__device__ __forceinline__ void debayer_block2x2(const SrcView& s, int y, int x, int (&out)[4][3]) {
  int v[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const unsigned off = __umul24((unsigned)(y - 1 + r), (unsigned)s.step) + (unsigned)(x - 1);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(s.base + (off & ~3u));
    const uint32_t w = __builtin_amdgcn_alignbyte(q[1], q[0], off & 3u);
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

// the 2x2 block of post-flip pixels at (y, x): fast window path for unflipped Bayer frames, else per pixel
__device__ __forceinline__ void fetch_block2x2(const SrcView& s, int angle, int y, int x, int y1, int x1, int (&out)[4][3]) {
  const bool fast = s.kind == SRC_BAYER && angle == 0 && y1 == y + 1 && x1 == x + 1 && y >= 1 && y1 <= s.rows - 2 && x >= 1 &&
                    x1 <= s.cols - 2 && (unsigned)(x + 6) < (unsigned)s.step && (reinterpret_cast<uintptr_t>(s.base) & 3u) == 0 &&
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

// Samples per frame: 360 x 270 (small_size_, convolutional_color_constancy.cpp:22).  kHistBlocks workgroups per frame
// walk them with the resize geometry and the log table in LDS (ten table reads per sample otherwise go to L2).
constexpr int kHistBlocks = 24;
__global__ __launch_bounds__(kBlock) void ccc_hist_kernel(CccParams p) {
  __shared__ int s_xofs[360], s_yofs[540];
  __shared__ short s_ialpha[720], s_ibeta[540];
  __shared__ float s_log[256];
  for (int i = threadIdx.x; i < 360; i += kBlock) s_xofs[i] = p.geom.xofs[i];
  for (int i = threadIdx.x; i < 540; i += kBlock) {
    s_yofs[i] = p.geom.yofs[i];
    s_ibeta[i] = p.geom.ibeta[i];
  }
  for (int i = threadIdx.x; i < 720; i += kBlock) s_ialpha[i] = p.geom.ialpha[i];
  s_log[threadIdx.x] = p.tabs->log_tab[threadIdx.x];
  __syncthreads();
  const int frame = blockIdx.y;
  SrcView s{p.src + (size_t)frame * p.src_frame_stride, p.src_step, p.rows, p.cols, p.src_kind, p.bayer_ry, p.bayer_rx};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < 360 * 270; i += kHistBlocks * kBlock) {
    const int dy = i / 360, dx = i - dy * 360;
    int sm[3];
    int t[4][3];  // taps (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    if (p.geom.area_fast) {
      fetch_block2x2(s, p.flip_angle, 2 * dy, 2 * dx, 2 * dy + 1, 2 * dx + 1, t);
#pragma unroll
      for (int c = 0; c < 3; c++) sm[c] = (t[0][c] + t[1][c] + t[2][c] + t[3][c] + 2) >> 2;
    } else {
      // cv::resize INTER_LINEAR, 8U: Q11 coefficients, two-pass integer arithmetic
      const int sx = s_xofs[dx];
      const int sx1 = sx + 1 < p.dcols ? sx + 1 : sx;
      const int a0 = s_ialpha[dx * 2], a1 = s_ialpha[dx * 2 + 1];
      const int y0 = s_yofs[dy * 2], y1 = s_yofs[dy * 2 + 1];
      const int b0 = s_ibeta[dy * 2], b1 = s_ibeta[dy * 2 + 1];
      fetch_block2x2(s, p.flip_angle, y0, sx, y1, sx1, t);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        int r0 = t[0][c] * a0 + t[1][c] * a1;
        int r1 = t[2][c] * a0 + t[3][c] * a1;
        sm[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
      }
    }
    // calculateHistogramFeature (:210-271)
    float fb = (float)sm[0], fg = (float)sm[1], fr = (float)sm[2];
    float gray = fb * 0.114f + fg * 0.587f + fr * 0.299f;
    bool ok = !(gray > p.upper) && (gray > p.lower);
    if (sm[0] == 0 || sm[1] == 0 || sm[2] == 0) ok = false;  // log(0) = -inf is skipped
    if (!ok) continue;
    const float bin_size = 1.0f / 64.0f, uv0 = -1.421875f;
    float lb = s_log[sm[0]], lg = s_log[sm[1]], lr = s_log[sm[2]];
    int u = (int)roundf((lg - lr - uv0) / bin_size);
    int v = (int)roundf((lg - lb - uv0) / bin_size);
    u = clampi(u, 0, 255);
    v = clampi(v, 0, 255);
    atomicAdd(&p.hist_counts[(size_t)frame * 65536 + u * 256 + v], 1u);
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

// Forward FFT of the columns, spectrum product + bias, inverse FFT of the columns.  A workgroup takes
// kFftCols = 16 adjacent columns: every row of its slab is 16 complex numbers = one 128-byte line, so the
// loads of the work buffer, of the filter / bias spectra and the stores move whole lines (one column per
// workgroup touched 8 bytes of every line: 1.9 GB of traffic for 0.27 GB of data, PMC).  The butterflies of a
// column are evaluated exactly as fft256_lds does, so the response is unchanged bit for bit.
constexpr int kFftCols = 16, kFftPitch = 257;  // odd pitch: the 16 columns of one element fall into 16 banks

// 256-point FFTs of kFftCols columns held as re/im[c * kFftPitch + i]; 256 threads: column t & 15, eight
// butterflies of every stage each
__device__ __forceinline__ void fft256_columns_lds(float* re, float* im, const float* twr, const float* twi, bool inverse) {
  const int c = threadIdx.x & (kFftCols - 1), b0 = threadIdx.x >> 4;
  float* cre = re + c * kFftPitch;
  float* cim = im + c * kFftPitch;
  for (int len = 2; len <= 256; len <<= 1) {
    const int half = len >> 1, tstep = 256 / len;
    float nr[8][2], ni[8][2];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int b = b0 + 16 * m;
      const int k = b & (half - 1), lo = (b / half) * len + k, hi = lo + half;
      const float wr = twr[k * tstep], wi = inverse ? -twi[k * tstep] : twi[k * tstep];
      const float xr = cre[hi], xi = cim[hi];
      const float tr = wr * xr - wi * xi;
      const float ti = wr * xi + wi * xr;
      const float ur = cre[lo], ui = cim[lo];
      nr[m][0] = ur + tr;
      ni[m][0] = ui + ti;
      nr[m][1] = ur - tr;
      ni[m][1] = ui - ti;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int b = b0 + 16 * m;
      const int k = b & (half - 1), lo = (b / half) * len + k, hi = lo + half;
      cre[lo] = nr[m][0];
      cim[lo] = ni[m][0];
      cre[hi] = nr[m][1];
      cim[hi] = ni[m][1];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void ccc_fft_cols_kernel(CccParams p) {
  __shared__ float re[kFftCols * kFftPitch], im[kFftCols * kFftPitch], twr[128], twi[128];
  const int col0 = blockIdx.x * kFftCols, frame = blockIdx.y, t = threadIdx.x;
  const int c = t & (kFftCols - 1), r0 = t >> 4;  // this thread moves rows r0, r0 + 16, ... of column col0 + c
  if (t < 128) {
    twr[t] = p.tabs->tw_re[t];
    twi[t] = p.tabs->tw_im[t];
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
  fft256_columns_lds(re, im, twr, twi, false);
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
  fft256_columns_lds(re, im, twr, twi, true);
#pragma unroll 4
  for (int i = r0; i < 256; i += 16) data[(size_t)i * 256] = make_float2(re[c * kFftPitch + i], im[c * kFftPitch + i]);
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

}  // namespace

void launch_ccc_estimate(const CccParams& p, hipStream_t stream) {
  if (p.n_frames <= 0) return;
  hipLaunchKernelGGL(ccc_hist_kernel, dim3(kHistBlocks, p.n_frames), dim3(kBlock), 0, stream, p);
  hipLaunchKernelGGL(ccc_fft_rows_kernel, dim3(256, p.n_frames), dim3(128), 0, stream, p);
  hipLaunchKernelGGL(ccc_fft_cols_kernel, dim3(256 / kFftCols, p.n_frames), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(ccc_ifft_rows_kernel, dim3(256, p.n_frames), dim3(128), 0, stream, p);
  hipLaunchKernelGGL(ccc_argmax_kernel, dim3(p.n_frames), dim3(256), 0, stream, p);
}

}  // namespace rip
