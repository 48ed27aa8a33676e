// rip_stats.hip -- white-balance statistics (grey-world sums, pca sums / maxima, SimpleWB histograms) and the
// on-device finalisation of the per-frame gains.
// Shared device code and the stage-by-stage reference citations: rip_device.hpp.
#include "rip_device.hpp"

namespace rip {
namespace {

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

// SimpleWB histograms in LDS: kHistRep interleaved copies of the 3 x 256 bins, lane l counts in copy l % kHistRep.
// Neighbouring pixels mostly fall into the same few bins, and LDS atomics of one instruction that hit the same address are
// served one after the other: with one copy a wave's 64 increments of a flat region take 64 LDS cycles, with eight copies
// (consecutive dwords = consecutive banks: no conflict between the copies) eight.  24 KB per workgroup.
constexpr int kHistRep = 8, kHistRepShift = 3;
constexpr int kHistWordsLds = 768 * kHistRep;
__device__ __forceinline__ void stat_add(const StatsParams& p, int b, int g, int r, StatAcc& a, unsigned* s_hist) {
  if (p.mode == WB_SIMPLE) {
    const unsigned copy = threadIdx.x & (kHistRep - 1);
    atomicAdd(&s_hist[((unsigned)b << kHistRepShift) + copy], 1u);
    atomicAdd(&s_hist[((256u + (unsigned)g) << kHistRepShift) + copy], 1u);
    atomicAdd(&s_hist[((512u + (unsigned)r) << kHistRepShift) + copy], 1u);
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
  for (int i = threadIdx.x; i < kHistWordsLds; i += kBlock) s_hist[i] = 0u;
  __syncthreads();
}

// white_balance.cpp:59-64 (GrayworldWB) and :73-136 (pca): a frame's sums / maxima -> its gains
__device__ __forceinline__ void solve2(float m00, float m01, float m10, float m11, float g0, float g1, float& o0, float& o1) {
  // Eigen::Matrix2f::inverse() * vec (white_balance.cpp:104-115)
  float det = m00 * m11 - m01 * m10;
  float invdet = 1.0f / det;
  float i00 = m11 * invdet, i01 = -m01 * invdet, i10 = -m10 * invdet, i11 = m00 * invdet;
  o0 = i00 * g0 + i01 * g1;
  o1 = i10 * g0 + i11 * g1;
}
__device__ __forceinline__ void wb_gains_from_sums(int mode, const StatShard& fs, FrameWb& w) {
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
}

__device__ __forceinline__ void stat_flush(const StatsParams& p, StatAcc& a, FrameStats* out, unsigned* s_hist, int frame) {
  if (p.mode == WB_SIMPLE) {
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += kBlock)
    {
      unsigned n = 0;
#pragma unroll
      for (int c = 0; c < kHistRep; c++) n += s_hist[i * kHistRep + c];
      if (n) atomicAdd(&p.hist3[(size_t)frame * 768 + i], n);
    }
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
  // With the fused finalisation the atomics return their old value: the thread then waits for it, i.e. until the update has
  // been performed at the device's coherence point (agent-scope atomics execute beyond the per-XCD L2s), before the
  // workgroup barrier below lets thread 0 draw the ticket.  A release fence instead (__threadfence, or an acq_rel ticket:
  // both make hipcc emit the agent-scope L2 write-back) made this kernel 13 x slower on gfx950.
  // HARDWARE ASSUMPTION, outside the HIP memory model: a returned agent-scope atomic HAS been performed where every other
  // agent-scope atomic on the same address will see it, and only atomics touch these records (the collector below reads
  // them with atomicExch, never with plain loads).  The asm below is both the use of the returned value (s_waitcnt vmcnt(0)
  // before it) and a compiler barrier ("memory": nothing moves across it).  tests/test_determinism_gpu.py re-runs the same
  // batches at several sizes and shard counts and fails on the first record that does not come back zeroed.
  const bool fused = p.wb_out != nullptr;
  StatShard* const mine = &out->shard[blockIdx.x % kStatShards];
  if (threadIdx.x < 5) {
    unsigned long long t = 0;
    for (int i = 0; i < kBlock / 64; i++) t += sh[threadIdx.x][i];
    if (t) {
      if (fused) {
        const unsigned long long old = atomicAdd(&mine->sum[threadIdx.x], t);
        asm volatile("" ::"v"(old) : "memory");
      } else {
        atomicAdd(&mine->sum[threadIdx.x], t);
      }
    }
  } else if (threadIdx.x < 8 && p.mode == WB_PCA) {
    unsigned t = 0;
    for (int i = 0; i < kBlock / 64; i++) t = max(t, sh[threadIdx.x][i]);
    if (t) {
      if (fused) {
        const unsigned old = atomicMax(&mine->mx[threadIdx.x - 5], t);
        asm volatile("" ::"v"(old) : "memory");
      } else {
        atomicMax(&mine->mx[threadIdx.x - 5], t);
      }
    }
  }
  if (!fused) return;
  // Fused finalisation: every workgroup of the frame (gridDim.x of them) takes a ticket once its own updates have been
  // performed; the one that draws the last ticket therefore sees all of them: it collects the shards (exchanging them
  // for zeros: the next batch starts from clean records without a memset), adds them up and writes the frame's gains.
  __shared__ unsigned s_ticket;
  __shared__ StatShard s_part[kStatShards];
  __syncthreads();
  if (threadIdx.x == 0) s_ticket = atomicAdd(&out->shard[0].done, 1u);
  __syncthreads();
  if (s_ticket != gridDim.x - 1) return;
  if (threadIdx.x < kStatShards) {
    StatShard* const src = &out->shard[threadIdx.x];
    StatShard& dst = s_part[threadIdx.x];
#pragma unroll
    for (int k = 0; k < 5; k++) dst.sum[k] = atomicExch(&src->sum[k], 0ull);
#pragma unroll
    for (int k = 0; k < 3; k++) dst.mx[k] = atomicExch(&src->mx[k], 0u);
    if (threadIdx.x == 0) atomicExch(&src->done, 0u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    StatShard fs = s_part[0];
    for (int i = 1; i < kStatShards; i++) {
#pragma unroll
      for (int k = 0; k < 5; k++) fs.sum[k] += s_part[i].sum[k];
#pragma unroll
      for (int k = 0; k < 3; k++) fs.mx[k] = max(fs.mx[k], s_part[i].mx[k]);
    }
    FrameWb w = {};
    wb_gains_from_sums(p.mode, fs, w);
    p.wb_out[frame] = w;
  }
}

// Grey-world statistics of four planar pixels, two pixels per instruction: the bytes are widened to 16-bit lanes;
// max/min with v_pk_max/min_u16.  GrayworldWB skips a pixel iff (max - min) * 255 > thresh255 * max, i.e. keeps it iff
// (255 - thresh255) * max < 255 * min + 1: both sides fit 16 bits (thresh255 <= 255 after the clamp below, which does not change
// the outcome: for thresh255 >= 255 no pixel is ever skipped), so the 0/1 keep flag is min(sat_sub(255 * min + 1, s * max), 1) --
// one packed multiply-add, one packed multiply, one saturating subtract, one min -- and the masked channel sums are one
// v_dot4_u32_u8 per channel on the packed pixels with the four flags as byte weights.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t gw_keep_pair(uint32_t mx, uint32_t mn, uint32_t s2, uint32_t c255) {
  // 0 / 1 in both 16-bit fields: 1 where 255 * min + 1 > (255 - thresh255) * max.  Written out: hipcc turns
  // min(sub_sat(lhs, rhs), 1) into two 16-bit compares, two selects and a byte merge (five instructions and two hazard nops)
  const u16x2 lhs = as_u16x2(mn) * as_u16x2(c255) + as_u16x2(0x00010001u);  // <= 65026
  const u16x2 rhs = as_u16x2(mx) * as_u16x2(s2);                            // <= 65025
  uint32_t keep;
  asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3"
      : "=&v"(keep)
      : "v"(__builtin_bit_cast(uint32_t, lhs)), "v"(__builtin_bit_cast(uint32_t, rhs)), "v"(0x00010001u));
  return keep;
}
__device__ __forceinline__ void grayworld_add_swar(const Planar& v, unsigned thresh255, StatAcc& a) {
  constexpr uint32_t M8 = 0x00FF00FFu;
  const uint32_t s2 = (255u - thresh255) * 0x00010001u, c255 = 0x00FF00FFu;
  // pixels 0 and 2: bytes widened to 16-bit fields
  const u16x2 be = as_u16x2(v.b & M8), ge = as_u16x2(v.g & M8), re = as_u16x2(v.r & M8);
  const uint32_t mxe = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_elementwise_max(be, ge), re));
  const uint32_t mne = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_min(be, ge), re));
  // pixels 1 and 3: a 16-bit max / min of the raw dwords is decided by the high bytes (the low byte only breaks ties
  // between equal high bytes), so the high byte of every field is the max / min of those pixels -- no masking of the inputs
  const u16x2 braw = as_u16x2(v.b), graw = as_u16x2(v.g), rraw = as_u16x2(v.r);
  const uint32_t mxo = (__builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_elementwise_max(braw, graw), rraw)) >> 8) & M8;
  const uint32_t mno = (__builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_min(braw, graw), rraw)) >> 8) & M8;
  // keep flags of the four pixels as bytes 0 / 1: the weights of one v_dot4_u32_u8 per channel on the packed pixels
  const uint32_t keep = gw_keep_pair(mxe, mne, s2, c255) | (gw_keep_pair(mxo, mno, s2, c255) << 8);
  a.s[0] = __builtin_amdgcn_udot4(v.b, keep, a.s[0], false);
  a.s[1] = __builtin_amdgcn_udot4(v.g, keep, a.s[1], false);
  a.s[2] = __builtin_amdgcn_udot4(v.r, keep, a.s[2], false);
}

// Statistics of a Bayer frame.  A wave owns a strip 64 groups (256 px) wide and walks down
// `pairs_per_task` row pairs of it: the row index is wave-uniform, so a row's byte offset lives in an
// SGPR (the buffer instruction's soffset) and the three per-lane column offsets never change -- no
// per-item address arithmetic -- and two of the four window rows (with their SWAR preparation) carry
// over from one row pair to the next.
template <int MODE>
__global__ __launch_bounds__(kBlock) void stats_fast_kernel(StatsParams p, int col_waves, int pairs_per_task, int n_tasks) {
  __shared__ unsigned s_hist[MODE == WB_SIMPLE ? kHistWordsLds : 1];
  p.mode = MODE;  // the per-pixel switch in stat_add folds away
  stat_hist_init(p, s_hist);
  const int frame = blockIdx.y;
  const unsigned step = (unsigned)p.src_step;
  const unsigned src_bytes = __umul24((unsigned)(p.rows - 1), step) + (unsigned)p.cols;
  const __amdgpu_buffer_rsrc_t src = frame_rsrc(p.src + (size_t)frame * p.src_frame_stride, src_bytes);
  const unsigned thresh255 = min(p.thresh255, 255u);
  const int lane = threadIdx.x & 63;
  // Block b runs on XCD b % 8 (observed dispatch order; speed only): every XCD takes a contiguous range of tasks, so that
  // neighbouring column strips -- whose 256-byte row segments straddle the same 128-byte lines when the row pitch is not
  // a multiple of 128 -- and vertically adjacent row ranges (two shared halo rows) meet in one L2.
  const int logical_block = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const int task = __builtin_amdgcn_readfirstlane(logical_block * (kBlock / 64) + (int)(threadIdx.x >> 6));
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
    // round 5: the six byte merges as v_perm_b32 with the pattern's column parity in an SGPR selector (were v_and + v_and_or
    // pairs and two copies for the row parity), and the wave-level border test without a ballot per row pair: the column test
    // depends on the lane only and is taken once, the row test is wave-uniform
    const DemosaicSel ds = demosaic_selectors(p.bayer_ry, p.bayer_rx, 0);
    const unsigned long long col_edge_lanes = __builtin_amdgcn_ballot_w64(x0 == 0 || x0 + 4 == p.cols);
    auto consume = [&](const RowPrep& r0, const RowPrep& r1, const RowPrep& r2, const RowPrep& r3, int y0) {
      Planar rowpx[2];
      debayer_rows_sel(r0, r1, r2, r3, ds, rowpx);
      debayer_fix_edges(y0, x0, p.rows, p.cols, rowpx, false, (y0 == 0 || y0 + 2 == p.rows) ? ~0ull : col_edge_lanes);
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
    // two row pairs per iteration so the carried rows change roles without register moves; the rows of the NEXT TWO
    // pairs are in flight while the current pair is reduced (one pair ahead left every wave waiting on HBM once per
    // pair: the reduction of a pair is ~400 issue cycles, the load latency under load several times that)
    int y0 = pair_begin * 2;
    RowPrep ra = prep(fetch_row(y0 - 1)), rb = prep(fetch_row(y0));
    RawRow n0 = fetch_row(y0 + 1), n1 = fetch_row(y0 + 2);
    RawRow m0 = fetch_row(y0 + 3), m1 = fetch_row(y0 + 4);
    for (int pair = pair_begin; pair < pair_end; pair += 2, y0 += 4) {
      const RowPrep rc = prep(n0), rd = prep(n1);
      n0 = fetch_row(y0 + 5);
      n1 = fetch_row(y0 + 6);
      consume(ra, rb, rc, rd, y0);
      if (pair + 1 >= pair_end) break;
      ra = prep(m0);
      rb = prep(m1);
      m0 = fetch_row(y0 + 7);
      m1 = fetch_row(y0 + 8);
      consume(rc, rd, ra, rb, y0 + 2);
    }
  }
  stat_flush(p, a, p.stats + frame, s_hist, frame);
}

// colour input (bgr8 / rgb8), 4 px per lane
__global__ __launch_bounds__(kBlock) void stats_color_kernel(StatsParams p, ItemMap im, int items_per_frame) {
  extern __shared__ unsigned s_hist[];  // kHistWordsLds words for SimpleWB, one otherwise (stats_hist_lds_bytes)
  stat_hist_init(p, s_hist);
  const int frame = blockIdx.y;
  const uint8_t* src = p.src + (size_t)frame * p.src_frame_stride;
  const bool rgb = p.src_kind == SRC_RGB;
  StatAcc a = {};
  for (int item = blockIdx.x * kBlock + threadIdx.x; item < items_per_frame; item += gridDim.x * kBlock) {
    int y, grp;
    im.split(item, y, grp);
    const uint3 in = *reinterpret_cast<const uint3*>(src + (__umul24((unsigned)y, (unsigned)p.src_step) + (unsigned)grp * 12u));
    if (p.mode == WB_Q8) {  // uniform: the grey-world keep test and masked sums on packed planes, as for Bayer frames
      // bytes 0 3 6 9 / 1 4 7 10 / 2 5 8 11 of the twelve (hipcc turns the masks and shifts into byte permutes)
      Planar v;
      v.b = (in.x & 0xFFu) | ((in.x >> 16) & 0xFF00u) | ((in.y & 0x00FF0000u)) | ((in.z & 0x0000FF00u) << 16);
      v.g = ((in.x >> 8) & 0xFFu) | ((in.y & 0xFFu) << 8) | ((in.y >> 8) & 0x00FF0000u) | ((in.z & 0x00FF0000u) << 8);
      v.r = ((in.x >> 16) & 0xFFu) | (in.y & 0xFF00u) | ((in.z & 0xFFu) << 16) | (in.z & 0xFF000000u);
      if (rgb) {  // cvtColor(RGB2BGR), debayer.cpp:72-73
        const uint32_t t = v.b;
        v.b = v.r;
        v.r = t;
      }
      grayworld_add_swar(v, min(p.thresh255, 255u), a);
      continue;
    }
    int q[4][3];
    unpack12(in, rgb, q);
#pragma unroll
    for (int k = 0; k < 4; k++) stat_add(p, q[k][0], q[k][1], q[k][2], a, s_hist);
  }
  stat_flush(p, a, p.stats + frame, s_hist, frame);
}

__global__ __launch_bounds__(kBlock) void stats_generic_kernel(StatsParams p) {
  extern __shared__ unsigned s_hist[];  // kHistWordsLds words for SimpleWB, one otherwise (stats_hist_lds_bytes)
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
__global__ void wb_finalize_kernel(int mode, const FrameStats* stats, const int* ccc_argmax, CccState* st,
                                   const DevTables* tabs, FrameWb* out, int n_frames, const unsigned* simple_hist,
                                   float simple_p, int simple_total, const float* ccc_row_best, int* ccc_argmax_out) {
  if (mode == WB_FLOAT) {
    // ccc (one workgroup): the temporal filter is sequential over the frames of the stream -- one lane walks it over
    // argmax values staged in LDS (a handful of float operations per frame) -- while loading the argmax pairs and turning
    // the filtered (u, v) into gains and FrameWb records is done by all lanes, a frame each.
    constexpr int kChunk = 1024;
    __shared__ int s_raw[kChunk][2], s_flt[kChunk][2];
    __shared__ CccState s_state;
    if (threadIdx.x == 0) s_state = *st;
    for (int f0 = 0; f0 < n_frames; f0 += kChunk) {
      const int n = min(kChunk, n_frames - f0);
      __syncthreads();
      if (ccc_row_best) {
        // small batches (ccc_argmax_in_finalize): cv::minMaxLoc over the response -- first maximum in row-major order -- from
        // the 256 row maxima of each frame, here instead of in a launch of its own (256 threads: one row each)
        __shared__ float bv[256];
        __shared__ int br[256];
        const int t = threadIdx.x;
        for (int i = 0; i < n; i++) {
          const float* rb = ccc_row_best + (size_t)(f0 + i) * 256 * 2;
          bv[t] = rb[2 * t];
          br[t] = t;
          __syncthreads();
          for (int off = 128; off > 0; off >>= 1) {
            if (t < off) {
              const float ov = bv[t + off];
              const int orow = br[t + off];
              if (ov > bv[t] || (ov == bv[t] && orow < br[t])) {
                bv[t] = ov;
                br[t] = orow;
              }
            }
            __syncthreads();
          }
          if (t == 0) {
            const int row = br[0];
            s_raw[i][0] = (int)rb[2 * row + 1];
            s_raw[i][1] = row;
            if (ccc_argmax_out) {
              ccc_argmax_out[2 * (f0 + i)] = s_raw[i][0];
              ccc_argmax_out[2 * (f0 + i) + 1] = row;
            }
          }
          __syncthreads();
        }
      } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
          s_raw[i][0] = ccc_argmax[2 * (f0 + i)];
          s_raw[i][1] = ccc_argmax[2 * (f0 + i) + 1];
        }
      }
      __syncthreads();
      // A = Q = I, H = h I, R = r I: the two components of (u, v) never mix, so lane a walks component a (half the
      // dependent float chain -- an IEEE division per step -- of a single lane doing both)
      if (threadIdx.x < 2) {
        const int a = threadIdx.x;
        const bool temporal = s_state.temporal != 0;
        bool first = s_state.first_frame != 0;
        float x = a == 0 ? s_state.st_x : s_state.st_y, pc = a == 0 ? s_state.p_x : s_state.p_y;
        const float kf_h = s_state.kf_h, kf_r = s_state.kf_r;
        int last = a == 0 ? s_state.uv_x : s_state.uv_y;
        // The gain does not see the data: p -> p' depends on (h, r) only.  Two cases keep the IEEE division (a dozen
        // dependent instructions on a single lane) out of most steps without changing a bit of the result:
        //  * h == 0 -- the filter the reference's pipeline actually runs (PARITY.md: the one-argument constructor
        //    leaves a default cv::KalmanFilter) -- gives t2 = +0, k = +0 / r = +0, p' = p + 1 and x' = x + 0 * innov = x;
        //  * otherwise the covariance runs into a fixed point of the float iteration after a few dozen frames, and from
        //    then on recomputing k would reproduce the same value.
        const bool degenerate = kf_h == 0.f && kf_r > 0.f;
        bool steady = false;
        float k = 0.f;
        for (int i = 0; i < n; i++) {
          const int z = s_raw[i][a];
          int o = z;
          if (temporal) {
            if (first) {
              first = false;
              x = (float)z;
            } else if (degenerate) {
              pc = pc + 1.0f;
              o = (int)x;
            } else {
              // cv::KalmanFilter(2,2,0) predict + correct
              const float x_pre = x;
              if (!steady) {
                const float p_pre = pc + 1.0f;
                const float t2 = kf_h * p_pre;
                const float t3 = t2 * kf_h + kf_r;
                k = t2 / t3;
                const float p_new = p_pre - k * t2;
                steady = p_new == pc;
                pc = p_new;
              }
              const float innov = (float)z - kf_h * x_pre;
              x = x_pre + k * innov;
              o = (int)x;
            }
          }
          s_flt[i][a] = o;
          last = o;
        }
        // both lanes have read s_state before either writes it: they run in lockstep within the wave
        __builtin_amdgcn_wave_barrier();
        if (a == 0) {
          s_state.st_x = x;
          s_state.p_x = pc;
          s_state.uv_x = last;
          if (n > 0 && temporal) s_state.first_frame = 0;
        } else {
          s_state.st_y = x;
          s_state.p_y = pc;
          s_state.uv_y = last;
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        FrameWb w = {};
        w.uv_raw[0] = s_raw[i][0];
        w.uv_raw[1] = s_raw[i][1];
        // Illuminant at log-chroma bin (u, v) -> channel gains (Barron 2015, as the reference uses it without the luminance
        // term): each channel is divided by its attenuation exp(-L) -- green is the anchor, L = 0 -- and the triple is scaled
        // so that the smallest gain is exactly one.  exp(-L) per bin is a host-built table (DevTables::exp_neg_tab), so no
        // device transcendental is involved; channel order B, G, R like every FrameWb field.
        const int bin[3] = {clampi(s_flt[i][1], 0, 255), -1, clampi(s_flt[i][0], 0, 255)};  // B <- v, G anchor, R <- u
        float smallest = 1.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          w.fg[c] = bin[c] < 0 ? 1.0f : 1.0f / tabs->exp_neg_tab[bin[c]];
          smallest = fminf(smallest, w.fg[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) w.fg[c] /= smallest;
        w.uv[0] = s_flt[i][0];
        w.uv[1] = s_flt[i][1];
        out[f0 + i] = w;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) *st = s_state;
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
  StatShard fs = stats[f].shard[0];
  for (int i = 1; i < kStatShards; i++) {
    for (int k = 0; k < 5; k++) fs.sum[k] += stats[f].shard[i].sum[k];
    for (int k = 0; k < 3; k++) fs.mx[k] = max(fs.mx[k], stats[f].shard[i].mx[k]);
  }
  wb_gains_from_sums(mode, fs, w);
  out[f] = w;
}

}  // namespace

static unsigned stats_hist_lds_bytes(int mode) { return (mode == WB_SIMPLE ? (unsigned)kHistWordsLds : 1u) * (unsigned)sizeof(unsigned); }

void launch_stats(const StatsParams& p, const Tunables& tn, hipStream_t stream) {
  if (p.n_frames <= 0) return;
  if (bayer_fast_geometry(p.src, p.src_step, p.src_frame_stride, p.rows, p.cols, p.src_kind)) {
    ItemMap im{p.cols / 4, 1.0f / (float)(p.cols / 4)};
    const int items = (p.rows / 2) * (p.cols / 4);
    // keep >= 1 block per 2^20 items so the 32-bit per-thread partial sums cannot overflow
    // wave tasks: 64 groups wide x pairs_per_task row pairs.  The wave total of the largest statistic
    // (pca: sum of squares) is 64 lanes * 8 px * 255^2 * pairs_per_task: 128 pairs keep it below 2^32
    const int groups = p.cols / 4, n_pairs = p.rows / 2;
    const int col_waves = (groups + 63) / 64;
    // per frame: 512 wave tasks when the batch fills the chip anyway, 2048 for a single frame (shorter row walks; with the
    // sums sharded over eight lines the workgroups no longer queue up on one line's atomics: 65.1 -> 63.9 us per call)
    const int budget = grid_multiple_of_8(tn.stats_blocks) * 4;
    const int target_tasks = std::max(8, std::min(budget / (p.n_frames == 1 ? 4 : 8), budget / std::max(1, std::min(p.n_frames, 16))));
    int pairs_per_task = std::max(2, (int)(((long long)col_waves * n_pairs + target_tasks - 1) / target_tasks));
    pairs_per_task = std::min((pairs_per_task + 1) & ~1, 128);  // even: the kernel consumes two pairs per iteration
    const int n_tasks = col_waves * ((n_pairs + pairs_per_task - 1) / pairs_per_task);
    const int task_blocks = (n_tasks + kBlock / 64 - 1) / (kBlock / 64);
    const dim3 grid((task_blocks + 7) / 8 * 8, p.n_frames);  // a multiple of 8: one contiguous task range per XCD
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
    hipLaunchKernelGGL(stats_color_kernel, dim3(per_frame, p.n_frames), dim3(kBlock), stats_hist_lds_bytes(p.mode), stream, p, im, items);
    return;
  }
  long long npix = (long long)p.rows * p.cols;
  int blocks = std::max(grid_blocks_for(npix, 1024), (int)((npix + (1 << 22) - 1) >> 22));
  hipLaunchKernelGGL(stats_generic_kernel, dim3(blocks, p.n_frames), dim3(kBlock), stats_hist_lds_bytes(p.mode), stream, p);
}

void launch_wb_finalize(int mode, const FrameStats* stats, const int* ccc_argmax, CccState* ccc_state,
                        const DevTables* tabs, FrameWb* out, int n_frames, hipStream_t stream, const unsigned* simple_hist,
                        float simple_p, int simple_total, const float* ccc_row_best, int* ccc_argmax_out) {
  if (n_frames <= 0) return;
  if (mode == WB_FLOAT) {
    hipLaunchKernelGGL(wb_finalize_kernel, dim3(1), dim3(256), 0, stream, mode, stats, ccc_argmax, ccc_state, tabs, out, n_frames,
                       simple_hist, simple_p, simple_total, ccc_row_best, ccc_argmax_out);
  } else {
    hipLaunchKernelGGL(wb_finalize_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, stream, mode, stats, ccc_argmax,
                       ccc_state, tabs, out, n_frames, simple_hist, simple_p, simple_total, nullptr, nullptr);
  }
}

}  // namespace rip
