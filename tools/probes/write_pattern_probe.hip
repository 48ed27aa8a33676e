// write_pattern_probe.hip -- what the memory system does with the ring remap's WRITE pattern, without the remap around it.
// The remap's stores alone take 1.17 ms for 3.85 GB (3.3 TB/s) where a linear fill reaches 6.9 TB/s (EXPERIMENTS.md round 6).
// Is that the address pattern (192-byte row segments of 64 x 16 tiles, 6 frames 15 MB apart per visit, 1 536 workgroups
// walking one band) or the kernel's structure (one store per wave and frame, then a counted wait and a barrier)?  This probe
// writes the same 256 x 2048 x 7344 bytes in several orders with NOTHING else in the kernel.
//   mode 0: linear fill, 16 B per lane, one short-lived workgroup per 4 KB (the library's fill probe)
//   mode 1: the remap's order -- persistent grid of 1 536 workgroups (XCD share b % 8), tiles dealt in runs of `run` tiles,
//           `fpv` frames per visit, every wave stores its 4 rows x 192 B of every frame back to back (no wait, no barrier)
//   mode 2: as 1 with a workgroup barrier and s_waitcnt vmcnt(1) per frame (the ring's cadence: at most two stores in flight per wave)
//   mode 3: as 1, short-lived workgroups: grid (1536, groups) like the product kernel
//   mode 4: as 3 with the cadence of mode 2
//   mode 5-7: the frame loop OUTSIDE the workgroup's tiles (frame-major): all workgroups of a frame group write the same frame at about the same time
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/write_pattern_probe.hip -o tools/probes/bin/write_pattern_probe
// usage: write_pattern_probe [tile_w_px=64] [tile_h=16] [fpv=6] [run_rows=4] [frames=256] [frame_stride_pad_bytes=0] [only_mode]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kW = 2448, kH = 2048, kPitch = kW * 3;
constexpr size_t kFrameBytes = (size_t)kPitch * kH;

__global__ __launch_bounds__(256) void fill_kernel(u32x4* dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = u32x4{1u, 2u, 3u, 4u};
}

struct P {
  uint8_t* dst;
  int tw, th, tiles_x, tiles_y, fpv, run, frames, cadence, persistent, frame_major;
  size_t frame_stride;  // bytes between frames (kFrameBytes + pad)
  unsigned* sync;       // [frames][8 shards x 32 dwords]: workgroups done with a frame, per XCD share; null = no frame lock
  int lead;             // a workgroup starts frame f only when every workgroup has finished frame f - lead (bounded spin: a hint, not a dependency)
};

__device__ __forceinline__ int dealt(int ti, int xcd, int run, int ntiles) {
  const int r = ti / run;
  const int t = (r * 8 + xcd) * run + (ti - r * run);
  return t < ntiles ? t : -1;
}

// persistent grid, static ownership of tiles, the frame loop outermost, and a LOOSE frame lock: per frame and XCD share a counter of
// finished workgroups; before frame f a workgroup waits (bounded) until all shares have finished frame f - lead
__global__ __launch_bounds__(256) void locked_kernel(P p) {
  const int ntiles = p.tiles_x * p.tiles_y;
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, share = gridDim.x >> 3;
  const int per_xcd = ((ntiles + p.run - 1) / p.run + 7) / 8 * p.run;
  const int lanes_per_row = p.tw / 4;
  const int lrow = threadIdx.x / lanes_per_row, lcol = threadIdx.x % lanes_per_row;
  for (int f = 0; f < p.frames; f++) {
    if (f >= p.lead && threadIdx.x < 8) {
      const unsigned* c = p.sync + ((size_t)(f - p.lead) * 8 + threadIdx.x) * 32;
      for (int spin = 0; spin < 20000; spin++) {
        const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64(v < (unsigned)share) == 0) break;  // lanes 0..7 active: all eight shares are through
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    for (int ti = wg; ti < per_xcd; ti += share) {
      const int tile = dealt(ti, xcd, p.run, ntiles);
      if (tile < 0) continue;
      const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
      const int y = ty * p.th + lrow, x = tx * p.tw + lcol * 4;
      if (y < kH && x < kW && lrow < p.th)
        *reinterpret_cast<u32x3*>(p.dst + (size_t)f * p.frame_stride + (size_t)y * kPitch + (size_t)x * 3) = u32x3{(uint32_t)f, (uint32_t)tile, (uint32_t)threadIdx.x};
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(p.sync + ((size_t)f * 8 + xcd) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void tile_kernel(P p) {
  const int ntiles = p.tiles_x * p.tiles_y;
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, share = gridDim.x >> 3;
  const int per_xcd = ((ntiles + p.run - 1) / p.run + 7) / 8 * p.run;
  const int groups = (p.frames + p.fpv - 1) / p.fpv;
  const int lanes_per_row = p.tw / 4;  // 4 px = 12 B per lane
  const int lrow = threadIdx.x / lanes_per_row, lcol = threadIdx.x % lanes_per_row;
  const int g0 = p.persistent ? 0 : blockIdx.y, g1 = p.persistent ? groups : blockIdx.y + 1;
  for (int g = g0; g < g1; g++) {
    const int f0 = g * p.fpv, f1 = min(p.frames, f0 + p.fpv);
    if (p.frame_major) {  // the workgroup's tiles inside the frame loop: every workgroup of the group is on the same frame at about the same time
      for (int f = f0; f < f1; f++)
        for (int ti = wg; ti < per_xcd; ti += share) {
          const int tile = dealt(ti, xcd, p.run, ntiles);
          if (tile < 0) continue;
          const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
          const int y = ty * p.th + lrow, x = tx * p.tw + lcol * 4;
          if (y < kH && x < kW && lrow < p.th)
            *reinterpret_cast<u32x3*>(p.dst + (size_t)f * p.frame_stride + (size_t)y * kPitch + (size_t)x * 3) = u32x3{(uint32_t)f, (uint32_t)tile, (uint32_t)threadIdx.x};
          if (p.cadence) {
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          }
        }
      continue;
    }
    for (int ti = wg; ti < per_xcd; ti += share) {
      const int tile = dealt(ti, xcd, p.run, ntiles);
      if (tile < 0) continue;
      const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
      const int y = ty * p.th + lrow, x = tx * p.tw + lcol * 4;
      const bool in = y < kH && x < kW && lrow < p.th;
      const size_t off = (size_t)y * kPitch + (size_t)x * 3;
      for (int f = f0; f < f1; f++) {
        if (in) *reinterpret_cast<u32x3*>(p.dst + (size_t)f * p.frame_stride + off) = u32x3{(uint32_t)f, (uint32_t)tile, (uint32_t)threadIdx.x};
        if (p.cadence) {
          asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
    }
  }
}

static float time_ms(hipStream_t s, int reps, const std::vector<float>& dummy, void (*launch)(void*), void* ctx) {
  (void)dummy;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch(ctx);
  hipStreamSynchronize(s);
  float best = 1e9f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(a, s);
    launch(ctx);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

struct Ctx {
  P p;
  int mode;
  size_t bytes;
};
static void launch(void* c) {
  Ctx* x = static_cast<Ctx*>(c);
  if (x->mode == 0) {
    const size_t n16 = x->bytes / 16;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<u32x4*>(x->p.dst), n16);
    return;
  }
  if (x->mode >= 8) {
    hipMemsetAsync(x->p.sync, 0, (size_t)x->p.frames * 8 * 32 * 4, 0);
    hipLaunchKernelGGL(locked_kernel, dim3(x->mode >= 11 ? 1024 : 1536), dim3(256), 0, 0, x->p);
    return;
  }
  const int groups = (x->p.frames + x->p.fpv - 1) / x->p.fpv;
  hipLaunchKernelGGL(tile_kernel, dim3(1536, x->p.persistent ? 1 : groups), dim3(256), 0, 0, x->p);
}

int main(int argc, char** argv) {
  const int tw = argc > 1 ? atoi(argv[1]) : 64, th = argc > 2 ? atoi(argv[2]) : 16, fpv = argc > 3 ? atoi(argv[3]) : 6;
  const int run_rows = argc > 4 ? atoi(argv[4]) : 4, frames = argc > 5 ? atoi(argv[5]) : 256;
  const size_t pad = argc > 6 ? (size_t)atoll(argv[6]) : 0;
  const int only = argc > 7 ? atoi(argv[7]) : -1;
  if (tw * th != 1024 || tw % 4) {
    printf("tile must hold 1024 pixels (256 lanes x 4 px)\n");
    return 1;
  }
  Ctx c;
  c.bytes = (size_t)frames * kFrameBytes;
  c.p.frame_stride = kFrameBytes + pad;
  hipMalloc(&c.p.dst, (size_t)frames * c.p.frame_stride + 4096);
  c.p.tw = tw;
  c.p.th = th;
  c.p.tiles_x = (kW + tw - 1) / tw;
  c.p.tiles_y = (kH + th - 1) / th;
  c.p.fpv = fpv;
  c.p.run = run_rows > 0 ? run_rows * c.p.tiles_x : (c.p.tiles_x * c.p.tiles_y + 7) / 8;
  c.p.frames = frames;
  const char* names[] = {"linear fill, 16 B lanes, short-lived workgroups", "remap order, persistent grid, stores back to back",
                         "remap order, persistent grid, barrier + vmcnt(1) per frame", "remap order, one workgroup per frame group, back to back",
                         "remap order, one workgroup per frame group, barrier + vmcnt(1) per frame",
                         "FRAME-MAJOR inside the workgroup (its 3-4 tiles per frame), one workgroup per frame group, back to back",
                         "FRAME-MAJOR inside the workgroup, one workgroup per frame group, barrier + vmcnt(1) per store",
                         "FRAME-MAJOR inside the workgroup, persistent grid, back to back",
                         "persistent, frames outermost, LOOSE FRAME LOCK (lead 1)", "persistent, frames outermost, loose frame lock (lead 2)",
                         "persistent, frames outermost, loose frame lock (lead 4)", "persistent 1024 workgroups, loose frame lock (lead 1)",
                         "persistent 1024 workgroups, loose frame lock (lead 2)"};
  printf("tile %d x %d px (%d-byte row segments), %d frames per visit, runs of %d tiles, %d frames, %.2f GB, frame stride + %zu B\n", tw, th, tw * 3, fpv, c.p.run, frames,
         c.bytes / 1e9, pad);
  hipMalloc(&c.p.sync, (size_t)frames * 8 * 32 * 4);
  for (int mode = 0; mode < 13; mode++) {
    if (only >= 0 && mode != only && mode != 0 && !(only == 8 && mode >= 8)) continue;
    c.mode = mode;
    c.p.persistent = mode == 1 || mode == 2 || mode == 7;
    c.p.cadence = mode == 2 || mode == 4 || mode == 6;
    c.p.frame_major = mode >= 5;
    c.p.lead = mode == 8 || mode == 11 ? 1 : (mode == 9 || mode == 12 ? 2 : 4);
    const float ms = time_ms(0, 5, {}, launch, &c);
    printf("mode %d  %-78s %7.3f ms  %6.0f GB/s\n", mode, names[mode], ms, c.bytes / (ms * 1e-3) / 1e9);
  }
  hipFree(c.p.dst);
  return 0;
}
