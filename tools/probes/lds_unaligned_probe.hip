// lds_unaligned_probe.hip -- does gfx950's LDS serve ds_read_b64 / ds_read_b32 at ANY byte address (unaligned access mode), and at
// what cost?  The remap gathers six interleaved BGR bytes per tap row at an arbitrary byte offset: today three aligned dwords +
// two v_alignbyte_b32; one unaligned ds_read_b64 would replace them.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_unaligned_probe.hip -o tools/probes/bin/lds_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void lds_b64(uint32_t& lo, uint32_t& hi, unsigned addr) {
  uint64_t v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  lo = (uint32_t)v;
  hi = (uint32_t)(v >> 32);
}
__device__ __forceinline__ uint32_t lds_b32(unsigned addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  return v;
}

__global__ void check_kernel(uint32_t* out, int* bad) {
  __shared__ __attribute__((aligned(16))) uint8_t s[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const unsigned base = (unsigned)reinterpret_cast<uintptr_t>(s);
  for (int k = 0; k < 16; k++) {
    const unsigned a = threadIdx.x * 13 + k;
    uint32_t lo, hi;
    lds_b64(lo, hi, base + a);
    const uint32_t w = lds_b32(base + a + 1);
    uint64_t exp = 0;
    for (int j = 7; j >= 0; j--) exp = (exp << 8) | s[a + j];
    uint32_t expw = 0;
    for (int j = 3; j >= 0; j--) expw = (expw << 8) | s[a + 1 + j];
    if (lo != (uint32_t)exp || hi != (uint32_t)(exp >> 32)) atomicAdd(&bad[0], 1);
    if (w != expw) atomicAdd(&bad[1], 1);
    if (threadIdx.x == 1 && k == 3) {
      out[0] = lo;
      out[1] = hi;
      out[2] = (uint32_t)exp;
      out[3] = (uint32_t)(exp >> 32);
    }
  }
}

// MODE 0: aligned read2_b32 + b32 + 2 alignbyte (today's lds_load6); 1: one unaligned ds_read_b64
template <int MODE>
__global__ void time_kernel(uint32_t* out, int iters, unsigned stride, unsigned mis) {
  __shared__ __attribute__((aligned(16))) uint8_t s[32768];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) s[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const unsigned base = (unsigned)reinterpret_cast<uintptr_t>(s);
  unsigned a = (threadIdx.x * stride + mis) & 16383u;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const unsigned addr = base + ((a + k * 232u) & 32767u & ~0u);
      if (MODE == 1) {
        uint64_t v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
        acc += (uint32_t)v ^ (uint32_t)(v >> 32);
      } else {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(s + ((addr - base) & ~3u));
        const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
        acc += __builtin_amdgcn_alignbyte(d1, d0, addr & 3u) ^ __builtin_amdgcn_alignbyte(d2, d1, addr & 3u);
      }
    }
    a = (a + acc % 5u * 6u + 6u) & 16383u;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  uint32_t* out;
  int* bad;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&bad, 8);
  hipMemset(bad, 0, 8);
  check_kernel<<<1, 256>>>(out, bad);
  int hb[2];
  uint32_t ho[4];
  hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
  printf("unaligned ds_read_b64: %d mismatches of 4096; unaligned ds_read_b32: %d mismatches (sample %08x %08x expected %08x %08x)\n", hb[0], hb[1], ho[0], ho[1], ho[2], ho[3]);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (unsigned stride : {6u, 12u, 24u}) {
    for (unsigned mis : {0u, 1u, 2u, 3u}) {
      float ms[2];
      for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 2; rep++) {
          hipEventRecord(e0);
          if (mode == 0)
            time_kernel<0><<<2048, 256>>>(out, 2000, stride, mis);
          else
            time_kernel<1><<<2048, 256>>>(out, 2000, stride, mis);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          hipEventElapsedTime(&ms[mode], e0, e1);
        }
      }
      printf("lane stride %2u B, byte misalignment %u: aligned 3 dwords + 2 alignbyte %.3f ms, one unaligned ds_read_b64 %.3f ms\n", stride, mis, ms[0], ms[1]);
    }
  }
  return hb[0] + hb[1] ? 1 : 0;
}
