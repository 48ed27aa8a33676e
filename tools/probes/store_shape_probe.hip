// Which access shapes reach the write-heavy streaming rate of the box?  A 1 : 3 read : write stream (the debayer-only
// chain's mix) with different per-lane load / store widths.  hipcc --offload-arch=gfx950 -O3 store_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A: dword load, dwordx3 store per lane (12 B lane stride): the chain's current shape
__global__ __launch_bounds__(256) void kA(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint32_t v = src[i];
    u32x3 o = {v, v ^ 1u, v ^ 2u};
    *reinterpret_cast<u32x3*>(dst + 3 * i) = o;
  }
}
// B: dwordx4 load, three dwordx4 stores per lane, lane-contiguous 48 B
__global__ __launch_bounds__(256) void kB(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const u32x4 v = src[i];
    dst[3 * i] = v;
    dst[3 * i + 1] = v ^ 1u;
    dst[3 * i + 2] = v ^ 2u;
  }
}
// C: dwordx4 load, three dwordx4 stores, each instruction wave-contiguous (1 KB per wave)
__global__ __launch_bounds__(256) void kC(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n4) {
  const int lane = threadIdx.x & 63;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const u32x4 v = src[i];
    const size_t wave0 = i - lane;  // first element of this wave's 64
    dst[3 * wave0 + lane] = v;
    dst[3 * wave0 + 64 + lane] = v ^ 1u;
    dst[3 * wave0 + 128 + lane] = v ^ 2u;
  }
}
// D: as A but non-temporal stores
__global__ __launch_bounds__(256) void kD(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint32_t v = src[i];
    __builtin_nontemporal_store(v, dst + 3 * i);
    __builtin_nontemporal_store(v ^ 1u, dst + 3 * i + 1);
    __builtin_nontemporal_store(v ^ 2u, dst + 3 * i + 2);
  }
}
template <typename F>
static void run(const char* name, F launch, double bytes) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 8; r++) {
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  printf("%-58s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
}
int main() {
  const size_t n = (size_t)1 << 30;  // source bytes
  uint32_t *src, *dst;
  hipMalloc(&src, n); hipMalloc(&dst, 3 * n);
  hipMemset(src, 1, n);
  const double bytes = 4.0 * n;
  for (int blocks : {2048, 8192, 65536, 1 << 20}) {
    printf("grid %d x 256\n", blocks);
    run("A dword load, dwordx3 store (12 B/lane)", [&] { hipLaunchKernelGGL(kA, dim3(blocks), dim3(256), 0, 0, src, dst, n / 4); }, bytes);
    run("B dwordx4 load, 3 x dwordx4 store, 48 B per lane contiguous", [&] { hipLaunchKernelGGL(kB, dim3(blocks), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, n / 16); }, bytes);
    run("C dwordx4 load, 3 x dwordx4 store, wave-contiguous", [&] { hipLaunchKernelGGL(kC, dim3(blocks), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, n / 16); }, bytes);
    run("D dword load, 3 x dword nt store", [&] { hipLaunchKernelGGL(kD, dim3(blocks), dim3(256), 0, 0, src, dst, n / 4); }, bytes);
  }
  return 0;
}
