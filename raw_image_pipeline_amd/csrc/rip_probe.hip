// rip_probe.hip -- streaming microbenchmarks behind rip_debug_hbm_probe (measurement hook, no reference counterpart):
// what THIS box's memory system delivers to the access shapes the pipeline's kernels are made of, so that bench.py can put
// a hand-written streaming rate -- not a framework's elementwise kernel -- beside the 8 TB/s spec figure
// (MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy).
//
// Every kernel is one item per lane, no loop, launched on a grid that covers the buffer: short-lived workgroups in address
// order are what streamed best in the round-3 probes (tools/probes/store_shape_probe.hip: 6.1 TB/s on a 64 k-workgroup
// grid against 5.2 TB/s with persistent ones).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rip_kernels.hpp"

namespace rip {
namespace {
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kProbeBlock = 256;

// 1 : 1 copy, 16 bytes per lane
__global__ __launch_bounds__(kProbeBlock) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * kProbeBlock + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
// read only: every lane folds its 16 bytes, one lane per wave writes the wave's fold (1 / 256 of the bytes read)
__global__ __launch_bounds__(kProbeBlock) void probe_read_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, size_t n16) {
  const size_t i = (size_t)blockIdx.x * kProbeBlock + threadIdx.x;
  uint32_t v = 0;
  if (i < n16) {
    const u32x4 a = src[i];
    v = a.x ^ a.y ^ a.z ^ a.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sink[i >> 6] = v;
}
// write only
__global__ __launch_bounds__(kProbeBlock) void probe_fill_kernel(u32x4* __restrict__ dst, size_t n16, uint32_t value) {
  const size_t i = (size_t)blockIdx.x * kProbeBlock + threadIdx.x;
  const u32x4 v = {value, value, value, value};
  if (i < n16) dst[i] = v;
}
// 1 : 3 expand, the fused chain's shape: one dword (four Bayer bytes) in, twelve bytes (four BGR pixels) out per lane
__global__ __launch_bounds__(kProbeBlock) void probe_expand13_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n4, int nt) {
  const size_t i = (size_t)blockIdx.x * kProbeBlock + threadIdx.x;
  if (i >= n4) return;
  const uint32_t v = src[i];
  const u32x3 o = {v, v ^ 0x01010101u, v ^ 0x02020202u};
  if (nt)
    __builtin_nontemporal_store(o, reinterpret_cast<u32x3*>(dst + 3 * i));
  else
    *reinterpret_cast<u32x3*>(dst + 3 * i) = o;
}
// 3 : 3 copy in 12-byte lanes: the remap's store shape fed by a contiguous read (its gather replaced by a stream)
__global__ __launch_bounds__(kProbeBlock) void probe_copy12_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n12) {
  const size_t i = (size_t)blockIdx.x * kProbeBlock + threadIdx.x;
  if (i >= n12) return;
  *reinterpret_cast<u32x3*>(dst + 3 * i) = *reinterpret_cast<const u32x3*>(src + 3 * i);
}
}  // namespace

// bytes: size of the SOURCE stream (copy, read, expand13, copy12) or of the destination (fill).  Returns the bytes the launch
// moves (read + written); 0 for an unknown kind.  src / dst must hold bytes and 3 * bytes respectively.
size_t launch_hbm_probe(int kind, const void* src, void* dst, size_t bytes, hipStream_t stream) {
  const auto blocks = [](size_t items) { return dim3((unsigned)((items + kProbeBlock - 1) / kProbeBlock)); };
  switch (kind) {
    case 0:
      hipLaunchKernelGGL(probe_copy_kernel, blocks(bytes / 16), dim3(kProbeBlock), 0, stream, static_cast<const u32x4*>(src), static_cast<u32x4*>(dst), bytes / 16);
      return 2 * (bytes / 16 * 16);
    case 1:
      hipLaunchKernelGGL(probe_read_kernel, blocks(bytes / 16), dim3(kProbeBlock), 0, stream, static_cast<const u32x4*>(src), static_cast<uint32_t*>(dst), bytes / 16);
      return bytes / 16 * 16 + bytes / 16 / 64 * 4;
    case 2:
      hipLaunchKernelGGL(probe_fill_kernel, blocks(bytes / 16), dim3(kProbeBlock), 0, stream, static_cast<u32x4*>(dst), bytes / 16, 0x5a5a5a5au);
      return bytes / 16 * 16;
    case 3:
    case 4:
      hipLaunchKernelGGL(probe_expand13_kernel, blocks(bytes / 4), dim3(kProbeBlock), 0, stream, static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst), bytes / 4,
                         kind == 4 ? 1 : 0);
      return 4 * (bytes / 4 * 4);
    case 5:
      hipLaunchKernelGGL(probe_copy12_kernel, blocks(bytes / 12), dim3(kProbeBlock), 0, stream, static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst), bytes / 12);
      return 2 * (bytes / 12 * 12);
    default:
      return 0;
  }
}
}  // namespace rip
