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
// read only: every lane folds four 16-byte loads (all in flight together, a wave covers 4 x 1 KB), one lane per wave writes
// the wave's fold (1 / 1024 of the bytes read).  One load per lane left the read rate at the workgroup launch rate.
constexpr int kReadUnroll = 4;
__global__ __launch_bounds__(kProbeBlock) void probe_read_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, size_t n16) {
  const size_t base = (size_t)blockIdx.x * (kProbeBlock * kReadUnroll) + threadIdx.x;
  u32x4 a[kReadUnroll];
#pragma unroll
  for (int k = 0; k < kReadUnroll; k++) {
    const size_t i = base + (size_t)k * kProbeBlock;
    a[k] = i < n16 ? src[i] : u32x4{0u, 0u, 0u, 0u};
  }
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < kReadUnroll; k++) v ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sink[((size_t)blockIdx.x * kProbeBlock + threadIdx.x) >> 6] = v;
}
// the same with eight loads in flight per lane and the non-temporal hint (the guide's LDS-DMA read streams reach 6.4 TB/s with
// the default policy and 6.5-6.8 with nt: is the four-load kernel above leaving rate on the table?)
constexpr int kReadUnrollNt = 8;
__global__ __launch_bounds__(kProbeBlock) void probe_read_nt_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, size_t n16) {
  const size_t base = (size_t)blockIdx.x * (kProbeBlock * kReadUnrollNt) + threadIdx.x;
  u32x4 a[kReadUnrollNt];
#pragma unroll
  for (int k = 0; k < kReadUnrollNt; k++) {
    const size_t i = base + (size_t)k * kProbeBlock;
    a[k] = i < n16 ? __builtin_nontemporal_load(src + i) : u32x4{0u, 0u, 0u, 0u};
  }
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < kReadUnrollNt; k++) v ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sink[((size_t)blockIdx.x * kProbeBlock + threadIdx.x) >> 6] = v;
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
// the same 1 : 3 expand with 16 bytes in and 48 contiguous bytes out per lane (three 16-byte stores): the shape a 16-pixel-wide
// demosaic item would have
__global__ __launch_bounds__(kProbeBlock) void probe_expand13_wide_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int nt) {
  const size_t i = (size_t)blockIdx.x * kProbeBlock + threadIdx.x;
  if (i >= n16) return;
  const u32x4 v = src[i];
  const u32x4 a = {v.x, v.x ^ 1u, v.x ^ 2u, v.y}, b = {v.y ^ 1u, v.y ^ 2u, v.z, v.z ^ 1u}, c = {v.z ^ 2u, v.w, v.w ^ 1u, v.w ^ 2u};
  if (nt) {
    __builtin_nontemporal_store(a, dst + 3 * i);
    __builtin_nontemporal_store(b, dst + 3 * i + 1);
    __builtin_nontemporal_store(c, dst + 3 * i + 2);
  } else {
    dst[3 * i] = a;
    dst[3 * i + 1] = b;
    dst[3 * i + 2] = c;
  }
}
// the same byte ratio with every store instruction of a wave covering 1 KB of contiguous bytes (what a kernel that transposes
// its output through LDS or lane shuffles would issue): is the 1 : 3 rate a property of the ratio or of the 12 / 48-byte lanes?
__global__ __launch_bounds__(kProbeBlock) void probe_expand13_coalesced_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int nt) {
  const size_t i = (size_t)blockIdx.x * kProbeBlock + threadIdx.x;
  if (i >= n16) return;
  const u32x4 v = src[i];
  const u32x4 a = {v.x, v.x ^ 1u, v.x ^ 2u, v.y}, b = {v.y ^ 1u, v.y ^ 2u, v.z, v.z ^ 1u}, c = {v.z ^ 2u, v.w, v.w ^ 1u, v.w ^ 2u};
  u32x4* out = dst + 3 * (i & ~(size_t)63) + (i & 63);  // the wave's 3 KB: three runs of 64 x 16 bytes
  if (nt) {
    __builtin_nontemporal_store(a, out);
    __builtin_nontemporal_store(b, out + 64);
    __builtin_nontemporal_store(c, out + 128);
  } else {
    out[0] = a;
    out[64] = b;
    out[128] = c;
  }
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
      hipLaunchKernelGGL(probe_read_kernel, blocks((bytes / 16 + kReadUnroll - 1) / kReadUnroll), dim3(kProbeBlock), 0, stream, static_cast<const u32x4*>(src),
                         static_cast<uint32_t*>(dst), bytes / 16);
      return bytes / 16 * 16 + bytes / 16 / 64 / kReadUnroll * 4;
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
    case 6:
    case 7:
      hipLaunchKernelGGL(probe_expand13_wide_kernel, blocks(bytes / 16), dim3(kProbeBlock), 0, stream, static_cast<const u32x4*>(src), static_cast<u32x4*>(dst), bytes / 16,
                         kind == 7 ? 1 : 0);
      return 4 * (bytes / 16 * 16);
    case 9:
    case 10:
      if ((bytes / 16) % 64) return 0;  // whole waves only
      hipLaunchKernelGGL(probe_expand13_coalesced_kernel, blocks(bytes / 16), dim3(kProbeBlock), 0, stream, static_cast<const u32x4*>(src), static_cast<u32x4*>(dst),
                         bytes / 16, kind == 10 ? 1 : 0);
      return 4 * (bytes / 16 * 16);
    case 8:
      hipLaunchKernelGGL(probe_read_nt_kernel, blocks((bytes / 16 + kReadUnrollNt - 1) / kReadUnrollNt), dim3(kProbeBlock), 0, stream, static_cast<const u32x4*>(src),
                         static_cast<uint32_t*>(dst), bytes / 16);
      return bytes / 16 * 16 + bytes / 16 / 64 / kReadUnrollNt * 4;
    default:
      return 0;
  }
}
}  // namespace rip
