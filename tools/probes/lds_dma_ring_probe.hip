#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
// probe: buffer_load_dwordx4 ... offen lds  (async global->LDS, M0 = wave-uniform LDS base)
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rsrc), "s"(lds_base) : "memory");
}
__global__ __launch_bounds__(256) void k(const uint8_t* src, uint32_t* out, int nframes, unsigned frame_bytes, unsigned total_chunks) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const unsigned stage_bytes = 2 * 256 * 16;
  const int NB = 3, D = 2;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;  // LDS byte address of the dynamic segment
  auto issue = [&](int f) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)f * frame_bytes), 0, (int)frame_bytes, 0x00020000);
    const unsigned base = lds0 + (unsigned)(f % NB) * stage_bytes;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const unsigned i = tid + j * 256;
      const unsigned wave_base = __builtin_amdgcn_readfirstlane(base + ((tid & ~63) + j * 256) * 16);
      if (i < total_chunks) glds16(r, i * 16, wave_base);
    }
  };
  for (int f = 0; f < D && f < nframes; f++) issue(f);
  uint32_t acc = 0;
  for (int f = 0; f < nframes; f++) {
    // loads(f) landed: all but the newest (D-1)*2 of my loads
    if (f + D - 1 < nframes) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (f + D < nframes) issue(f + D);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(lds + (f % NB) * stage_bytes);
    // read something written by another wave
    const unsigned idx = ((tid * 7 + 13) % (total_chunks * 4));
    acc += w[idx] * (f + 1);
  }
  out[blockIdx.x * 256 + tid] = acc;
}
int main() {
  const int nframes = 16; const unsigned total_chunks = 400, frame_bytes = total_chunks * 16 + 64;
  std::vector<uint8_t> h((size_t)nframes * frame_bytes);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 2654435761u >> 13);
  uint8_t* d; uint32_t* o;
  hipMalloc(&d, h.size()); hipMalloc(&o, 256 * 64 * 4);
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  k<<<64, 256, 3 * 2 * 256 * 16>>>(d, o, nframes, frame_bytes, total_chunks);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  std::vector<uint32_t> ho(256 * 64);
  hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < 64; b++) for (int t = 0; t < 256; t++) {
    uint32_t acc = 0;
    for (int f = 0; f < nframes; f++) {
      unsigned idx = ((t * 7 + 13) % (total_chunks * 4));
      uint32_t v; memcpy(&v, &h[(size_t)f * frame_bytes + idx * 4], 4);
      acc += v * (f + 1);
    }
    if (acc != ho[b * 256 + t]) bad++;
  }
  printf("mismatches: %d of %d\n", bad, 256 * 64);
  return bad != 0;
}
