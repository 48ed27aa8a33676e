// What do the d16 LDS byte loads leave in the other half of their destination on gfx950 (SRAM-ECC on)?
// hipcc --offload-arch=gfx950 -O2 d16_probe.hip -o d16_probe && ./d16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  __shared__ unsigned char t[256];
  t[threadIdx.x] = (unsigned char)(threadIdx.x * 3 + 1);
  __syncthreads();
  unsigned a = 0xAAAAAAAAu, b = 0xBBBBBBBBu, addr = (unsigned)(uintptr_t)&t[0] + threadIdx.x;
  asm volatile("ds_read_u8_d16 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(addr));
  asm volatile("ds_read_u8_d16_hi %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(b) : "v"(addr));
  unsigned c = 0xCCCCCCCCu;
  asm volatile("ds_read_u8_d16 %0, %1\n ds_read_u8_d16_hi %0, %2\n s_waitcnt lgkmcnt(0)" : "+v"(c) : "v"(addr), "v"(addr + 1));
  out[threadIdx.x * 3] = a;
  out[threadIdx.x * 3 + 1] = b;
  out[threadIdx.x * 3 + 2] = c;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 64 * 3 * 4);
  k<<<1, 64>>>(d);
  unsigned h[192];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; i++) printf("lane %d: d16 %08x  d16_hi %08x  both %08x\n", i, h[i * 3], h[i * 3 + 1], h[i * 3 + 2]);
  return 0;
}
