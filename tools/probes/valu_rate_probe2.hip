// Probe 2: VALU issue cost by OPERAND FORM and in mixes (gfx950), with the shader clock measured in-kernel
// (s_memtime vs the 100 MHz s_memrealtime) instead of assumed.  Cycles are per wave64 instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate_probe2.hip -o bin/valu_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* clk, int iters, unsigned seed, float sf, unsigned su) {
  unsigned a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 0x55, d = a + 7;
  float fa = (float)a, fb = 1.0001f, fc = 0.5f, fd = 3.0f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {1.f, 1.f, 1.f, 1.f};
  __shared__ unsigned lds[1024];
  lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b; lds[threadIdx.x + 512] = c; lds[threadIdx.x + 768] = d;
  __syncthreads();
  unsigned la = (threadIdx.x * 4) & 4095;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP16(asm volatile("v_mul_f32 %0, %2, %0\n v_mul_f32 %1, %2, %1" : "+v"(fa), "+v"(fc) : "s"(sf));) }
    if (OP == 1) { REP16(asm volatile("v_add_f32 %0, 0x4b400000, %0\n v_add_f32 %1, 0x4b400000, %1" : "+v"(fa), "+v"(fc));) }
    if (OP == 2) { REP16(asm volatile("v_fmamk_f32 %0, %0, 0x3a200000, %2\n v_fmamk_f32 %1, %1, 0x3a200000, %2" : "+v"(fa), "+v"(fc) : "v"(fb));) }
    if (OP == 3) { REP16(asm volatile("v_fma_f32 %0, %0, %2, 0.5\n v_fma_f32 %1, %1, %2, 0.5" : "+v"(fa), "+v"(fc) : "v"(fb));) }
    if (OP == 4) { REP16(asm volatile("v_add_u32 %0, %2, %0\n v_add_u32 %1, %2, %1" : "+v"(a), "+v"(c) : "s"(su));) }
    if (OP == 5) { REP16(asm volatile("v_and_b32 %0, 0x00ff00ff, %0\n v_and_b32 %1, 0x00ff00ff, %1" : "+v"(a), "+v"(c));) }
    if (OP == 6) { REP16(asm volatile("v_add_u32 %0, %0, %2\n v_mad_u32_u24 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 7) { REP16(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_mad_u32_u24 %3, %3, %2, %4" : "+v"(a), "+v"(b), "+v"(b), "+v"(d) : "v"(c));) }
    if (OP == 8) { REP16(asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(a), "+v"(c) : "v"(b) : "vcc");) }
    if (OP == 9) { REP16(acc = __builtin_amdgcn_mfma_f32_4x4x1f32(fb, fa, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(fb, fc, acc2, 0, 0, 0);) }
    if (OP == 10) { REP16(acc = __builtin_amdgcn_mfma_f32_4x4x1f32(fb, fa, acc, 0, 0, 0); asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 11) { REP16(asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(c) : "v"(la));) }
    if (OP == 12) { REP16(asm volatile("ds_read_b32 %0, %2\n v_add_u32 %3, %3, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %3, %3, %4\n ds_read_b32 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(c) : "v"(la), "v"(d), "v"(b));) }
    if (OP == 13) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(fa), "+v"(fc) : "v"(fb));) }
    if (OP == 14) { REP16(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 15) { REP16(asm volatile("v_add_u32 %0, %0, %2\n v_mul_f32 %1, %1, %3" : "+v"(a), "+v"(fc) : "v"(b), "v"(fb));) }
    if (OP == 16) { REP16(asm volatile("v_lshlrev_b32 %0, 2, %0\n v_ashrrev_i32 %1, 14, %1" : "+v"(a), "+v"(c));) }
    if (OP == 17) { REP16(asm volatile("v_mul_f32_e64 %0, %0, %2\n v_add_f32_e64 %1, %1, %2" : "+v"(fa), "+v"(fc) : "v"(fb));) }
    if (OP == 18) { REP16(asm volatile("v_sub_f32 %0, %0, %2\n v_fma_f32 %1, %1, %2, %3" : "+v"(fa), "+v"(fc) : "v"(fb), "v"(fd));) }
    if (OP == 19) { REP16(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2" : "+v"(a), "+v"(c), "+v"(c), "+v"(d), "+v"(la) : "v"(b));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (unsigned)fa + (unsigned)fb + (unsigned)fc + (unsigned)fd + (unsigned)(acc[0] + acc2[1]);
}
template <int OP>
void run(const char* name, int instr_per_rep, int blocks_per_cu = 8) {
  unsigned* o;
  unsigned long long* clk;
  (void)hipMalloc(&o, 256 * 1024 * 8 * 4);
  (void)hipMalloc(&clk, 16);
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  k<OP><<<blocks, 256>>>(o, clk, 10, 1, 1.0001f, 3u);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(o, clk, iters, 1, 1.0001f, 3u);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;  // s_memtime ticks per second
  double winstr = (double)blocks * 4 * iters * 16 * instr_per_rep;  // wave instructions
  double per_simd = winstr / (256 * 4);
  printf("%-34s waves/SIMD %d  %8.3f ms  clock %.2f GHz  cycles per wave-instr per SIMD = %.2f\n", name, blocks_per_cu, ms, ghz,
         ms * 1e-3 * ghz * 1e9 / per_simd);
  (void)hipFree(o);
  (void)hipFree(clk);
}
int main() {
  run<14>("v_add_u32 vgpr", 2); run<14>("v_add_u32 vgpr", 2, 4); run<14>("v_add_u32 vgpr", 2, 2); run<14>("v_add_u32 vgpr", 2, 1);
  run<19>("v_add_u32 x4 independent", 4, 1); run<19>("v_add_u32 x4 independent", 4, 2);
  run<13>("v_mul_f32 vgpr", 2); run<0>("v_mul_f32 sgpr operand", 2); run<1>("v_add_f32 literal", 2); run<2>("v_fmamk_f32 literal", 2);
  run<3>("v_fma_f32 inline const", 2); run<4>("v_add_u32 sgpr operand", 2); run<5>("v_and_b32 literal", 2);
  run<17>("v_mul/add_f32 e64", 2); run<18>("v_sub_f32 + v_fma_f32", 2); run<16>("v_lshlrev + v_ashrrev", 2); run<15>("v_add_u32 + v_mul_f32", 2);
  run<6>("add + mad_u24 (1 fast + 1 slow)", 2); run<7>("2 add + 1 mad_u24", 3);
  run<6>("add + mad_u24 (1 fast + 1 slow)", 2, 4);
  run<8>("v_cndmask_b32 vcc", 2);
  run<9>("mfma_4x4x1 x2 chains", 2); run<9>("mfma_4x4x1 x2 chains", 2, 4);
  run<10>("1 mfma_4x4x1 + 4 v_add_u32 (per 5)", 5); run<10>("1 mfma_4x4x1 + 4 v_add_u32 (per 5)", 5, 4);
  run<11>("2 ds_read_b32 + wait (per 2)", 2); run<12>("2 ds_read_b32 + 4 v_add (per 6)", 6);
  return 0;
}
