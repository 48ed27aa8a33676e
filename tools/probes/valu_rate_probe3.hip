// Probe 3: single-class issue costs that probe 1 lumped together (shifts, compares, selects), clock measured in-kernel.
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate_probe3.hip -o bin/valu_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* clk, int iters, unsigned seed, float sf, unsigned su) {
  unsigned a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 0x55, d = a + 7;
  float fa = (float)a, fb = 1.0001f, fc = 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP16(asm volatile("v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 3, %1" : "+v"(a), "+v"(c));) }
    if (OP == 1) { REP16(asm volatile("v_ashrrev_i32 %0, 2, %0\n v_ashrrev_i32 %1, 3, %1" : "+v"(a), "+v"(c));) }
    if (OP == 2) { REP16(asm volatile("v_lshrrev_b32 %0, 2, %0\n v_lshrrev_b32 %1, 3, %1" : "+v"(a), "+v"(c));) }
    if (OP == 3) { REP16(asm volatile("v_lshlrev_b32 %0, %2, %0\n v_lshlrev_b32 %1, %2, %1" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 4) { REP16(asm volatile("v_cndmask_b32 %0, %0, %2, %3\n v_cndmask_b32 %1, %1, %2, %3" : "+v"(a), "+v"(c) : "v"(b), "s"(0x5555555555555555ull));) }
    if (OP == 5) { REP16(asm volatile("v_cmp_gt_u32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_gt_u32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(a), "+v"(c) : "v"(b) : "vcc");) }
    if (OP == 6) { REP16(asm volatile("v_min_i32 %0, %0, %2\n v_max_i32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 7) { REP16(asm volatile("v_sub_u32 %0, %0, %2\n v_subrev_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 8) { REP16(asm volatile("v_or_b32 %0, %0, %2\n v_xor_b32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 9) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3" : "+v"(a), "+v"(c) : "s"(su), "v"(b));) }
    if (OP == 10) { REP16(asm volatile("v_dot2c_i32_i16 %0, 0x12345678, %2\n v_dot2c_i32_i16 %1, 0x12345678, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 11) { REP16(asm volatile("v_mul_f32 %0, 0x3f800123, %0\n v_mul_f32 %1, 0x3f800123, %1" : "+v"(fa), "+v"(fc));) }
    if (OP == 12) { REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, %0, %2\n v_readlane_b32 s21, %1, 5\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b) : "s20", "s21");) }
    if (OP == 13) { REP16(asm volatile("v_cvt_pk_u8_f32 %0, %2, 0, 0\n v_cvt_pk_u8_f32 %1, %3, 0, 0" : "+v"(a), "+v"(b) : "v"(fa), "v"(fb));) }
    if (OP == 14) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2" : "+v"(a), "+v"(c) : "s"(su));) }
    if (OP == 15) { REP16(asm volatile("v_bfi_b32 %0, %2, %0, %3\n v_bfi_b32 %1, %2, %1, %3" : "+v"(a), "+v"(c) : "s"(su), "v"(b));) }
    if (OP == 16) { REP16(asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %3" : "+v"(a), "+v"(c) : "v"(b), "v"(d));) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (unsigned)fa + (unsigned)fb + (unsigned)fc;
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
  const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
  double winstr = (double)blocks * 4 * iters * 16 * instr_per_rep;
  double per_simd = winstr / (256 * 4);
  printf("%-34s waves/SIMD %d  %8.3f ms  clock %.2f GHz  cycles per wave-instr per SIMD = %.2f\n", name, blocks_per_cu, ms, ghz,
         ms * 1e-3 * ghz * 1e9 / per_simd);
  (void)hipFree(o);
  (void)hipFree(clk);
}
int main() {
  run<0>("v_lshlrev_b32 imm", 2); run<1>("v_ashrrev_i32 imm", 2); run<2>("v_lshrrev_b32 imm", 2); run<3>("v_lshlrev_b32 vgpr amount", 2);
  run<4>("v_cndmask_b32 sgpr-pair mask", 2); run<5>("v_cmp + v_cndmask vcc (per 2)", 4); run<6>("v_min/max_i32", 2);
  run<7>("v_sub/subrev_u32", 2); run<8>("v_or/xor_b32", 2); run<9>("v_mad_u32_u24 sgpr operand", 2); run<10>("v_dot2c_i32_i16 literal", 2);
  run<11>("v_mul_f32 literal", 2); run<12>("v_readlane + v_add (per 2)", 4); run<13>("v_cvt_pk_u8_f32 (0,0)", 2); run<14>("v_mul_u32_u24 sgpr", 2);
  run<15>("v_bfi_b32 sgpr mask", 2); run<16>("v_mov_b32", 2);
  return 0;
}
