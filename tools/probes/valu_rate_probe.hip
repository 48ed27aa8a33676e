// Probe: issue rate of VALU instruction classes on gfx950 (cycles per wave64 instruction per SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate_probe.hip -o bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
  unsigned a = threadIdx.x + seed, b = a * 3 + 1, c = b ^ 0x55, d = a + 7;
  float fa = (float)a, fb = 1.0001f, fc = 0.5f, fd = 3.0f;
  double da = 1.0 + a, db = 2.5;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 1) { REP16(asm volatile("v_mad_i32_i24 %0, %0, %1, %2\n v_mad_i32_i24 %3, %3, %1, %2" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 2) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 3) { REP16(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(fa), "+v"(fd) : "v"(fb), "v"(fc));) }
    if (OP == 4) { REP16(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1" : "+v"(a), "+v"(b));) }
    if (OP == 5) { REP16(asm volatile("v_and_b32 %0, %0, %2\n v_and_b32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 6) { REP16(asm volatile("v_cvt_f32_u32 %0, %2\n v_cvt_f32_u32 %1, %3" : "+v"(fa), "+v"(fc) : "v"(a), "v"(b));) }
    if (OP == 7) { REP16(asm volatile("v_cvt_pk_u8_f32 %0, %2, 0, %0\n v_cvt_pk_u8_f32 %1, %3, 1, %1" : "+v"(a), "+v"(b) : "v"(fa), "v"(fb));) }
    if (OP == 8) { REP16(asm volatile("v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 9) { REP16(asm volatile("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8" : "+v"(a), "+v"(b));) }
    if (OP == 10) { REP16(asm volatile("v_add3_u32 %0, %0, %2, %3\n v_add3_u32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 11) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(fa), "+v"(fc) : "v"(fb));) }
    if (OP == 12) { REP16(asm volatile("v_med3_i32 %0, %0, %2, %3\n v_med3_i32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 13) { REP16(asm volatile("v_pk_add_u16 %0, %0, %2\n v_pk_add_u16 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 14) { REP16(asm volatile("v_mul_hi_u32_u24 %0, %0, %2\n v_mul_hi_u32_u24 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 15) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(da) : "v"(db));) }
    if (OP == 16) { REP16(asm volatile("v_cvt_f32_ubyte1 %0, %2\n v_cvt_f32_ubyte2 %1, %3" : "+v"(fa), "+v"(fc) : "v"(a), "v"(b));) }
    if (OP == 17) { REP16(asm volatile("v_add_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 18) { REP16(asm volatile("v_sad_u8 %0, %2, %3, %0\n v_sad_u8 %1, %2, %3, %1" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 19) { REP16(asm volatile("v_dot4_u32_u8 %0, %2, %3, %0\n v_dot4_u32_u8 %1, %2, %3, %1" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 20) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(da) : "v"(db));) }
    if (OP == 21) { REP16(asm volatile("v_alignbyte_b32 %0, %0, %2, 1\n v_alignbyte_b32 %1, %1, %2, 3" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 22) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %2\n v_pk_mad_u16 %1, %1, %2, %0" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 23) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 8, %2\n v_lshl_or_b32 %1, %1, 8, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 24) { REP16(asm volatile("v_cvt_u32_f32 %0, %2\n v_cvt_u32_f32 %1, %3" : "+v"(a), "+v"(b) : "v"(fa), "v"(fb));) }
    if (OP == 25) { REP16(asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(a), "+v"(c) : "v"(b) : "vcc");) }
    if (OP == 26) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 27) { REP16(asm volatile("v_min3_u32 %0, %0, %2, %3\n v_max3_u32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 28) { REP16(asm volatile("v_max_u32 %0, %0, %2\n v_min_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 29) { REP16(asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 30) { REP16(asm volatile("v_cmp_gt_u32 vcc, %0, %2\n v_cmp_gt_u32 vcc, %1, %2" : "+v"(a), "+v"(c) : "v"(b) : "vcc");) }
    if (OP == 31) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 32) { REP16(asm volatile("v_max_f32 %0, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(fa), "+v"(fc) : "v"(fb));) }
    if (OP == 33) { REP16(asm volatile("v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %1, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 34) { REP16(asm volatile("v_mul_u32_u24_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_u32_u24_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 35) { REP16(asm volatile("v_pk_max_u16 %0, %0, %2\n v_pk_min_u16 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 36) { REP16(asm volatile("v_bfi_b32 %0, %2, %0, %3\n v_bfi_b32 %1, %2, %1, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 37) { REP16(asm volatile("v_and_or_b32 %0, %0, %2, %3\n v_and_or_b32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 38) { REP16(asm volatile("v_lerp_u8 %0, %0, %2, %3\n v_lerp_u8 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 39) { REP16(asm volatile("v_cvt_f32_i32 %0, %2\n v_cvt_f32_i32 %1, %3" : "+v"(fa), "+v"(fc) : "v"(a), "v"(b));) }
    if (OP == 40) { REP16(asm volatile("v_sub_f32 %0, %0, %2\n v_sub_f32 %1, %1, %2" : "+v"(fa), "+v"(fc) : "v"(fb));) }
    if (OP == 41) { REP16(asm volatile("v_xor_b32 %0, %0, %2\n v_or_b32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 42) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(da) : "v"(db));) }
    if (OP == 43) { REP16(asm volatile("v_sub_u32 %0, %0, %2\n v_subrev_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 44) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 2, %2\n v_lshl_add_u32 %1, %1, 2, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 45) { REP16(asm volatile("v_pk_lshrrev_b16 %0, 1, %0\n v_pk_lshrrev_b16 %1, 2, %1" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 46) { REP16(asm volatile("v_msad_u8 %0, %2, %3, %0\n v_msad_u8 %1, %2, %3, %1" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 47) { REP16(asm volatile("v_dot2_i32_i16 %0, %2, %3, %0\n v_dot2_i32_i16 %1, %2, %3, %1" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 48) { REP16(asm volatile("v_mad_u16 %0, %0, %2, %3\n v_mad_u16 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }
    if (OP == 49) { REP16(asm volatile("v_add_f32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(fa), "+v"(fc) : "v"(fb));) }
    if (OP == 50) { REP16(asm volatile("v_add_u32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(c) : "v"(b));) }
    if (OP == 51) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %2\n v_cvt_f32_ubyte3 %1, %3" : "+v"(fa), "+v"(fc) : "v"(a), "v"(b));) }
    if (OP == 52) { REP16(asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3" : "+v"(fa), "+v"(fd) : "v"(fb), "v"(fc));) }
    if (OP == 53) { REP16(asm volatile("v_mul_i32_i24 %0, %0, %2\n v_mul_i32_i24 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (unsigned)fa + (unsigned)fb + (unsigned)fc + (unsigned)fd + (unsigned)da;
}
template <int OP>
void run(const char* name, int instr_per_rep) {
  unsigned* o;
  (void)hipMalloc(&o, 256 * 1024 * 8 * 4);
  const int iters = 2000, blocks = 256 * 8;
  k<OP><<<blocks, 256>>>(o, 10, 1);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(o, iters, 1);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)blocks * 4 * iters * 16 * instr_per_rep;  // wave instructions
  double per_simd = winstr / (256 * 4);
  printf("%-22s %8.3f ms  %7.2f T lane-ops/s  cycles per wave-instr per SIMD @2.4GHz = %.2f\n", name, ms, winstr * 64 / ms / 1e9,
         ms * 1e-3 * 2.4e9 / per_simd);
  (void)hipFree(o);
}
int main() {
  run<0>("v_add_u32", 2); run<1>("v_mad_i32_i24", 2); run<26>("v_mad_u32_u24", 2); run<2>("v_mul_lo_u32", 2); run<3>("v_fma_f32", 2);
  run<4>("v_lshrrev_b32", 2); run<5>("v_and_b32", 2); run<6>("v_cvt_f32_u32", 2); run<24>("v_cvt_u32_f32", 2); run<7>("v_cvt_pk_u8_f32", 2);
  run<8>("v_perm_b32", 2); run<9>("v_bfe_u32", 2); run<10>("v_add3_u32", 2); run<11>("v_mul/add_f32", 2);
  run<12>("v_med3_i32", 2); run<27>("v_min3/max3_u32", 2); run<13>("v_pk_add_u16", 2); run<14>("v_mul_hi_u32_u24", 2); run<15>("v_fma_f64", 1);
  run<16>("v_cvt_f32_ubyteN", 2); run<17>("v_add_u32_sdwa", 2); run<18>("v_sad_u8", 2); run<19>("v_dot4_u32_u8", 2);
  run<20>("v_pk_fma_f32", 1); run<21>("v_alignbyte_b32", 2); run<22>("v_pk_mul/mad_u16", 2); run<23>("v_lshl_or_b32", 2);
  run<25>("v_cndmask_b32", 2);
  run<28>("v_max/min_u32", 2);
  run<29>("v_mov_b32", 2);
  run<30>("v_cmp_gt_u32", 2);
  run<31>("v_mul_u32_u24", 2);
  run<32>("v_max/min_f32", 2);
  run<33>("v_lshlrev_b32_sdwa", 2);
  run<34>("v_mul_u32_u24_sdwa", 2);
  run<35>("v_pk_max/min_u16", 2);
  run<36>("v_bfi_b32", 2);
  run<37>("v_and_or_b32", 2);
  run<38>("v_lerp_u8", 2);
  run<39>("v_cvt_f32_i32", 2);
  run<40>("v_sub_f32", 2);
  run<41>("v_xor/or_b32", 2);
  run<42>("v_pk_mul_f32", 1);
  run<43>("v_sub_u32", 2);
  run<44>("v_lshl_add_u32", 2);
  run<45>("v_pk_lshrrev_b16", 2);
  run<46>("v_msad_u8", 2);
  run<47>("v_dot2_i32_i16", 2);
  run<48>("v_mad_u16", 2);
  run<49>("v_add_f32_sdwa", 2);
  run<50>("v_add_u32_dpp", 2);
  run<51>("v_cvt_f32_ubyte0_sdwa?", 2);
  run<52>("v_fmac_f32", 2);
  run<53>("v_mul_i32_i24", 2);
  return 0;
}
