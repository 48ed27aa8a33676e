// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the access widths the pipeline uses: streaming reads of a
// 2 GiB buffer (far beyond L2 + Infinity Cache) with 4-byte and 16-byte loads per lane, global and buffer forms.
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib_probe.hip -o bin/fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ __launch_bounds__(256) void read_b32_global(const uint32_t* p, size_t n, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[i];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void read_b128_global(const uint4* p, size_t n, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// buffer_load_dword with a per-row scalar offset, as the statistics kernel reads Bayer rows
__global__ __launch_bounds__(256) void read_b32_buffer_rows(const uint8_t* p, unsigned row_bytes, unsigned rows, uint32_t* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)0x7fffffff, 0x00020000);
  uint32_t acc = 0;
  for (unsigned y = blockIdx.x; y < rows; y += gridDim.x)
    for (unsigned x = threadIdx.x * 4; x < row_bytes; x += 1024) acc ^= __builtin_amdgcn_raw_buffer_load_b32(r, (int)x, (int)(y * row_bytes), 0);
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const size_t bytes = 2ull << 30;
  uint8_t* d; uint32_t* o;
  (void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 4);
  (void)hipMemset(d, 1, bytes);
  for (int rep = 0; rep < 3; rep++) {
    read_b32_global<<<4096, 256>>>((const uint32_t*)d, bytes / 4, o);
    read_b128_global<<<4096, 256>>>((const uint4*)d, bytes / 16, o);
    read_b32_buffer_rows<<<4096, 256>>>(d, 2448, (unsigned)(((1ull << 31) - 4096) / 2448), o);
  }
  (void)hipDeviceSynchronize();
  printf("bytes per launch: global %zu, buffer rows %zu\n", bytes, (size_t)2448 * (size_t)(((1ull << 31) - 4096) / 2448));
  return 0;
}
