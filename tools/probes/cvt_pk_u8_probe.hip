// Probe: is v_cvt_pk_u8_f32 == saturate_cast<uchar>(float) (round half to even, clamp, NaN -> 0)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
  std::vector<float> h;
  for (int i = -64; i <= 8 * 260; i++) h.push_back(i * 0.125f);
  for (int i = 0; i < 256; i++) { h.push_back(std::nextafterf(i + 0.5f, 0.f)); h.push_back(std::nextafterf(i + 0.5f, 1e9f)); }
  h.push_back(NAN); h.push_back(INFINITY); h.push_back(-INFINITY); h.push_back(1e9f); h.push_back(-1e9f); h.push_back(3e9f);
  int n = (int)h.size();
  float* d; unsigned* o;
  (void)hipMalloc(&d, n * 4); (void)hipMalloc(&o, n * 4);
  (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(d, o, n);
  std::vector<unsigned> r(n);
  (void)hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) {
    float v = h[i];
    int ref;
    if (std::isnan(v)) ref = 0; else { float c = fminf(fmaxf(v, 0.f), 255.f); ref = (int)lrintf(c); }
    if ((int)r[i] != ref) { if (bad < 12) printf("v=%.9g got %u ref %d\n", v, r[i], ref); bad++; }
  }
  printf("cvt_pk_u8_f32 mismatches vs RNE+saturate: %d of %d\n", bad, n);
  return 0;
}
