// v_cvt_scalef32_pk_bf8_f16 against the two-step conversion the epilogues use (f16 -> f32, clamp to +-57344, v_cvt_pk_bf8_f32), over all
// 65536 f16 bit patterns: which scale operand multiplies by 2^10, and does the instruction saturate like the clamp?
//   hipcc --offload-arch=gfx950 -O2 tools/scratch/probe_cvt_bf8.hip -o tools/scratch/probe_cvt_bf8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ unsigned ref2(float a, float b) {
  a = __builtin_amdgcn_fmed3f(a, -57344.f, 57344.f);
  b = __builtin_amdgcn_fmed3f(b, -57344.f, 57344.f);
  return (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffffu;
}
__global__ void k(unsigned* out, float scale, float mul) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;      // f16 bit pattern pair (i, i ^ 0x8000)
  const unsigned short b0 = (unsigned short)i, b1 = (unsigned short)(i ^ 0x8000u);
  h2 v = {__builtin_bit_cast(_Float16, b0), __builtin_bit_cast(_Float16, b1)};
  s2 old = {0, 0};
  s2 r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(old, v, scale, false);
  out[2 * i] = (unsigned)__builtin_bit_cast(int, r) & 0xffffu;
  out[2 * i + 1] = ref2((float)v[0] * mul, (float)v[1] * mul);
}
int main() {
  unsigned* d;
  hipMalloc(&d, 65536 * 2 * 4);
  std::vector<unsigned> h(65536 * 2);
  const float cases[][2] = {{1.f, 1.f}, {1024.f, 1024.f}, {1.f / 1024.f, 1024.f}, {1024.f, 1.f / 1024.f}, {1.f / 1024.f, 1.f / 1024.f}};
  for (auto& c : cases) {
    k<<<256, 256>>>(d, c[0], c[1]);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0, nan_in = 0;
    for (unsigned i = 0; i < 65536; ++i) {
      const bool isnan = ((i & 0x7c00) == 0x7c00) && (i & 0x3ff);
      if (isnan) { ++nan_in; continue; }
      if (h[2 * i] != h[2 * i + 1]) {
        if (bad < 6) printf("   f16 0x%04x: cvt_scalef32_pk_bf8_f16 -> 0x%04x, f32 path (x * %g) -> 0x%04x\n", i, h[2 * i], c[1], h[2 * i + 1]);
        ++bad;
      }
    }
    printf("scale operand %g vs two-step conversion of x * %g: %d of %d non-NaN patterns differ\n", c[0], c[1], bad, 65536 - nan_in);
  }
  return 0;
}
