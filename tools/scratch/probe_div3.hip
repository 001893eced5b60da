// (an experiment of round 6, not in the library: DESIGN.md 5.4)
// (1) v / 3 by the reciprocal with one fma correction (Markstein) against the compiler's correctly rounded division, all 2^32 patterns;
// (2) e5m2 of four f16 hi values: packed clamp + v_cvt_scalef32_pk_bf8_f16 (scale 1) against the f32 path of pack_e5m2x4, all 2^32 pairs.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I sa-toolkit_amd/csrc -I include tools/scratch/probe_div3.hip -o tools/scratch/probe_div3 && tools/scratch/probe_div3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "conv_common.h"
using namespace sat;

__device__ __forceinline__ float div_fast(float v, float d, float r) {
  const float q = v * r;
  const float rem = __builtin_fmaf(-d, q, v);
  return __builtin_copysignf(__builtin_fmaf(rem, r, q), v);
}
template <class H2>
__device__ __forceinline__ unsigned pack_e5m2_hi_x4(H2 h01, H2 h23) {
  const f16x2_t lim = {(_Float16)57344.f, (_Float16)57344.f};
  const f16x2_t a = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(f16x2_t, h01), -lim), lim);
  const f16x2_t b = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(f16x2_t, h23), -lim), lim);
  s16x2_t r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r, a, 1.0f, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r, b, 1.0f, true);
  return (unsigned)__builtin_bit_cast(int, r);
}

// classes: 0 normal in / normal out, 1 zero, 2 denormal in or out, 3 inf / nan, 4 |v| >= 2^126
__global__ void probe_div(unsigned long long* bad, unsigned* first, float d, float r) {
  const unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
  for (unsigned k = 0; k < 256u; ++k) {
    const unsigned vb = (unsigned)(i0 + k);
    const float v = __builtin_bit_cast(float, vb);
    const float ref = v / d;
    const float got = div_fast(v, d, r);
    const unsigned rb = __builtin_bit_cast(unsigned, ref), gb = __builtin_bit_cast(unsigned, got);
    if (rb == gb || (ref != ref && got != got)) continue;
    const unsigned e = (vb >> 23) & 0xffu, er = (rb >> 23) & 0xffu;
    int cls = 0;
    if ((vb << 1) == 0u) cls = 1;
    else if (e == 0xffu) cls = 3;
    else if (e == 0u || er == 0u) cls = 2;
    else if (e >= 253u) cls = 4;
    if (atomicAdd(bad + cls, 1ull) == 0) first[cls] = vb;
  }
}

__global__ void probe_e5m2(unsigned long long* bad, unsigned* first) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const unsigned a = blockIdx.x;
  for (unsigned b = threadIdx.x; b < 65536u; b += blockDim.x) {
    const unsigned w = a | (b << 16);
    const h2 h01 = __builtin_bit_cast(h2, w), h23 = __builtin_bit_cast(h2, (b ^ 0x1234u) | ((a ^ 0x4321u) << 16));
    if (h01[0] != h01[0] || h01[1] != h01[1] || h23[0] != h23[0] || h23[1] != h23[1]) continue;     // (NaN: not compared)
    const unsigned ref = pack_e5m2x4((float)h01[0], (float)h01[1], (float)h23[0], (float)h23[1]);
    const unsigned got = pack_e5m2_hi_x4(h01, h23);
    if (ref != got && atomicAdd(bad, 1ull) == 0) first[0] = w, first[1] = ref, first[2] = got;
  }
}

int main() {
  unsigned long long* bad;
  unsigned* first;
  hipMalloc(&bad, 64);
  hipMalloc(&first, 64);
  const float ds[7] = {3.0f, 2.0f, 4.0f, 5.0f, 6.0f, 7.0f, 8.0f};
  for (int t = 0; t < 7; ++t) {
    hipMemset(bad, 0, 64);
    hipMemset(first, 0, 64);
    const float d = ds[t], r = 1.0f / d;
    hipLaunchKernelGGL(probe_div, dim3(65536), dim3(256), 0, 0, bad, first, d, r);
    unsigned long long nb[5];
    unsigned f[5];
    hipMemcpy(nb, bad, 40, hipMemcpyDeviceToHost);
    hipMemcpy(f, first, 20, hipMemcpyDeviceToHost);
    printf("v / %g over all 2^32 patterns, mismatches by class: normal %llu (%08x)  zero %llu (%08x)  denormal in / out %llu (%08x)  inf / nan %llu (%08x)  |v| >= 2^126 %llu (%08x)\n",
           d, nb[0], f[0], nb[1], f[1], nb[2], f[2], nb[3], f[3], nb[4], f[4]);
  }
  hipMemset(bad, 0, 64);
  hipMemset(first, 0, 64);
  hipLaunchKernelGGL(probe_e5m2, dim3(65536), dim3(256), 0, 0, bad, first);
  unsigned long long nb = 0;
  unsigned f[3] = {0, 0, 0};
  hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(f, first, 12, hipMemcpyDeviceToHost);
  printf("e5m2 of hi halves (packed clamp + cvt_scalef32 vs f32 path): %llu mismatches of 2^32 pairs (first: %08x ref %08x got %08x)\n", nb, f[0], f[1], f[2]);
  return 0;
}
