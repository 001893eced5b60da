// v_fma_mix_f32 against v_cvt_f32_f16 + v_add_f32 / v_sub_f32 (csrc/common.h: mix_add_halves, mix_sub_half), bit for bit:
//   add: every pair of halves (2^32), both lane halves;  sub: every half x 2^12 floats (a stride over all bit patterns + the specials).
// hipcc --offload-arch=gfx950 -O3 -I sa-toolkit_amd/csrc -I include tools/scratch/probe_fma_mix.hip -o tools/scratch/probe_fma_mix && tools/scratch/probe_fma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "common.h"
using namespace sat;

__global__ void probe_add(unsigned long long* bad, unsigned* first) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const unsigned h = blockIdx.x;                       // 65536 blocks: the hi half pattern
  for (unsigned l = threadIdx.x; l < 65536u; l += blockDim.x) {
    const unsigned hw = h | (l << 16), lw = l | (h << 16);       // lower half of hw = h, upper = l; lw the other way round
    const h2 a = __builtin_bit_cast(h2, hw), b = __builtin_bit_cast(h2, lw);
    const float r_lo = (float)a[0] + (float)b[0], r_hi = (float)a[1] + (float)b[1];
    const float m_lo = mix_add_halves<false>(hw, lw), m_hi = mix_add_halves<true>(hw, lw);
    const bool ok_lo = __builtin_bit_cast(unsigned, r_lo) == __builtin_bit_cast(unsigned, m_lo) || (r_lo != r_lo && m_lo != m_lo);
    const bool ok_hi = __builtin_bit_cast(unsigned, r_hi) == __builtin_bit_cast(unsigned, m_hi) || (r_hi != r_hi && m_hi != m_hi);
    if (!ok_lo || !ok_hi) {
      if (atomicAdd(bad, 1ull) == 0) first[0] = hw, first[1] = lw;
    }
  }
}

__global__ void probe_sub(unsigned long long* bad, unsigned* first) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const unsigned h = blockIdx.x;
  for (unsigned i = threadIdx.x; i < 4096u + 16u; i += blockDim.x) {
    unsigned vb = i < 4096u ? i * 1048583u + h * 2654435761u : 0u;                 // scattered patterns
    const unsigned sp[16] = {0u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x00800000u,
                             0x3f800000u, 0xbf800000u, 0x477fe000u, 0xc77fe000u, 0x33800000u, 0x38800000u, 0x387fc000u, 0x7f7fffffu};
    if (i >= 4096u) vb = sp[i - 4096u];
    const float v = __builtin_bit_cast(float, vb);
    const unsigned w = h | ((h ^ 0x5a5au) << 16);
    const h2 a = __builtin_bit_cast(h2, w);
    const float r_lo = v - (float)a[0], r_hi = v - (float)a[1];
    const float m_lo = mix_sub_half<false>(v, w), m_hi = mix_sub_half<true>(v, w);
    const bool ok_lo = __builtin_bit_cast(unsigned, r_lo) == __builtin_bit_cast(unsigned, m_lo) || (r_lo != r_lo && m_lo != m_lo);
    const bool ok_hi = __builtin_bit_cast(unsigned, r_hi) == __builtin_bit_cast(unsigned, m_hi) || (r_hi != r_hi && m_hi != m_hi);
    if (!ok_lo || !ok_hi) {
      if (atomicAdd(bad, 1ull) == 0) first[0] = w, first[1] = vb;
    }
  }
}

int main() {
  unsigned long long* bad;
  unsigned* first;
  hipMalloc(&bad, 16);
  hipMalloc(&first, 16);
  for (int which = 0; which < 2; ++which) {
    hipMemset(bad, 0, 16);
    hipMemset(first, 0, 16);
    if (which == 0) hipLaunchKernelGGL(probe_add, dim3(65536), dim3(256), 0, 0, bad, first);
    else hipLaunchKernelGGL(probe_sub, dim3(65536), dim3(256), 0, 0, bad, first);
    unsigned long long nb = 0;
    unsigned f[2] = {0, 0};
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(f, first, 8, hipMemcpyDeviceToHost);
    printf("%s: %llu mismatches of %s (first: %08x %08x)\n", which ? "v - half (fma_mix vs cvt + sub)" : "half + half (fma_mix vs cvt + add)", nb,
           which ? "2^16 halves x 4112 floats x 2 lane halves" : "2^32 pairs x 2 lane halves", f[0], f[1]);
  }
  return 0;
}
