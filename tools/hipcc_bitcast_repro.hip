// hipcc (ROCm 7.2.0) miscompile met in csrc/pair32s.hip: a 2 x u32 ext-vector loaded from LDS whose ELEMENTS are bit-cast to
// half2 and widened — the code generator reads 4 bytes instead of 8 and uses element 0 for both (rv[2..3] == rv[0..1]), at -O1
// to -O3.  `hipcc --offload-arch=gfx950 --cuda-device-only -O3 -S tools/hipcc_bitcast_repro.hip -o -` shows one
// `ds_read2st64_b32` and four v_cvt of the same two registers.  Reading the 8 bytes as a 4 x half vector is compiled correctly.
#include <hip/hip_runtime.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, int g) {
  extern __shared__ uint4 lds[];
  asm volatile("s_nop 0" ::: "memory");
  const u32x2* src = (const u32x2*)(lds + threadIdx.x) + (g & 1);
  const u32x2 rh = src[0], rl = src[1024];
  const h2 a = __builtin_bit_cast(h2, rh[0]), b = __builtin_bit_cast(h2, rh[1]);
  const h2 c = __builtin_bit_cast(h2, rl[0]), d = __builtin_bit_cast(h2, rl[1]);
  float rv[4] = {(float)a[0] + (float)c[0], (float)a[1] + (float)c[1], (float)b[0] + (float)d[0], (float)b[1] + (float)d[1]};
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = rv[i];
}
