// hipcc (ROCm 7.2, gfx950, -O3): bit_cast<float ext-vector of 4>(__builtin_amdgcn_raw_buffer_load_b128(...)) is narrowed to ONE
// buffer_load_dword whose value is used for all four elements (through uint4 or an unsigned ext-vector the load stays 16 bytes).
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/scratch/b128_bitcast_repro.hip -o - | grep buffer_load
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* dst, int n) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)n, 0x00020000);
  f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, 0, 0));
  const auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, u[0]), __builtin_bit_cast(unsigned, u[2]), false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, u[1]), __builtin_bit_cast(unsigned, u[3]), false, false);
  dst[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, s0[0]);
  dst[threadIdx.x * 4 + 1] = __builtin_bit_cast(float, s1[0]);
  dst[threadIdx.x * 4 + 2] = __builtin_bit_cast(float, s0[1]);
  dst[threadIdx.x * 4 + 3] = __builtin_bit_cast(float, s1[1]);
}
