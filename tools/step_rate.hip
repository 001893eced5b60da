// Diagnostic (not part of the library): what does one "step" of the ring kernels cost on a SIMD when nothing but its matrix
// instructions is left?  512-thread blocks, one per CU (two waves per SIMD), every wave: 60 v_mfma_f32_16x16x32_f16 as 20
// dependent triples per iteration (the step of conv1d_f16x3_ring16_kernel), optionally an s_barrier per iteration and 18
// ds_read_b128 per iteration.  Prints cycles per iteration (ideal: 2 x 60 x 16 = 1920) and the clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/step_rate.hip -o tools/step_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int BARRIER, int READS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(const uint4* src, float* out, long long* stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 64 * WAVES) lds[i] = src[(blockIdx.x * 8192 + i) & 0xffff];
  __syncthreads();
  h8 fa[4][2], fb[2][2];
  for (int m = 0; m < 4; ++m) for (int j = 0; j < 2; ++j) fa[m][j] = __builtin_bit_cast(h8, lds[(tid + 64 * (2 * m + j)) & 8191]);
  for (int n = 0; n < 2; ++n) for (int j = 0; j < 2; ++j) fb[n][j] = __builtin_bit_cast(h8, lds[(tid + 640 + 64 * (2 * n + j)) & 8191]);
  f32x4 acc[4][5];
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 5; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
  const uint4* base = lds + (tid & 63) + 1024 * (tid >> 6);
  for (int it = 0; it < iters; ++it) {
    if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int n = 0; n < 5; ++n) {
      __builtin_amdgcn_sched_barrier(0);
      if (READS) {
        // the step's 18 fragment reads: 2 (+ 2 from column 1 on) per column
        fb[(n + 1) & 1][0] = __builtin_bit_cast(h8, base[(it * 7 + n * 64) & 511]);
        fb[(n + 1) & 1][1] = __builtin_bit_cast(h8, base[(it * 7 + n * 64 + 256) & 511]);
        if (n >= 1) {
          fa[n - 1][0] = __builtin_bit_cast(h8, base[(it * 5 + n * 64 + 128) & 511]);
          fa[n - 1][1] = __builtin_bit_cast(h8, base[(it * 5 + n * 64 + 384) & 511]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m][n]) : "v"(fa[m][1]), "v"(fb[n & 1][0]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m][n]) : "v"(fa[m][0]), "v"(fb[n & 1][1]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m][n]) : "v"(fa[m][0]), "v"(fb[n & 1][0]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 5; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if ((tid & 63) == 0) {
    stamps[(blockIdx.x * WAVES + (tid >> 6)) * 2] = (long long)(t1 - t0);
    stamps[(blockIdx.x * WAVES + (tid >> 6)) * 2 + 1] = (long long)(r1 - r0);
  }
}

template <int BARRIER, int READS, int WAVES>
int run(const char* name, const uint4* src, float* out, long long* stamps, int iters) {
  auto kern = k<BARRIER, READS, WAVES>;
  const size_t lds = 8192 * 16 + (WAVES == 8 ? 24 * 1024 : 0);    // 128 / 152 KB: one block per CU
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), lds, 0, src, out, stamps, iters);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(256 * WAVES * 2);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (int i = 0; i < 256 * WAVES; ++i) { cyc.push_back((double)h[2 * i] / iters); clk.push_back((double)h[2 * i] / h[2 * i + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  printf("%-44s waves/block %d: %7.0f cycles per iteration (median; ideal %d), clock %.2f GHz -> %.0f TFLOP/s useful\n", name, WAVES, cyc[cyc.size() / 2],
         WAVES == 8 ? 1920 : 960, clk[clk.size() / 2], 256.0 * WAVES * 60 * 16 * 16 * 32 * 2 / 3 / (cyc[cyc.size() / 2] / (clk[clk.size() / 2] * 1e9)) / 1e12);
  return 0;
}

int main() {
  uint4* src; float* out; long long* stamps;
  CK(hipMalloc(&src, 65536 * 16)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&stamps, 256 * 8 * 16));
  std::vector<unsigned short> h(65536 * 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));   // random f16 in +-[0.125, 0.25)
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const int iters = 2000;
  run<0, 0, 8>("mfma only", src, out, stamps, iters);
  run<1, 0, 8>("mfma + barrier per iteration", src, out, stamps, iters);
  run<0, 1, 8>("mfma + 18 ds_read_b128", src, out, stamps, iters);
  run<1, 1, 8>("mfma + reads + barrier", src, out, stamps, iters);
  run<0, 0, 4>("mfma only", src, out, stamps, iters);
  run<1, 1, 4>("mfma + reads + barrier", src, out, stamps, iters);
  return 0;
}
