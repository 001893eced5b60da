// Issue interval of single MFMA shapes on gfx950, independent accumulators (16 of 16x16 / 4 of 32x32), operands in registers.
//   hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_issue.hip -o tools/scratch/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const uint4* src, float* out, long long* stamps, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  h8 ah[2], bh[2];
  i32x8 a8[2], b8[2];
  for (int i = 0; i < 2; ++i) {
    ah[i] = __builtin_bit_cast(h8, src[(tid * 16 + i) & 0xffff]);
    bh[i] = __builtin_bit_cast(h8, src[(tid * 16 + 4 + i) & 0xffff]);
    uint4 u0 = src[(tid * 16 + 8 + 2 * i) & 0xffff], u1 = src[(tid * 16 + 9 + 2 * i) & 0xffff];
    a8[i] = i32x8{(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w, (int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
    uint4 v0 = src[(tid * 16 + 12 + 2 * i) & 0xffff], v1 = src[(tid * 16 + 13 + 2 * i) & 0xffff];
    b8[i] = i32x8{(int)v0.x, (int)v0.y, (int)v0.z, (int)v0.w, (int)v1.x, (int)v1.y, (int)v1.z, (int)v1.w};
  }
  f32x16 acc[2][2];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  f32x4 acc4[4][4];
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) acc4[m][n][r] = 0.f;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if constexpr (MODE == 0) acc4[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m & 1], bh[n & 1], acc4[m][n], 0, 0, 0);
        if constexpr (MODE == 1) acc4[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[m & 1], b8[n & 1], acc4[m][n], 0, 0, 0, 121, 0, 117);
        if constexpr (MODE == 2) acc4[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[m & 1], b8[n & 1], acc4[m][n], 2, 2, 0, 121, 0, 117);
        if constexpr (MODE == 3) acc4[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[m & 1], b8[n & 1], acc4[m][n], 4, 4, 0, 121, 0, 117);
        if constexpr (MODE == 4) if (m < 2 && n < 2) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[m], b8[n], acc[m][n], 0, 0, 0, 121, 0, 117);
        if constexpr (MODE == 5) if (m < 2 && n < 2) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
        if constexpr (MODE == 7) acc4[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(*(long*)&a8[m & 1], *(long*)&b8[n & 1], acc4[m][n], 0, 0, 0);
      }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
  float s = 0.f;
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc4[m][n][r];
  out[tid] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[2 * (tid >> 6)] = (long long)(t1 - t0);
    stamps[2 * (tid >> 6) + 1] = (long long)(r1 - r0);
  }
}

template <int MODE>
void run(const char* name, int per_iter, const uint4* src, float* out, long long* stamps, int wps) {
  const int threads = 256, blocks = 256 * wps, iters = 20000;
  for (int i = 0; i < 20; ++i) k<MODE><<<blocks, threads>>>(src, out, stamps, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(2 * blocks * 4);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> mhz, cyc;
  for (int w = 0; w < blocks * 4; ++w) { mhz.push_back(h[2 * w] / (double)h[2 * w + 1] * 100.0); cyc.push_back((double)h[2 * w] / iters / per_iter); }
  std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
  printf("%-40s %d wave/SIMD: %6.1f cycles per instruction per wave (x%d waves = %6.1f per SIMD)  clock %5.0f MHz\n", name, wps, cyc[cyc.size() / 2], wps, cyc[cyc.size() / 2] / wps * 1.0, mhz[mhz.size() / 2]);
}

int main() {
  uint4* src; float* out; long long* stamps;
  std::vector<unsigned> h(65536 * 4);
  unsigned s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    unsigned r = s ^ (s >> 13);
    v = (r & 0x83ff83ffu) | 0x38003800u;
    v &= 0xbfffbfffu;
    v &= ~0x40404040u | 0x3f3f3f3fu;
  }
  hipMalloc(&src, h.size() * 4); hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, 256 * 2 * 256 * 4); hipMalloc(&stamps, 256 * 2 * 4 * 16);
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_mfma_f32_16x16x32_f16", 16, src, out, stamps, w);
    run<1>("v_mfma_scale_f32_16x16x128 fp8 e4m3", 16, src, out, stamps, w);
    run<2>("v_mfma_scale_f32_16x16x128 fp6 e2m3", 16, src, out, stamps, w);
    run<3>("v_mfma_scale_f32_16x16x128 fp4", 16, src, out, stamps, w);
    run<7>("v_mfma_f32_16x16x32_fp8_fp8", 16, src, out, stamps, w);
    run<4>("v_mfma_scale_f32_32x32x64 fp8 e4m3", 4, src, out, stamps, w);
    run<5>("v_mfma_f32_32x32x16_f16", 4, src, out, stamps, w);
  }
  return 0;
}
