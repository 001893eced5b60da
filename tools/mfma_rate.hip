// Bare MFMA-rate experiment (diagnostic, not part of the library): what does one split product cost
// on the matrix cores, as 3 f16 MFMAs (hi*hi + hi*lo + lo*hi) versus 1 f16 MFMA + block-scaled
// fp8 / fp6 MFMAs (K = 64) for the two cross terms?  Operands random, in registers; the chip's own
// clock management is part of the answer (MI355X_MICROARCH.md, DVFS give-back), so the in-kernel
// clock is reported next to the wall time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rate.hip -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// one "tap" = the products of a 64x64 wave tile (2x2 MFMA tiles) with one 16-channel K slice
template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(const uint4* src, float* out, long long* stamps, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  // operand registers: 2 A tiles + 2 B tiles, hi and lo (f16) and 32-byte fp8/fp6 forms
  h8 ah[2], al[2], bh[2], bl[2];
  i32x8 a8[2], b8[2];
  for (int i = 0; i < 2; ++i) {
    ah[i] = __builtin_bit_cast(h8, src[(tid * 16 + i) & 0xffff]);
    al[i] = __builtin_bit_cast(h8, src[(tid * 16 + 2 + i) & 0xffff]);
    bh[i] = __builtin_bit_cast(h8, src[(tid * 16 + 4 + i) & 0xffff]);
    bl[i] = __builtin_bit_cast(h8, src[(tid * 16 + 6 + i) & 0xffff]);
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
  // `iters` pairs of taps
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int tap = 0; tap < 2; ++tap)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          }
    } else if constexpr (MODE == 1 || MODE == 2) {
      constexpr int FMT = MODE == 1 ? 0 : 2;   // 0 = fp8 e4m3, 2 = fp6 e2m3
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bl[n], acc[m][n], 0, 0, 0);   // second tap's hi*hi (other registers)
          acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[m], b8[n], acc[m][n], FMT, FMT, 0, 121, 0, 117);
        }
    } else if constexpr (MODE == 3) {
      // 3 f16 products on the 16x16x32 shape: the same 64x64 wave tile = 4x4 tiles, K = 32 per
      // instruction = one pair of taps
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          acc4[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m & 1], bh[n & 1], acc4[m][n], 0, 0, 0);
          acc4[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m & 1], bl[n & 1], acc4[m][n], 0, 0, 0);
          acc4[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m & 1], bh[n & 1], acc4[m][n], 0, 0, 0);
        }
    } else {
      // 16x16 shape with fp8 cross terms: 16x16x32 f16 (K = 32: a tap pair) + 16x16x128 fp8 (two tap pairs x 2 terms)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          acc4[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m & 1], bh[n & 1], acc4[m][n], 0, 0, 0);
          if ((it & 1) == 0)
            acc4[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[m & 1], b8[n & 1], acc4[m][n], 0, 0, 0, 121, 0, 117);
        }
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
int run(const char* name, const uint4* src, float* out, long long* stamps, int waves_per_simd) {
  const int threads = 256, blocks = 256 * waves_per_simd, iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // settle the clock: ~1.5 s of back-to-back launches, then time 5
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    const int n = rep ? 5 : 40;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) rate_kernel<MODE><<<blocks, threads>>>(src, out, stamps, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= n;
  }
  std::vector<long long> h(2 * blocks * 4);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> mhz, cyc;
  for (int w = 0; w < blocks * 4; ++w) { mhz.push_back(h[2 * w] / (double)h[2 * w + 1] * 100.0); cyc.push_back((double)h[2 * w] / iters); }
  std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
  // useful products per tap pair and wave: 2 taps x 64 x 64 x 16
  const double prod = 2.0 * 64 * 64 * 16 * iters * blocks * 4.0;
  printf("%-44s %d wave/SIMD: %8.3f ms  %7.1f useful TFLOP/s (2*products/s)  %6.0f cycles per tap pair  clock %5.0f MHz\n", name,
         waves_per_simd, ms, 2.0 * prod / (ms * 1e-3) / 1e12, cyc[cyc.size() / 2], mhz[mhz.size() / 2]);
  return 0;
}

int main() {
  uint4* src; float* out; long long* stamps;
  std::vector<unsigned> h(65536 * 4);
  unsigned s = 12345;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    unsigned r = s ^ (s >> 13);
    v = (r & 0x83ff83ffu) | 0x38003800u;   // pairs of f16 in +-[0.5, 1); as fp8 bytes: random finite values
    v &= 0xbfffbfffu;
    v &= ~0x40404040u | 0x3f3f3f3fu;
  }
  CK(hipMalloc(&src, h.size() * 4)); CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 256 * 2 * 256 * 4)); CK(hipMalloc(&stamps, 256 * 2 * 4 * 16));
  for (int w = 1; w <= 2; ++w) {
    if (run<0>("3 x f16 32x32x16", src, out, stamps, w)) return 1;
    if (run<1>("1 x f16 32x32x16 + fp8 32x32x64 cross", src, out, stamps, w)) return 1;
    if (run<2>("1 x f16 32x32x16 + fp6 32x32x64 cross", src, out, stamps, w)) return 1;
    if (run<3>("3 x f16 16x16x32", src, out, stamps, w)) return 1;
    if (run<4>("1 x f16 16x16x32 + fp8 16x16x128 cross", src, out, stamps, w)) return 1;
  }
  return 0;
}
