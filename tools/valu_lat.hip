// Dependent-issue latency of the VALU forms the YAAPT prefilter recursion uses (one wave, s_memtime stamps).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_lat.hip -o tools/valu_lat && tools/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int N = 4096;
template <int MODE>
__global__ void k(const float* f, float* out, long long* cyc, float c1, float c2) {
  float p1 = f[0], p2n = f[1], d = f[2];
  float acc = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 64
  for (int i = 0; i < N; ++i) {
    const float fn = f[3 + (i & 63)];
    if (MODE == 0) {            // sub, sub, mul, mul, med3 (plain)
      const float v = d - p1; d = fn - p2n; p1 = c1 * v; p2n = c2 * v; acc += fminf(fmaxf(v, -1.f), 1.f);
    } else if (MODE == 1) {     // pk_mul
      const float v = d - p1; d = fn - p2n; const v2f q = (v2f){c1, c2} * (v2f){v, v}; p1 = q.x; p2n = q.y; acc += fminf(fmaxf(v, -1.f), 1.f);
    } else if (MODE == 2) {     // dependent v_sub chain
      d = d - fn;
    } else if (MODE == 3) {     // dependent pk_mul chain
      const v2f q = (v2f){c1, c2} * (v2f){p1, p1}; p1 = q.x + 0.f * q.y;
    } else if (MODE == 4) {     // dependent mul -> sub pair
      p1 = c1 * p1; p1 = fn - p1;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = p1 + p2n + d + acc;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float *f, *out; long long* cyc;
  hipMalloc(&f, 1024); hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
  float h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.001f * i;
  hipMemcpy(f, h, 512, hipMemcpyHostToDevice);
  const char* names[] = {"sub,sub,mul,mul,med3 (plain)", "sub,sub,pk_mul,med3", "dependent v_sub", "dependent pk_mul(+fma)", "dependent mul->sub pair"};
  for (int rep = 0; rep < 2; ++rep) {
    long long c;
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, f, out, cyc, 0.5f, 0.25f); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-32s %.2f cycles per step\n", names[M], (double)c / N);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
  }
  return 0;
}
