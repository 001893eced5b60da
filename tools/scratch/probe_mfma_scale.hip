// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands): which lane / byte holds which k, and which lane / byte of the scale
// operands scales it.  hipcc --offload-arch=gfx950 -O2 tools/scratch/probe_mfma_scale.hip -o tools/scratch/probe_mfma_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ i32x8 onehot(bool mine, int byte) {      // e4m3 1.0 (0x38) at `byte` of the lane's 32, no dynamic vector indexing
  i32x8 v;
#pragma unroll
  for (int w = 0; w < 8; ++w) v[w] = (mine && (byte >> 2) == w) ? (int)(0x38u << (8 * (byte & 3))) : 0;
  return v;
}
// block = one wave; each block probes one (ga, pa): A = 1.0 at lane (row 0, group ga) byte pa; B = 1.0 at lane (col 0, group gb) byte pb for
// every (gb, pb) in turn -> out[(ga*32+pa)*128 + gb*32+pb] = D[0][0]
__global__ void probe_k(float* out) {
  const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  const int ga = blockIdx.x >> 5, pa = blockIdx.x & 31;
  const i32x8 a = onehot(li == 0 && lg == ga, pa);
  for (int kb = 0; kb < 128; ++kb) {
    const int gb = kb >> 5, pb = kb & 31;
    const i32x8 b = onehot(li == 0 && lg == gb, pb);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    if (lane == 0) out[blockIdx.x * 128 + kb] = c[0];
  }
}
// A = B = 1.0 at (row/col 0, group g, byte p) -> product 1; scale_a = 127 everywhere but lane (0, gs) whose scale VGPR byte bs = 128:
// out[((g*32+p)*4 + gs)*4 + bs] = D[0][0]
__global__ void probe_scale(float* out, int which) {
  const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  const int g = blockIdx.x >> 5, p = blockIdx.x & 31;
  const i32x8 a = onehot(li == 0 && lg == g, p);
  for (int s = 0; s < 16; ++s) {
    const int gs = s >> 2, bs = s & 3;
    int sc = 0x7f7f7f7f;
    if (li == 0 && lg == gs) sc = (sc & ~(0xff << (8 * bs))) | (128 << (8 * bs));
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (which == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, a, c, 0, 0, 0, sc, 0, 0x7f7f7f7f);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, a, c, 0, 0, 0, 0x7f7f7f7f, 0, sc);
    if (lane == 0) out[blockIdx.x * 16 + s] = c[0];
  }
}
int main() {
  float* d;
  hipMalloc(&d, 128 * 128 * 4);
  std::vector<float> h(128 * 128);
  probe_k<<<128, 64>>>(d);
  hipMemcpy(h.data(), d, 128 * 128 * 4, hipMemcpyDeviceToHost);
  int ident = 1;
  for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) if ((h[i * 128 + j] != 0.f) != (i == j)) ident = 0;
  printf("A (group, byte) meets B (group, byte) exactly at the same (group, byte): %d\n", ident);
  if (!ident) for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) if (h[i * 128 + j] != 0.f) printf("  A (%d,%d) x B (%d,%d) = %g\n", i >> 5, i & 31, j >> 5, j & 31, h[i * 128 + j]);
  for (int which = 0; which < 2; ++which) {
    probe_scale<<<128, 64>>>(d, which);
    hipMemcpy(h.data(), d, 128 * 16 * 4, hipMemcpyDeviceToHost);
    printf("scale_%c: element (group g, byte p) is scaled by scale VGPR of lane (row 0, group gs) byte bs:\n", which ? 'b' : 'a');
    for (int g = 0; g < 4; ++g) {
      for (int p = 0; p < 32; ++p) {
        printf("  g%d p%2d:", g, p);
        for (int s = 0; s < 16; ++s) if (h[(g * 32 + p) * 16 + s] != 1.f) printf(" (gs %d, byte %d: x%g)", s >> 2, s & 3, h[(g * 32 + p) * 16 + s]);
        printf("\n");
      }
    }
  }
  return 0;
}
