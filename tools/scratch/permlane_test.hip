// what v_permlane16_swap_b32 does on gfx950 (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  p[threadIdx.x] = r[0]; p[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("first : "); for (int i = 0; i < 64; i += 4) printf("%u ", h[i]); printf("\nsecond: "); for (int i = 0; i < 64; i += 4) printf("%u ", h[64 + i]); printf("\n");
  return 0;
}
