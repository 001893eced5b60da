// Diagnostic build of the ring GEMM (never part of the library): what does one step cost without its barrier / without
// its DMA issue?  Results of the ablated variants are wrong; only the time is read.
//   for v in 0 1 2 3; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSAT_RING_ABLATE=$v -Iinclude \
//         tools/ablate_ring.hip -o tools/ablate_ring_$v; done;  tools/ablate_ring_0 1024 1024 249 32
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sa-toolkit_amd/csrc/api.hip"
#include "../sa-toolkit_amd/csrc/gemm_ring.hip"

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void fill(unsigned* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) {
    unsigned v = (unsigned)i * 2654435761u + seed;
    v ^= v >> 15;
    p[i] = (v & 0x83ff83ffu) | 0x38003800u;   // two f16 in [-1, 1)
  }
}

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 1024, cout = argc > 2 ? atoi(argv[2]) : 1024;
  const int T = argc > 3 ? atoi(argv[3]) : 249, B = argc > 4 ? atoi(argv[4]) : 32;
  const size_t nx = (size_t)B * cin * T, ny = (size_t)B * cout * T, wbytes = (size_t)(cin / 16) * cout * 64;
  void *xs, *w; float *y, *bias;
  CK(hipMalloc(&xs, nx * 4)); CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&bias, cout * 4));
  CK(hipMemset(bias, 0, cout * 4));
  fill<<<(nx + 255) / 256, 256>>>((unsigned*)xs, nx, 1);
  fill<<<(wbytes / 4 + 255) / 256, 256>>>((unsigned*)w, wbytes / 4, 3);
  sat::ConvArgs a{};
  a.w = (const float*)w; a.y = y; a.bias = bias; a.x16 = xs;
  a.y_bs = (long long)cout * T; a.y_cs = T;
  a.cin_g = cin; a.T_in = T; a.rows_g = cout; a.cout_g = cout; a.T_q = T;
  a.ksize = 1; a.dil = 1; a.stride = 1; a.up = 1; a.cin_pad = cin; a.co_pad = cout;
  a.w_gs = (long long)wbytes; a.res_tstride = 1; a.fast_epi = 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) if (sat::launch_f16x3_ring16(a, B, nullptr) != 0) { printf("error: %s\n", sat_last_error()); return 1; }
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) sat::launch_f16x3_ring16(a, B, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms / 20 < best ? ms / 20 : best;
  }
  const double fl = 2.0 * cin * cout * (double)B * T;
  printf("SAT_RING_ABLATE=%d (bits: 1 no barrier, 2 no DMA issue in the loop, 4 no fragment reads in the loop): %d -> %d, T=%d, B=%d: %.1f us per launch, %.0f TFLOP/s useful\n",
         SAT_RING_ABLATE, cin, cout, T, B, best * 1e3, fl / (best * 1e-3) / 1e12);
  return 0;
}
