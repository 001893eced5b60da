// Diagnostic build of the split-f16 conv kernel with s_memtime phase stamps (never part of the
// library): where does a K-chunk of a thick generator layer spend its cycles?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSAT_STAMPS -Iinclude \
//         tools/stamp_conv.hip -o tools/stamp_conv && tools/stamp_conv C T k dil [B]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sa-toolkit_amd/csrc/api.hip"
#include "../sa-toolkit_amd/csrc/conv1d_mfma.hip"

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void fill(unsigned* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) {
    unsigned v = (unsigned)i * 2654435761u + seed;
    v ^= v >> 15;
    // two f16 in [-1, 1): sign + exponent <= 14
    p[i] = (v & 0x83ff83ffu) | 0x38003800u;
  }
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 1250;
  const int k = argc > 3 ? atoi(argv[3]) : 11, dil = argc > 4 ? atoi(argv[4]) : 5, B = argc > 5 ? atoi(argv[5]) : 32;
  sat::g_stamp_variant = argc > 6 ? atoi(argv[6]) : 0;   // 1: no output stores, 2: no residual, 3: neither
  const size_t n = (size_t)B * C * T;
  float *res, *y;
  void *xs, *ys, *w;
  const size_t wbytes = (size_t)(C / 16) * k * round_up(C, 64) * 64;
  CK(hipMalloc(&res, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&xs, n * 4)); CK(hipMalloc(&ys, n * 4)); CK(hipMalloc(&w, wbytes));
  float* bias; CK(hipMalloc(&bias, C * 4)); CK(hipMemset(bias, 0, C * 4));
  fill<<<(n + 255) / 256, 256>>>((unsigned*)xs, n, 1);
  fill<<<(n + 255) / 256, 256>>>((unsigned*)res, n, 2);
  fill<<<(wbytes / 4 + 255) / 256, 256>>>((unsigned*)w, wbytes / 4, 3);
  const int nchunk = C / 16;
  const int co_tiles = (C + 63) / 64, t_tiles = (T + 255) / 256;
  const size_t nblocks = (size_t)B * co_tiles * t_tiles;
  long long* dbg; CK(hipMalloc(&dbg, nblocks * nchunk * 64)); CK(hipMemset(dbg, 0, nblocks * nchunk * 64));
  sat_conv1d_desc d{};
  d.B = B; d.C_in = C; d.T_in = T; d.C_out = C; d.T_q = T; d.ksize = k; d.dilation = dil; d.stride = 1; d.groups = 1; d.up = 1;
  d.pad_left = (k * dil - dil) / 2; d.mode = SAT_CONV_F16X3;
  d.x_cstride = T; d.x_bstride = (int64_t)C * T; d.y_cstride = T; d.y_bstride = (int64_t)C * T;
  // the generator's conv2 of a ResBlock step: residual from split planes, planes out, no f32 store
  d.res_split = res; d.res_split_slope = 0.1f; d.res_scale = 1.f; d.res_tstride = 1; d.no_y = 1;
  d.bias = bias; d.x_split = xs; d.y_split = ys; d.y_split_slope = 0.1f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int pass = 0; pass < 2; ++pass) {
    sat::g_stamp_buffer = pass ? dbg : nullptr;
    for (int i = 0; i < 3; ++i) if (sat_conv1d_f32(&d, nullptr, w, y, nullptr) != 0) { printf("error: %s\n", sat_last_error()); return 1; }
    CK(hipEventRecord(e0));
    const int reps = pass ? 1 : 20;
    for (int i = 0; i < reps; ++i) sat_conv1d_f32(&d, nullptr, w, y, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: %.1f us per launch (C=%d T=%d k=%d d=%d B=%d, %zu blocks)\n", pass ? "stamped" : "plain", ms / reps * 1e3, C, T, k, dil, B, nblocks);
  }
  std::vector<long long> h(nblocks * nchunk * 8);
  CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
  // per-phase medians over (block, chunk)
  const char* names[6] = {"(unused)", "wait for loads + ds_write", "barrier after staging", "issue next loads + MFMA phase", "barrier at loop top (next chunk)", "epilogue"};
  std::vector<long long> ph[6], life;
  for (size_t b = 0; b < nblocks; ++b) {
    const long long* s = &h[b * nchunk * 8];
    for (int c = 0; c < nchunk; ++c) {
      const long long* q = s + c * 8;
      ph[0].push_back(0); ph[1].push_back(q[2] - q[0]); ph[2].push_back(q[3] - q[2]); ph[3].push_back(q[4] - q[3]);
      if (c + 1 < nchunk) ph[4].push_back(q[8] - q[4]);
    }
    ph[5].push_back(s[(nchunk - 1) * 8 + 5] - s[(nchunk - 1) * 8 + 4]);
    life.push_back(s[(nchunk - 1) * 8 + 5] - s[0]);
  }
  auto med = [](std::vector<long long>& v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
  for (int i = 0; i < 6; ++i) printf("  %-34s p10 %7lld  median %7lld  p90 %7lld cycles\n", names[i], med(ph[i], .1), med(ph[i], .5), med(ph[i], .9));
  printf("  %-34s p10 %7lld  median %7lld  p90 %7lld cycles;  MFMA issue floor per chunk = %d\n", "block lifetime", med(life, .1), med(life, .5), med(life, .9), k * 12 * 32);
  if (nchunk >= 2) {
    // wall clock (100 MHz) per block, shader clock, placement
    long long t0 = h[6], t1 = 0;
    std::vector<double> mhz;
    std::vector<int> per_cu(8 * 64 * 4, 0);
    for (size_t b = 0; b < nblocks; ++b) {
      const long long* s = &h[b * nchunk * 8];
      t0 = std::min(t0, s[6]); t1 = std::max(t1, s[14]);
      mhz.push_back((double)(s[(nchunk - 1) * 8 + 5] - s[0]) / (double)(s[14] - s[6]) * 100.0);
      const unsigned hw = (unsigned)s[7], xcc = (unsigned)(s[7] >> 32) & 15;
      const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // gfx9 HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
      per_cu[((xcc * 8 + se) * 2 + sh) * 16 + cu]++;
    }
    std::sort(mhz.begin(), mhz.end());
    printf("  wall span of all blocks: %.1f us;  shader clock inside blocks: median %.0f MHz (p10 %.0f, p90 %.0f)\n", (t1 - t0) / 100.0, mhz[mhz.size() / 2], mhz[mhz.size() / 10], mhz[mhz.size() * 9 / 10]);
    int hist[16] = {0}, used = 0;
    for (int c : per_cu) if (c) { hist[std::min(c, 15)]++; used++; }
    printf("  CUs used: %d; blocks per CU histogram:", used);
    for (int i = 1; i < 16; ++i) if (hist[i]) printf(" %dx%d", hist[i], i);
    printf("\n");
  }
  return 0;
}
