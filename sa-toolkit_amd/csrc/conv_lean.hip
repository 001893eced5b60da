// Three-blocks-per-CU form of the k-tap conv tile on split planes (3 / 7 / 11 taps, no folded BatchNorm: the generator's
// resblock convs).
#include "conv_common.h"

#include <type_traits>

namespace sat {

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ------------------------------------------------------------------------------------------------
// conv1d_f16x3_planes_kernel (conv1d_mfma.hip) holds 235 VGPRs — two fragment sets and 64 residual-prefetch registers —
// and up to 65 KB of LDS: two blocks per CU, and its launches wait on memory 14-35 % of their cycles.  This form trades
// the deeper per-wave pipelining for a third co-resident block: one flowing fragment set, the residual loaded in the
// epilogue, the folded-BatchNorm step compiled out (<= 168 VGPRs), and for 11 taps the weights of a chunk pass through
// LDS in two tap groups (6 + 5: 24 KB instead of 45), so that a block needs 33 / 49 / 45 KB of LDS at 3 / 7 / 11 taps.
// Same tile (64 rows x 256 positions, four waves of 64 x 64), same LDS images, same order of operations per
// accumulator (chunk, tap, lo*hi, hi*lo, hi*hi) and the same epilogue arithmetic: bit-identical results
// (tests/test_hip_parity.py).  The fixed costs of a block (first loads, epilogue) and, at C = 256, the 640 blocks on
// 512 slots of the two-block form (now 768 slots) are what the third block hides.
// ------------------------------------------------------------------------------------------------
template <int KS, int TG>
__global__ void __launch_bounds__(256, 3) conv1d_f16x3_planes_lean_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int MT = 2, NT = 2, CO_B = 64, T_B = 256, XWI = 5, XWP = 64 * XWI;
  constexpr int G = (KS + TG - 1) / TG;
  uint4* ldsx = lds4;                     // [4][XWP]
  uint4* ldsw = lds4 + 4 * XWP;           // [TG][4][CO_B]

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Plain grid: (column tile, row tile, utterance).  Balanced grid (p.bal): a 1-D grid whose first blocks are the full tiles and
  // whose last blocks are the ragged end of every (row tile, utterance) as 128-column HALF tiles (waves 2, 3 idle): the
  // dispatcher hands blocks out in id order, so the small blocks come last.  (Tried for balance — 640 tiles on 768 slots at
  // C = 256, T = 1250 as 512 full + 256 half tiles, 2.5 per CU: level, a CU with 3 blocks is not 1.5 x slower than one with 2.)
  int b, co_w, q_b;
  bool active = true;
  if (p.bal) {
    const int id = blockIdx.x, nfull = p.bal_fpp * p.bal;     // p.bal = number of (row tile, utterance) pairs
    int pair;
    if (id < nfull) {
      pair = id / p.bal_fpp;
      q_b = (id - pair * p.bal_fpp) * T_B;
    } else {
      const int hid = id - nfull;
      pair = hid / p.bal_hpp;
      q_b = p.bal_fpp * T_B + (hid - pair * p.bal_hpp) * (T_B / 2);
      active = wave < 2;
    }
    pair = __builtin_amdgcn_readfirstlane(pair);
    q_b = __builtin_amdgcn_readfirstlane(q_b);
    b = __builtin_amdgcn_readfirstlane(pair / p.co_tiles_g);
    co_w = (pair - b * p.co_tiles_g) * CO_B;
  } else {
    b = blockIdx.z;
    co_w = blockIdx.y * CO_B;
    q_b = blockIdx.x * T_B;
  }
  const int q_w = q_b + wave * (32 * NT);
  const int xi0 = q_b - p.pad_left;
  const int nch = p.cin_pad / CI_CHUNK;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x16 + (long long)b * p.cin_g * p.T_in * 4), 0, (unsigned)(p.cin_g * p.T_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;
  const int x_chunk_bytes = 4 * p.T_in * 16;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // staging: wave w copies segment w (part * 2 + half) of every tap of the group and plane w of the chunk
  uint4 wst[TG], xst[XWI];
  const unsigned w_voff = (unsigned)((co_w + lane) * 16 + wave * seg_bytes);
  auto issue_w = [&](int chunk, auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value, t0 = g * TG, nt = (KS - t0 < TG) ? KS - t0 : TG;
#pragma unroll
    for (int i = 0; i < nt; ++i)
      wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                             wrs, w_voff, __builtin_amdgcn_readfirstlane(((chunk * KS + t0 + i) * 4) * seg_bytes), 0));
  };
  auto issue_x = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < XWI; ++it) {
      const int xi = xi0 + lane + 64 * it;
      const unsigned voff = (xi >= 0 && xi < p.T_in) ? (unsigned)((wave * p.T_in + xi) * 16) : 0x80000000u;
      xst[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, __builtin_amdgcn_readfirstlane(chunk * x_chunk_bytes), 0));
    }
  };
  auto publish = [&](auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value, t0 = g * TG, nt = (KS - t0 < TG) ? KS - t0 : TG;
#pragma unroll
    for (int i = 0; i < nt; ++i) ldsw[(i * 4 + wave) * CO_B + lane] = wst[i];
    if constexpr (g == 0) {
#pragma unroll
      for (int it = 0; it < XWI; ++it) ldsx[wave * XWP + lane + 64 * it] = xst[it];
    }
  };
  const uint4* xb0 = ldsx + lh * XWP + wave * (32 * NT) + l31;
  const uint4* wb0 = ldsw + lh * CO_B + l31;
  // Fragments FLOW through one register set + one spare A pair (gemm_f16x3_ring16_kernel): the next tap's A pair of
  // row m is read once row m - 1 has been multiplied, its B pair of column n once the last row has used column n, so
  // no tap but a group's first starts on an LDS round trip — at 40 fragment registers instead of 64 for two full sets.
  auto read_a = [&](h8 (&dst)[2], int tl, int m) __attribute__((always_inline)) {
    dst[0] = __builtin_bit_cast(h8, wb0[(tl * 4 + 0) * CO_B + m * 32]);
    dst[1] = __builtin_bit_cast(h8, wb0[(tl * 4 + 2) * CO_B + m * 32]);
  };
  auto read_b = [&](h8 (&dst)[2], int t, int n) __attribute__((always_inline)) {
    const uint4* xt = xb0 + t * p.dil + n * 32;
    dst[0] = __builtin_bit_cast(h8, xt[0]);
    dst[1] = __builtin_bit_cast(h8, xt[2 * XWP]);
  };
  auto mfma_group = [&](auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value, t0 = g * TG, nt = (KS - t0 < TG) ? KS - t0 : TG;
    h8 fa[MT][2], fb[NT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) read_a(fa[m], 0, m);
#pragma unroll
    for (int n = 0; n < NT; ++n) read_b(fb[n], t0, n);
#pragma unroll
    for (int tl = 0; tl < nt; ++tl) {
      h8 an[MT][2], bn[NT][2];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        __builtin_amdgcn_sched_barrier(0);
        if (tl + 1 < nt) read_a(an[m], tl + 1, m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m][1], fb[n][0], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m][0], fb[n][1], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m][0], fb[n][0], acc[m][n], 0, 0, 0);
          if (m == MT - 1 && tl + 1 < nt) {
            __builtin_amdgcn_sched_barrier(0);
            read_b(bn[n], t0 + tl + 1, n);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (tl + 1 < nt) {
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[m][0] = an[m][0], fa[m][1] = an[m][1];
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n][0] = bn[n][0], fb[n][1] = bn[n][1];
      }
    }
  };

  issue_x(0);
  issue_w(0, std::integral_constant<int, 0>{});
  for (int chunk = 0; chunk < nch; ++chunk) {
    static_for<G>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      __syncthreads();                       // every wave is done reading the previous stage's tiles; this stage's loads are back
      publish(gc);
      __syncthreads();
      if constexpr (g + 1 < G) {
        issue_w(chunk, std::integral_constant<int, g + 1>{});
      } else {
        if (chunk + 1 < nch) {
          issue_x(chunk + 1);
          issue_w(chunk + 1, std::integral_constant<int, 0>{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (active) mfma_group(gc);
    });
  }
  if (active) conv_epilogue<MT, NT, false, false>(p, acc, b, 0, co_w, q_w, l31, lh);
}

// 0 = plain grid; 1 = half tile where ONE covers a row's ragged end (never more blocks than the plain grid, less wasted work);
// 2 = always (two half tiles for ends of 129 .. 255 columns: measured level to 10 % slower, a half tile stages what a full one does)
static int g_lean_balance = 1;
void lean_set_balance(int v) { g_lean_balance = v < 0 ? 0 : v > 2 ? 2 : v; }

bool lean_supports(const ConvArgs& a) {
  return a.x16 && !a.f8 && !a.y16_f8 && !a.poly_planes && !a.ch_scale && a.fast_epi && a.up == 1 && a.stride == 1 &&
         (a.ksize == 3 || a.ksize == 7 || a.ksize == 11) && a.rows_g > 32 && a.co_pad % 64 == 0 && 256 + (a.ksize - 1) * a.dil <= 320;
}

template <int KS, int TG>
static int launch_lean(const ConvArgs& a, int B, hipStream_t s) {
  ConvArgs p = a;
  p.xw = 256 + (p.ksize - 1) * p.dil;
  p.co_tiles_g = ceil_div(p.rows_g, 64);
  const size_t lds_bytes = ((size_t)4 * 320 + (size_t)TG * 4 * 64) * 16;
  auto kern = conv1d_f16x3_planes_lean_kernel<KS, TG>;
  dim3 grid(ceil_div(p.T_q, 256), p.co_tiles_g, B);
  const int rem = p.T_q % 256;
  if (rem != 0 && (g_lean_balance == 2 || (g_lean_balance == 1 && rem <= 128))) {
    p.bal = p.co_tiles_g * B;
    p.bal_fpp = p.T_q / 256;
    p.bal_hpp = rem > 128 ? 2 : 1;
    grid = dim3((unsigned)((p.bal_fpp + p.bal_hpp) * p.bal), 1, 1);
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("conv1d_f16x3_planes_lean_kernel");
  return SAT_OK;
}

int launch_f16x3_lean(const ConvArgs& a, int B, hipStream_t s) {
  switch (a.ksize) {
    case 3: return launch_lean<3, 3>(a, B, s);
    case 7: return launch_lean<7, 7>(a, B, s);
    case 11: return launch_lean<11, 6>(a, B, s);
  }
  set_error("conv1d(lean): %d taps not instantiated", a.ksize);
  return SAT_ERR_INVALID;
}

}  // namespace sat
