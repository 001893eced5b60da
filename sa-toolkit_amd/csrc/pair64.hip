// Fused ResBlock1 step for C = 64, split planes end to end (the generator's third stage: 20 000 frames per utterance,
// where the two-launch form of the 3-tap steps is HBM-bound).
#include "conv_common.h"

#include <type_traits>

namespace sat {

// ------------------------------------------------------------------------------------------------
// x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 in one launch: the intermediate activation never leaves the CU.  Two
// launches move 20 bytes per element (x in, t1 out; t1 in, residual in, planes out); this one moves 8 — and at 64
// channels x 20 000 frames x 32 utterances the 3-tap launches ran at 4.3 TB/s, i.e. on the HBM roof.
//   Block = 128 - (KS - 1) output positions x all 64 channels, four waves.  conv1 (dilation d) computes t1 on the 128 columns
//   [t0 - h2, t0 - h2 + 128): wave w owns columns 32 w .. 32 w + 31 for both 32-row tiles, K = 4 chunks x KS taps; t1 =
//   lrelu(. + b1) is written to LDS as split planes (zero outside the utterance: conv2's own zero padding) OVER the
//   input chunk, which is dead by then; conv2 (dilation 1) reads column o + tap of it for output column o, its last KS - 1
//   columns of the 128 are not stored (q_end).  Weights stream through LDS per 16-channel chunk in groups of up to four
//   taps (16 KB), the next group's loads in flight in registers during a group's MFMAs (conv2's first under conv1's last).
//   LDS 45-49 KB, 122 VGPRs: three blocks per CU.  Used for 3 and 7 taps (generator -1.2 % and -1.0 %); at 11 taps the
//   recomputed halo (128 columns of t1 for 118 outputs) and the short blocks cost more than the traffic saved.
// Order of operations per accumulator (chunk, tap, lo*hi, hi*lo, hi*hi), the t1 split (leaky-relu, round-toward-zero
// pack) and the epilogue are those of the two-launch path on the conv tile: the same bits.
// ------------------------------------------------------------------------------------------------
// KS taps; the weights of a chunk pass through LDS in groups of TG taps (7 taps: 4 + 3) so that a block stays at 48 KB.
constexpr int P64_W1 = 128;     // t1 window: [t0 - h2, t0 - h2 + 128), h2 = (KS - 1) / 2; 128 - (KS - 1) output positions per block

template <int N, class F>
__device__ __forceinline__ void p64_static_for(F&& f) {
  if constexpr (N > 0) {
    p64_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int KS, int TG, bool RPRE>
__global__ void __launch_bounds__(256, 3) resblock_pair64_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int NCH = 4, W1 = P64_W1, TO = W1 - (KS - 1), H2 = (KS - 1) / 2, XWI = 3, XWP = 64 * XWI, W_UNITS = TG * 4 * 64, W_IT = TG;
  constexpr int G = (KS + TG - 1) / TG;
  uint4* ldsw = lds4;                     // [TG][4][64]     weight group (conv1, then conv2)
  uint4* ldst = lds4 + W_UNITS;           // [NCH][4][W1]    t1, all channels
  uint4* ldsx = ldst;                     // [4][XWP]        input chunk, over the head of the t1 image

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * TO;
  const int xi0 = t0 - H2 - p.pad_left;   // input position of staged column 0 (pad_left = conv1's halo)

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x16 + (long long)b * 64 * p.T_in * 4), 0, (unsigned)(64 * p.T_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;

  // Two staging register sets: the loads of stage s + 2 are issued while stage s is multiplied, so that a stage's loads
  // have two stage times to land (a stage is short here: 36-48 MFMAs per wave); the barriers wait for LDS only
  // (`s_waitcnt lgkmcnt(0)` + `s_barrier`: __syncthreads() would drain the loads in flight), the compiler counts vmcnt
  // down to the set a publish reads.
  // Stage s of 2 * NCH * G: conv s / (NCH G), chunk (s / G) % NCH, tap group s % G.
  constexpr int NS = 2 * NCH * G;
  uint4 xst[2][XWI], wst[2][W_IT];
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto issue = [&](auto sc) __attribute__((always_inline)) {
    constexpr int st = decltype(sc)::value, set = st & 1, conv = st / (NCH * G), ch = (st / G) % NCH, g = st % G;
    constexpr int tg0 = g * TG, nt = (KS - tg0 < TG) ? KS - tg0 : TG;
    if constexpr (conv == 0 && g == 0) {
#pragma unroll
      for (int it = 0; it < XWI; ++it) {
        const int xi = xi0 + lane + 64 * it;
        // (p.xw: the 128-column window of conv1 plus the halo of its dilated taps — the rest of the 192 staged columns is never multiplied)
        const unsigned voff = (xi >= 0 && xi < p.T_in && lane + 64 * it < p.xw) ? (unsigned)((wave * p.T_in + xi) * 16) : 0x80000000u;
        xst[set][it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, ch * 4 * p.T_in * 16, 0));
      }
    }
    // wave w copies segment w (part * 2 + half) of every tap of the group
#pragma unroll
    for (int i = 0; i < nt; ++i)
      wst[set][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(conv ? w2rs : w1rs, lane * 16 + wave * seg_bytes,
                                                                                     ((ch * KS + tg0 + i) * 4) * seg_bytes, 0));
  };
  auto publish = [&](auto sc) __attribute__((always_inline)) {
    constexpr int st = decltype(sc)::value, set = st & 1, conv = st / (NCH * G), g = st % G;
    constexpr int tg0 = g * TG, nt = (KS - tg0 < TG) ? KS - tg0 : TG;
    if constexpr (conv == 0 && g == 0) {
#pragma unroll
      for (int it = 0; it < XWI; ++it) ldsx[wave * XWP + lane + 64 * it] = xst[set][it];
    }
#pragma unroll
    for (int i = 0; i < nt; ++i) ldsw[(i * 4 + wave) * 64 + lane] = wst[set][i];
  };
  f32x16 acc[2][1];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;
  };
  // one tap group of MFMAs: B fragments at base + tap * step in an image of plane pitch `pitch`
  auto mfma_group = [&](const uint4* base, int pitch, int step, auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value, tg0 = g * TG, nt = (KS - tg0 < TG) ? KS - tg0 : TG;
    const uint4* wb = ldsw + lh * 64 + l31;
#pragma unroll
    for (int t = 0; t < nt; ++t) {
      const uint4* xt = base + (tg0 + t) * step;
      const h8 b_hi = __builtin_bit_cast(h8, xt[0]);
      const h8 b_lo = __builtin_bit_cast(h8, xt[2 * pitch]);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const h8 a_hi = __builtin_bit_cast(h8, wb[(t * 4 + 0) * 64 + m * 32]);
        const h8 a_lo = __builtin_bit_cast(h8, wb[(t * 4 + 2) * 64 + m * 32]);
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[m][0], 0, 0, 0);
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[m][0], 0, 0, 0);
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[m][0], 0, 0, 0);
      }
    }
  };
  // t1 = lrelu(conv1 + b1) -> LDS as conv2's B operand (hi | lo planes of the four chunks), zero outside the utterance
  auto write_t1 = [&]() __attribute__((always_inline)) {
    const int col = wave * 32 + l31;
    const int pos = t0 - H2 + col;
    const bool inside = pos >= 0 && pos < p.T_in;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = __builtin_fmaf(acc[m][0][4 * rg + k], p.w_descale1, p.bias1[32 * m + 8 * rg + 4 * lh + k]);
          t = lrelu_max(t, p.in_slope);
          v[k] = inside ? t : 0.f;
        }
        const auto h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
        const auto h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
        const auto l01 = split_lo2(h01, v[0], v[1]);
        const auto l23 = split_lo2(h23, v[2], v[3]);
        // rows 32 m + 8 rg + 4 lh + k: chunk 2 m + rg / 2, half rg & 1, bytes 8 lh .. of the unit
        const int chunk = 2 * m + (rg >> 1);
        ((uint2*)(ldst + ((chunk * 4 + 0 + (rg & 1)) * W1 + col)))[lh] =
            make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
        ((uint2*)(ldst + ((chunk * 4 + 2 + (rg & 1)) * W1 + col)))[lh] =
            make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
      }
  };

  // RPRE: the residual words of the wave's output tile (the block's own input planes) are requested when conv1 is done, a whole conv2 of
  // matrix work ahead of the epilogue that adds them (32 registers; the same loads, earlier: the same bits)
  float rpre[2][1][16];
  zero_acc();
  issue(std::integral_constant<int, 0>{});
  issue(std::integral_constant<int, 1>{});
  const uint4* xbase = ldsx + lh * XWP + wave * 32 + l31;
  p64_static_for<NS>([&](auto sc) __attribute__((always_inline)) {
    constexpr int st = decltype(sc)::value, conv = st / (NCH * G), ch = (st / G) % NCH, g = st % G;
    lds_barrier();                          // every wave is done reading the previous stage's tiles (first conv2 stage: t1 complete)
    publish(sc);
    lds_barrier();
    if constexpr (st + 2 < NS) issue(std::integral_constant<int, st + 2>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (conv == 0) mfma_group(xbase, XWP, p.dil, std::integral_constant<int, g>{});
    else mfma_group(ldst + (ch * 4 + lh) * W1 + wave * 32 + l31, W1, 1, std::integral_constant<int, g>{});
    if constexpr (st == NCH * G - 1) {      // conv1 done
      lds_barrier();                        // every wave is done reading the input chunk: t1 goes over it
      write_t1();
      zero_acc();
      if constexpr (RPRE) epilogue_prefetch_res<2, 1>(p, rpre, b, 0, 0, t0 + wave * 32, l31, lh, 32, t0 + TO);
    }
  });
  // residual = the block input (from its planes), bias, MRF accumulate, f32 and / or planes out; the last KS - 1 columns dropped
  if constexpr (RPRE) conv_epilogue<2, 1, true, false>(p, acc, b, 0, 0, t0 + wave * 32, l31, lh, 32, t0 + TO, rpre);
  else conv_epilogue<2, 1, false, false>(p, acc, b, 0, 0, t0 + wave * 32, l31, lh, 32, t0 + TO);
}

bool pair64_supports(const ConvArgs& a) {
  return a.cin_g == 64 && a.rows_g == 64 && (a.ksize == 3 || a.ksize == 7 || a.ksize == 11) && a.x16 && a.res16 && a.fast_epi &&
         !a.ch_scale && a.co_pad == 64 && P64_W1 + (a.ksize - 1) * a.dil <= 192;
}

static int g_pair64_rpre = 1;      // option "pair64_rpre"
void pair64_rpre_set(int v) { g_pair64_rpre = v != 0; }

template <int KS, int TG>
static int launch_p64(const ConvArgs& a, int B, hipStream_t s) {
  // (+ KS - 1 units: conv2's last, unstored, columns read t1 columns 128 ..)
  const size_t lds_bytes = ((size_t)TG * 4 * 64 + (size_t)4 * 4 * P64_W1 + KS - 1) * 16;
  dim3 grid(ceil_div(a.T_q, P64_W1 - (KS - 1)), 1, B);
  if (g_pair64_rpre && a.res16 && !a.res) hipLaunchKernelGGL((resblock_pair64_kernel<KS, TG, true>), grid, dim3(256), lds_bytes, s, a);
  else hipLaunchKernelGGL((resblock_pair64_kernel<KS, TG, false>), grid, dim3(256), lds_bytes, s, a);
  SAT_LAUNCH_CHECK("resblock_pair64_kernel");
  return SAT_OK;
}

int launch_pair64(const ConvArgs& a, int B, hipStream_t s) {
  switch (a.ksize) {
    case 3: return launch_p64<3, 3>(a, B, s);
    case 7: return launch_p64<7, 4>(a, B, s);
    default: return launch_p64<11, 4>(a, B, s);
  }
}

}  // namespace sat
