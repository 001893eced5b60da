// YAAPT pitch tracking on gfx950, batched over (utterance, frame).
//
// Reference: satools/satools/hifigan/yaapt.py:795-951 (`_yaapt`/`yaapt`: a serial Python loop over
// the batch, forced onto the CPU) with the options of egs/vc/libritts/local/tuning/hifigan.py:31-36.
// Kernel by kernel:
//   prefilter     SignalObj.filtered_version :42-51 (torchaudio biquads; third-party)
//   nlfer_frames  nlfer :148-176                    one 8192-point FFT in LDS per frame
//   energy_norm   PitchObj.set_energy :125-128
//   spec_frames   spec_track :209-238 + peaks :383-497   FFT -> |X| -> SHC -> candidate peaks
//   spec_post     spec_track :241-312, dynamic5 :506-523, path1 :530-570, medfilt :54-69
//   frame_means   the in-place mean subtraction on overlapping frame views, time_track :711-714 + :589
//   nccf_frames   crs_corr :577-602 + cmp_rate :609-673 + merit weighting :724-727
//   refine_dp     refine :732-784, dynamic :321-370, path1
// Decision rules (strict/non-strict comparisons, first/last extremum on ties, stable ordering) are
// kept exactly; f32 operation order is kept wherever the reference's order is defined by its source.
#include <cmath>
#include <type_traits>

#include "common.h"

namespace sat {

typedef sat_yaapt_plan Plan;

constexpr int FFT_N = 8192;
constexpr int FFT_LOG = 13;
constexpr int MAXP = 4;   // shc_maxpeaks
constexpr int NC = 6;     // refine candidates: 2 tracks x nccf_maxcands

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wmin_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}

// block-wide sum of one float per thread (256 threads), result broadcast; `red` = 8 floats of LDS
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = v;   // waves past the fourth (FFT helpers) only read
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wmax(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// torch.maximum / torch.minimum propagate NaN
__device__ __forceinline__ float tmax(float a, float b) { return (a != a || b != b) ? NAN : fmaxf(a, b); }
__device__ __forceinline__ float tmin(float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); }

// ragged batches: U[b] = {samples, padded length L, frames, tda frames} of utterance b (host-computed with the
// plan's formulas); null = every utterance has the plan's length.  Row strides stay the plan's (the longest).
__device__ __forceinline__ int ulen(const int* __restrict__ U, int b, int k, int dflt) { return U ? U[b * 4 + k] : dflt; }

// ------------------------------------------------------------------------------------------------
// prefilter: zero-pad, (square), low-pass biquad -> clamp -> high-pass biquad -> clamp   (yaapt.py:42-51).
// torchaudio's lfilter order (third party, restated in oracle/biquad.py, order "torchaudio"):
//   FIR   f[t] = fma(b0', x[t], fma(b1', x[t-1], b2' * x[t-2]))      b' = b / a0  (conv1d's FMA chain in tap order)
//   IIR   v = f[t] - c2 * y[t-2];  v = v - c1 * y[t-1];  y[t] = v    c = a / a0   (multiply, subtract; no fma)
//   out   clamp(y[t], -1, 1); the recursion itself runs on unclamped values
// The recursion is sequential in t and its f32 rounding order is part of the result, so the only parallelism is
// ACROSS chains: a chain = (utterance, signal in {x, x^2}), and a lane of the IIR waves owns one chain (one
// wave-instruction advances 64 chains by one sample).  A block takes 32 utterances = 64 chains through a six-stage
// software pipeline over tiles of 64 samples, one role per wave, one barrier per tile, the tile's LDS buffer
// (row = chain, 68 floats: 16-byte rows, ds_read_b128 of 64 rows conflict-free) handed down the stages and
// transformed IN PLACE:
//   L  load 32 utterances (coalesced, registers one tile ahead), write rows x | x^2        lane = 4 samples of a row
//   F1 low-pass FIR    (3 VALU per sample)                                                  lane = chain
//   I1 low-pass IIR + clamp (sub, sub, 2 mul, clamp: the critical wave, ~18 cycles/sample)  lane = chain
//   F2 high-pass FIR, I2 high-pass IIR + clamp                                              lane = chain
//   S  store rows (zero past the chain's own padded length: the zero extension spec_track reads)
// A workgroup's waves are dealt to the CU's four SIMDs cyclically (k and k + 4 share one), so the roles are
// numbered to leave each IIR wave alone on its SIMD: waves 0 / 1 = I1 / I2, 2 / 3 = F1 / F2, 4 / 5 idle,
// 6 = L (beside F1), 7 = S (beside F2; placement checked with s_getreg HW_ID).  Measured (s_memtime stamps per role,
// 32 x 5 s): 2.2k cycles per tile in the IIR waves (34 cycles per sample: sub, sub, pk_mul — 8 issue cycles, no
// gain over two v_mul —, clamp, plus the row's LDS traffic; a lone wave issues one VALU per ~4.2 cycles,
// tools/valu_lat.hip), 2.0-2.5k in the others; 1 260 tiles = 1.25 ms at the 2.4 GHz the chip holds under this
// one-CU kernel, for any batch up to 32 (round 1: one block of two waves per chain with wave-uniform recursions,
// 1.7-2.5 ms, 64 blocks).
// ------------------------------------------------------------------------------------------------
constexpr int PF_T = 64;
constexpr int PF_PITCH = 68;
constexpr int PF_RING = 6;
constexpr int PF_THREADS = 512;

__device__ __forceinline__ void pf_fir_tile(float* __restrict__ row, float b0, float b1, float b2, float& x1, float& x2) {
  float4 xq[PF_T / 4];
#pragma unroll
  for (int q = 0; q < PF_T / 4; ++q) xq[q] = *reinterpret_cast<const float4*>(row + 4 * q);
#pragma unroll
  for (int q = 0; q < PF_T / 4; ++q) {
    const float4 x = xq[q];
    float4 f;
    f.x = __builtin_fmaf(b0, x.x, __builtin_fmaf(b1, x1, b2 * x2));
    f.y = __builtin_fmaf(b0, x.y, __builtin_fmaf(b1, x.x, b2 * x1));
    f.z = __builtin_fmaf(b0, x.z, __builtin_fmaf(b1, x.y, b2 * x.x));
    f.w = __builtin_fmaf(b0, x.w, __builtin_fmaf(b1, x.z, b2 * x.y));
    x2 = x.z;
    x1 = x.w;
    *reinterpret_cast<float4*>(row + 4 * q) = f;
  }
}

// One step of the recursion.  State: p1 = c1*y[t-1], p2n = c2*y[t-1], d = f[t] - c2*y[t-2] (the first subtraction, taken
// one step ahead: it does not depend on y[t-1], and in program order right behind `v = d - p1` it fills that
// instruction's result latency instead of standing in front of it).  Dependent chain per sample: sub -> pk_mul.
// The operations and their order per sample are exactly  v = f - c2*y2;  v = v - c1*y1.
__device__ __forceinline__ float pf_iir_step(float f_next, float c1, float c2, float& p1, float& p2n, float& d) {
  const float v = d - p1;
  d = f_next - p2n;
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f q = (v2f){c1, c2} * (v2f){v, v};      // one v_pk_mul_f32: the same two IEEE products
  p1 = q.x;
  p2n = q.y;
  return fminf(fmaxf(v, -1.f), 1.f);
}
// `d` enters holding f[first sample of this tile] - c2*y[t-2] ... which needs this tile's first f: the caller passes
// `dp2` = c2*y[t-2] pending from the previous tile instead, and the tile finishes the subtraction itself.
__device__ __forceinline__ void pf_iir_tile(float* __restrict__ row, float c1, float c2, float& p1, float& p2n, float& dp2) {
  // the whole row up front (64 registers): one LDS round trip per tile instead of one per group of reads
  float4 fq[PF_T / 4];
#pragma unroll
  for (int q = 0; q < PF_T / 4; ++q) fq[q] = *reinterpret_cast<const float4*>(row + 4 * q);
  float d = fq[0].x - dp2;
#pragma unroll
  for (int q = 0; q < PF_T / 4; ++q) {
    const float4 f = fq[q];
    float4 u;
    u.x = pf_iir_step(f.y, c1, c2, p1, p2n, d);
    u.y = pf_iir_step(f.z, c1, c2, p1, p2n, d);
    u.z = pf_iir_step(f.w, c1, c2, p1, p2n, d);
    if (q + 1 < PF_T / 4) {
      u.w = pf_iir_step(fq[q + 1 < PF_T / 4 ? q + 1 : q].x, c1, c2, p1, p2n, d);
    } else {                                    // last sample of the tile: the next f is not here yet
      const float v = d - p1;
      dp2 = p2n;                                // c2*y[t-1] of this sample = c2*y[t-2] of the next tile's first
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v2f qq = (v2f){c1, c2} * (v2f){v, v};
      p1 = qq.x;
      p2n = qq.y;
      u.w = fminf(fmaxf(v, -1.f), 1.f);
    }
    *reinterpret_cast<float4*>(row + 4 * q) = u;
  }
}

// VEC: rows of wav / filt are 16-byte aligned (n, pad, Lz multiples of 4): 16-byte global accesses.
// Loads and stores go through buffer descriptors over the block's rows: a lane outside its row's valid range gets
// an out-of-range offset (loads return 0, stores are dropped), so both roles are branch-free and the compiler
// batches a tile's LDS reads and memory operations instead of one round trip per row.
template <bool VEC>
__global__ void __launch_bounds__(PF_THREADS) yaapt_prefilter_kernel(const float* __restrict__ wav, float* __restrict__ filt,
                                                                     const int* __restrict__ U, const Plan P, int B) {
  __shared__ __attribute__((aligned(16))) float ring[PF_RING][64 * PF_PITCH];
  __shared__ int s_n[32], s_L[32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ub = blockIdx.x * 32;                       // first utterance of the block
  const int rows = min(32, B - ub);
  if (threadIdx.x < 32) {
    const int b = ub + threadIdx.x;
    s_n[threadIdx.x] = b < B ? ulen(U, b, 0, P.n) : 0;
    s_L[threadIdx.x] = b < B ? ulen(U, b, 1, P.L) : 0;
  }
  __syncthreads();
  // role -> pipeline stage
  const int stage = wave == 6 ? 0 : wave == 2 ? 1 : wave == 0 ? 2 : wave == 3 ? 3 : wave == 1 ? 4 : wave == 7 ? 5 : -1;
  const int ntiles = (P.Lz + PF_T - 1) / PF_T;
  const float* kc = (wave == 0 || wave == 2) ? P.lp : P.hp;
  const float kb0 = kc[0], kb1 = kc[1], kb2 = kc[2], c1 = kc[4], c2 = kc[5];
  float x1 = 0.f, x2 = 0.f;                 // FIR history
  float p1 = 0.f, p2n = 0.f, dp2 = 0.f;     // IIR history (products of zeros)
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(wav + (size_t)ub * P.n), 0, (unsigned)((size_t)rows * P.n * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(filt + (size_t)ub * 2 * P.Lz), 0, (unsigned)((size_t)rows * 2 * P.Lz * 4), 0x00020000);
  const int rsub = lane >> 4, jq = lane & 15;
  // per-lane row constants of the loader (NL rows per tile and lane) and of the storer (NS)
  constexpr int NL = VEC ? 8 : 32, NS = VEC ? 16 : 64;
  int l_n[NL], s_len[NS];
  unsigned l_off[NL], s_off[NS];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int r = VEC ? i * 4 + rsub : i;
    l_n[i] = stage == 0 ? s_n[r] : 0;
    l_off[i] = r < rows ? (unsigned)r * (unsigned)P.n * 4u : OOB;            // OOB + a row offset stays out of range
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int c = VEC ? i * 4 + rsub : i;
    s_len[i] = stage == 5 ? s_L[c >> 1] : 0;
    s_off[i] = (c >> 1) < rows ? (unsigned)c * (unsigned)P.Lz * 4u : OOB;     // no such chain: nothing stored
  }
  int n_min = P.n, L_min = P.L;          // over the block's utterances: tiles inside them need no per-lane masks
  for (int r = 0; r < rows; ++r) {
    n_min = min(n_min, s_n[r]);
    L_min = min(L_min, s_L[r]);
  }
  constexpr int NPRE = VEC ? 8 : 32;
  typedef typename std::conditional<VEC, float4, float>::type pre_t;
  constexpr int PD = 2;                  // tiles in flight in the loader's registers (two tile periods of latency)
  pre_t pre[PD][NPRE];
  auto load_tile = [&](int tile, pre_t* pre) {
    const int src = tile * PF_T + (VEC ? 4 * jq : lane) - P.pad;   // VEC: a multiple of 4, a quad starts inside [0, n) or outside
    const bool interior = tile * PF_T - P.pad >= 0 && tile * PF_T + PF_T - P.pad <= n_min;     // wave-uniform
    if (interior) {
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const unsigned off = l_off[i] + (unsigned)src * 4u;
        if constexpr (VEC) pre[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
        else pre[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, off, 0, 0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const unsigned off = (tile < ntiles && src >= 0 && src < l_n[i]) ? l_off[i] + (unsigned)src * 4u : OOB;
        if constexpr (VEC) {
          float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
          if (src + 1 >= l_n[i]) v.y = 0.f;      // ragged tail inside the row pitch
          if (src + 2 >= l_n[i]) v.z = 0.f;
          if (src + 3 >= l_n[i]) v.w = 0.f;
          pre[i] = v;
        } else {
          pre[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, off, 0, 0));
        }
      }
    }
  };
  if (stage == 0) {
#pragma unroll
    for (int k = 0; k < PD; ++k) load_tile(k, pre[k]);
  }
  auto loader = [&](int tile, float* buf, pre_t* pk) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      if constexpr (VEC) {
        const int r = i * 4 + rsub;
        const float4 v = pk[i];
        *reinterpret_cast<float4*>(buf + (2 * r) * PF_PITCH + 4 * jq) = v;
        *reinterpret_cast<float4*>(buf + (2 * r + 1) * PF_PITCH + 4 * jq) = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
      } else {
        buf[(2 * i) * PF_PITCH + lane] = pk[i];
        buf[(2 * i + 1) * PF_PITCH + lane] = pk[i] * pk[i];
      }
    }
    load_tile(tile + PD, pk);              // the same registers: lands during the next PD tile periods
  };
  for (int it = 0; it < ntiles + PF_RING - 1; ++it) {
    const int tile = it - stage;
    if (stage >= 0 && tile >= 0 && tile < ntiles) {
      float* buf = ring[tile % PF_RING];
      if (stage == 0) {
        switch (tile & (PD - 1)) {          // compile-time register set per case
          case 0: loader(tile, buf, pre[0]); break;
          default: loader(tile, buf, pre[1]); break;
        }
      } else if (stage == 1 || stage == 3) {
        pf_fir_tile(buf + lane * PF_PITCH, kb0, kb1, kb2, x1, x2);
      } else if (stage == 2 || stage == 4) {
        pf_iir_tile(buf + lane * PF_PITCH, c1, c2, p1, p2n, dp2);
      } else {
        typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
        const int t = tile * PF_T + (VEC ? 4 * jq : lane);   // VEC: Lz % 4 == 0, a quad is inside or outside as a whole
        const bool interior = tile * PF_T + PF_T <= L_min;   // wave-uniform (L_min <= L <= Lz)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const unsigned off = (interior || t < P.Lz) ? s_off[i] + (unsigned)t * 4u : OOB;
          if constexpr (VEC) {
            float4 v = *reinterpret_cast<const float4*>(buf + (i * 4 + rsub) * PF_PITCH + 4 * jq);
            if (!interior) {
              if (t + 0 >= s_len[i]) v.x = 0.f;         // zero extension past the chain's own padded length
              if (t + 1 >= s_len[i]) v.y = 0.f;
              if (t + 2 >= s_len[i]) v.z = 0.f;
              if (t + 3 >= s_len[i]) v.w = 0.f;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), frs, off, 0, 0);
          } else {
            float v = buf[i * PF_PITCH + lane];
            if (!interior && t >= s_len[i]) v = 0.f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), frs, off, 0, 0);
          }
        }
      }
    }
    // LDS-only barrier: __syncthreads() also drains vmcnt, i.e. it would wait every tile for the loader's prefetch
    // and for the storer's write acknowledgements
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// ------------------------------------------------------------------------------------------------
// 8192-point complex FFT in LDS, radix-2 decimation in time, 256 threads.  Input already stored in
// bit-reversed order.  tw[k] = exp(-2*pi*i*k/8192), k < 4096 (host table, f64 -> f32).
// ------------------------------------------------------------------------------------------------
constexpr int FFT_THREADS = 1024;
__device__ __forceinline__ int brev13(int j) { return (int)(__brev((unsigned)j) >> (32 - FFT_LOG)); }

// One pass = NST consecutive radix-2 stages (s .. s+NST-1) done in registers on groups of 2^NST elements
// spaced 2^(s-1) apart: the butterflies, their twiddles and their order of operations are exactly those of the
// stage-by-stage loop, so the result is bit-identical to it — only the LDS round trips and barriers between
// the stages of a pass are gone (13 -> 4-5 passes).
template <int NST>
__device__ __forceinline__ void fft8192_pass(float* re, float* im, const float2* __restrict__ tw, int s) {
  constexpr int R = 1 << NST;
  const int h = 1 << (s - 1);
  // FFT_THREADS threads: the FFT kernels run 1024 (16 waves; LDS holds two such blocks per CU, so the waves of
  // a block are what hides the LDS round trips of a pass); what follows the FFT keeps its 256-thread shape
#pragma unroll 2
  for (int k = 0; k < (FFT_N / R + FFT_THREADS - 1) / FFT_THREADS; ++k) {
    const int i = threadIdx.x + FFT_THREADS * k;
    if (FFT_N / R < FFT_THREADS && i >= FFT_N / R) break;
    const int pos = i & (h - 1);
    const int base = ((i >> (s - 1)) << (s - 1 + NST)) + pos;
    float xr[R], xi[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      xr[j] = re[base + j * h];
      xi[j] = im[base + j * h];
    }
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int d = 1 << st;                              // partner distance in group slots at stage s + st
      // staged table (yaapt_stage_twiddles_kernel): stage s + st's 2^(s + st - 1) twiddles tw[p * tstep] are contiguous from
      // entry 2^(s + st - 1) - 1 on, so consecutive lanes (consecutive pos) read consecutive entries instead of a gather at
      // stride tstep (64 cache lines per wave instruction at the middle stages)
      const float2* __restrict__ ts = tw + ((h << st) - 1);
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (j & d) continue;                              // j = the "a" element, j + d = the "c" element
        const int p = pos + (j & (d - 1)) * h;            // index of a inside its half-block of stage s + st
        const float2 w = ts[p];
        const float tr = xr[j + d] * w.x - xi[j + d] * w.y;
        const float ti = xr[j + d] * w.y + xi[j + d] * w.x;
        const float ur = xr[j], ui = xi[j];
        xr[j] = ur + tr;
        xi[j] = ui + ti;
        xr[j + d] = ur - tr;
        xi[j + d] = ui - ti;
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      re[base + j * h] = xr[j];
      im[base + j * h] = xi[j];
    }
  }
  __syncthreads();
}

// stages first .. 13 of the 8192-point radix-2 DIT FFT (input in bit-reversed order).
// (Round 3, measured and not kept: the twiddles of a radix-8 pass in registers, requested one pass ahead — nlfer 268 -> 418 us;
// the last pass computing only the wanted output bins, 7 of 24 butterfly results and 1 of 8 LDS writes — 229 -> 238 us at
// the same occupancy.  What did pay: the per-stage contiguous twiddle table below, 268 -> 229 and 304 -> 222 us.)
__device__ void fft8192_from(float* re, float* im, const float2* __restrict__ tw, int first) {
  int s = first;
  const int lead = (FFT_LOG - first + 1) % 3;
  if (lead == 1) { fft8192_pass<1>(re, im, tw, s); s += 1; }
  if (lead == 2) { fft8192_pass<2>(re, im, tw, s); s += 2; }
  for (; s <= FFT_LOG; s += 3) fft8192_pass<3>(re, im, tw, s);
}

// Zero-padded real input of L <= 8192 >> z samples: in bit-reversed order the samples sit at multiples of 2^z
// and the first z stages only copy each one over its block of 2^z (x + w*0 and x - w*0).  Fills re/im with that
// state and returns the first stage left to do.  get(j) = sample j.
template <typename Get>
__device__ __forceinline__ int fft8192_load_padded(float* re, float* im, int L, Get get) {
  const int z = L <= 1024 ? 3 : (L <= 2048 ? 2 : (L <= 4096 ? 1 : 0));
  for (int i = threadIdx.x; i < FFT_N; i += FFT_THREADS) im[i] = 0.f;
  // Element (j, r) = re[brev13(j) + r], j < 8192 >> z, r < 2^z.  The LDS bank of brev13(j) is set by the TOP 6 - z bits of j
  // (bits 7 .. 12 - z reversed), so a wave that walks 64 consecutive j writes all its lanes to ONE bank (64-way conflict on
  // every store; with 8000 blocks per launch this fill was a third of the FFT kernels' time).  Here a lane is (r, q):
  // r = the low z bits, q = the top 6 - z bits of j; the 7 low bits of j come from (wave, round) — 64 distinct banks per store.
  static_assert(FFT_THREADS == 1024, "16 waves x 8 rounds = the 128 low-bit values of j");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & ((1 << z) - 1), q = lane >> z;
#pragma unroll
  for (int round = 0; round < 8; ++round) {
    const int j = (q << 7) | (wave * 8 + round);
    const float v = j < L ? get(j) : 0.f;
    re[brev13(j) + r] = v;
  }
  __syncthreads();
  return z + 1;
}

// staged twiddles: out[(h - 1) + p] = tw[p * (4096 / h)] for every stage half-size h = 1, 2, ... 4096 and p < h (8191 entries):
// the same table values, laid out so that a stage's twiddles are contiguous
__global__ void __launch_bounds__(256) yaapt_stage_twiddles_kernel(const float2* __restrict__ tw, float2* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // 0 .. 8190
  if (i >= FFT_N - 1) return;
  const int h = 1 << (31 - __clz(i + 1));                // largest power of two <= i + 1
  const int p = i + 1 - h;
  out[i] = tw[p * ((FFT_N / 2) / h)];
}

// nlfer: frame (560 samples) x hann -> FFT -> sum |X[nl_lo:nl_hi]|
__global__ void __launch_bounds__(FFT_THREADS) yaapt_nlfer_kernel(const float* __restrict__ filt, const float* __restrict__ hann,
                                                         const float2* __restrict__ tw, float* __restrict__ e_raw,
                                                         const int* __restrict__ U, const Plan P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* re = lds;
  float* im = lds + FFT_N;
  __shared__ float red[8];
  const int f = blockIdx.x, b = blockIdx.y;
  if (f >= ulen(U, b, 2, P.nframes)) return;
  const float* x = filt + ((size_t)b * 2 + 0) * P.Lz + (size_t)f * P.frame_jump;
  const int first = fft8192_load_padded(re, im, P.frame_size, [&](int j) { return x[j] * hann[j]; });
  fft8192_from(re, im, tw, first);
  if (threadIdx.x >= 256) return;          // whole waves: the reduction keeps its 256-thread order
  float part = 0.f;
  for (int k = P.nl_lo + threadIdx.x; k < P.nl_hi; k += 256) part += hypotf(re[k], im[k]);
  const float tot = block_sum256(part, red);
  if (threadIdx.x == 0) e_raw[(size_t)b * P.nframes + f] = tot;
}

// energy / mean(energy), vuv = energy > nlfer_thresh1   (one block per utterance)
__global__ void __launch_bounds__(256) yaapt_energy_norm_kernel(const float* __restrict__ e_raw, float* __restrict__ energy,
                                                               int* __restrict__ vuv, const int* __restrict__ U, const Plan P) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  const int nf = ulen(U, b, 2, P.nframes);
  const float* e = e_raw + (size_t)b * P.nframes;
  float part = 0.f;
  for (int f = threadIdx.x; f < nf; f += 256) part += e[f];
  const float mean = block_sum256(part, red) / (float)nf;
  for (int f = threadIdx.x; f < P.nframes; f += 256) {
    const float v = f < nf ? e[f] / mean : 0.f;
    energy[(size_t)b * P.nframes + f] = v;
    vuv[(size_t)b * P.nframes + f] = (f < nf && v > P.nlfer_thresh1) ? 1 : 0;
  }
}

// spectral track, per voiced frame: 1120 samples x kaiser, minus mean -> FFT -> |X| -> SHC -> peaks
// cand layout: [b][8][nframes] = pitch[0..3], merit[0..3]
// Two kernels (round 3): the FFT block holds 64 KB of LDS and 16 waves, the SHC / peak search after it is a 256-thread,
// then one-thread, latency-bound tail of about the same duration — kept in one kernel it pinned those 64 KB for twice as
// long, and two such blocks leave no room on a CU for the generator's conv blocks of the other job streams.  The FFT kernel
// now ends at the magnitude window (n_mag <= 1280 floats per frame, through the workspace); the tail runs with 6 KB of LDS.
// Same arithmetic in the same order: the F0 track is unchanged bit for bit.
constexpr int SPEC_MAGP = 5 * 256;      // floats per frame of the magnitude hand-over (the kernels' 5 x 256 window)
__global__ void __launch_bounds__(FFT_THREADS) yaapt_spec_kernel(const float* __restrict__ filt, const float* __restrict__ kaiser,
                                                        const float2* __restrict__ tw, const int* __restrict__ vuv,
                                                        float* __restrict__ magbuf, const int* __restrict__ U, const Plan P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* re = lds;
  float* im = lds + FFT_N;
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const int f = blockIdx.x, b = blockIdx.y;
  const int nf = P.nframes;                       // row stride
  if (f >= ulen(U, b, 2, nf)) return;
  if (!vuv[(size_t)b * nf + f]) return;           // unvoiced: the tail kernel writes the defaults
  const float* x = filt + ((size_t)b * 2 + 1) * P.Lz + (size_t)f * P.frame_jump;
  // windowed slice and its mean (nframe_size <= 5*256)
  float part = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int j = tid + 256 * k;
    if (tid < 256 && j < P.nframe_size) part += x[j] * kaiser[j];
  }
  const float mean = block_sum256(part, red) / (float)P.nframe_size;
  const int first = fft8192_load_padded(re, im, P.nframe_size, [&](int j) { return x[j] * kaiser[j] - mean; });
  const int n_mag = P.min_shc * (P.nharm + 1) + (P.max_shc - P.min_shc) * (P.nharm + 1) + P.wl;  // exclusive bound
  fft8192_from(re, im, tw, first);
  if (tid >= 256) return;
  // magnitude[i] = i < half_wl ? 0 : |X[i - half_wl]|
  float* mo = magbuf + ((size_t)b * nf + f) * SPEC_MAGP;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = tid + 256 * k;
    float v = 0.f;
    if (i < n_mag && i >= P.half_wl) v = hypotf(re[i - P.half_wl], im[i - P.half_wl]);
    if (i < n_mag) mo[i] = v;
  }
}

__global__ void __launch_bounds__(256) yaapt_spec_peaks_kernel(const float* __restrict__ magbuf, const int* __restrict__ vuv,
                                                               float* __restrict__ cand, const int* __restrict__ U, const Plan P) {
  __shared__ float mag[SPEC_MAGP];
  __shared__ float red[8];
  __shared__ float s_shc[256];
  __shared__ unsigned char s_flag[256];
  const int tid = threadIdx.x;
  const int f = blockIdx.x, b = blockIdx.y;
  const int nf = P.nframes;                       // row stride
  if (f >= ulen(U, b, 2, nf)) return;
  float* cp = cand + (size_t)b * 8 * nf;
  float* cm = cp + 4 * (size_t)nf;
  if (!vuv[(size_t)b * nf + f]) {
    if (tid < MAXP) { cp[(size_t)tid * nf + f] = 0.f; cm[(size_t)tid * nf + f] = 1.f; }
    return;
  }
  const int n_mag = P.min_shc * (P.nharm + 1) + (P.max_shc - P.min_shc) * (P.nharm + 1) + P.wl;  // exclusive bound
  const float* mi = magbuf + ((size_t)b * nf + f) * SPEC_MAGP;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = tid + 256 * k;
    if (i < n_mag) mag[i] = mi[i];
  }
  __syncthreads();
  // SHC[min_shc-1 + r] = sum_w prod_h mag[min_shc*(h+1) + r*(h+1) + w]
  const int rows = P.max_shc - P.min_shc + 1;
  float shc = 0.f;
  if (tid >= P.min_shc - 1 && tid < P.min_shc - 1 + rows) {
    const int r = tid - (P.min_shc - 1);
    float s = 0.f;
    for (int w = 0; w < P.wl; ++w) {
      float pr = mag[P.min_shc + r + w];
      for (int h = 1; h <= P.nharm; ++h) pr = pr * mag[(P.min_shc + r) * (h + 1) + w];
      s += pr;
    }
    shc = s;
  }
  // ---- peaks() on the 256-vector (yaapt.py:383-497) ----
  const int lo = P.pk_min_lag, hi = P.pk_max_lag, c = P.pk_center;
  const bool in_rng = tid >= lo && tid <= hi;
  const float mx = block_max256(in_rng ? shc : -INFINITY, red);
  if (mx > 1e-14f) shc = shc / mx;
  s_shc[tid] = shc;
  const float avg = block_sum256(in_rng ? shc : 0.f, red) / (float)(hi - lo + 1);
  bool ispk = false;
  if (tid >= lo + c + 1 && tid <= hi - c) {
    const float v = s_shc[tid];
    if (v > s_shc[tid - 1] && v > s_shc[tid + 1] && v > P.shc_thresh2 * avg) {
      ispk = true;  // argmax(data[n-c : n+c+1]) == c  <=>  strictly above everything before, >= everything after
      for (int i = tid - c; i < tid; ++i) ispk = ispk && (s_shc[i] < v);
      for (int i = tid + 1; i <= tid + c; ++i) ispk = ispk && (s_shc[i] <= v);
    }
  }
  s_flag[tid] = ispk ? 1 : 0;
  __syncthreads();
  if (tid != 0) return;
  float pit[MAXP], mer[MAXP];
  for (int i = 0; i < MAXP; ++i) { pit[i] = 0.f; mer[i] = 1.f; }
  bool done = avg > P.inv_shc_thresh1;
  if (!done) {
    // collect peaks in ascending lag; keep the MAXP largest merits, earlier index first on ties
    float sp[MAXP], sm[MAXP];
    int np = 0, total = 0;
    float maxm = 0.f;
    for (int n = lo + c + 1; n <= hi - c; ++n) {
      if (!s_flag[n]) continue;
      const float m = s_shc[n];
      const float pv = (float)n * P.delta;
      ++total;
      maxm = total == 1 ? m : fmaxf(maxm, m);
      int pos = np;
      while (pos > 0 && sm[pos - 1] < m) --pos;   // stable descending insertion
      if (pos < MAXP) {
        const int last = np < MAXP ? np : MAXP - 1;
        for (int i = last; i > pos; --i) { sp[i] = sp[i - 1]; sm[i] = sm[i - 1]; }
        sp[pos] = pv;
        sm[pos] = m;
        if (np < MAXP) ++np;
      }
    }
    if (total == 0) maxm = 0.f;
    if (maxm / avg < P.shc_thresh1) {
      done = true;
    } else if (np > 0) {
      for (int i = 0; i < MAXP; ++i) { pit[i] = i < np ? sp[i] : 0.f; mer[i] = i < np ? sm[i] : 0.f; }
      int k = np;
      if (pit[0] > P.f0_double) { k = k + 1 < MAXP ? k + 1 : MAXP; pit[k - 1] = pit[0] / 2.0f; mer[k - 1] = P.merit_extra; }
      if (pit[0] < P.f0_half) { k = k + 1 < MAXP ? k + 1 : MAXP; pit[k - 1] = pit[0] * 2.0f; mer[k - 1] = P.merit_extra; }
      for (int i = k; i < MAXP; ++i) { pit[i] = pit[0]; mer[i] = mer[0]; }
    }
  }
  for (int i = 0; i < MAXP; ++i) { cp[(size_t)i * nf + f] = pit[i]; cm[(size_t)i * nf + f] = mer[i]; }
}

// ------------------------------------------------------------------------------------------------
// per-utterance helpers: ONE wave per utterance, data in LDS, all 64 lanes work (frames are spread
// over lanes; only the Viterbi recursion itself is sequential in time, and it runs its C x C
// transitions on C*C lanes).  Callers separate phases with __syncthreads() (a one-wave block).
// ------------------------------------------------------------------------------------------------
__device__ float median_small(float* v, int k) {  // k odd, <= 7: middle order statistic
  for (int i = 1; i < k; ++i) {
    const float x = v[i];
    int j = i - 1;
    while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; }
    v[j + 1] = x;
  }
  return v[k / 2];
}
__device__ void medfilt_par(const float* in, float* out, int n, int k) {  // zero-padded sliding median
  const int pad = k / 2;
  for (int i = threadIdx.x; i < n; i += 64) {
    float w[7];
    for (int j = 0; j < 7; ++j) {
      const int s = i + j - pad;
      w[j] = (j < k && s >= 0 && s < n) ? in[s] : 0.f;
    }
    out[i] = median_small(w, k);
  }
}
__device__ double wsum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ float mean_par(const float* v, int n) {   // f64 accumulation, rounded once
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)v[i];
  return (float)(wsum_d(s) / (double)n);   // n == 0 -> NaN like torch.mean of an empty tensor
}
__device__ float std_unbiased_par(const float* v, int n) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += (double)v[i];
  const double m = wsum_d(s) / (double)n;
  double q = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) { const double d = (double)v[i] - m; q += d * d; }
  return (float)sqrt(wsum_d(q) / (double)(n - 1));  // n == 1 -> NaN
}

// path1 (yaapt.py:530-570) on one wave.  local [C][T] (row stride ls), trans(i, j, t) supplied by
// a functor.  aux[i][j] = PCOST[j] + trans[i][j][t]; K[i] = LAST argmin_j; CCOST[i] = PCOST[K[i]] +
// trans[K[i]][i][t] + local[i][t]; p_small = LAST argmin_i CCOST.  Back-trace P[t] = PRED[P[t+1]][t+1].
// Lane (i*C + j) evaluates transition (i, j); lane l < C carries PCOST[l].
// torch.argmin treats NaN as the smallest value and returns the FIRST one; the reference takes it on the flipped row
// ("last minimum"), so a NaN cost beats every number and the LAST NaN of a row wins.  NaN costs are real: an utterance
// whose median-filtered best track is all zero (noise: `rand` inputs) has mean_pitch = mean(empty) = NaN in
// dynamic() (yaapt.py:326), and every voiced-to-voiced transition cost is NaN.
__device__ __forceinline__ bool path1_takes(float cand, float best) {
  return (cand != cand) || (!(best != best) && cand <= best);
}
template <int C, class TransFn>
__device__ void path1_wave(const float* local, int ls, int T, TransFn trans, unsigned char* pred,
                           unsigned char* path_out) {
  const int lane = threadIdx.x;
  const bool act = lane < C * C;
  const int i = act ? lane / C : 0;
  const int j = act ? lane % C : 0;
  float pc = lane < C ? local[lane * ls] : 0.f;
  for (int t = 1; t < T; ++t) {
    const float pcj = __shfl(pc, j, 64);
    const float v = pcj + trans(i, j, t);
    int K = 0;
    float bv = __shfl(v, i * C, 64);
#pragma unroll
    for (int jj = 1; jj < C; ++jj) {
      const float vv = __shfl(v, i * C + jj, 64);
      if (path1_takes(vv, bv)) { bv = vv; K = jj; }   // last minimum wins
    }
    const float pcK = __shfl(pc, K, 64);
    const float cc = (pcK + trans(K, i, t)) + local[i * ls + t];
    if (act && j == 0) pred[(size_t)i * T + t] = (unsigned char)K;
    pc = __shfl(cc, (lane * C) & 63, 64);   // lane l < C <- CCOST[l] (held by lane l*C)
  }
  __syncthreads();
  // p_small[T-1] = last argmin_i PCOST (0 when T == 1); every lane computes it from the shuffles
  int p = 0;
  {
    float jv = __shfl(pc, 0, 64);
#pragma unroll
    for (int ii = 1; ii < C; ++ii) {
      const float cv = __shfl(pc, ii, 64);
      if (path1_takes(cv, jv)) { jv = cv; p = ii; }
    }
    if (T <= 1) p = 0;
  }
  if (lane == 0) {
    path_out[T - 1] = (unsigned char)p;
    for (int t = T - 2; t >= 0; --t) {
      p = pred[(size_t)p * T + (t + 1)];
      path_out[t] = (unsigned char)p;
    }
  }
  __syncthreads();
}

// stable compaction of the frames selected by `flag(f)` (ascending frame order): returns the count,
// writes the selected frame indices to idx[]
template <class Pred>
__device__ int compact_frames(int nf, Pred flag, short* idx) {
  int base = 0;
  for (int f0 = 0; f0 < nf; f0 += 64) {
    const int f = f0 + threadIdx.x;
    const bool on = f < nf && flag(f);
    const unsigned long long mask = __ballot(on);
    if (on) idx[base + __popcll(mask & ((1ull << threadIdx.x) - 1ull))] = (short)f;
    base += __popcll(mask);
  }
  return base;
}

// spec_track after the per-frame candidates: selection, smoothing, Viterbi, interpolation.
// LDS (floats): vcp[4*nf] vcm[4*nf] a[nf] b[nf] spec[nf]; shorts: vidx[nf], index[nf]; bytes pred[4*nf] path[nf]
__global__ void __launch_bounds__(64) yaapt_spec_post_kernel(const float* __restrict__ cand, float* __restrict__ spec_out,
                                                            float* __restrict__ scal, int* __restrict__ status,
                                                            const int* __restrict__ U, const Plan P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x;
  const int nfs = P.nframes;                      // row stride of the per-frame arrays in global memory
  const int nf = ulen(U, b, 2, nfs);              // frames of this utterance (also the LDS row pitch)
  const int lane = threadIdx.x;
  float* vcp = lds;
  float* vcm = vcp + 4 * (size_t)nf;
  float* ta = vcm + 4 * (size_t)nf;
  float* tb = ta + nf;
  float* spec = tb + nf;
  short* vidx = (short*)(spec + nf);
  short* index = vidx + nf;
  unsigned char* pred = (unsigned char*)(index + nf);
  unsigned char* path = pred + 4 * (size_t)nf;
  const float* cp = cand + (size_t)b * 8 * nfs;
  const float* cm = cp + 4 * (size_t)nfs;
  for (int f = lane; f < nf; f += 64) spec[f] = cp[f];
  const int nv = compact_frames(nf, [&](int f) { return cp[f] > 0.f; }, vidx);
  __syncthreads();
  for (int i = lane; i < nv; i += 64) {
    const int f = vidx[i];
    for (int c = 0; c < 4; ++c) { vcp[c * nf + i] = cp[(size_t)c * nfs + f]; vcm[c * nf + i] = cm[(size_t)c * nfs + f]; }
  }
  __syncthreads();
  float pitch_avg, pitch_std;
  if (nv == 0) {
    // the reference fails here (medfilt of an empty tensor, yaapt.py:257 -> :54-69); report it
    if (lane == 0) status[b] = 1;
    pitch_avg = 150.f;
    pitch_std = NAN;
  } else {
    const float avg_v = mean_par(vcp, nv);
    const float std_v = std_unbiased_par(vcp, nv);
    const float ref = 0.8f * avg_v;
    for (int i = lane; i < nv; i += 64) {
      int bi = 0;
      float bv = fabsf(vcp[i] - ref) * (3.f - vcm[i]);
      for (int c = 1; c < 4; ++c) {
        const float d = fabsf(vcp[c * nf + i] - ref) * (3.f - vcm[c * nf + i]);
        if (d < bv) { bv = d; bi = c; }   // first minimum
      }
      index[i] = (short)bi;
      ta[i] = vcp[bi * nf + i];
    }
    __syncthreads();
    const int mk = P.median_value - 2 > 1 ? P.median_value - 2 : 1;
    medfilt_par(ta, tb, nv, mk);
    __syncthreads();
    for (int i = lane; i < nv; i += 64) vcp[index[i] * nf + i] = tb[i];
    __syncthreads();
    const float k1 = (P.dp5_k1 * std_v) / avg_v;
    if (nv > 2) {
      // dynamic5 (yaapt.py:506-523): local = 1 - merit (in place), trans = k1*(0.05*d + d*d), d = |p_j(t) - p_i(t-1)|/f0_min
      for (int c = 0; c < 4; ++c)
        for (int i = lane; i < nv; i += 64) vcm[c * nf + i] = 1.f - vcm[c * nf + i];
      __syncthreads();
      const float f0min = P.f0_min;
      auto tr = [&](int i, int j, int t) {
        const float d = fabsf(vcp[j * nf + t] - vcp[i * nf + (t - 1)]) / f0min;
        return k1 * (0.05f * d + d * d);
      };
      path1_wave<4>(vcm, nf, nv, tr, pred, path);
      for (int t = lane; t < nv; t += 64) tb[t] = vcp[path[t] * nf + t];
      __syncthreads();
      medfilt_par(tb, ta, nv, mk);
    } else {
      for (int i = lane; i < nv; i += 64) ta[i] = 150.f;
    }
    __syncthreads();
    pitch_avg = mean_par(ta, nv);
    pitch_std = tmax(std_unbiased_par(ta, nv), pitch_avg * P.spec_pitch_min_std);
    for (int i = lane; i < nv; i += 64) spec[vidx[i]] = ta[i];
  }
  __syncthreads();
  if (lane == 0) {
    if (spec[0] < pitch_avg / 2.f) spec[0] = pitch_avg;
    if (spec[nf - 1] < pitch_avg / 2.f) spec[nf - 1] = pitch_avg;
  }
  __syncthreads();
  // F.interpolate(non-zero values, size=nf, mode='linear', align_corners=False)
  const int nz = compact_frames(nf, [&](int f) { return spec[f] != 0.f; }, vidx);
  __syncthreads();
  for (int i = lane; i < nz; i += 64) tb[i] = spec[vidx[i]];
  __syncthreads();
  if (nz == nf) {
    for (int f = lane; f < nf; f += 64) ta[f] = tb[f];
  } else {
    const float scale = (float)nz / (float)nf;
    for (int f = lane; f < nf; f += 64) {
      // torch's CPU kernel contracts both expressions into fused multiply-adds (checked bit for
      // bit against F.interpolate): src = fma(scale, f + 0.5, -0.5), out = fma(l0, x0, l1*x1)
      float src = fmaf(scale, (float)f + 0.5f, -0.5f);
      if (src < 0.f) src = 0.f;
      int i0 = (int)floorf(src);
      if (i0 > nz - 1) i0 = nz - 1;
      float l1 = src - (float)i0;
      l1 = fminf(fmaxf(l1, 0.f), 1.f);
      const int i1 = i0 + (i0 < nz - 1 ? 1 : 0);
      const float l0 = 1.f - l1;
      ta[f] = fmaf(l0, tb[i0], l1 * tb[i1]);
    }
  }
  __syncthreads();
  if (lane == 0 && nf >= 4) { ta[0] = ta[2]; ta[1] = ta[3]; }
  __syncthreads();
  float* out = spec_out + (size_t)b * nfs;
  for (int f = lane; f < nf; f += 64) out[f] = ta[f];
  if (lane == 0) scal[b * 4 + 0] = pitch_std;
}

// frame means of time_track's in-place subtraction on overlapping views: frame k's first `ov`
// samples already carry frame k-1's mean, so  mean_k = (sum_{j<ov} (x_j - mean_{k-1}) + sum_{j>=ov} x_j) / n.
// One block per (utterance, signal): all four waves first reduce the recurrence-free tails of every
// frame (LDS), then wave 0 walks the frames with only the `ov`-sample head on the critical path
// (next frame's head prefetched).
__global__ void __launch_bounds__(256) yaapt_frame_means_kernel(const float* __restrict__ filt, float* __restrict__ fmean,
                                                               const int* __restrict__ U, const Plan P) {
  __shared__ float s_tail[2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x, sig = blockIdx.y;
  const float* x = filt + ((size_t)b * 2 + sig) * P.Lz;
  float* m = fmean + ((size_t)b * 2 + sig) * P.nframes;
  const int ov = P.tda_len - P.frame_jump;
  const int T = ulen(U, b, 3, P.tda_nframes);
  for (int k = wave; k < T; k += 4) {
    float part = 0.f;
    for (int j = ov + lane; j < P.tda_len; j += 64) part += x[(size_t)k * P.frame_jump + j];
    part = wsum(part);
    if (lane == 0) s_tail[k] = part;
  }
  __syncthreads();
  if (wave != 0) return;
  auto head = [&](int k, int j) { return (k < T && j < ov) ? x[(size_t)k * P.frame_jump + j] : 0.f; };
  float h0 = head(0, lane), h1 = head(0, lane + 64);
  float prev = 0.f;
  for (int k = 0; k < T; ++k) {
    const float n0 = head(k + 1, lane), n1 = head(k + 1, lane + 64);   // prefetch
    float v0 = h0, v1 = h1;
    if (k > 0) {
      if (lane < ov) v0 = v0 - prev;
      if (lane + 64 < ov) v1 = v1 - prev;
    }
    const float mean = (wsum(v0 + v1) + s_tail[k]) / (float)P.tda_len;
    if (lane == 0) m[k] = mean;
    prev = mean;
    h0 = n0;
    h1 = n1;
  }
}

// NCCF candidate of one (utterance, signal, frame)
__global__ void __launch_bounds__(256) yaapt_nccf_kernel(const float* __restrict__ filt, const float* __restrict__ fmean,
                                                        const float* __restrict__ spec, const float* __restrict__ scal,
                                                        float* __restrict__ tp, float* __restrict__ tm,
                                                        int* __restrict__ status, const int* __restrict__ U, const Plan P) {
  __shared__ float d[1024];
  __shared__ float phi[1024];
  __shared__ float red[8];
  __shared__ int s_first;
  const int tid = threadIdx.x;
  const int k = blockIdx.x, sig = blockIdx.y, b = blockIdx.z;
  const int nf = P.nframes;                       // row stride
  if (k >= ulen(U, b, 2, nf)) return;
  const size_t oidx = ((size_t)b * 2 + sig) * nf + k;
  const float sp = spec[(size_t)b * nf + k];
  const float pstd = scal[b * 4 + 0];
  const float fthr = 5.0f * pstd;
  const float rlo = tmax(sp - 2.0f * pstd, P.f0_min);
  const float rhi = tmin(sp + 2.0f * pstd, P.f0_max);
  const float fa = floorf(P.fs / rhi), fb = floorf(P.fs / rlo);
  float pitch = 0.f, merit = 0.f;
  const int n = P.tda_len;
  if (!(fa != fa) && !(fb != fb)) {
    const int lag_min = (int)fa - P.nccf_center;
    const int lag_max = (int)fb + P.nccf_center;
    const int N = n - lag_max;
    if (N <= 0 || lag_min < 1) {
      if (tid == 0) status[b] = 2;  // the reference asserts N > 0 (yaapt.py:586)
    } else {
      const float* x = filt + ((size_t)b * 2 + sig) * P.Lz + (size_t)k * P.frame_jump;
      const float* fm = fmean + ((size_t)b * 2 + sig) * nf;
      const float mk = fm[k];
      const float mprev = k > 0 ? fm[k - 1] : 0.f;
      const int ov = P.tda_len - P.frame_jump;
      for (int j = tid; j < n; j += 256) {
        float v = x[j];
        if (k > 0 && j < ov) v = v - mprev;
        d[j] = v - mk;
        phi[j] = 0.f;
      }
      __syncthreads();
      float part = 0.f;
      for (int j = tid; j < N; j += 256) part += d[j] * d[j];
      const float pw = block_sum256(part, red);
      const int nl = lag_max - lag_min;
      for (int l = tid; l < nl; l += 256) {
        const float* row = d + lag_min + l;
        float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
        int j = 0;
        for (; j + 4 <= N; j += 4) {
          n0 = fmaf(row[j], d[j], n0); q0 = fmaf(row[j], row[j], q0);
          n1 = fmaf(row[j + 1], d[j + 1], n1); q1 = fmaf(row[j + 1], row[j + 1], q1);
          n2 = fmaf(row[j + 2], d[j + 2], n2); q2 = fmaf(row[j + 2], row[j + 2], q2);
          n3 = fmaf(row[j + 3], d[j + 3], n3); q3 = fmaf(row[j + 3], row[j + 3], q3);
        }
        for (; j < N; ++j) { n0 = fmaf(row[j], d[j], n0); q0 = fmaf(row[j], row[j], q0); }
        const float nume = (n0 + n1) + (n2 + n3);
        const float den = (q0 + q1) + (q2 + q3);
        phi[lag_min + l] = nume / sqrtf(den * pw + 0.0f);
      }
      __syncthreads();
      // cmp_rate: first index with phi[n] > phi[n-1], > phi[n+1], > nccf_thresh1 inside [lag_min+c, lag_max-c]
      const int c = P.nccf_center;
      int first = 1 << 30;
      float amax = 0.f;  // phi is zero outside the lag window
      for (int i = tid; i < n; i += 256) {
        const float v = phi[i];
        amax = fmaxf(amax, v);   // NaN entries are skipped by fmaxf, like they never win a '>' test
        if (i >= lag_min + c && i <= lag_max - c && v > phi[i - 1] && v > phi[i + 1] && v > P.nccf_thresh1)
          first = min(first, i);
      }
      amax = block_max256(amax, red);
      first = wmin_i(first);
      if (tid == 0) s_first = 1 << 30;
      __syncthreads();
      if ((tid & 63) == 0) atomicMin(&s_first, first);
      __syncthreads();
      first = s_first;
      if (first < (1 << 30)) {
        bool take = amax > P.nccf_thresh2;
        if (!take) {
          const float v = phi[first];
          take = true;  // argmax(phi[first-c : first+c+1]) == c
          for (int i = first - c; i < first; ++i) take = take && (phi[i] < v);
          for (int i = first + 1; i <= first + c; ++i) take = take && (phi[i] <= v);
        }
        if (take) {
          pitch = (float)((double)P.fs / (double)(first + 1));
          merit = phi[first];
          if (merit > 1.0f) merit = merit / merit;
        }
      }
    }
  }
  if (tid == 0) {
    const float diff = fabsf(pitch - sp);
    const float m1 = diff < fthr ? 1.f : 0.f;
    const float match = (1.f - diff / fthr) * m1;
    tp[oidx] = pitch;
    tm[oidx] = (P.merit_boost1 * merit) * match;
  }
}

// refine + dynamic + path1 -> final pitch.  One wave per utterance; frames spread over lanes.
// LDS floats: rp[6*nf] rm[6*nf] ta[nf] tb[nf]; bytes pred[6*nf] path[nf]
__global__ void __launch_bounds__(64) yaapt_refine_dp_kernel(const float* __restrict__ tp, const float* __restrict__ tm,
                                                            const float* __restrict__ spec, const float* __restrict__ energy,
                                                            const int* __restrict__ vuv, float* __restrict__ f0,
                                                            const int* __restrict__ U, const Plan P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int nfs = P.nframes;                      // row stride in global memory
  const int nf = ulen(U, b, 2, nfs);              // frames of this utterance (also the LDS row pitch)
  float* rp = lds;
  float* rm = rp + NC * (size_t)nf;
  float* ta = rm + NC * (size_t)nf;
  float* en = ta + nf;
  unsigned char* pred = (unsigned char*)(en + nf);
  unsigned char* path = pred + NC * (size_t)nf;
  const float* tp1 = tp + ((size_t)b * 2 + 0) * nfs;
  const float* tp2 = tp + ((size_t)b * 2 + 1) * nfs;
  const float* tm1 = tm + ((size_t)b * 2 + 0) * nfs;
  const float* tm2 = tm + ((size_t)b * 2 + 1) * nfs;
  const float* sp = spec + (size_t)b * nfs;
  const int* vv = vuv + (size_t)b * nfs;
  const int nt = ulen(U, b, 3, P.tda_nframes);
  // concatenate the two tracks (rows 0 and 3 carry the single NCCF candidate), order by merit
  // descending with a stable sort (torch CPU argsort keeps equal keys in index order)
  for (int k = lane; k < nf; k += 64) {
    en[k] = energy[(size_t)b * nfs + k];
    float p[NC] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, m[NC] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k < nt) { p[0] = tp1[k]; m[0] = tm1[k]; p[3] = tp2[k]; m[3] = tm2[k]; }
    int ord[NC];
    for (int i = 0; i < NC; ++i) {
      int pos = i;
      // key = -m ascending, NaN last; insertion keeps earlier rows first on ties
      while (pos > 0) {
        const float a = m[ord[pos - 1]], c = m[i];
        const bool a_nan = a != a, c_nan = c != c;
        const bool c_before_a = c_nan ? false : (a_nan ? true : (c > a));
        if (!c_before_a) break;
        ord[pos] = ord[pos - 1];
        --pos;
      }
      ord[pos] = i;
    }
    // values: flip(sort ascending) ; NaN (sorted last ascending) comes first after the flip
    float ms[NC];
    {
      float t[NC];
      for (int i = 0; i < NC; ++i) t[i] = m[i];
      for (int i = 1; i < NC; ++i) {
        const float x = t[i];
        int j = i - 1;
        while (j >= 0 && ((t[j] != t[j]) ? !(x != x) : (x == x && t[j] > x))) { t[j + 1] = t[j]; --j; }
        t[j + 1] = x;
      }
      for (int i = 0; i < NC; ++i) ms[i] = t[NC - 1 - i];
    }
    for (int i = 0; i < NC; ++i) { rp[i * nf + k] = p[ord[i]]; rm[i * nf + k] = ms[i]; }
  }
  __syncthreads();
  // best_pitch = medfilt(time_pitch[0], median_value) * vuv
  medfilt_par(rp, ta, nf, P.median_value);
  __syncthreads();
  const float th2 = P.nlfer_thresh2;
  double acc = 0.0;
  int cnt = 0;
  for (int k = lane; k < nf; k += 64) {
    const float best = ta[k] * (vv[k] ? 1.f : 0.f);
    ta[k] = best;
    if (best > 0.f) { acc += (double)best; ++cnt; }
    const float e = en[k];
    const float tp0 = rp[k];
    const bool i1 = e <= th2;
    const bool i2 = (e > th2) && (tp0 > 0.f);
    const bool i3 = (e > th2) && (tp0 <= 0.f);
    bool mm[NC];
    for (int r = 0; r < NC; ++r) mm[r] = (r >= 1 && r <= NC - 2) && (rp[r * nf + k] == 0.f) && i2;
    if (i1) for (int r = 0; r < NC; ++r) { rp[r * nf + k] = 0.f; rm[r * nf + k] = P.merit_pivot; }
    if (i2) { rp[(NC - 1) * nf + k] = 0.f; rm[(NC - 1) * nf + k] = 1.0f - rm[k]; }
    for (int r = 0; r < NC; ++r) if (mm[r]) rm[r * nf + k] = 0.f;
    if (i3) {
      rp[k] = sp[k];
      rm[k] = tmin(1.f, e / 2.0f);
      for (int r = 1; r < NC; ++r) { rp[r * nf + k] = 0.f; rm[r * nf + k] = 1.0f - rm[k]; }
    }
    rp[(NC - 2) * nf + k] = best;
    if (best > 0.f) rm[(NC - 2) * nf + k] = rm[k];
    else rm[(NC - 2) * nf + k] = 1.0f - tmin(1.f, e / 2.0f);
    rp[(NC - 3) * nf + k] = sp[k];
    rm[(NC - 3) * nf + k] = e / 5.0f;
    for (int r = 0; r < NC; ++r) rm[r * nf + k] = 1.f - rm[r * nf + k];   // local cost of dynamic()
  }
  // dynamic (yaapt.py:321-370)
  acc = wsum_d(acc);
  cnt = (int)wsum((float)cnt);
  const float mean_pitch = (float)(acc / (double)cnt);
  __syncthreads();
  const float w1 = P.dp_w1, w2 = P.dp_w2, w3 = P.dp_w3, w4 = P.dp_w4;
  auto tr = [&](int i, int j, int t) {
    const float p1 = rp[j * nf + t], p2 = rp[i * nf + (t - 1)];
    float v = 1.f;
    if (p1 > 0.f && p2 > 0.f) v = w1 * (fabsf(p1 - p2) / mean_pitch);
    else if ((p1 == 0.f && p2 > 0.f) || (p1 > 0.f && p2 == 0.f)) v = w2 * (1.f - tmin(1.f, fabsf(en[t - 1] - en[t])));
    else if (p1 == 0.f && p2 == 0.f) v = w3;
    return v / w4;
  };
  path1_wave<NC>(rm, nf, nf, tr, pred, path);
  float* out = f0 + (size_t)b * nfs;
  for (int k = lane; k < nf; k += 64) out[k] = rp[path[k] * nf + k];
  for (int k = nf + lane; k < nfs; k += 64) out[k] = 0.f;     // shorter utterance of a ragged batch: zero-padded track
}

}  // namespace sat

using namespace sat;

static size_t yaapt_ws_floats(const Plan& P, int B) {
  const size_t nf = P.nframes;
  return (size_t)B * (2 * (size_t)P.Lz + 3 * nf /*e_raw, energy, vuv*/ + 8 * nf /*cand*/ + nf /*spec*/ + 4 /*scal*/ +
                      2 * nf /*fmean*/ + 4 * nf /*tp, tm*/ + (size_t)SPEC_MAGP * nf /*magnitude windows*/) + 2 * FFT_N /*staged twiddles*/ + 8 + 64;
}

extern "C" size_t sat_yaapt_workspace_bytes(const sat_yaapt_plan* plan, int B) {
  if (!plan || B <= 0) return 0;
  return align_up(yaapt_ws_floats(*plan, B) * sizeof(float), 256);
}

static int yaapt_run(const sat_yaapt_plan* plan, const float* wav, const int32_t* U, float* f0, int32_t* status,
                     const float* hann, const float* kaiser, const float* twiddle, void* workspace,
                     size_t workspace_bytes, int B, void* stream) {
  SAT_REQUIRE(plan && wav && f0 && status && hann && kaiser && twiddle && workspace, "yaapt: null pointer");
  const Plan& P = *plan;
  SAT_REQUIRE(B > 0 && P.n > 0, "yaapt: empty batch");
  SAT_REQUIRE(P.nfft == FFT_N, "yaapt: fft_length %d not supported (8192 only)", P.nfft);
  SAT_REQUIRE(P.maxpeaks == MAXP && P.maxcands * 2 == NC, "yaapt: shc_maxpeaks/nccf_maxcands must be 4/3");
  SAT_REQUIRE(P.nharm >= 1 && P.nharm <= 7, "yaapt: shc_numharms out of range");
  // reference asserts 15 < frame_size < 2048 (yaapt.py:885-886)
  SAT_REQUIRE(P.frame_size > 15, "Frame length value %d is too short.", P.frame_size);
  SAT_REQUIRE(P.frame_size < 2048, "Frame length value %d exceeds the limit.", P.frame_size);
  SAT_REQUIRE(P.nframe_size <= 5 * 256 && P.frame_size <= FFT_N, "yaapt: frame_length too long for the spectral kernel");
  SAT_REQUIRE(P.max_shc <= 256 && P.pk_max_lag + 1 < 256 && P.pk_min_lag + P.pk_center + 1 >= 1 &&
                  P.pk_min_lag - 0 >= 0, "yaapt: SHC range out of the 256-bin kernel window");
  SAT_REQUIRE(P.min_shc * (P.nharm + 1) + (P.max_shc - P.min_shc) * (P.nharm + 1) + P.wl <= 5 * 256,
              "yaapt: SHC harmonics exceed the spectral kernel's magnitude window");
  SAT_REQUIRE(P.tda_len <= 1024 && P.tda_len > P.frame_jump && P.tda_len - P.frame_jump <= 128,
              "yaapt: tda_frame_length / frame_space combination not supported");
  SAT_REQUIRE(P.median_value >= 1 && P.median_value <= 7 && (P.median_value & 1), "yaapt: median_value must be odd <= 7");
  SAT_REQUIRE(P.nframes >= 4 && P.nframes <= 2048, "yaapt: %d frames not supported (4..2048, i.e. up to ~40 s)", P.nframes);
  SAT_REQUIRE(P.tda_nframes == P.nframes, "yaapt: tda frame count differs from the analysis frame count");
  SAT_REQUIRE_WORKSPACE(workspace_bytes >= sat_yaapt_workspace_bytes(plan, B), "yaapt: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t nf = P.nframes;
  float* w = (float*)workspace;
  float* filt = w;            w += (size_t)B * 2 * P.Lz;
  float* e_raw = w;           w += (size_t)B * nf;
  float* energy = w;          w += (size_t)B * nf;
  int* vuv = (int*)w;         w += (size_t)B * nf;
  float* cand = w;            w += (size_t)B * 8 * nf;
  float* spec = w;            w += (size_t)B * nf;
  float* scal = w;            w += (size_t)B * 4;
  float* fmean = w;           w += (size_t)B * 2 * nf;
  float* tp = w;              w += (size_t)B * 2 * nf;
  float* tm = w;              w += (size_t)B * 2 * nf;
  float* magbuf = w;          w += (size_t)B * nf * SPEC_MAGP;
  w += (4 - ((uintptr_t)w / 4) % 4) % 4;                     // 16-byte alignment of the float2 table
  float2* stw = (float2*)w;   w += 2 * FFT_N;
  const float2* tw = (const float2*)twiddle;
  SAT_HIP(hipMemsetAsync(status, 0, sizeof(int32_t) * B, s));

  const bool pf_vec = (P.n % 4 == 0) && (P.pad % 4 == 0) && (P.Lz % 4 == 0) && (((uintptr_t)wav & 15) == 0) && (((uintptr_t)filt & 15) == 0);
  SAT_REQUIRE((size_t)32 * P.n * 4 < ((size_t)1 << 31) && (size_t)64 * P.Lz * 4 < ((size_t)1 << 31), "yaapt: utterance too long for the prefilter's 31-bit row offsets");
  if (pf_vec) hipLaunchKernelGGL(yaapt_prefilter_kernel<true>, dim3((B + 31) / 32), dim3(PF_THREADS), 0, s, wav, filt, U, P, B);
  else hipLaunchKernelGGL(yaapt_prefilter_kernel<false>, dim3((B + 31) / 32), dim3(PF_THREADS), 0, s, wav, filt, U, P, B);
  SAT_LAUNCH_CHECK("yaapt_prefilter_kernel");
  const size_t fft_lds = 2 * FFT_N * sizeof(float);
  static std::atomic<uint64_t> attr_done{0};      // per device
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)yaapt_nlfer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds));
    SAT_HIP(hipFuncSetAttribute((const void*)yaapt_spec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds));
    SAT_HIP(hipFuncSetAttribute((const void*)yaapt_spec_post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SAT_HIP(hipFuncSetAttribute((const void*)yaapt_refine_dp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done_on_device(attr_done, dev);
  }
  hipLaunchKernelGGL(yaapt_stage_twiddles_kernel, dim3(FFT_N / 256), dim3(256), 0, s, tw, stw);
  SAT_LAUNCH_CHECK("yaapt_stage_twiddles_kernel");
  hipLaunchKernelGGL(yaapt_nlfer_kernel, dim3(P.nframes, B), dim3(FFT_THREADS), fft_lds, s, filt, hann, stw, e_raw, U, P);
  SAT_LAUNCH_CHECK("yaapt_nlfer_kernel");
  hipLaunchKernelGGL(yaapt_energy_norm_kernel, dim3(B), dim3(256), 0, s, e_raw, energy, vuv, U, P);
  SAT_LAUNCH_CHECK("yaapt_energy_norm_kernel");
  hipLaunchKernelGGL(yaapt_spec_kernel, dim3(P.nframes, B), dim3(FFT_THREADS), fft_lds, s, filt, kaiser, stw, vuv, magbuf, U, P);
  SAT_LAUNCH_CHECK("yaapt_spec_kernel");
  hipLaunchKernelGGL(yaapt_spec_peaks_kernel, dim3(P.nframes, B), dim3(256), 0, s, magbuf, vuv, cand, U, P);
  SAT_LAUNCH_CHECK("yaapt_spec_peaks_kernel");
  const size_t post_lds = (size_t)nf * (4 + 4 + 3) * sizeof(float) + nf * 2 * sizeof(short) + nf * 5;
  hipLaunchKernelGGL(yaapt_spec_post_kernel, dim3(B), dim3(64), post_lds, s, cand, spec, scal, status, U, P);
  SAT_LAUNCH_CHECK("yaapt_spec_post_kernel");
  hipLaunchKernelGGL(yaapt_frame_means_kernel, dim3(B, 2), dim3(256), 0, s, filt, fmean, U, P);
  SAT_LAUNCH_CHECK("yaapt_frame_means_kernel");
  hipLaunchKernelGGL(yaapt_nccf_kernel, dim3(P.nframes, 2, B), dim3(256), 0, s, filt, fmean, spec, scal, tp, tm, status, U, P);
  SAT_LAUNCH_CHECK("yaapt_nccf_kernel");
  const size_t dp_lds = (size_t)nf * (2 * NC + 2) * sizeof(float) + nf * (NC + 1);
  hipLaunchKernelGGL(yaapt_refine_dp_kernel, dim3(B), dim3(64), dp_lds, s, tp, tm, spec, energy, vuv, f0, U, P);
  SAT_LAUNCH_CHECK("yaapt_refine_dp_kernel");
  return SAT_OK;
}

extern "C" int sat_yaapt_f32(const sat_yaapt_plan* plan, const float* wav, float* f0, int32_t* status,
                             const float* hann, const float* kaiser, const float* twiddle, void* workspace,
                             size_t workspace_bytes, int B, void* stream) {
  return yaapt_run(plan, wav, nullptr, f0, status, hann, kaiser, twiddle, workspace, workspace_bytes, B, stream);
}

extern "C" int sat_yaapt_ragged_f32(const sat_yaapt_plan* plan, const float* wav, const int32_t* utt_dims, float* f0,
                                    int32_t* status, const float* hann, const float* kaiser, const float* twiddle,
                                    void* workspace, size_t workspace_bytes, int B, void* stream) {
  SAT_REQUIRE(utt_dims, "yaapt_ragged: null utt_dims");
  return yaapt_run(plan, wav, utt_dims, f0, status, hann, kaiser, twiddle, workspace, workspace_bytes, B, stream);
}
