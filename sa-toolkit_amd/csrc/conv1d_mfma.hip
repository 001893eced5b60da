// Fused conv1d as an implicit GEMM on the gfx950 f32 matrix cores.
//
//   GEMM view:  M = output rows (C_out*up, per group), N = time positions, K = C_in/g * ksize
//   MFMA:       v_mfma_f32_32x32x2_f32 — A[i][k] from lane (i = l&31, k = l>>5),
//               B[k][j] from lane (k = l>>5, j = l&31), D: col = l&31, row = (r&3)+8(r>>2)+4(l>>5)
//               (cdna_hip_programming.md §3).  k = a PAIR of input channels at one tap.
//   B operand:  the input tile [16 channels][tile width + halo] staged in LDS; a fragment read is
//               32 consecutive floats of one channel row per half-wave -> conflict-free ds_read_b32.
//   A operand:  packed weights w[g][ci][tap][co] (co fastest) read straight from L2: one coalesced
//               128-B row per half-wave; all blocks share the same few MB of weights.
//   Epilogue:   bias, residual / bypass, folded BatchNorm, ReLU, MRF accumulation, polyphase store.
//
// Replaces (reference): torch Conv1d/ConvTranspose1d calls of hifigan/archi.py:77-91 and
// hifigan/nn.py:179-186; unfold+matmul/addmm of chain/nn.py:267-292 + BatchNorm/ReLU :338-347.
#include "common.h"
#include "conv_common.h"
#include <cstring>

namespace sat {

// sat_conv_set_option("k1_gemm", v): 1x1 convs on split planes through 0 = the conv tile, 1 = conv1d_f16x3_k1_kernel,
// 2 = gemm_f16x3_ring_kernel where its 256-column tiles fit (else 1), 3 = the 16x16x32-shape ring kernel where its
// epilogue subset covers the call (else 2)
static int g_k1_gemm = 3;
// sat_conv_set_option("lean3" | "lean7" | "lean11", v): 3- / 7- / 11-tap convs on split planes (no folded BatchNorm) through
// the three-blocks-per-CU form of the tile (1, conv_lean.hip) or the two-block form (0)
static int g_lean3 = 1, g_lean7 = 1, g_lean11 = 1;
static int g_trim_halo = 1; // the fused pair kernels load only the columns of their staged window that conv1 reads (window + halo of the dilation)
static int g_half_tile7 = 1;   // conv_pre on 64 x 128 tiles when the 64 x 256 ones are at most one per CU
static int g_pair32s = 1;   // the 3-tap fused step at C = 32 on the streaming kernel (pair32s.hip)

template <int MT, int NT, int WM, int WN, int KS, bool STRIDE1, int XWI>
__global__ void __launch_bounds__(256, 2) conv1d_mfma_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CO_B = 32 * MT * WM;
  constexpr int T_B = 32 * NT * WN;
  static_assert(WM * WN == 4, "4 waves per block");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = wave / WM;
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  const int b = blockIdx.z;
  const int g = blockIdx.y / p.co_tiles_g;
  const int cot = blockIdx.y - g * p.co_tiles_g;
  const int co_w = cot * CO_B + wm * (32 * MT);  // first row of this wave inside the group
  const int q_b = blockIdx.x * T_B;              // first output position of the block
  const int q_w = q_b + wn * (32 * NT);
  const bool wave_active = co_w < p.rows_g;  // co_pad is a multiple of 64 >= rows_g, so the wave's rows stay inside the packed weights

  const int ks = KS > 0 ? KS : p.ksize;
  const int st = STRIDE1 ? 1 : p.stride;
  const int XW = XWI > 0 ? 64 * XWI : p.xw;  // LDS row pitch (columns past the tile are never read)
  const int xi0 = q_b * st - p.pad_left;

  const float* __restrict__ xg = p.x + (long long)b * p.x_bs + (long long)(g * p.cin_g) * p.x_cs;
  // weights of this group behind one buffer descriptor; per-lane part of the A-fragment address in
  // ONE 32-bit VGPR (channel half lh, row l31), everything else (chunk, pair, tap, m) scalar/immediate
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.w + (long long)g * p.w_gs), 0, (unsigned)(p.w_gs * 4), 0x00020000);
  const int w_ci_bytes = ks * p.co_pad * 4;            // bytes between input channels
  const int a_voff = lh * w_ci_bytes + (co_w + l31) * 4;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;


  // XWI > 0 (compile-time trip counts): the input rows of chunk c + 1 are requested into registers BEFORE the MFMA phase of chunk c and
  // stored to LDS behind it (round 6): a launch with few blocks per CU — one or two utterances on the exact-f32 kernels, what the
  // near-tie guard of the VQ runs — no longer pays one exposed global round trip per 16-channel chunk.  Same chunks in the same order:
  // the same bits.
  constexpr int XWR = XWI > 0 ? XWI : 1;
  float stg[CI_CHUNK / 4][XWR];
  auto request_rows = [&](int c0) __attribute__((always_inline)) {
    // every global load of the chunk is issued before the first LDS store, so the loads overlap each other
    const int voff = (xi0 + lane) * 4;  // byte offset inside the row; negative / past-the-end -> 0 by the range check
#pragma unroll
    for (int rr = 0; rr < CI_CHUNK / 4; ++rr) {
      const int ci = c0 + wave + rr * 4;
      // one buffer descriptor per (utterance, channel) row: the hardware range check supplies the
      // conv's zero padding on both sides; descriptor inputs are wave-uniform (readfirstlane, T20)
      const unsigned long long a = (unsigned long long)(xg + (long long)ci * p.x_cs);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
      const unsigned nbytes = __builtin_amdgcn_readfirstlane(ci < p.cin_g ? (unsigned)p.T_in * 4u : 0u);
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(((unsigned long long)hi << 32) | lo), 0, nbytes, 0x00020000);
#pragma unroll
      for (int it = 0; it < XWR; ++it)
        stg[rr][it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 256 * it, 0, 0));
    }
  };
  if constexpr (XWI > 0) request_rows(0);

  for (int c0 = 0; c0 < p.cin_pad; c0 += CI_CHUNK) {
    __syncthreads();
    // ---- stage the input tile: 16 channels x XW columns, pre-activation fused ----
    if constexpr (XWI > 0) {
#pragma unroll
      for (int rr = 0; rr < CI_CHUNK / 4; ++rr) {
        float* dst = lds + (wave + rr * 4) * XW;
#pragma unroll
        for (int it = 0; it < XWI; ++it) {
          const int col = lane + 64 * it;
          float v = stg[rr][it];
          if (p.in_lrelu) v = v > 0.f ? v : v * p.in_slope;
          dst[col] = v;
        }
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < CI_CHUNK / 4; ++rr) {
        const int r = wave + rr * 4;
        const int ci = c0 + r;
        const bool cok = ci < p.cin_g;
        const float* __restrict__ src = xg + (long long)ci * p.x_cs;
        float* dst = lds + r * XW;
        for (int col = lane; col < XW; col += 64) {
          const int xi = xi0 + col;
          float v = 0.f;
          if (cok && xi >= 0 && xi < p.T_in) {
            v = src[xi];
            if (p.in_lrelu) v = v > 0.f ? v : v * p.in_slope;
          }
          dst[col] = v;
        }
      }
    }
    __syncthreads();
    // (the next chunk's rows are requested BEHIND this chunk's first weight loads: vmcnt retires in issue order, so a wait for a
    // weight fragment also waits for every older load)
    auto request_next = [&]() __attribute__((always_inline)) {
      if constexpr (XWI > 0) {
        if (c0 + CI_CHUNK < p.cin_pad) request_rows(c0 + CI_CHUNK);      // in flight during this chunk's MFMA phase
      }
    };
    if (!wave_active) {
      request_next();
      continue;
    }

    const int wc_soff = c0 * w_ci_bytes;   // scalar byte offset of this chunk's first channel
    const float* xrow = lds + lh * XW + (wn * (32 * NT) + l31) * st;
#define SAT_LOAD_A(pair, tap, m) \
  __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32( \
      wrsrc, a_voff + (m) * 128, wc_soff + (2 * (pair)) * w_ci_bytes + (tap) * p.co_pad * 4, 0))

    if constexpr (KS > 0 && KS * MT <= 4) {
      // few taps and row tiles (the 1x1 GEMM shapes of the encoders, TDNNF linearA / linearB): ALL weight fragments of the chunk are
      // requested up front (8 channel pairs x KS x MT <= 32 registers) — one round trip per chunk instead of one per channel pair,
      // which is what a launch of a few blocks per CU waits for (round 6; the same products in the same order)
      float a_all[CI_CHUNK / 2][KS][MT];
#pragma unroll
      for (int pr = 0; pr < CI_CHUNK / 2; ++pr)
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m) a_all[pr][t][m] = SAT_LOAD_A(pr, t, m);
      request_next();
#pragma unroll
      for (int pr = 0; pr < CI_CHUNK / 2; ++pr) {
        const float* xp = xrow + (2 * pr) * XW;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          float bf[NT];
          const float* xt = xp + t * p.dil;
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = xt[n * 32 * st];
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_all[pr][t][m], bf[n], acc[m][n], 0, 0, 0);
        }
      }
    } else if constexpr (KS > 0) {
      float a_cur[KS][MT];
#pragma unroll
      for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[t][m] = SAT_LOAD_A(0, t, m);
      request_next();
#pragma unroll
      for (int pr = 0; pr < CI_CHUNK / 2; ++pr) {
        float a_nxt[KS][MT];
        if (pr + 1 < CI_CHUNK / 2) {
#pragma unroll
          for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) a_nxt[t][m] = SAT_LOAD_A(pr + 1, t, m);
        }
        const float* xp = xrow + (2 * pr) * XW;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          float bf[NT];
          const float* xt = xp + t * p.dil;
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = xt[n * 32 * st];
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t][m], bf[n], acc[m][n], 0, 0, 0);
        }
        if (pr + 1 < CI_CHUNK / 2) {
#pragma unroll
          for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) a_cur[t][m] = a_nxt[t][m];
        }
        // keep the software pipeline one channel pair deep: without this fence the scheduler hoists
        // the weight loads of all 8 unrolled pairs to the top of the chunk and spills
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // runtime tap count (k = 2, 10, 128 ...): taps looped, one k-pair of channels at a time
      request_next();
#pragma unroll 1
      for (int pr = 0; pr < CI_CHUNK / 2; ++pr) {
        const float* xp = xrow + (2 * pr) * XW;
        float a_nx[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a_nx[m] = SAT_LOAD_A(pr, 0, m);
#pragma unroll 2
        for (int t = 0; t < ks; ++t) {
          float a[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = a_nx[m];
          if (t + 1 < ks) {
#pragma unroll
            for (int m = 0; m < MT; ++m) a_nx[m] = SAT_LOAD_A(pr, t + 1, m);
          }
          float bf[NT];
          const float* xt = xp + t * p.dil;
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = xt[n * 32 * st];
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bf[n], acc[m][n], 0, 0, 0);
        }
      }
    }
  }

#undef SAT_LOAD_A
  if (!wave_active) return;

  conv_epilogue<MT, NT>(p, acc, b, g, co_w, q_w, l31, lh);
}

// ------------------------------------------------------------------------------------------------
// Split-f16 variant: every f32 operand is carried as hi + lo f16 (hi = x rounded to f16, lo = x - hi,
// together 22 significand bits) and a product is hi*hi + hi*lo + lo*hi on the f16 matrix cores with
// f32 accumulation (v_mfma_f32_32x32x16_f16: 16 input channels per instruction in 32 cycles, i.e.
// 16x the f32-MFMA rate, x3 instructions).  The dropped lo*lo term and the lo rounding are ~2^-21
// relative per product, two orders of magnitude below the path's 1e-4 RMS parity bar and at the
// level of the f32 re-association noise measured against the CPU oracle.  Used for the generator
// (activations O(1), inside f16 range); the TDNNF/VQ path stays on the exact-f32 kernel because its
// arg-min decisions are sensitive to 1e-6 perturbations.
//
// At these MFMA rates the weight fragments can no longer be streamed from L2 per wave (that alone
// would need ~43 B/clk/CU): a block = 4 waves side by side in time over ONE 32*MT-row weight tile, and
// per 16-channel chunk both operands are staged in LDS:
//   W  [tap][hi|lo][channel half][row]      x 16 B (8 f16)  — a straight copy of the packed weights
//   X  [hi|lo][channel half][column]        x 16 B          — lrelu + hi/lo split while staging
// so every A and B fragment is one conflict-free ds_read_b128 and the main loop issues no global
// loads.  Weights arrive packed as w16[g][chunk][tap][hi|lo][half][co_pad][8] f16.
// ------------------------------------------------------------------------------------------------

template <int MT, int NT, int KS, int XWI>
__global__ void __launch_bounds__(256, 2) conv1d_f16x3_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int CO_B = 32 * MT;
  constexpr int T_B = 128 * NT;          // 4 waves x NT x 32 positions
  constexpr int XWP = 64 * XWI;
  constexpr int NIT = (XWI + 1) / 2;
  constexpr int W_UNITS = KS * 4 * CO_B;  // 16-byte units of one weight chunk
  constexpr int W_IT = (W_UNITS + 255) / 256;
  uint4* ldsx = lds4;                     // [4][XWP]
  uint4* ldsw = lds4 + 4 * XWP;           // [KS][2][2][CO_B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  const int b = blockIdx.z;
  const int g = blockIdx.y / p.co_tiles_g;
  const int cot = blockIdx.y - g * p.co_tiles_g;
  const int co_w = cot * CO_B;            // every wave of the block works on the same rows
  const int q_b = blockIdx.x * T_B;
  const int q_w = q_b + wave * (32 * NT);
  const int xi0 = q_b - p.pad_left;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + (long long)b * p.x_bs + (long long)(g * p.cin_g) * p.x_cs), 0,
      (unsigned)((long long)p.cin_g * p.x_cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.w + (long long)g * p.w_gs), 0, (unsigned)p.w_gs, 0x00020000);
  const int sh = __builtin_amdgcn_readfirstlane(wave & 1);   // channel half staged by this wave
  const int sp = __builtin_amdgcn_readfirstlane(wave >> 1);  // parity of the 64-column slices it stages
  const int x_row_bytes = (int)p.x_cs * 4;
  const int seg_bytes = p.co_pad * 16;                        // one (tap, part, half) segment of all rows

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  int chunk = 0;
  for (int c0 = 0; c0 < p.cin_pad; c0 += CI_CHUNK, ++chunk) {
    __syncthreads();
    {
      // ---- issue every load of the chunk first: weights (copy) then the input tile ----
      uint4 wst[W_IT];
#pragma unroll
      for (int i = 0; i < W_IT; ++i) {
        const int u = tid + 256 * i;
        const int seg = u / CO_B, r = u % CO_B;
        wst[i] = make_uint4(0, 0, 0, 0);
        if (u < W_UNITS)
          wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 wrs, (co_w + r) * 16 + seg * seg_bytes, chunk * (KS * 4) * seg_bytes, 0));
      }
      {
        float stg[NIT][8];
#pragma unroll
        for (int ii = 0; ii < NIT; ++ii) {
          const int it = 2 * ii + sp;
          const int xi = xi0 + lane + 64 * it;
          const unsigned voff = (it < XWI && xi >= 0 && xi < p.T_in) ? (unsigned)(xi * 4) : 0x80000000u;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ci = c0 + 8 * sh + j;
            float v = 0.f;
            if (ci < p.cin_g) v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, voff, ci * x_row_bytes, 0));
            stg[ii][j] = v;
          }
        }
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
          const int u = tid + 256 * i;
          if (u < W_UNITS) ldsw[u] = wst[i];
        }
#pragma unroll
        for (int ii = 0; ii < NIT; ++ii) {
          const int it = 2 * ii + sp;
          if (it < XWI) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float x0 = stg[ii][2 * j], x1 = stg[ii][2 * j + 1];
              if (p.in_lrelu) {
                x0 = lrelu_max(x0, p.in_slope);
                x1 = lrelu_max(x1, p.in_slope);
              }
              const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
              const auto l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
              hi[j] = __builtin_bit_cast(unsigned, h);
              lo[j] = __builtin_bit_cast(unsigned, l);
            }
            const int col = lane + 64 * it;
            ldsx[(0 * 2 + sh) * XWP + col] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            ldsx[(1 * 2 + sh) * XWP + col] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
          }
        }
      }
    }
    __syncthreads();

    const uint4* xb = ldsx + lh * XWP + wave * (32 * NT) + l31;
    const uint4* wb = ldsw + lh * CO_B + l31;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      h8 a_hi[MT], a_lo[MT], b_hi[NT], b_lo[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        a_hi[m] = __builtin_bit_cast(h8, wb[(t * 4 + 0) * CO_B + m * 32]);
        a_lo[m] = __builtin_bit_cast(h8, wb[(t * 4 + 2) * CO_B + m * 32]);
      }
      const uint4* xt = xb + t * p.dil;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        b_hi[n] = __builtin_bit_cast(h8, xt[n * 32]);
        b_lo[n] = __builtin_bit_cast(h8, xt[2 * XWP + n * 32]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[m], b_hi[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[m], b_lo[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[m], b_hi[n], acc[m][n], 0, 0, 0);
        }
    }
  }
  conv_epilogue<MT, NT>(p, acc, b, g, co_w, q_w, l31, lh);

}

// ------------------------------------------------------------------------------------------------
// Split-plane input (x16): the B operand arrives as ready hi|lo f16 planes, staging is a 16-byte copy.
// Software-pipelined over K-chunks through registers: chunk c+1's global loads (weights + planes,
// W_IT + XWI 16-byte loads per lane) are issued right after chunk c's tile is published in LDS and
// stay in flight during its MFMA phase; during the LAST chunk's MFMA phase the same registers
// prefetch the residual the epilogue adds.  Measured with s_memtime stamps (tools/stamp_conv.hip) the
// un-pipelined form spent 1.7-2.7k cycles per chunk waiting on loads and 22-32k cycles in the epilogue
// against a 4.2k-cycle MFMA phase (k = 11).
// ------------------------------------------------------------------------------------------------
// ---- polyphase (transposed-conv) output straight to split planes.  The accumulator rows are (channel, phase)
// pairs and a lane holds a column q of them, but a plane unit is 8 CHANNELS at one output time t = q*up + phase:
// the tile goes through LDS once (f32, pitch +1 against bank conflicts between phases) and is read back in
// plane order — 16-byte stores, consecutive lanes on consecutive times.  Replaces the f32 store of the
// upsampled tensor and the separate split pass (12 bytes per element of HBM traffic).
// Needs 32*MT % (8*up) == 0 (a block's rows are whole 8-channel groups).
template <int MT, int NT>
__device__ __forceinline__ void polyphase_planes_epilogue(const ConvArgs& p, f32x16 (&acc)[MT][NT], float* tile, int b,
                                                          int co_w, int q_b, int wave, int l31, int lh) {
  constexpr int CO_B = 32 * MT, T_B = 128 * NT, PITCH = T_B + 1;
  const int up = p.up;
  __syncthreads();                                   // every wave is done with the operand tiles
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int grow = co_w + row;
        const float bias = (p.bias && grow < p.rows_g) ? p.bias[grow / up] : 0.f;
        tile[row * PITCH + wave * (32 * NT) + n * 32 + l31] = __builtin_fmaf(acc[m][n][r], p.w_descale, bias);
      }
  __syncthreads();
  const int T_out = p.T_q * up;
  uint4* yb = (uint4*)p.y16 + (long long)b * (p.cout_g / 4) * T_out;     // cout_g * T_out * 4 bytes per utterance
  const int n_tt = T_B * up;                            // output times of the block
  const int n_cg = CO_B / (8 * up);                     // 8-channel groups of the block
  for (int it = threadIdx.x; it < n_cg * n_tt; it += 256) {
    const int cg = it / n_tt, tt = it - cg * n_tt;
    const int ql = tt / up, ph = tt - ql * up;
    const int t = q_b * up + tt;
    const int c0 = co_w / up + cg * 8;                  // first channel of the group
    if (t >= T_out || c0 >= p.cout_g) continue;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0 = tile[((cg * 8 + 2 * j) * up + ph) * PITCH + ql];
      float x1 = tile[((cg * 8 + 2 * j + 1) * up + ph) * PITCH + ql];
      x0 = lrelu_max(x0, p.y16_slope);
      x1 = lrelu_max(x1, p.y16_slope);
      const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
      const auto l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
      hi[j] = __builtin_bit_cast(unsigned, h);
      lo[j] = __builtin_bit_cast(unsigned, l);
    }
    const long long u = ((long long)((c0 >> 4) * 4 + ((c0 >> 3) & 1))) * T_out + t;
    yb[u] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    yb[u + 2LL * T_out] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// S: 16-channel sub-chunks per pipeline stage.  With few taps the matrix work of a 16-channel chunk (1152
// cycles at 3 taps) is dwarfed by the ~2800 cycles of barriers, LDS stores and load issue around it: S = 2
// halves the number of stages.
template <int MT, int NT, int KS, int XWI, bool F8, int S>
__global__ void __launch_bounds__(256, 2) conv1d_f16x3_planes_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int CO_B = 32 * MT;
  constexpr int T_B = 128 * NT;
  constexpr int XWP = 64 * XWI;
  constexpr int W_UNITS = KS * 4 * CO_B;
  constexpr int W_IT = (W_UNITS + 255) / 256;
  uint4* ldsx = lds4;                     // [S][4][XWP]
  uint4* ldsw = lds4 + S * 4 * XWP;       // [S][KS][2][2][CO_B]
  static_assert(!F8 || S == 1, "e4m3 cross terms: one sub-chunk per stage");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int lh = lane >> 5;
  const int b = blockIdx.z;
  const int co_w = blockIdx.y * CO_B;
  const int q_b = blockIdx.x * T_B;
  const int q_w = q_b + wave * (32 * NT);
  const int xi0 = q_b - p.pad_left;
  const int nch = p.cin_pad / CI_CHUNK / S;   // pipeline stages

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x16 + (long long)b * p.cin_g * p.T_in * 4), 0, (unsigned)(p.cin_g * p.T_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;
  const int pl = __builtin_amdgcn_readfirstlane(wave);    // plane (part*2 + half) this wave copies
  const int x_chunk_bytes = 4 * p.T_in * 16;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  uint4 wst[S][W_IT], xst[S][XWI];
  auto issue_loads = [&](int stage) {
#pragma unroll
    for (int sc = 0; sc < S; ++sc) {
      const int chunk = stage * S + sc;
#pragma unroll
      for (int i = 0; i < W_IT; ++i) {
        const int u = tid + 256 * i;
        const int seg = u / CO_B, r = u % CO_B;
        wst[sc][i] = make_uint4(0, 0, 0, 0);
        if (u < W_UNITS)
          wst[sc][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      wrs, (co_w + r) * 16 + seg * seg_bytes, chunk * (KS * 4) * seg_bytes, 0));
      }
#pragma unroll
      for (int it = 0; it < XWI; ++it) {
        const int xi = xi0 + lane + 64 * it;
        const unsigned voff = (xi >= 0 && xi < p.T_in) ? (unsigned)((pl * p.T_in + xi) * 16) : 0x80000000u;
        xst[sc][it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, chunk * x_chunk_bytes, 0));
      }
    }
  };
  auto publish = [&]() {
#pragma unroll
    for (int sc = 0; sc < S; ++sc) {
#pragma unroll
      for (int i = 0; i < W_IT; ++i) {
        const int u = tid + 256 * i;
        if (u < W_UNITS) ldsw[sc * W_UNITS + u] = wst[sc][i];
      }
#pragma unroll
      for (int it = 0; it < XWI; ++it) ldsx[(sc * 4 + pl) * XWP + lane + 64 * it] = xst[sc][it];
    }
  };
  const uint4* xb0 = ldsx + lh * XWP + wave * (32 * NT) + l31;
  const uint4* wb0 = ldsw + lh * CO_B + l31;
  // F8: per-lane E8M0 scale bytes of the cross-term MFMA (lanes 0-31: W_lo8 . x_hi8, lanes 32-63: W_hi8 . x_lo8)
  const int sc_a = lh ? F8_E_WHI : F8_E_WLO;
  const int sc_b = lh ? F8_E_XLO : F8_E_XHI;
  auto mfma_phase = [&]() {
   if constexpr (!F8) {
    // fragments of tap idx + 1 are read from LDS ahead of tap idx's MFMAs (double-buffered registers; the
    // scheduler would otherwise sink the reads to their first use and every tap would start on an LDS round trip)
    h8 fa[2][MT][2], fb[2][NT][2];
    auto ld = [&](int buf, int sc, int t) {
      const uint4* wb = wb0 + sc * W_UNITS;
      const uint4* xt = xb0 + sc * 4 * XWP + t * p.dil;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        fa[buf][m][0] = __builtin_bit_cast(h8, wb[(t * 4 + 0) * CO_B + m * 32]);
        fa[buf][m][1] = __builtin_bit_cast(h8, wb[(t * 4 + 2) * CO_B + m * 32]);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        fb[buf][n][0] = __builtin_bit_cast(h8, xt[n * 32]);
        fb[buf][n][1] = __builtin_bit_cast(h8, xt[2 * XWP + n * 32]);
      }
    };
    ld(0, 0, 0);
#pragma unroll
    for (int idx = 0; idx < S * KS; ++idx) {
      if (idx + 1 < S * KS) ld((idx + 1) & 1, (idx + 1) / KS, (idx + 1) % KS);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[idx & 1][m][1], fb[idx & 1][n][0], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[idx & 1][m][0], fb[idx & 1][n][1], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[idx & 1][m][0], fb[idx & 1][n][0], acc[m][n], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
   }
#pragma unroll
   for (int sc = 0; sc < S; ++sc) {
    const uint4* xb = xb0 + sc * 4 * XWP;
    const uint4* wb = wb0 + sc * W_UNITS;
    if constexpr (F8) {
      // hi*hi on the f16 MFMA per tap; both cross terms of a PAIR of taps in one block-scaled e4m3
      // MFMA (K = 64 = 2 terms x 2 taps x 16 channels) at twice the f16 rate per K
#pragma unroll
      for (int tp = 0; tp < (KS + 1) / 2; ++tp) {
        const int t0 = 2 * tp, t1 = 2 * tp + 1;
        const bool two = t1 < KS;
        h8 a_h0[MT], a_h1[MT], b_h0[NT], b_h1[NT];
        uint4 a8_0[MT], a8_1[MT], b8_0[NT], b8_1[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          a_h0[m] = __builtin_bit_cast(h8, wb[(t0 * 4 + 0) * CO_B + m * 32]);
          a8_0[m] = wb[(t0 * 4 + 2) * CO_B + m * 32];
          if (two) {
            a_h1[m] = __builtin_bit_cast(h8, wb[(t1 * 4 + 0) * CO_B + m * 32]);
            a8_1[m] = wb[(t1 * 4 + 2) * CO_B + m * 32];
          } else {
            a8_1[m] = make_uint4(0, 0, 0, 0);   // phantom tap: zero weights
          }
        }
        const uint4* x0 = xb + t0 * p.dil;
        const uint4* x1 = xb + t1 * p.dil;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          b_h0[n] = __builtin_bit_cast(h8, x0[n * 32]);
          b8_0[n] = x0[2 * XWP + n * 32];
          if (two) {
            b_h1[n] = __builtin_bit_cast(h8, x1[n * 32]);
            b8_1[n] = x1[2 * XWP + n * 32];
          } else {
            b8_1[n] = b8_0[n];                  // finite bytes against the zero weights
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h0[m], b_h0[n], acc[m][n], 0, 0, 0);
            if (two) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h1[m], b_h1[n], acc[m][n], 0, 0, 0);
            const i32x8 a8 = {(int)a8_0[m].x, (int)a8_0[m].y, (int)a8_0[m].z, (int)a8_0[m].w,
                              (int)a8_1[m].x, (int)a8_1[m].y, (int)a8_1[m].z, (int)a8_1[m].w};
            const i32x8 b8 = {(int)b8_0[n].x, (int)b8_0[n].y, (int)b8_0[n].z, (int)b8_0[n].w,
                              (int)b8_1[n].x, (int)b8_1[n].y, (int)b8_1[n].z, (int)b8_1[n].w};
            acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[m][n], 0, 0, 0, sc_a, 0, sc_b);
          }
      }
    } else {
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        h8 a_hi[MT], a_lo[MT], b_hi[NT], b_lo[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          a_hi[m] = __builtin_bit_cast(h8, wb[(t * 4 + 0) * CO_B + m * 32]);
          a_lo[m] = __builtin_bit_cast(h8, wb[(t * 4 + 2) * CO_B + m * 32]);
        }
        const uint4* xt = xb + t * p.dil;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          b_hi[n] = __builtin_bit_cast(h8, xt[n * 32]);
          b_lo[n] = __builtin_bit_cast(h8, xt[2 * XWP + n * 32]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[m], b_hi[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[m], b_lo[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[m], b_hi[n], acc[m][n], 0, 0, 0);
          }
      }
    }
   }
  };

  int chunk = 0;
  SAT_STAMP_BLOCK(6, __builtin_amdgcn_s_memrealtime());
  SAT_STAMP_BLOCK(7, ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492));
  issue_loads(0);
  for (; chunk < nch - 1; ++chunk) {
    __syncthreads();   // every wave is done reading the previous tile
    SAT_STAMP(0);
    publish();
    SAT_STAMP(2);
    __syncthreads();
    SAT_STAMP(3);
    issue_loads(chunk + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_phase();
    SAT_STAMP(4);
  }
  __syncthreads();
  SAT_STAMP(0);
  publish();
  SAT_STAMP(2);
  __syncthreads();
  SAT_STAMP(3);
  float rpre[MT][NT][16];
  if (p.res || p.res16) epilogue_prefetch_res<MT, NT>(p, rpre, b, 0, co_w, q_w, l31, lh);
  __builtin_amdgcn_sched_barrier(0);
  mfma_phase();
  SAT_STAMP(4);
  if (p.poly_planes) {                                  // wave-uniform: polyphase upsampler writing planes only
    polyphase_planes_epilogue<MT, NT>(p, acc, (float*)lds4, b, co_w, q_b, wave, l31, lh);
    return;
  }
  conv_epilogue<MT, NT, true>(p, acc, b, 0, co_w, q_w, l31, lh, 32, 0x7fffffff, rpre);
  SAT_STAMP(5);
  SAT_STAMP_BLOCK(14, __builtin_amdgcn_s_memrealtime());
}

template <int MT, int NT, int KS>
static int launch_f16x3(const ConvArgs& a, int B, int groups, hipStream_t s) {
  constexpr int CO_B = 32 * MT;
  constexpr int T_B = 128 * NT;
  constexpr int XWI = (T_B + (KS - 1) * 5 + 63) / 64;
  ConvArgs p = a;
#ifdef SAT_STAMPS
  p.dbg = g_stamp_buffer;
  if (g_stamp_variant & 1) p.y16 = nullptr;
  if (g_stamp_variant & 2) p.res16 = nullptr, p.res = nullptr;
#endif
  p.xw = T_B + (p.ksize - 1) * p.dil;
  if (p.xw > 64 * XWI) {
    set_error("conv1d(f16x3): dilation %d too large for the %d-tap kernel", p.dil, p.ksize);
    return SAT_ERR_INVALID;
  }
  p.co_tiles_g = ceil_div(p.rows_g, CO_B);
  size_t lds_bytes = ((size_t)4 * 64 * XWI + (size_t)KS * 4 * CO_B) * 16;
  void (*kern)(const ConvArgs) = conv1d_f16x3_kernel<MT, NT, KS, XWI>;
  if (p.x16) {
    kern = nullptr;
    // split-plane input: the tap counts of the generator (3, 7, 11), of the TDNNF stack (1, 3) and of the wav2vec2
    // feature extractor's polyphase stride-2 convs (2, 1)
    if constexpr (KS == 3 || KS == 7 || KS == 11)
      kern = p.f8 ? conv1d_f16x3_planes_kernel<MT, NT, KS, XWI, true, 1> : conv1d_f16x3_planes_kernel<MT, NT, KS, XWI, false, 1>;
    if constexpr ((KS == 1 || KS == 2 || KS == 3) && MT == 2) {
      // few taps: two 16-channel sub-chunks per pipeline stage
      if (!p.f8 && (p.cin_pad / CI_CHUNK) % 2 == 0) {
        kern = conv1d_f16x3_planes_kernel<MT, NT, KS, XWI, false, 2>;
        lds_bytes *= 2;
      }
    }
    if constexpr (KS == 1 || KS == 2) {
      if (!kern && !p.f8) kern = conv1d_f16x3_planes_kernel<MT, NT, KS, XWI, false, 1>;
    }
    if (!kern) {
      set_error("conv1d(split planes): %d taps not instantiated", KS);
      return SAT_ERR_INVALID;
    }
    if (p.poly_planes) lds_bytes = std::max(lds_bytes, (size_t)CO_B * (T_B + 1) * 4);   // the f32 tile of the transposed epilogue
  }
  if (lds_bytes > 64 * 1024)
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  dim3 grid(ceil_div(p.T_q, T_B), p.co_tiles_g * groups, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("conv1d_f16x3_kernel");
  return SAT_OK;
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution on split planes = a GEMM (TDNNF linearA/B, wav2vec2 Linear layers, ASR output affines).  With
// one tap nothing is shared between columns, so the B operand (activations) never goes through LDS: a lane loads
// its own two 16-byte fragments per 16-channel chunk straight from the planes (coalesced, 512 B per half wave).
// Only the weight tile is shared (4 waves side by side in time): 128 rows x 64 channels per stage, double-buffered
// in LDS (2 x 32 KB), ONE barrier per stage; the next stage's fragments and weights fly in registers during this
// stage's 48 MFMAs per wave.  Block tile = 128 rows x 128 positions (wave: 128 x 32, 64 accumulator registers):
// half the activation traffic per MFMA of the 64-row conv tile.  Blocks are numbered so that the co tiles of one
// (utterance, time tile) land on the same XCD and share the activations through its L2.
// ------------------------------------------------------------------------------------------------
template <int SC>
__global__ void __launch_bounds__(256, 2) conv1d_f16x3_k1_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int CO_B = 128, T_B = 128, MT = 4;
  constexpr int WST = SC * 4 * CO_B;        // 16-byte units of one stage's weight tile: [chunk][part*2 + half][row]
  constexpr int W_IT = WST / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  // block -> (co tile, time tile, utterance): id = xcd + 8 * (co + n_co * (g >> 3)), g = 8 * (g >> 3) + xcd
  const int n_co = p.co_tiles_g, n_tt = p.pp_tiles_t;
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int co_t = __builtin_amdgcn_readfirstlane(rest % n_co);
  const int g = __builtin_amdgcn_readfirstlane((rest / n_co) * 8 + xcd);
  if (g >= p.pp_total) return;
  const int b = __builtin_amdgcn_readfirstlane(g / n_tt);     // (division runs on the vector ALU: keep the results scalar)
  const int co_w = co_t * CO_B;
  const int q_w = (g - b * n_tt) * T_B + wave * 32;
  const int nst = p.cin_pad / CI_CHUNK / SC;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x16 + (long long)b * p.cin_g * p.T_in * 4), 0, (unsigned)(p.cin_g * p.T_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;
  const int x_chunk_bytes = 4 * p.T_in * 16;
  const int xi = q_w + l31 - p.pad_left;
  const bool xok = xi >= 0 && xi < p.T_in;
  const unsigned xo_hi = xok ? (unsigned)(((0 + lh) * p.T_in + xi) * 16) : 0x80000000u;
  const unsigned xo_lo = xok ? (unsigned)(((2 + lh) * p.T_in + xi) * 16) : 0x80000000u;

  f32x16 acc[MT][1];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;

  uint4 wst[W_IT], bst[SC][2], bcur[SC][2];
  auto issue = [&](int st) {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      // unit u = tid + 256 i of the stage tile [chunk][seg][128 rows]: chunk = i / 2 for every thread (a scalar offset)
      const int cl = i >> 1, seg = (tid >> 7) + 2 * (i & 1), r = tid & 127;
      wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (co_w + r) * 16 + seg * seg_bytes,
                                                                                (st * SC + cl) * 4 * seg_bytes, 0));
    }
#pragma unroll
    for (int cl = 0; cl < SC; ++cl) {
      bst[cl][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo_hi, (st * SC + cl) * x_chunk_bytes, 0));
      bst[cl][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo_lo, (st * SC + cl) * x_chunk_bytes, 0));
    }
  };
  issue(0);
  for (int st = 0; st < nst; ++st) {
    uint4* wl = lds4 + (st & 1) * WST;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) wl[tid + 256 * i] = wst[i];
#pragma unroll
    for (int cl = 0; cl < SC; ++cl) {
      bcur[cl][0] = bst[cl][0];
      bcur[cl][1] = bst[cl][1];
    }
    __syncthreads();       // this stage's weights are visible; every wave has left the stage before, whose buffer the next one overwrites
    if (st + 1 < nst) issue(st + 1);
    __builtin_amdgcn_sched_barrier(0);
    const uint4* wb = wl + lh * CO_B + l31;
    // A fragments of chunk cl + 1 are read from LDS ahead of chunk cl's MFMAs (double-buffered registers): no
    // MFMA waits on a read issued just before it
    h8 af[2][MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      af[0][m][0] = __builtin_bit_cast(h8, wb[(0 * 4 + 0) * CO_B + m * 32]);
      af[0][m][1] = __builtin_bit_cast(h8, wb[(0 * 4 + 2) * CO_B + m * 32]);
    }
#pragma unroll
    for (int cl = 0; cl < SC; ++cl) {
      const h8 b_hi = __builtin_bit_cast(h8, bcur[cl][0]);
      const h8 b_lo = __builtin_bit_cast(h8, bcur[cl][1]);
      if (cl + 1 < SC) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          af[(cl + 1) & 1][m][0] = __builtin_bit_cast(h8, wb[((cl + 1) * 4 + 0) * CO_B + m * 32]);
          af[(cl + 1) & 1][m][1] = __builtin_bit_cast(h8, wb[((cl + 1) * 4 + 2) * CO_B + m * 32]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // the scheduler would sink those reads to their first use
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cl & 1][m][1], b_hi, acc[m][0], 0, 0, 0);
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cl & 1][m][0], b_lo, acc[m][0], 0, 0, 0);
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cl & 1][m][0], b_hi, acc[m][0], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  conv_epilogue<MT, 1>(p, acc, b, 0, co_w, q_w, l31, lh);
}

static int launch_f16x3_k1(const ConvArgs& a, int B, hipStream_t s) {
  constexpr int SC = 4;
  ConvArgs p = a;
  p.xw = 128;
  p.co_tiles_g = ceil_div(p.rows_g, 128);
  p.pp_tiles_t = ceil_div(p.T_q, 128);
  p.pp_total = p.pp_tiles_t * B;
  const size_t lds_bytes = (size_t)2 * SC * 4 * 128 * 16;
  auto kern = conv1d_f16x3_k1_kernel<SC>;
  static std::atomic<uint64_t> attr_done{0};      // per device (one static per SC instantiation)
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  dim3 grid(8 * p.co_tiles_g * ceil_div(p.pp_total, 8), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("conv1d_f16x3_k1_kernel");
  return SAT_OK;
}

template <int MT, int NT>
static int launch_f16x3_ks(const ConvArgs& a, int B, int groups, hipStream_t s) {
  switch (a.ksize) {
    case 1: return launch_f16x3<MT, NT, 1>(a, B, groups, s);
    case 2: return launch_f16x3<MT, NT, 2>(a, B, groups, s);
    case 3: return launch_f16x3<MT, NT, 3>(a, B, groups, s);
    case 7: return launch_f16x3<MT, NT, 7>(a, B, groups, s);
    case 11: return launch_f16x3<MT, NT, 11>(a, B, groups, s);
    default:
      set_error("conv1d(f16x3): kernel size %d not instantiated (1, 2, 3, 7, 11)", a.ksize);
      return SAT_ERR_INVALID;
  }
}

// ------------------------------------------------------------------------------------------------
// Fused ResBlock1 step for the thin generator stages (C <= 32), split-f16 arithmetic:
//     out = conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x        (hifigan/nn.py:179-186)
// Unfused, a step moves five stage-sized tensors through HBM (x, t1 out, t1 in, residual, out) and the
// thin stages sit on the HBM roofline; fused, the intermediate t1 lives only in LDS (already split into
// hi|lo f16 planes, ready to be the B operand of conv2) and the step reads x once and writes out once.
// A block produces 224 output positions: conv1 is evaluated on the 256-position window that conv2
// needs (t1 positions outside the utterance are zero, they are conv2's zero padding), 8 sub-tiles of
// 32 over 4 waves; conv2 on 7 sub-tiles.  One 32-row weight tile (C = 16 is padded to 32 rows).
// ------------------------------------------------------------------------------------------------
constexpr int FP_TO = 224;    // output positions per block
constexpr int FP_W1 = 256;    // t1 window: [t0 - 16, t0 + 240)
constexpr int FP_OFF = 16;

template <int KS, int XWI, bool X16IN>
__global__ void __launch_bounds__(256, 2) resblock_pair_f16x3_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int CO_B = 32;
  constexpr int XWP = 64 * XWI;
  constexpr int NIT = (XWI + 1) / 2;
  constexpr int W_UNITS = KS * 4 * CO_B;
  constexpr int W_IT = (W_UNITS + 255) / 256;
  uint4* ldsx = lds4;                       // [4][XWP]            input chunk, hi|lo x half
  uint4* ldsw = lds4 + 4 * XWP;             // [KS][2][2][32]      weight chunk
  uint4* ldst = ldsw + W_UNITS;             // [nchunk][4][FP_W1]  t1, all channels

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int lh = lane >> 5;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * FP_TO;
  const int h2 = (KS - 1) / 2;
  const int xi0 = t0 - FP_OFF - p.pad_left;   // first input position of the staged tile (pad_left = conv1 halo)

  const __amdgpu_buffer_rsrc_t xrs = X16IN
      ? __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x16 + (long long)b * p.cin_g * p.T_in * 4), 0,
                                          (unsigned)(p.cin_g * p.T_in * 4), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.x + (long long)b * p.x_bs), 0, (unsigned)((long long)p.cin_g * p.x_cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (unsigned)p.w_gs, 0x00020000);
  const int sh = __builtin_amdgcn_readfirstlane(wave & 1);
  const int sp = __builtin_amdgcn_readfirstlane(wave >> 1);
  const int x_row_bytes = (int)p.x_cs * 4;
  const int seg_bytes = p.co_pad * 16;
  const int nchunk = p.cin_pad / CI_CHUNK;

  auto stage_w = [&](const __amdgpu_buffer_rsrc_t& rs, int chunk) {
    uint4 wst[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int u = tid + 256 * i;
      wst[i] = make_uint4(0, 0, 0, 0);
      if (u < W_UNITS)
        wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                               rs, (u % CO_B) * 16 + (u / CO_B) * seg_bytes, chunk * (KS * 4) * seg_bytes, 0));
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int u = tid + 256 * i;
      if (u < W_UNITS) ldsw[u] = wst[i];
    }
  };

  // ================= phase 1: t1 = lrelu(conv1(lrelu(x)) + b1) on the 256-position window =================
  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int c0 = chunk * CI_CHUNK;
    __syncthreads();
    if constexpr (X16IN) {
      uint4 xst[XWI];
      const int pl = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
      for (int it = 0; it < XWI; ++it) {
        const int xi = xi0 + lane + 64 * it;
        const unsigned voff = (xi >= 0 && xi < p.T_in) ? (unsigned)((pl * p.T_in + xi) * 16) : 0x80000000u;
        xst[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, chunk * 4 * p.T_in * 16, 0));
      }
      stage_w(w1rs, chunk);
#pragma unroll
      for (int it = 0; it < XWI; ++it) ldsx[pl * XWP + lane + 64 * it] = xst[it];
    } else
    {
      float stg[NIT][8];
#pragma unroll
      for (int ii = 0; ii < NIT; ++ii) {
        const int it = 2 * ii + sp;
        const int xi = xi0 + lane + 64 * it;
        const unsigned voff = (it < XWI && xi >= 0 && xi < p.T_in) ? (unsigned)(xi * 4) : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ci = c0 + 8 * sh + j;
          float v = 0.f;
          if (ci < p.cin_g) v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, voff, ci * x_row_bytes, 0));
          stg[ii][j] = v;
        }
      }
      stage_w(w1rs, chunk);
#pragma unroll
      for (int ii = 0; ii < NIT; ++ii) {
        const int it = 2 * ii + sp;
        if (it < XWI) {
          unsigned hi[4], lo[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x0 = stg[ii][2 * j], x1 = stg[ii][2 * j + 1];
            x0 = lrelu_max(x0, p.in_slope);
            x1 = lrelu_max(x1, p.in_slope);
            const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
            const auto l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
            hi[j] = __builtin_bit_cast(unsigned, h);
            lo[j] = __builtin_bit_cast(unsigned, l);
          }
          const int col = lane + 64 * it;
          ldsx[(0 * 2 + sh) * XWP + col] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
          ldsx[(1 * 2 + sh) * XWP + col] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
      }
    }
    __syncthreads();
    const uint4* wb = ldsw + lh * CO_B + l31;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const h8 a_hi = __builtin_bit_cast(h8, wb[(t * 4 + 0) * CO_B]);
      const h8 a_lo = __builtin_bit_cast(h8, wb[(t * 4 + 2) * CO_B]);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const uint4* xt = ldsx + lh * XWP + (wave + 4 * n) * 32 + l31 + t * p.dil;
        const h8 b_hi = __builtin_bit_cast(h8, xt[0]);
        const h8 b_lo = __builtin_bit_cast(h8, xt[2 * XWP]);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[n], 0, 0, 0);
      }
    }
  }
  // t1 -> LDS as conv2's B operand.  D layout: lane (col = l31, lh) holds rows 8*rg + 4*lh + (r&3), rg = r>>2:
  // four consecutive channels = 8 bytes of the 16-byte unit (chunk' = rg>>1, half' = rg&1) of its column.
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = (wave + 4 * n) * 32 + l31;
    const int pos = t0 - FP_OFF + col;
    const bool inside = pos >= 0 && pos < p.T_in;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      if ((rg >> 1) >= nchunk) continue;   // C = 16: rows 16..31 are padding, no t1 plane for them
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 8 * rg + 4 * lh + k;
        float t = __builtin_fmaf(acc[n][4 * rg + k], p.w_descale1, row < p.rows_g ? p.bias1[row] : 0.f);
        t = lrelu_max(t, p.in_slope);
        v[k] = inside ? t : 0.f;
      }
      const auto h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
      const auto h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
      const auto l01 = split_lo2(h01, v[0], v[1]);
      const auto l23 = split_lo2(h23, v[2], v[3]);
      uint2* dh = (uint2*)(ldst + (((rg >> 1) * 4 + 0 * 2 + (rg & 1)) * FP_W1 + col)) + lh;
      uint2* dl = (uint2*)(ldst + (((rg >> 1) * 4 + 1 * 2 + (rg & 1)) * FP_W1 + col)) + lh;
      *dh = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
      *dl = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    }
  }

  // ================= phase 2: out = conv2(t1) + b2 + x =================
  f32x16 acc2[1][2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[0][n][r] = 0.f;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();
    stage_w(w2rs, chunk);
    __syncthreads();
    const uint4* wb = ldsw + lh * CO_B + l31;
    const uint4* tb = ldst + (chunk * 4 + lh) * FP_W1 + FP_OFF - h2 + l31;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const h8 a_hi = __builtin_bit_cast(h8, wb[(t * 4 + 0) * CO_B]);
      const h8 a_lo = __builtin_bit_cast(h8, wb[(t * 4 + 2) * CO_B]);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (wave + 4 * n < FP_TO / 32) {   // wave-uniform: 7 output sub-tiles over 4 waves
          const uint4* xt = tb + (wave + 4 * n) * 32 + t;
          const h8 b_hi = __builtin_bit_cast(h8, xt[0]);
          const h8 b_lo = __builtin_bit_cast(h8, xt[2 * FP_W1]);
          acc2[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc2[0][n], 0, 0, 0);
          acc2[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc2[0][n], 0, 0, 0);
          acc2[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc2[0][n], 0, 0, 0);
        }
      }
    }
  }
  conv_epilogue<1, 2>(p, acc2, b, 0, 0, t0 + wave * 32, l31, lh, 128, t0 + FP_TO);
}

template <int KS>
static int launch_pair(const ConvArgs& a, int B, hipStream_t s) {
  constexpr int XWI = (FP_W1 + (KS - 1) * 5 + 63) / 64;
  ConvArgs p = a;
  if (FP_W1 + (p.ksize - 1) * p.dil > 64 * XWI) {
    set_error("resblock_pair: dilation %d too large", p.dil);
    return SAT_ERR_INVALID;
  }
  p.co_tiles_g = 1;
  const size_t lds_bytes = ((size_t)4 * 64 * XWI + (size_t)KS * 4 * 32 + (size_t)(p.cin_pad / CI_CHUNK) * 4 * FP_W1) * 16;
  auto kern = p.x16 ? resblock_pair_f16x3_kernel<KS, XWI, true> : resblock_pair_f16x3_kernel<KS, XWI, false>;
  if (lds_bytes > 64 * 1024)
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  dim3 grid(ceil_div(p.T_q, FP_TO), 1, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("resblock_pair_f16x3_kernel");
  return SAT_OK;
}

// ------------------------------------------------------------------------------------------------
// ResBlock1 step for the C = 16 stage on the 16x16x32 MFMA shape: 16 rows = the 16 channels exactly
// (the 32x32 tile pads them to 32 and wastes half the matrix work), K = 32 = a PAIR of taps x 16
// channels.  Split planes in, split planes out.  Persistent blocks: both convs' A fragments stay in
// registers (every wave holds all 16 rows), the input tile and the intermediate t1 live in LDS (36 KB,
// 3-4 blocks per CU), the next tile's input is prefetched into registers under this tile's matrix
// work, and the residual comes from the input tile the block already holds.
//   lanes: A[row l&15][k = 8(l>>4)..], B[k = 8(l>>4)..][col l&15], D col = l&15, rows 4(l>>4) + r
//   k-group g = l>>4: tap (g>>1) of the pair, channel half (g&1) -> the same 16-byte units as before
// ------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int KS>
__global__ void __launch_bounds__(256, KS == 11 ? 2 : 3) resblock_pair16_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int XWI = 5, XWP = 64 * XWI;
  constexpr int NTP = (KS + 1) / 2;         // tap pairs; an odd tap count gets a zero phantom tap
  uint4* ldsx = lds4;                       // [4][XWP]
  uint4* ldst = ldsx + 4 * XWP;             // [4][FP_W1]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j16 = lane & 15;
  const int g = lane >> 4;
  const int gh = g & 1, gt = g >> 1;        // channel half / tap of the pair
  const int h2 = (KS - 1) / 2;
  const unsigned OOB = 0x80000000u;

  const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;
  const int pl = __builtin_amdgcn_readfirstlane(wave);

  // persistent walk: workgroups go round-robin to the 8 XCDs, so XCD x (= blockIdx.x & 7) owns the contiguous tile
  // range [x, x+1) * pp_per_xcd and its pp_nslots blocks take consecutive tiles of it at the same time — neighbouring
  // tiles share their 96-column halo through that XCD's L2.  Weights are staged once per block; the input tile of
  // the next step is loaded into registers while this one computes.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_end = min((xcd + 1) * p.pp_per_xcd, p.pp_total);
  int tile = xcd * p.pp_per_xcd + slot;
  if (tile >= tile_end) return;

  uint4 xst[XWI];
  auto issue_x = [&](int tl) {
    const int ub = __builtin_amdgcn_readfirstlane(tl / p.pp_tiles_t);
    const int xi_0 = (tl - ub * p.pp_tiles_t) * FP_TO - FP_OFF - p.pad_left;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x16 + (long long)ub * 16 * p.T_in * 4), 0, (unsigned)(16 * p.T_in * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < XWI; ++it) {
      const int xi = xi_0 + lane + 64 * it;
      const unsigned voff = (xi >= 0 && xi < p.T_in) ? (unsigned)((pl * p.T_in + xi) * 16) : OOB;
      xst[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, 0, 0));
    }
  };
  issue_x(tile);
  // both convs' A fragments live in registers for the block's whole walk (the 16 rows are all the channels, so every
  // wave holds the same ones): packed units [tap][hi|lo][half][16 rows], the phantom tap of an odd count reads zeros
  h8 w1h[NTP], w1l[NTP], w2h[NTP], w2l[NTP];
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) {
    const int tap = 2 * tp + gt;
    const unsigned vh = tap < KS ? (unsigned)(j16 * 16 + (tap * 4 + 0 + gh) * seg_bytes) : OOB;
    const unsigned vl = tap < KS ? (unsigned)(j16 * 16 + (tap * 4 + 2 + gh) * seg_bytes) : OOB;
    w1h[tp] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(w1rs, vh, 0, 0));
    w1l[tp] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(w1rs, vl, 0, 0));
    w2h[tp] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, vh, 0, 0));
    w2l[tp] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, vl, 0, 0));
  }
  float bias1[4], bias2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bias1[k] = p.bias1[4 * g + k];
    bias2[k] = p.bias[4 * g + k];
  }
  for (;;) {
  const int b = __builtin_amdgcn_readfirstlane(tile / p.pp_tiles_t);     // (division runs on the vector ALU: keep it scalar)
  const int t0 = (tile - b * p.pp_tiles_t) * FP_TO;
  __syncthreads();            // the previous step's readers of the input tile and of t1 are done
#pragma unroll
  for (int it = 0; it < XWI; ++it) ldsx[pl * XWP + lane + 64 * it] = xst[it];
  __syncthreads();
  const int next = tile + p.pp_nslots;
  const bool more = next < tile_end;
  if (more) issue_x(next);
  __builtin_amdgcn_sched_barrier(0);

  // ================= phase 1: t1 = lrelu(conv1(x planes) + b1) on the 256-column window =================
  f32x4v acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) acc[s] = f32x4v{0.f, 0.f, 0.f, 0.f};
  {
    const uint4* xl = ldsx + gh * XWP + wave * 64 + j16;
#pragma unroll
    for (int tp = 0; tp < NTP; ++tp) {
      const int tap = 2 * tp + gt;
      const h8 a_hi = w1h[tp], a_lo = w1l[tp];
      const uint4* xt = xl + tap * p.dil;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const h8 b_hi = __builtin_bit_cast(h8, xt[16 * s]);
        const h8 b_lo = __builtin_bit_cast(h8, xt[2 * XWP + 16 * s]);
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi, acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo, acc[s], 0, 0, 0);
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi, acc[s], 0, 0, 0);
      }
    }
  }
  // t1 -> LDS planes: this lane's four consecutive channels 4g .. 4g+3 = 8 bytes of the unit (half g>>1)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int col = wave * 64 + 16 * s + j16;
    const int pos = t0 - FP_OFF + col;
    const bool inside = pos >= 0 && pos < p.T_in;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = __builtin_fmaf(acc[s][k], p.w_descale1, bias1[k]);
      t = lrelu_max(t, p.in_slope);
      v[k] = inside ? t : 0.f;      // t1 outside the utterance is conv2's zero padding
    }
    const auto h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
    const auto h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    const auto l01 = split_lo2(h01, v[0], v[1]);
    const auto l23 = split_lo2(h23, v[2], v[3]);
    ((uint2*)(ldst + (0 + gt) * FP_W1 + col))[gh] = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    ((uint2*)(ldst + (2 + gt) * FP_W1 + col))[gh] = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
  }
  __syncthreads();

  // ================= phase 2: out = conv2(t1) + b2 + x on 224 columns (14 sub-tiles: 4, 4, 4, 2) =================
  const int ns = wave < 3 ? 4 : 2;
#pragma unroll
  for (int s = 0; s < 4; ++s) acc[s] = f32x4v{0.f, 0.f, 0.f, 0.f};
  {
    const uint4* tl = ldst + gh * FP_W1 + wave * 64 + j16 + FP_OFF - h2;
#pragma unroll
    for (int tp = 0; tp < NTP; ++tp) {
      const int tap = 2 * tp + gt;
      const h8 a_hi = w2h[tp], a_lo = w2l[tp];
      const uint4* xt = tl + tap;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < ns) {     // wave-uniform
          const h8 b_hi = __builtin_bit_cast(h8, xt[16 * s]);
          const h8 b_lo = __builtin_bit_cast(h8, xt[2 * FP_W1 + 16 * s]);
          acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi, acc[s], 0, 0, 0);
          acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo, acc[s], 0, 0, 0);
          acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi, acc[s], 0, 0, 0);
        }
      }
    }
  }

  // ================= epilogue: + b2 + x (from the input tile in LDS), MRF sum, stores =================
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.y + (long long)b * p.y_bs), 0, (unsigned)(16 * p.y_cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.y16 ? (char*)p.y16 + (long long)b * 16 * p.T_q * 4 : (char*)p.y), 0, p.y16 ? (unsigned)(16 * p.T_q * 4) : 0u, 0x00020000);
  const int y_rb = (int)p.y_cs * 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s >= ns) continue;
    const int o = wave * 64 + 16 * s + j16;
    const int q = t0 + o;
    const bool ok = q < p.T_q;
    // residual: the block input at this position, still in the LDS tile (planes of lrelu(x): undo it)
    const int xcol = o + FP_OFF + p.pad_left;
    const uint2 rh = ((const uint2*)(ldsx + (0 + gt) * XWP + xcol))[gh];
    const uint2 rl = ((const uint2*)(ldsx + (2 + gt) * XWP + xcol))[gh];
    float r[4];
    decode_res16(__builtin_bit_cast(float, rh.x), __builtin_bit_cast(float, rh.y), __builtin_bit_cast(float, rl.x),
                 __builtin_bit_cast(float, rl.y), p.res16_inv, r);
    const unsigned yoff = ok ? (unsigned)((4 * g) * y_rb + q * 4) : OOB;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(acc[s][k], p.w_descale, bias2[k]) + r[k];
    if (p.accum) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, yoff + k * y_rb, 0, 0)) + v[k];
    }
    if (p.accum_div != 0.f) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] / p.accum_div;
    }
    if (!p.no_y && !p.no_store) {
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), yrs, yoff + k * y_rb, 0, 0);
    }
    if (p.y16) {
      float u[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = lrelu_max(v[k], p.y16_slope);
      const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
      const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
      const auto l01 = split_lo2(h01, u[0], u[1]);
      const auto l23 = split_lo2(h23, u[2], u[3]);
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      u32x2 hv, lv;
      hv[0] = __builtin_bit_cast(unsigned, h01); hv[1] = __builtin_bit_cast(unsigned, h23);
      lv[0] = __builtin_bit_cast(unsigned, l01); lv[1] = __builtin_bit_cast(unsigned, l23);
      const unsigned off = ok ? (unsigned)(((0 + gt) * p.T_q + q) * 16 + 8 * gh) : OOB;
      __builtin_amdgcn_raw_buffer_store_b64(hv, y16rs, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(lv, y16rs, off, 2 * p.T_q * 16, 0);
    }
  }
  if (!more) break;
  tile = next;
  }
}

template <int KS>
static int launch_pair16(const ConvArgs& a, int B, hipStream_t s) {
  ConvArgs p = a;
  if (FP_W1 + (p.ksize - 1) * p.dil > 320) {
    set_error("resblock_pair: dilation %d too large", p.dil);
    return SAT_ERR_INVALID;
  }
  const size_t lds_bytes = ((size_t)4 * 320 + 4 * FP_W1) * 16;
  auto kern = resblock_pair16_kernel<KS>;
  if (lds_bytes > 64 * 1024)
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  // persistent: as many blocks as fit the chip at once (LDS-limited), a multiple of the 8 XCDs
  const int per_cu = KS == 11 ? 2 : KS == 7 ? 3 : 4;    // register-limited: 16 B x 4 x ceil(KS/2) weight fragments per lane
  p.pp_tiles_t = ceil_div(p.T_q, FP_TO);
  p.pp_total = p.pp_tiles_t * B;
  p.pp_per_xcd = ceil_div(p.pp_total, 8);
  p.pp_nslots = std::max(1, std::min(32 * per_cu, p.pp_per_xcd));
  dim3 grid(8 * p.pp_nslots, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("resblock_pair16_kernel");
  return SAT_OK;
}

// ------------------------------------------------------------------------------------------------
// ResBlock1 step for C = 32 with split planes end to end: the fused pair kernel above with its four
// global-load round trips (x + W1 chunk 0, x + W1 chunk 1, W2 chunk 0, W2 chunk 1) pipelined through
// registers — each is issued before the previous step's MFMA phase — and the residual (the input
// planes again, L2-resident) prefetched under the last one.
// ------------------------------------------------------------------------------------------------
template <int KS>
__global__ void __launch_bounds__(256, 3) resblock_pair32_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int CO_B = 32;
  constexpr int XWI = 5, XWP = 64 * XWI;
  constexpr int W_UNITS = KS * 4 * CO_B;
  constexpr int W_IT = (W_UNITS + 255) / 256;
  // Three blocks per CU (168 VGPRs allow it; the kernel waits on memory 45 % of its cycles at two): the t1 image lies
  // OVER the input chunk, which is dead once conv1's last MFMA phase has read it (one barrier in between), and for 11
  // taps the t1 window is trimmed to the 234 columns conv2 reads — 22.5 KB of weights + 29.5 KB = 52 KB per block.
  constexpr int W1 = KS == 11 ? 236 : FP_W1;      // t1 window [t0 - OFF, t0 - OFF + W1)
  constexpr int OFF = KS == 11 ? 6 : FP_OFF;
  static_assert(OFF >= (KS - 1) / 2 && OFF - (KS - 1) / 2 + FP_TO + KS - 1 <= W1, "conv2 reads t1 columns OFF - h2 .. OFF - h2 + 224 + 2 h2");
  uint4* ldsw = lds4;                       // [KS][4][32]    weight chunk
  uint4* ldsx = lds4 + W_UNITS;             // [4][XWP]       input chunk
  uint4* ldst = ldsx;                       // [2][4][W1]     t1, both chunks (over the input chunk)
  static_assert(2 * 4 * W1 >= 4 * XWP, "the LDS size of the launcher is that of the t1 image");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int lh = lane >> 5;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * FP_TO;
  const int h2 = (KS - 1) / 2;
  const int xi0 = t0 - OFF - p.pad_left;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x16 + (long long)b * 32 * p.T_in * 4), 0, (unsigned)(32 * p.T_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;
  const int pl = __builtin_amdgcn_readfirstlane(wave);

  uint4 xst[XWI], wst[W_IT];
  auto issue_x = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < XWI; ++it) {
      const int xi = xi0 + lane + 64 * it;
      // (p.xw: the columns conv1 reads — its W1-column window plus the halo of the dilated taps; the rest of the 320 is never multiplied)
      const unsigned voff = (xi >= 0 && xi < p.T_in && lane + 64 * it < p.xw) ? (unsigned)((pl * p.T_in + xi) * 16) : 0x80000000u;
      xst[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, chunk * 4 * p.T_in * 16, 0));
    }
  };
  auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rs, int chunk) {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int u = tid + 256 * i;
      const unsigned voff = u < W_UNITS ? (unsigned)((u % CO_B) * 16 + (u / CO_B) * seg_bytes) : 0x80000000u;
      wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, chunk * (KS * 4) * seg_bytes, 0));
    }
  };
  auto publish_x = [&]() {
#pragma unroll
    for (int it = 0; it < XWI; ++it) ldsx[pl * XWP + lane + 64 * it] = xst[it];
  };
  auto publish_w = [&]() {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int u = tid + 256 * i;
      if (u < W_UNITS) ldsw[u] = wst[i];
    }
  };
  f32x16 acc[1][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][n][r] = 0.f;
  };
  // one chunk of MFMAs: B fragments at `base + n*128 + tap*step` in an image of plane pitch `pitch`
  auto mfma_chunk = [&](const uint4* base, int pitch, int step, bool seven) {
    const uint4* wb = ldsw + lh * CO_B + l31;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const h8 a_hi = __builtin_bit_cast(h8, wb[(t * 4 + 0) * CO_B]);
      const h8 a_lo = __builtin_bit_cast(h8, wb[(t * 4 + 2) * CO_B]);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (!seven || wave + 4 * n < FP_TO / 32) {   // wave-uniform: conv2 has 7 output sub-tiles over 4 waves
          const uint4* xt = base + n * 128 + t * step;
          const h8 b_hi = __builtin_bit_cast(h8, xt[0]);
          const h8 b_lo = __builtin_bit_cast(h8, xt[2 * pitch]);
          acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[0][n], 0, 0, 0);
          acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[0][n], 0, 0, 0);
          acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[0][n], 0, 0, 0);
        }
      }
    }
  };
  const uint4* xbase = ldsx + lh * XWP + wave * 32 + l31;

  // ---- conv1, chunk 0 and 1 ----
  issue_x(0);
  issue_w(w1rs, 0);
  zero_acc();
  publish_x();
  publish_w();
  __syncthreads();
  issue_x(1);
  issue_w(w1rs, 1);
  __builtin_amdgcn_sched_barrier(0);
  mfma_chunk(xbase, XWP, p.dil, false);
  __syncthreads();
  publish_x();
  publish_w();
  __syncthreads();
  issue_w(w2rs, 0);
  __builtin_amdgcn_sched_barrier(0);
  mfma_chunk(xbase, XWP, p.dil, false);
  __syncthreads();                          // every wave is done reading the input chunk: t1 goes over it
  // t1 -> LDS as conv2's B operand (hi|lo planes of both chunks)
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = (wave + 4 * n) * 32 + l31;
    const int pos = t0 - OFF + col;
    const bool inside = pos >= 0 && pos < p.T_in;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = __builtin_fmaf(acc[0][n][4 * rg + k], p.w_descale1, p.bias1[8 * rg + 4 * lh + k]);
        t = lrelu_max(t, p.in_slope);
        v[k] = inside ? t : 0.f;
      }
      const auto h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
      const auto h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
      const auto l01 = split_lo2(h01, v[0], v[1]);
      const auto l23 = split_lo2(h23, v[2], v[3]);
      if (col < W1) {
        ((uint2*)(ldst + (((rg >> 1) * 4 + 0 + (rg & 1)) * W1 + col)))[lh] =
            make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
        ((uint2*)(ldst + (((rg >> 1) * 4 + 2 + (rg & 1)) * W1 + col)))[lh] =
            make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
      }
    }
  }
  // ---- conv2, chunk 0 and 1 ----
  zero_acc();
  __syncthreads();
  publish_w();
  __syncthreads();
  issue_w(w2rs, 1);
  __builtin_amdgcn_sched_barrier(0);
  const uint4* tbase = ldst + lh * W1 + OFF - h2 + wave * 32 + l31;
  mfma_chunk(tbase, W1, 1, true);
  __syncthreads();
  publish_w();
  __syncthreads();
  float rpre[1][2][16];
  epilogue_prefetch_res<1, 2>(p, rpre, b, 0, 0, t0 + wave * 32, l31, lh, 128, t0 + FP_TO);
  __builtin_amdgcn_sched_barrier(0);
  mfma_chunk(tbase + 4 * W1, W1, 1, true);
  conv_epilogue<1, 2, true>(p, acc, b, 0, 0, t0 + wave * 32, l31, lh, 128, t0 + FP_TO, rpre);
}

template <int KS>
static int launch_pair32(const ConvArgs& a, int B, hipStream_t s) {
  ConvArgs p = a;
  if (FP_W1 + (p.ksize - 1) * p.dil > 320) {
    set_error("resblock_pair: dilation %d too large", p.dil);
    return SAT_ERR_INVALID;
  }
  const size_t lds_bytes = ((size_t)KS * 4 * 32 + (size_t)2 * 4 * (KS == 11 ? 236 : FP_W1)) * 16;   // weights + t1 (over the input chunk)
  p.xw = g_trim_halo ? (KS == 11 ? 236 : FP_W1) + (p.ksize - 1) * p.dil : 320;
  auto kern = resblock_pair32_kernel<KS>;
  if (lds_bytes > 64 * 1024)
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  dim3 grid(ceil_div(p.T_q, FP_TO), 1, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("resblock_pair32_kernel");
  return SAT_OK;
}

// f32 [B][C][T] -> split planes of lrelu(x, slope): thread = (utterance, 8-channel group, position);
// 8 coalesced dword loads, two coalesced 16-byte stores (HBM-streaming)
__global__ void __launch_bounds__(256) act_split_kernel(const float* __restrict__ x, uint4* __restrict__ y,
                                                        int C, int T, float slope, int f8) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int cg = blockIdx.y;            // 8-channel group: chunk = cg >> 1, half = cg & 1
  const int b = blockIdx.z;
  if (t >= T) return;
  const float* xr = x + ((long long)b * C + cg * 8) * T + t;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = xr[(long long)j * T];
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x0 = v[2 * j], x1 = v[2 * j + 1];
    x0 = lrelu_max(x0, slope);
    x1 = lrelu_max(x1, slope);
    const auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const auto l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]);
    hi[j] = __builtin_bit_cast(unsigned, h);
    lo[j] = __builtin_bit_cast(unsigned, l);
  }
  uint4* yb = y + (long long)b * (C / 4) * T;                     // C*T*4 bytes per utterance
  const long long u = ((long long)((cg >> 1) * 4 + (cg & 1))) * T + t;
  yb[u] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  if (!f8) {
    yb[u + 2LL * T] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    return;
  }
  // planes 2 / 3: e4m3(hi) and e4m3(lo * 2^10), one byte per channel; this thread owns 8 of the 16 bytes
  float hf[8], lf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float xv = v[j];
    xv = lrelu_max(xv, slope);
    const auto hh = __builtin_bit_cast(decltype(__builtin_amdgcn_cvt_pkrtz(0.f, 0.f)), hi[j >> 1]);
    hf[j] = (float)hh[j & 1];
    lf[j] = (xv - hf[j]) * F8_XLO_SCALE;
  }
  uint2* y8 = (uint2*)(yb + ((long long)((cg >> 1) * 4 + 2)) * T + t) + (cg & 1);
  *y8 = make_uint2(pack_e4m3x4(hf[0], hf[1], hf[2], hf[3]), pack_e4m3x4(hf[4], hf[5], hf[6], hf[7]));
  y8[2LL * T] = make_uint2(pack_e4m3x4(lf[0], lf[1], lf[2], lf[3]), pack_e4m3x4(lf[4], lf[5], lf[6], lf[7]));
}

template <int MT, int NT, int WM, int WN, int KS, bool S1, int XWI>
static int launch_cfg(const ConvArgs& a, int B, int groups, hipStream_t s) {
  constexpr int CO_B = 32 * MT * WM;
  constexpr int T_B = 32 * NT * WN;
  ConvArgs p = a;
  p.xw = (T_B - 1) * p.stride + (p.ksize - 1) * p.dil + 1;
  p.co_tiles_g = ceil_div(p.rows_g, CO_B);
  const size_t lds_bytes = (size_t)CI_CHUNK * (XWI > 0 ? 64 * XWI : p.xw) * sizeof(float);
  auto kern = conv1d_mfma_kernel<MT, NT, WM, WN, KS, S1, XWI>;
  if (lds_bytes > 160 * 1024) {
    set_error("conv1d: input tile of %zu bytes does not fit LDS (stride %d, ksize %d, dilation %d)",
              lds_bytes, p.stride, p.ksize, p.dil);
    return SAT_ERR_INVALID;
  }
  if (lds_bytes > 64 * 1024) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes));
  }
  dim3 grid(ceil_div(p.T_q, T_B), p.co_tiles_g * groups, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("conv1d_mfma_kernel");
  return SAT_OK;
}

// batched staging needs a compile-time bound on the tile width: tile + (KS-1)*dilation columns,
// instantiated for dilation <= 5 (every dilated layer of the generator); wider halos fall back to
// the dynamic staging loop
template <int MT, int NT, int WM, int WN, int KS>
static int launch_xwi(const ConvArgs& a, int B, int groups, hipStream_t s) {
  constexpr int T_B = 32 * NT * WN;
  constexpr int XWI = (T_B + (KS - 1) * 5 + 63) / 64;
  const int xw = T_B + (a.ksize - 1) * a.dil;
  if (xw <= 64 * XWI) return launch_cfg<MT, NT, WM, WN, KS, true, XWI>(a, B, groups, s);
  return launch_cfg<MT, NT, WM, WN, KS, true, 0>(a, B, groups, s);
}

template <int MT, int NT, int WM, int WN>
static int launch_ks(const ConvArgs& a, int B, int groups, hipStream_t s) {
  if (a.stride == 1) {
    switch (a.ksize) {
      case 1: return launch_xwi<MT, NT, WM, WN, 1>(a, B, groups, s);
      case 2: return launch_xwi<MT, NT, WM, WN, 2>(a, B, groups, s);
      case 3: return launch_xwi<MT, NT, WM, WN, 3>(a, B, groups, s);
      case 7: return launch_xwi<MT, NT, WM, WN, 7>(a, B, groups, s);
      case 11: return launch_xwi<MT, NT, WM, WN, 11>(a, B, groups, s);
      default: return launch_cfg<MT, NT, WM, WN, 0, true, 0>(a, B, groups, s);
    }
  }
  return launch_cfg<MT, NT, WM, WN, 0, false, 0>(a, B, groups, s);
}

}  // namespace sat

using namespace sat;

extern "C" int sat_conv1d_packed_dims(int C_in, int C_out, int up, int groups, int* cin_pad,
                                      int* co_pad) {
  SAT_REQUIRE(C_in > 0 && C_out > 0 && up > 0 && groups > 0, "conv1d_packed_dims: bad sizes");
  SAT_REQUIRE(C_in % groups == 0 && C_out % groups == 0, "conv1d_packed_dims: groups must divide channels");
  if (cin_pad) *cin_pad = round_up(C_in / groups, CI_CHUNK);
  if (co_pad) *co_pad = round_up(C_out / groups * up, 64);
  return SAT_OK;
}

extern "C" int sat_upsample_grouped_supported(int C_in, int C_out, int ksize, int stride, int padding) {
  if (stride != 4 || C_in <= 0 || C_out <= 0 || ksize < stride || padding < 0) return 0;
  int lo = 1 << 30, hi = -(1 << 30);
  for (int r = 0; r < stride; ++r)
    for (int dl = -ksize; dl <= ksize; ++dl) {
      const int j = r + padding - stride * dl;
      if (j >= 0 && j < ksize) lo = std::min(lo, dl), hi = std::max(hi, dl);
    }
  const int slots = hi - lo + 1;
  return C_in % 64 == 0 && C_out % 16 == 0 && C_out * 4 > 128 && slots == 3;
}

extern "C" uint32_t sat_convtranspose_zero_taps(int ksize, int stride, int padding) {
  if (stride < 1 || stride > 4 || ksize < 1 || padding < 0) return 0;
  int lo = 1 << 30, hi = -(1 << 30);
  for (int r = 0; r < stride; ++r)
    for (int dl = -ksize; dl <= ksize; ++dl) {
      const int j = r + padding - stride * dl;
      if (j >= 0 && j < ksize) lo = std::min(lo, dl), hi = std::max(hi, dl);
    }
  if (hi - lo + 1 > 8) return 0;
  uint32_t mask = 0;
  for (int slot = 0; slot <= hi - lo; ++slot)
    for (int r = 0; r < stride; ++r) {
      const int j = r + padding - stride * (slot + lo);
      if (j < 0 || j >= ksize) mask |= 1u << (slot * 4 + r);
    }
  return mask;
}

// descriptor -> launch arguments (every check of sat_conv1d_f32 but the choice of the kernel)
static int conv1d_prepare(const sat_conv1d_desc* d, const float* x, const void* w_packed, float* y, ConvArgs& a) {
  SAT_REQUIRE(d && w_packed && (x || d->x_split) && (y || (d->no_y && d->y_split)), "conv1d: null pointer");
  SAT_REQUIRE(d->B > 0 && d->C_in > 0 && d->C_out > 0 && d->T_in > 0 && d->T_q > 0, "conv1d: empty shape");
  SAT_REQUIRE(d->ksize >= 1 && d->dilation >= 1 && d->stride >= 1 && d->up >= 1 && d->groups >= 1,
              "conv1d: bad ksize/dilation/stride/up/groups");
  SAT_REQUIRE(d->C_in % d->groups == 0 && d->C_out % d->groups == 0, "conv1d: groups must divide channels");
  SAT_REQUIRE(d->up == 1 || d->stride == 1, "conv1d: polyphase output requires stride 1");
  SAT_REQUIRE(!d->up_grouped || (d->mode == SAT_CONV_F16X3 && d->up == 4), "conv1d: up_grouped is a SAT_CONV_F16X3 layout of up = 4");
  a = ConvArgs{};
  a.x = x;
  a.w = (const float*)w_packed;
  a.y = y;
  a.bias = d->bias;
  a.res = d->res;
  a.ch_scale = d->ch_scale;
  a.ch_shift = d->ch_shift;
  SAT_REQUIRE((d->ch_scale == nullptr) == (d->ch_shift == nullptr), "conv1d: ch_scale and ch_shift go together");
  a.x_bs = d->x_bstride;
  a.x_cs = d->x_cstride;
  a.y_bs = d->y_bstride;
  a.y_cs = d->y_cstride;
  a.r_bs = d->res_bstride;
  a.r_cs = d->res_cstride;
  a.cin_g = d->C_in / d->groups;
  a.cout_g = d->C_out / d->groups;
  a.rows_g = a.cout_g * d->up;
  a.T_in = d->T_in;
  a.T_q = d->T_q;
  a.ksize = d->ksize;
  a.dil = d->dilation;
  a.stride = d->stride;
  a.pad_left = d->pad_left;
  a.up = d->up;
  a.cin_pad = round_up(a.cin_g, CI_CHUNK);
  a.co_pad = round_up(a.rows_g, 64);
  a.w_gs = (long long)a.cin_pad * a.ksize * a.co_pad;
  a.in_lrelu = d->in_lrelu;
  a.in_slope = d->in_slope;
  a.relu = d->relu;
  a.relu_first = d->relu_first;
  a.gelu = d->gelu;
  a.res_after = d->res_after_act;
  a.accum = d->accum;
  a.accum_div = d->accum_div;
  a.no_store = d->accum_no_store;
  SAT_REQUIRE(!d->accum_no_store || (d->accum && y && d->y_split && !d->no_y && d->up == 1),
              "conv1d: accum_no_store goes with accum (y is read), y_split (the output that is written) and up 1");
  a.res_scale = d->res_scale;
  a.res_toff = d->res_toff;
  a.res_tstride = d->res_tstride > 0 ? d->res_tstride : 1;
  a.w_descale = d->w_descale != 0.f ? d->w_descale : 1.f;
  a.w_descale1 = 1.f;
  SAT_REQUIRE(d->mode != SAT_CONV_F32 || a.w_descale == 1.f, "conv1d: w_descale is a split-f16 option");
  {
    const long long ylim = (long long)a.rows_g * a.y_cs * 4;
    const long long rlim = d->res ? ((long long)a.rows_g * a.r_cs + (long long)a.T_q * a.res_tstride + a.res_toff) * 4 : 0;
    a.fast_epi = d->up == 1 && ylim < (1LL << 31) && rlim < (1LL << 31) && ylim > 0;
  }
  SAT_REQUIRE(d->mode != SAT_CONV_F32 || (!d->x_split && !d->y_split && !d->no_y), "conv1d: split planes need a split-f16 mode");
  if (d->mode == SAT_CONV_F16X3 || d->mode == SAT_CONV_F16F8 || d->mode == SAT_CONV_F16F8R) {
    a.f8 = d->mode == SAT_CONV_F16F8;
    a.f8r = d->mode == SAT_CONV_F16F8R;
    SAT_REQUIRE(!a.f8 || d->x_split, "conv1d(f16f8): the input must be split planes (SAT_SPLIT_F8)");
    SAT_REQUIRE(!a.f8r || (d->x_split && d->x_split8 && d->y_split_format <= 1 && d->ksize >= 3 && !d->up_grouped && d->up == 1),
                "conv1d(f16f8r): needs x_split (SAT_SPLIT_F16 planes) with its e4m3 sidecar x_split8, SAT_SPLIT_F16 output planes, ksize >= 3 (the ring kernel's taps), up 1");
    SAT_REQUIRE(!d->y_split8 || (d->y_split && d->mode != SAT_CONV_F16F8 && d->y_split_format <= 1), "conv1d: y_split8 is the sidecar of SAT_SPLIT_F16 planes y_split");
    SAT_REQUIRE(!d->y_split_hi_only || d->y_split8, "conv1d: y_split_hi_only goes with y_split8");
    a.x8 = d->x_split8;
    a.y8 = d->y_split8;
    a.y16_hi_only = d->y_split_hi_only;
    SAT_REQUIRE(d->y_split_format >= 0 && d->y_split_format <= 2, "conv1d: unknown y_split_format");
    a.y16_f8 = d->y_split_format == 0 ? a.f8 : d->y_split_format == 2;
    SAT_REQUIRE(d->stride == 1, "conv1d(f16x3): stride 1 only");
    SAT_REQUIRE((long long)a.cin_g * a.x_cs * 4 < (1LL << 31), "conv1d(f16x3): input slab too large for 31-bit offsets");
    a.x16 = d->x_split;
    a.y16 = d->y_split;
    a.y16_slope = d->y_split_slope;
    a.no_y = d->no_y;
    // the split-f16 kernels apply a leaky-relu as max(v, slope v) and undo one as min(r, r / slope) (common.h lrelu_max)
    SAT_REQUIRE((!d->in_lrelu || (d->in_slope >= 0.f && d->in_slope <= 1.f)) && (!d->y_split || (d->y_split_slope >= 0.f && d->y_split_slope <= 1.f)) &&
                    (!d->res_split || d->res_split_slope <= 1.f),
                "conv1d(f16x3): leaky-relu slopes must lie in [0, 1]");
    if (a.x16)
      SAT_REQUIRE(d->groups == 1 && a.cin_g % 16 == 0 && (long long)a.cin_g * a.T_in * 4 < (1LL << 31),
                  "conv1d(f16x3): x_split needs groups 1, C_in %% 16 == 0 and a slab below 2 GiB");
    if (d->x_wrap_channels) {
      SAT_REQUIRE(a.x16 && !a.f8 && d->ksize == 1 && d->groups == 1 && d->x_wrap_channels > 0 && d->x_wrap_channels % 32 == 0 &&
                      d->x_wrap_channels < d->C_in && d->C_in <= 2 * d->x_wrap_channels && d->C_in % 32 == 0 && a.rows_g % 128 == 0,
                  "conv1d: x_wrap_channels needs a 1x1 split-f16 conv on split planes, channel counts multiples of 32 with "
                  "x_wrap < C_in <= 2 x_wrap, C_out %% 128 == 0");
      a.k1_wrap = d->x_wrap_channels / CI_CHUNK;
      a.cin_g = d->x_wrap_channels;            // channels behind x_split (the descriptor range of the planes)
    }
    if (d->res_split) {
      SAT_REQUIRE(!d->res && a.fast_epi && d->groups == 1 && a.rows_g % 16 == 0 && d->res_split_slope > 0.f &&
                  (long long)a.rows_g * a.T_q * 4 < (1LL << 31), "conv1d(f16x3): res_split needs up 1, groups 1, C_out %% 16 == 0, no f32 res");
      a.res16 = d->res_split;
      a.res16_inv = 1.0f / d->res_split_slope;
    }
    if (d->up_grouped) {
      // polyphase upsampler of stride 4 with its rows grouped by phase: the LDS-DMA ring writes the planes itself
      a.up_grouped = 1;
      a.up_zero_taps = d->up_zero_taps;
      a.y = (float*)a.y16;
      SAT_REQUIRE(d->mode == SAT_CONV_F16X3 && d->groups == 1 && !d->res && !d->res_split && !d->accum && !d->relu && !d->gelu && !d->ch_scale &&
                      (long long)a.rows_g * a.T_q * 4 < (1LL << 31) && convring_ups_supports(a),
                  "conv1d(f16x3): up_grouped needs up 4, split planes in and out with no_y, C_in %% 64 == 0, C_out %% 16 == 0, C_out * 4 > 128, "
                  "3 tap slots, a bias and a plain epilogue");
    } else if (d->up > 1 && (a.y16 || a.no_y)) {
      // polyphase upsampler straight to planes (LDS-transposed epilogue)
      const int co_b = a.rows_g > 32 ? 64 : 32;
      SAT_REQUIRE(a.x16 && a.y16 && a.no_y && !a.f8 && !a.y16_f8 && d->groups == 1 && co_b % (8 * d->up) == 0 && a.cout_g % 16 == 0 &&
                      !d->res && !d->res_split && !d->accum && !d->relu && !d->gelu && !d->ch_scale &&
                      (long long)a.rows_g * a.T_q * 4 < (1LL << 31),
                  "conv1d(f16x3): up > 1 with y_split needs split-plane input, no_y, up in {2, 4}, C_out %% 16 == 0, a plain epilogue");
      a.poly_planes = 1;
      a.y = (float*)a.y16;
    } else if (a.y16 || a.no_y) {
      SAT_REQUIRE(a.fast_epi && d->groups == 1 && a.rows_g % 16 == 0 && (long long)a.rows_g * a.T_q * 4 < (1LL << 31),
                  "conv1d(f16x3): y_split / no_y need up 1, groups 1, C_out %% 16 == 0 and a slab below 2 GiB");
      if (a.no_y) a.y = (float*)a.y16, a.accum = 0;   // descriptor base only; nothing is loaded or stored through it
      SAT_REQUIRE(!(d->no_y && d->accum), "conv1d: no_y with accum");
    }
    a.w_gs = (long long)(a.cin_pad / CI_CHUNK) * a.ksize * a.co_pad * 64;   // bytes per group
    // (SAT_CONV_F16F8R packing: [2 ceil(C_in / 32 x k / 2) steps][8 planes][co_pad][16 B])
    if (a.f8r) a.w_gs = (long long)(2 * (((a.cin_pad / (2 * CI_CHUNK)) * a.ksize + 1) / 2)) * 8 * a.co_pad * 16;
    return SAT_OK;
  }
  SAT_REQUIRE(d->mode == SAT_CONV_F32, "conv1d: unknown mode %d", d->mode);
  return SAT_OK;
}

static int conv1d_dispatch(const sat_conv1d_desc* d, ConvArgs& a, hipStream_t s);

extern "C" int sat_conv1d_f32(const sat_conv1d_desc* d, const float* x, const void* w_packed,
                              float* y, void* stream) {
  ConvArgs a;
  int st = conv1d_prepare(d, x, w_packed, y, a);
  if (st != SAT_OK) return st;
  return conv1d_dispatch(d, a, (hipStream_t)stream);
}

// the kernel choice for a prepared conv (sat_conv1d_f32; sat_tdnnf_layer_f32 enters here with the bypass planes' geometry set)
static int conv1d_dispatch(const sat_conv1d_desc* d, ConvArgs& a, hipStream_t s) {
  if (d->mode == SAT_CONV_F16F8R) {
    SAT_REQUIRE(d->groups == 1 && convring_supports(a, d->B),
                "conv1d(f16f8r): served by the LDS-DMA ring kernel only (C_out > 64, C_in %% 32 == 0, ksize >= 3, halo <= 64, bias, plain / "
                "ResBlock epilogue: sat_conv1d_f8r_supported)");
    return launch_f16x3_convring(a, d->B, s);
  }
  if (d->mode == SAT_CONV_F16X3 || d->mode == SAT_CONV_F16F8) {
    // an e4m3 sidecar of the output is written by the epilogues of conv_ring16.hip only
    SAT_REQUIRE(!a.y8 || a.up_grouped || (d->groups == 1 && convring_supports(a, d->B)),
                "conv1d: y_split8 needs a shape the LDS-DMA ring kernel serves (or sat_planes_f8_sidecar after the launch)");
    // 1x1 on split planes with enough rows: the GEMM kernel (activation fragments straight from the planes)
    if (a.x16 && a.ksize == 1 && !a.f8 && !a.poly_planes && a.fast_epi && d->groups == 1 && a.rows_g >= 128 &&
        (a.cin_pad / CI_CHUNK) % 4 == 0 && g_k1_gemm) {
      // ring kernel: whole 128-row weight tiles, and 256-column tiles that pad the time axis (nearly) no more than
      // 128-column ones (249 frames: 256 either way; 49 frames: the 128-column kernel)
      const long long c256 = (long long)ceil_div(a.T_q, 256) * 256, c128 = (long long)ceil_div(a.T_q, 128) * 128;
      if (a.k1_wrap) {                         // only the 16x16x32 ring kernel reads wrapped K chunks
        SAT_REQUIRE(ring16_supports(a), "conv1d: x_wrap_channels needs the epilogue of the ring GEMM (up 1, split-f16 planes)");
        return launch_f16x3_ring16(a, d->B, s);
      }
      if (g_k1_gemm >= 2 && a.co_pad % 128 == 0 && c256 * 8 <= c128 * 9) {
        // more tiles than CUs (the 1024 -> 4096 layer: four per CU): the persistent walk of the same ring (gemm_walk16.hip)
        if (g_k1_gemm >= 3 && gemm_walk_supports(a) && gemm_walk_wanted((long long)ceil_div(a.rows_g, 128) * ceil_div(a.T_q, 256) * d->B))
          return launch_f16x3_gemm_walk(&a, 1, d->B, s);
        return g_k1_gemm >= 3 && ring16_supports(a) ? launch_f16x3_ring16(a, d->B, s) : launch_f16x3_ring(a, d->B, s);
      }
      return launch_f16x3_k1(a, d->B, s);
    }
    SAT_REQUIRE(!a.k1_wrap, "conv1d: x_wrap_channels is only served by the 1x1 GEMM path (C_in %% 64 == 0, up 1, 31-bit slabs)");
    if (a.up_grouped) return launch_f16x3_convring_ups(a, d->B, s);
    // the generator's resblock convs at C >= 128: the LDS-DMA ring on the 16x16x32 shape (conv_ring16.hip)
    if (d->groups == 1 && convring_supports(a, d->B)) return launch_f16x3_convring(a, d->B, s);
    // few blocks (TDNNF linearB: 128 rows x 250 frames x 32 utterances = 64 tiles of 64 x 256 on 256 CUs): half-width
    // tiles double the blocks of these latency-bound launches
    if (a.x16 && a.ksize == 3 && !a.f8 && !a.poly_planes && a.rows_g > 32 && a.T_q > 128 &&
        (long long)ceil_div(a.rows_g, 64) * ceil_div(a.T_q, 256) * d->B * d->groups < 256)
      return launch_f16x3<2, 1, 3>(a, d->B, d->groups, s);
    // the same for the generator's conv_pre (f32 input, 7 taps, 512 rows x 250 frames x 32 utterances = 256 tiles of 64 x 256)
    if (g_half_tile7 && !a.x16 && a.ksize == 7 && a.rows_g > 32 && a.T_q > 128 &&
        (long long)ceil_div(a.rows_g, 64) * ceil_div(a.T_q, 256) * d->B * d->groups <= 256)
      return launch_f16x3<2, 1, 7>(a, d->B, d->groups, s);
    // the generator's resblock convs: the three-blocks-per-CU form of the tile (conv_lean.hip)
    if (d->groups == 1 && (a.ksize == 3 ? g_lean3 : a.ksize == 7 ? g_lean7 : g_lean11) && lean_supports(a))
      return launch_f16x3_lean(a, d->B, s);
    if (a.rows_g > 32) return launch_f16x3_ks<2, 2>(a, d->B, d->groups, s);   // 64 rows x 256 positions
    return launch_f16x3_ks<1, 4>(a, d->B, d->groups, s);                       // 32 rows x 512 positions
  }
  // tile shape by output rows per group: wide-in-time tiles for thin layers
  if (a.rows_g > 64) {
    // few, long GEMM-like layers (TDNNF: 128 rows x ~530 frames per utterance, K = 3072) would
    // launch fewer 128x128 blocks than there are CUs: cut the time tile to 32 positions instead
    const long long blocks128 = (long long)ceil_div(a.T_q, 128) * ceil_div(a.rows_g, 128) * d->groups * d->B;
    if (blocks128 < 2 * 256) return launch_ks<1, 1, 4, 1>(a, d->B, d->groups, s);   // 128 rows x 32 positions
    return launch_ks<2, 2, 2, 2>(a, d->B, d->groups, s);                              // 128 rows x 128 positions
  }
  if (a.rows_g > 32) return launch_ks<2, 2, 1, 4>(a, d->B, d->groups, s);   //  64 rows x 256 positions
  return launch_ks<1, 4, 1, 4>(a, d->B, d->groups, s);                       //  32 rows x 512 positions
}

// One TDNNF layer per call (chain/nn.py:336-347 TDNNFBatchNorm.forward = TDNNF.forward 267-292 with its bypass -> BatchNorm1d -> ReLU): the
// two launches sat_conv1d_f32 makes for linearB (context_len taps over frames, feat -> bottleneck) and linearA (1x1, bottleneck -> out,
// bias, bypass from the layer input identity_lidx frames in, folded BatchNorm, ReLU) behind ONE descriptor — the host composes nothing
// between them, and the bottleneck never has to exist as f32 when the layer runs on split planes.
extern "C" int sat_tdnnf_layer_f32(const sat_tdnnf_layer_desc* d, void* stream) {
  SAT_REQUIRE(d && d->wB_packed && d->wA_packed, "tdnnf_layer: null pointer");
  SAT_REQUIRE(d->y || (d->y_split && d->mode == SAT_CONV_F16X3), "tdnnf_layer: y, or (split planes) y_split alone");
  SAT_REQUIRE(d->B > 0 && d->feat_dim > 0 && d->bottleneck_dim > 0 && d->out_dim > 0 && d->context_len >= 1 && d->T_in >= d->context_len,
              "tdnnf_layer: unsupported shape");
  SAT_REQUIRE(d->mode == SAT_CONV_F32 || d->mode == SAT_CONV_F16X3, "tdnnf_layer: mode must be SAT_CONV_F32 or SAT_CONV_F16X3");
  SAT_REQUIRE(d->x || d->x_split, "tdnnf_layer: x or x_split");
  const int T_q = d->T_in - (d->context_len - 1);
  const bool planes = d->mode == SAT_CONV_F16X3 && d->z_split && d->bottleneck_dim % 16 == 0;
  SAT_REQUIRE(planes || d->z, "tdnnf_layer: the bottleneck needs z (f32) or, on split planes, z_split");
  const bool bypass_planes = d->bypass_scale != 0.f && !d->x;      // no f32 input: the bypass is rebuilt from the input planes (hi + lo)
  SAT_REQUIRE(d->bypass_scale == 0.f || d->out_dim == d->feat_dim, "tdnnf_layer: the bypass adds the layer input (out_dim == feat_dim)");
  SAT_REQUIRE(!bypass_planes || (d->x_split && d->mode == SAT_CONV_F16X3 && d->feat_dim % 16 == 0),
              "tdnnf_layer: a bypass without the f32 input needs x_split (SAT_CONV_F16X3, feat_dim %% 16 == 0)");
  sat_conv1d_desc b{};
  b.B = d->B, b.C_in = d->feat_dim, b.T_in = d->T_in, b.C_out = d->bottleneck_dim, b.T_q = T_q;
  b.ksize = d->context_len, b.dilation = 1, b.stride = 1, b.pad_left = 0, b.mode = d->mode, b.groups = 1, b.up = 1;
  b.x_cstride = d->T_in, b.x_bstride = (int64_t)d->feat_dim * d->T_in;
  b.y_cstride = T_q, b.y_bstride = (int64_t)d->bottleneck_dim * T_q;
  b.res_tstride = 1;
  b.bias = d->bB;
  b.w_descale = d->wB_descale;
  b.x_split = d->mode == SAT_CONV_F16X3 ? d->x_split : nullptr;
  if (planes) b.y_split = d->z_split, b.y_split_slope = 1.0f, b.no_y = 1;
  int st = sat_conv1d_f32(&b, d->x, d->wB_packed, planes ? (float*)d->z_split : d->z, stream);
  if (st != SAT_OK) return st;
  sat_conv1d_desc a{};
  a.B = d->B, a.C_in = d->bottleneck_dim, a.T_in = T_q, a.C_out = d->out_dim, a.T_q = T_q;
  a.ksize = 1, a.dilation = 1, a.stride = 1, a.pad_left = 0, a.mode = d->mode, a.groups = 1, a.up = 1;
  a.x_cstride = T_q, a.x_bstride = (int64_t)d->bottleneck_dim * T_q;
  a.y_cstride = T_q, a.y_bstride = (int64_t)d->out_dim * T_q;
  a.bias = d->bA, a.ch_scale = d->bn_scale, a.ch_shift = d->bn_shift, a.relu = 1;
  a.w_descale = d->wA_descale;
  a.res_tstride = 1;
  const int lidx = d->context_len == 2 ? 1 : d->context_len / 2;      // identity_lidx, chain/nn.py:233-247
  if (d->bypass_scale != 0.f && !bypass_planes) {
    a.res = d->x, a.res_scale = d->bypass_scale;
    a.res_toff = lidx;
    a.res_cstride = d->T_in, a.res_bstride = (int64_t)d->feat_dim * d->T_in;
  } else if (bypass_planes) {
    a.res_split = d->x_split, a.res_split_slope = 1.0f, a.res_scale = d->bypass_scale;
  }
  if (planes) a.x_split = d->z_split;
  if (d->y_split && d->mode == SAT_CONV_F16X3) a.y_split = d->y_split, a.y_split_slope = 1.0f;
  if (!d->y) a.no_y = 1;
  ConvArgs args;
  st = conv1d_prepare(&a, planes ? (const float*)d->z_split : d->z, d->wA_packed, d->y, args);
  if (st != SAT_OK) return st;
  if (bypass_planes) {
    // the bypass planes are the layer INPUT: rows of T_in positions, output column 0 <-> position identity_lidx (the common epilogues
    // of the 1x1 GEMM kernels take both; conv_common.h res16_T / res16_toff)
    SAT_REQUIRE((long long)d->feat_dim * d->T_in * 4 < (1LL << 31), "tdnnf_layer: input slab too large for 31-bit offsets");
    args.res16_T = d->T_in, args.res16_toff = lidx;
  }
  return conv1d_dispatch(&a, args, (hipStream_t)stream);
}

extern "C" int sat_conv1d_multi_f32(const sat_conv1d_desc* d, const float* const* x, const void* const* w_packed, float* const* y,
                                    int n, void* stream) {
  SAT_REQUIRE(d && x && w_packed && y && n >= 1 && n <= 3, "conv1d_multi: 1..3 convolutions");
  ConvArgs a[3];
  bool ring = true, writes_read = false, partial = false;
  for (int j = 0; j < n; ++j) {
    int st = conv1d_prepare(&d[j], x[j], w_packed[j], y[j], a[j]);
    if (st != SAT_OK) return st;
    ring = ring && (d[j].mode == SAT_CONV_F16X3 || d[j].mode == SAT_CONV_F16F8R) && d[j].groups == 1 && d[j].B == d[0].B && convring_supports(a[j], d[j].B) && convring_same_shape(a[j], a[0]);
  }
  // the order of the jobs may rotate from block to block unless a job reads (accumulates into, takes its residual or
  // input from) what another one writes, or two jobs write the same bytes: compared as BYTE RANGES (views of one buffer at
  // different offsets overlap without being equal; round-4 advisor item)
  struct Span { const char* lo; const char* hi; };
  auto span = [](const void* p, long long bytes) { return Span{(const char*)p, p ? (const char*)p + (bytes > 0 ? bytes : 0) : (const char*)p}; };
  auto overlap = [](const Span& u, const Span& v) { return u.lo && v.lo && u.lo < v.hi && v.lo < u.hi; };
  auto yspan = [&](const ConvArgs& c, int B) {
    return span(c.no_y ? nullptr : (const void*)c.y, ((long long)(B - 1) * c.y_bs + (long long)(c.rows_g - 1) * c.y_cs + (long long)c.T_q * c.up) * 4);
  };
  auto writes = [&](const ConvArgs& c, int B, Span (&w)[3]) {
    w[0] = c.no_store ? Span{nullptr, nullptr} : yspan(c, B);      // (accum_no_store: y is only read)
    w[1] = span(c.y16, (long long)B * c.rows_g * c.T_q * c.up * 4);
    w[2] = span(c.y8, (long long)B * c.rows_g * c.T_q * c.up * 2);
  };
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      if (i == j) continue;
      const int B = d[0].B;
      Span wi[3], wj[3];
      writes(a[i], B, wi);
      writes(a[j], B, wj);
      const Span rd[6] = {a[j].accum ? yspan(a[j], B) : Span{nullptr, nullptr},
                          span(a[j].res, ((long long)(B - 1) * a[j].r_bs + (long long)(a[j].rows_g - 1) * a[j].r_cs + (long long)a[j].T_q * a[j].res_tstride + a[j].res_toff) * 4),
                          span(a[j].res16, (long long)B * a[j].rows_g * a[j].T_q * 4), span(a[j].x16, (long long)B * a[j].cin_g * a[j].T_in * 4),
                          span(a[j].x, ((long long)(B - 1) * a[j].x_bs + (long long)(a[j].cin_g - 1) * a[j].x_cs + a[j].T_in) * 4),
                          span(a[j].x8, (long long)B * a[j].cin_g * a[j].T_in * 2)};
      // the SAME buffer read / written by two jobs: kept in index order inside a block (a tile of job j + 1 touches what its block's
      // tile of job j touched); buffers that overlap at DIFFERENT bases: tiles of different blocks would meet — single launches
      for (const Span& r : rd)
        for (const Span& w : wi)
          if (overlap(r, w)) (r.lo == w.lo ? writes_read : partial) = true;
      for (const Span& u : wi)
        for (const Span& v : wj)
          if (overlap(u, v)) (u.lo == v.lo ? writes_read : partial) = true;
    }
  if (partial) ring = false;
  if (ring && n > 1) return launch_f16x3_convring_multi(a, n, !writes_read, d[0].B, (hipStream_t)stream);
  // 1x1 convs of one shape on split planes (q | k | v of an attention layer): one launch of the persistent ring GEMM
  bool walk = n > 1 && !writes_read && !partial && g_k1_gemm >= 3;
  for (int j = 0; j < n && walk; ++j) {
    const long long c256 = (long long)ceil_div(a[j].T_q, 256) * 256, c128 = (long long)ceil_div(a[j].T_q, 128) * 128;
    walk = d[j].mode == SAT_CONV_F16X3 && d[j].groups == 1 && d[j].B == d[0].B && gemm_walk_supports(a[j]) && gemm_walk_same_shape(a[j], a[0]) &&
           c256 * 8 <= c128 * 9;
  }
  if (walk) return launch_f16x3_gemm_walk(a, n, d[0].B, (hipStream_t)stream);
  for (int j = 0; j < n; ++j) {
    int st = sat_conv1d_f32(&d[j], x[j], w_packed[j], y[j], stream);
    if (st != SAT_OK) return st;
  }
  return SAT_OK;
}


extern "C" int sat_resblock_pair_f16x3(const sat_conv1d_desc* d, const float* x, const void* w1_packed,
                                       const float* bias1, const void* w2_packed, float* y, void* stream) {
  return sat_resblock_pair_scaled_f16x3(d, x, w1_packed, bias1, 1.f, w2_packed, y, stream);
}

extern "C" int sat_resblock_pair_scaled_f16x3(const sat_conv1d_desc* d, const float* x, const void* w1_packed,
                                              const float* bias1, float w1_descale, const void* w2_packed, float* y, void* stream) {
  SAT_REQUIRE(d && w1_packed && w2_packed && bias1 && d->bias && (x || d->x_split) && (y || (d->no_y && d->y_split)),
              "resblock_pair: null pointer");
  SAT_REQUIRE(d->C_in == d->C_out && d->C_in % 16 == 0 &&
                  (d->C_in <= 32 || (d->C_in == 64 && d->x_split && d->res_split)),
              "resblock_pair: C must be 16 or 32 (64 with 3 taps and split planes end to end)");
  SAT_REQUIRE(d->groups == 1 && d->up == 1 && d->stride == 1 && d->T_q == d->T_in, "resblock_pair: same-length conv pair only");
  SAT_REQUIRE(d->ksize == 3 || d->ksize == 7 || d->ksize == 11, "resblock_pair: kernel size %d not instantiated", d->ksize);
  SAT_REQUIRE(d->in_lrelu, "resblock_pair: both convs take a leaky-relu input");
  SAT_REQUIRE((d->res && d->res == x && !d->res_split) || (!d->res && d->res_split && d->res_split == d->x_split),
              "resblock_pair: the residual is the block input (res == x, or res_split == x_split)");
  ConvArgs a{};
  a.x = x;
  a.w = (const float*)w1_packed;
  a.w2 = w2_packed;
  a.bias1 = bias1;
  a.y = y;
  a.bias = d->bias;                 // bias of the second conv (the epilogue's)
  a.res = d->res;
  a.x_bs = d->x_bstride; a.x_cs = d->x_cstride;
  a.y_bs = d->y_bstride; a.y_cs = d->y_cstride;
  a.r_bs = d->res_bstride; a.r_cs = d->res_cstride;
  a.cin_g = d->C_in; a.cout_g = d->C_out; a.rows_g = d->C_out;
  a.T_in = d->T_in; a.T_q = d->T_q;
  a.ksize = d->ksize; a.dil = d->dilation; a.stride = 1; a.up = 1;
  a.pad_left = (d->ksize * d->dilation - d->dilation) / 2;   // 'same' padding of the dilated first conv
  a.cin_pad = round_up(a.cin_g, CI_CHUNK);
  a.co_pad = 64;
  a.w_gs = (long long)(a.cin_pad / CI_CHUNK) * a.ksize * a.co_pad * 64;
  a.in_lrelu = 1; a.in_slope = d->in_slope;
  SAT_REQUIRE(d->in_slope >= 0.f && d->in_slope <= 1.f && (!d->y_split || (d->y_split_slope >= 0.f && d->y_split_slope <= 1.f)) &&
                  (!d->res_split || d->res_split_slope <= 1.f),
              "resblock_pair: leaky-relu slopes must lie in [0, 1] (common.h lrelu_max)");
  a.accum = d->accum; a.accum_div = d->accum_div;
  a.no_store = d->accum_no_store;
  SAT_REQUIRE(!d->accum_no_store || (d->accum && y && d->y_split && !d->no_y), "resblock_pair: accum_no_store goes with accum and y_split");
  a.res_scale = 1.f; a.res_toff = 0; a.res_tstride = 1;
  a.w_descale = d->w_descale != 0.f ? d->w_descale : 1.f;        // second conv (the epilogue's)
  a.w_descale1 = w1_descale != 0.f ? w1_descale : 1.f;
  SAT_REQUIRE((long long)a.cin_g * a.x_cs * 4 < (1LL << 31) && (long long)a.rows_g * a.y_cs * 4 < (1LL << 31),
              "resblock_pair: slab too large for 31-bit offsets");
  a.fast_epi = 1;
  a.x16 = d->x_split;
  a.y16 = d->y_split;
  a.y16_slope = d->y_split_slope;
  a.no_y = d->no_y;
  SAT_REQUIRE(!(d->no_y && d->accum), "resblock_pair: no_y with accum");
  if (a.no_y) a.y = (float*)a.y16;
  if (d->res_split) {
    SAT_REQUIRE(d->res_split_slope > 0.f, "resblock_pair: res_split_slope");
    a.res16 = d->res_split;
    a.res16_inv = 1.0f / d->res_split_slope;
  }
  SAT_REQUIRE((long long)a.rows_g * a.T_q * 4 < (1LL << 31), "resblock_pair: slab too large");
  hipStream_t s = (hipStream_t)stream;
  if (a.cin_g == 16 && a.x16 && a.res16) {
    // C = 16 with split planes end to end: the 16-row MFMA shape, everything resident
    switch (a.ksize) {
      case 3: return launch_pair16<3>(a, d->B, s);
      case 7: return launch_pair16<7>(a, d->B, s);
      default: return launch_pair16<11>(a, d->B, s);
    }
  }
  if (a.cin_g == 64 && g_pair32s && pair32s_supports(a)) return launch_pair32s(a, d->B, s);
  if (a.cin_g == 64) {
    SAT_REQUIRE(pair64_supports(a), "resblock_pair(C = 64): split planes in, residual from planes, (ksize - 1) * dilation <= 64");
    a.xw = g_trim_halo ? 128 + (a.ksize - 1) * a.dil : 192;      // staged columns conv1 reads (pair64.hip)
    return launch_pair64(a, d->B, s);
  }
  if (a.cin_g == 32 && a.x16 && a.res16) {
    if (g_pair32s && pair32s_supports(a)) return launch_pair32s(a, d->B, s);
    switch (a.ksize) {
      case 3: return launch_pair32<3>(a, d->B, s);
      case 7: return launch_pair32<7>(a, d->B, s);
      default: return launch_pair32<11>(a, d->B, s);
    }
  }
  switch (a.ksize) {
    case 3: return launch_pair<3>(a, d->B, s);
    case 7: return launch_pair<7>(a, d->B, s);
    default: return launch_pair<11>(a, d->B, s);
  }
}

extern "C" int sat_conv1d_f8r_supported(const sat_conv1d_desc* d) {
  if (!d || d->mode != SAT_CONV_F16F8R || !d->x_split || !d->x_split8 || d->groups != 1) return 0;
  ConvArgs a;
  float dummy;
  if (conv1d_prepare(d, nullptr, &dummy, d->no_y ? nullptr : &dummy, a) != SAT_OK) return 0;
  return convring_supports(a, d->B) ? 1 : 0;
}

// 8-bit sidecar of SAT_SPLIT_F16 planes: thread = (utterance, chunk, position); reads the chunk's four 16-byte units at t, writes
// e5m2(hi) and e5m2(lo * 2^10) of its 16 channels as two 16-byte units (HBM-streaming)
__global__ void __launch_bounds__(256) planes_f8_sidecar_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int C, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const uint4* xb = x + ((long long)b * (C / 16) + chunk) * 4 * T + t;
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  const h8v h0 = __builtin_bit_cast(h8v, xb[0]), h1 = __builtin_bit_cast(h8v, xb[T]);
  const h8v l0 = __builtin_bit_cast(h8v, xb[2LL * T]), l1 = __builtin_bit_cast(h8v, xb[3LL * T]);
  uint4 oh, ol;
  oh.x = pack_e5m2x4((float)h0[0], (float)h0[1], (float)h0[2], (float)h0[3]);
  oh.y = pack_e5m2x4((float)h0[4], (float)h0[5], (float)h0[6], (float)h0[7]);
  oh.z = pack_e5m2x4((float)h1[0], (float)h1[1], (float)h1[2], (float)h1[3]);
  oh.w = pack_e5m2x4((float)h1[4], (float)h1[5], (float)h1[6], (float)h1[7]);
  const float k = F8_XLO_SCALE;
  ol.x = pack_e5m2x4((float)l0[0] * k, (float)l0[1] * k, (float)l0[2] * k, (float)l0[3] * k);
  ol.y = pack_e5m2x4((float)l0[4] * k, (float)l0[5] * k, (float)l0[6] * k, (float)l0[7] * k);
  ol.z = pack_e5m2x4((float)l1[0] * k, (float)l1[1] * k, (float)l1[2] * k, (float)l1[3] * k);
  ol.w = pack_e5m2x4((float)l1[4] * k, (float)l1[5] * k, (float)l1[6] * k, (float)l1[7] * k);
  uint4* yb = y + ((long long)b * (C / 16) + chunk) * 2 * T + t;
  yb[0] = oh;
  yb[T] = ol;
}

extern "C" int sat_planes_f8_sidecar(const void* x_split, void* x_split8, int B, int C, int T, void* stream) {
  SAT_REQUIRE(x_split && x_split8, "planes_f8_sidecar: null pointer");
  SAT_REQUIRE(B > 0 && C > 0 && T > 0 && C % 16 == 0 && B < 65536 && C / 16 < 65536, "planes_f8_sidecar: unsupported shape B=%d C=%d T=%d", B, C, T);
  dim3 grid(ceil_div(T, 256), C / 16, B);
  hipLaunchKernelGGL(planes_f8_sidecar_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)x_split, (uint4*)x_split8, C, T);
  SAT_LAUNCH_CHECK("planes_f8_sidecar_kernel");
  return SAT_OK;
}

extern "C" int sat_act_split_f32(const float* x, void* x_split, int B, int C, int T, float slope, int format, void* stream) {
  SAT_REQUIRE(x && x_split, "act_split: null pointer");
  SAT_REQUIRE(format == SAT_SPLIT_F16 || format == SAT_SPLIT_F8, "act_split: unknown format");
  SAT_REQUIRE(B > 0 && C > 0 && T > 0 && C % 16 == 0 && B < 65536, "act_split: unsupported shape B=%d C=%d T=%d", B, C, T);
  SAT_REQUIRE(slope >= 0.f && slope <= 1.f, "act_split: slope must lie in [0, 1]");
  dim3 grid(ceil_div(T, 256), C / 8, B);
  hipLaunchKernelGGL(act_split_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (uint4*)x_split, C, T, slope, format == SAT_SPLIT_F8);
  SAT_LAUNCH_CHECK("act_split_kernel");
  return SAT_OK;
}

extern "C" int sat_pair32_debug_stamps(int64_t* buf) { return pair32_debug_stamps((long long*)buf); }
extern "C" int sat_convring_debug_stamps(int64_t* buf) { return convring_debug_stamps((long long*)buf); }

extern "C" int sat_conv_set_option(const char* name, int value) {
  SAT_REQUIRE(name, "conv_set_option: null name");
  if (!strcmp(name, "lean3")) { g_lean3 = value != 0; return SAT_OK; }
  if (!strcmp(name, "lean7")) { g_lean7 = value != 0; return SAT_OK; }
  if (!strcmp(name, "lean11")) { g_lean11 = value != 0; return SAT_OK; }
  if (!strcmp(name, "half_tile7")) { g_half_tile7 = value != 0; return SAT_OK; }
  if (!strcmp(name, "pair32s")) { g_pair32s = value != 0; return SAT_OK; }
  if (!strcmp(name, "trim_halo")) { g_trim_halo = value != 0; return SAT_OK; }
  if (!strcmp(name, "pair32w")) { pair32w_set(value); return SAT_OK; }
  if (!strcmp(name, "pair64w")) { pair64w_set(value); return SAT_OK; }
  if (!strcmp(name, "pair64_rpre")) { pair64_rpre_set(value); return SAT_OK; }
  if (!strcmp(name, "convpost_quad")) { convpost_set_quad(value); return SAT_OK; }
  if (!strcmp(name, "convring")) { convring_set(value); return SAT_OK; }
  if (!strcmp(name, "convring_blocks")) { convring_set_blocks(value); return SAT_OK; }
  if (!strcmp(name, "convring_wr")) { convring_set_wr(value); return SAT_OK; }
  if (!strcmp(name, "gemm_walk")) { gemm_walk_set(value); return SAT_OK; }
  if (!strcmp(name, "lean_balance")) { lean_set_balance(value); return SAT_OK; }
  if (!strcmp(name, "pair32s_waves")) { pair32s_set_waves(value); return SAT_OK; }
  if (!strcmp(name, "k1_gemm")) { g_k1_gemm = value < 0 ? 0 : value > 3 ? 3 : value; return SAT_OK; }
  set_error("conv_set_option: unknown option '%s'", name);
  return SAT_ERR_INVALID;
}
