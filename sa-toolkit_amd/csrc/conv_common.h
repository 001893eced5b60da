// Shared by the conv / GEMM kernel files: the launch descriptor (ConvArgs), the split-f16 typedefs and the common
// epilogue (bias, residual / bypass, folded BatchNorm, ReLU / GELU, MRF accumulation, f32 and split-plane stores).
#pragma once
#include "common.h"

namespace sat {


typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// power-of-two scales of the e4m3 cross-term operands (exact): x_lo * 2^10, W_hi * 2^6, W_lo * 2^16;
// E8M0 scale bytes of v_mfma_scale (value 2^(e - 127)) undo them: lanes 0-31 carry the K block
// "W_lo8 . x_hi8", lanes 32-63 the block "W_hi8 . x_lo8"
constexpr float F8_XLO_SCALE = 1024.f;
constexpr int F8_E_XHI = 127, F8_E_XLO = 127 - 10, F8_E_WHI = 127 - 6, F8_E_WLO = 127 - 16;
// SAT_CONV_F16F8R (conv_ring16.hip): the weights carry the SAT_CONV_F16X3 layer scale (largest |w'| in [2^9, 2^10)), so
// e4m3(W_hi * 2^-2) stays below 448 and e4m3(W_lo * 2^9) (|W_lo| < 2^-1) below 256
constexpr int F8R_E_WHI = 127 + 2, F8R_E_WLO = 127 - 9;

__device__ __forceinline__ unsigned pack_e4m3x4(float a, float b, float c, float d) {
  // OCP e4m3 saturates at 448; clamp first so an out-of-range activation degrades gracefully (v_med3_f32: one instruction)
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

// the same for e5m2 (bf8: f16's exponent range with two mantissa bits; saturates at 57344) — the ACTIVATION operands of SAT_CONV_F16F8R:
// e4m3's normal range starts at 2^-6, which a layer's inner activations can sit far below (a ResBlock whose conv1 is scaled down against its
// conv2: the cross terms then degrade to plain-f16 accuracy, 2e-5 RMS on the waveform); e5m2 carries every f16 magnitude at 2^-3 relative
__device__ __forceinline__ unsigned pack_e5m2x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -57344.f, 57344.f);
  b = __builtin_amdgcn_fmed3f(b, -57344.f, 57344.f);
  c = __builtin_amdgcn_fmed3f(c, -57344.f, 57344.f);
  d = __builtin_amdgcn_fmed3f(d, -57344.f, 57344.f);
  int w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
  return (unsigned)w;
}

// e5m2(lo * 2^10) of two lo halves straight from their packed f16 pair: v_cvt_scalef32_pk_bf8_f16 converts src / scale (measured on all
// 65536 patterns, tools/scratch/probe_cvt_bf8.hip: identical to the f32 path except that it overflows to Inf where the clamp saturates),
// and a lo half — |lo| <= ulp(hi) / 2... < 2^-10 |x| — times 2^10 never reaches e5m2's 57344 while |x| is an f16: no clamp needed.
// `old` carries the other half of the dword (op_sel picks the destination half).
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
template <class H2>       // (the 2 x f16 vector type v_cvt_pkrtz returns)
__device__ __forceinline__ unsigned pack_e5m2_lo_x4(H2 l01, H2 l23) {
  s16x2_t r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r, __builtin_bit_cast(f16x2_t, l01), 1.0f / F8_XLO_SCALE, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r, __builtin_bit_cast(f16x2_t, l23), 1.0f / F8_XLO_SCALE, true);
  return (unsigned)__builtin_bit_cast(int, r);
}

constexpr int CI_CHUNK = 16;  // input channels staged per K-chunk (8 MFMA k-pairs)

struct ConvArgs {
  const float* x;
  const float* w;   // exact-f32 packing, or (split-f16 mode) the f16 hi|lo packing reinterpreted
  float* y;
  const float* bias;
  const float* res;
  const float* ch_scale;
  const float* ch_shift;
  long long x_bs, x_cs, y_bs, y_cs, r_bs, r_cs;
  long long w_gs;  // packed weight elements per group
  int cin_g, T_in, rows_g, cout_g, T_q;
  int ksize, dil, stride, pad_left, up;
  int cin_pad, co_pad, xw, co_tiles_g;
  int bal, bal_fpp, bal_hpp;   // conv_lean.hip: 1-D balanced grid (bal = (row tile, utterance) pairs, 0 = plain 3-D grid) — full 256-column tiles first (fpp per (row tile, utterance)), then the ragged end of every row as 1 or 2 half tiles
  int in_lrelu, relu, accum, gelu, res_after, relu_first;
  float in_slope, accum_div, res_scale;
  int res_toff, res_tstride;
  const void* w2;        // fused pair: packed split-f16 weights of the second conv
  const float* bias1;    // fused pair: bias of the first conv
  int fast_epi;  // up == 1 and every (utterance, group) slab addressable with 31-bit byte offsets
  const void* x16;       // input as split planes (see satools_hip.h), or null
  void* y16;             // output as split planes, or null
  float y16_slope;
  int no_y;
  const void* res16;     // residual as SAT_SPLIT_F16 planes of lrelu(r, res16_slope) (inverted on the fly), or null
  float res16_inv;       // 1 / slope
  int res16_T, res16_toff;   // row length of the residual planes (0 = T_q) and the position of output column 0 in them (the TDNNF bypass: the layer input, identity_lidx frames in)
  int f8;                // planes kernels: cross terms hi*lo + lo*hi on the block-scaled e4m3 MFMA
  int poly_planes;       // up > 1, planes only: the LDS-transposed polyphase epilogue
  int up_grouped;        // up = 4, rows (16-channel group, phase, channel): conv_ring16.hip's upsampler form
  unsigned up_zero_taps; // its all-zero (tap slot, phase) pairs, bit slot * 4 + phase
  int y16_f8;            // output planes carry (hi f16 | e4m3(hi) | e4m3(lo * 2^10)) instead of (hi f16 | lo f16)
  int k1_wrap;           // ring16 GEMM: K chunks >= k1_wrap read plane chunk (c - k1_wrap) one position later; 0 = none
  int pp_tiles_t, pp_total, pp_per_xcd, pp_nslots;   // persistent pair kernel: tiles per utterance / in all / per XCD, blocks per XCD
  float w_descale;       // split-f16: the packed weights carry a power-of-two factor (packing.py: their largest magnitude moved
                         // into the middle of the f16 range, so that lo = f16(w - hi) is a normal number for every weight
                         // that matters); the accumulator is multiplied by its inverse before the bias — exact
  float w_descale1;      // fused pair: the same for the first conv
  int f8r;               // SAT_CONV_F16F8R (conv_ring16.hip): hi*hi on the f16 MFMA, the cross terms of a pair of taps on the e4m3 MFMA
  const void* x8;        // its e4m3 sidecar of x16
  void* y8;              // e4m3 sidecar of y16 to write, or null
  int y16_hi_only;       // with y8: the lo units of y16 are not stored
  int no_store;          // with accum: y is read (the running MRF sum) but the new sum is not stored (sat_conv1d_desc.accum_no_store)
#ifdef SAT_STAMPS
  long long* dbg;        // diagnostic build (tools/stamp_conv.hip): per-block, per-chunk phase time stamps
#endif
};
#ifdef SAT_STAMPS
long long* g_stamp_buffer = nullptr;
int g_stamp_variant = 0;   // 1: no output stores, 2: no residual, 3: neither
#define SAT_STAMP(i) do { \
    __builtin_amdgcn_sched_barrier(0); \
    unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); \
    if (p.dbg && tid == 0) p.dbg[((long long)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (p.cin_pad / CI_CHUNK) + chunk) * 8 + (i)] = (long long)t_; \
  } while (0)
// block-level record in the slots 6/7 of chunk 0 and 1: 100 MHz wall clock at start/end, HW_ID, XCC_ID
#define SAT_STAMP_BLOCK(i, expr) do { \
    if (p.dbg && tid == 0) p.dbg[((long long)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (p.cin_pad / CI_CHUNK)) * 8 + (i)] = (long long)(expr); \
  } while (0)
#else
#define SAT_STAMP(i)
#define SAT_STAMP_BLOCK(i, expr)
#endif

// ---- epilogue shared by the exact-f32 and the split-f16 kernels: bias, residual / bypass, folded
// BatchNorm, ReLU, MRF accumulation, (polyphase) store ----
// residual of the whole wave tile fetched ahead of the epilogue (fast path only): issued before the last
// K-chunk's MFMA phase, the HBM round trip hides under the matrix work instead of stalling the epilogue
template <int MT, int NT>
__device__ __forceinline__ void epilogue_prefetch_res(const ConvArgs& p, float (&rpre)[MT][NT][16], int b, int g, int co_w,
                                                      int q_w, int l31, int lh, int q_step = 32, int q_end = 0x7fffffff) {
  const long long rb = (long long)b * p.r_bs + (long long)(g * p.cout_g) * p.r_cs;
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res ? p.res + rb : p.y), 0, p.res ? (unsigned)(p.rows_g * p.r_cs * 4) : 0u, 0x00020000);
  const int r_rb = (int)p.r_cs * 4;
  if (p.res16) {
    // residual from the split planes: per 4 consecutive rows the 8-byte hi and lo words of this column (raw)
    const int rT = p.res16_T > 0 ? p.res16_T : p.T_q, rO = p.res16_toff;
    const __amdgpu_buffer_rsrc_t r16rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.res16 + (long long)b * p.rows_g * rT * 4), 0, (unsigned)(p.rows_g * rT * 4), 0x00020000);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int q = q_w + n * q_step + l31;
        const bool qok = q < p.T_q && q < q_end;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int chunk = ((co_w + m * 32) >> 4) + (rg >> 1);
          const unsigned off = (qok && chunk * 16 < p.rows_g) ? (unsigned)(((chunk * 4 + (rg & 1)) * rT + q + rO) * 16 + 8 * lh) : 0x80000000u;
          const uint4 hv4 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r16rs, off, 0, 0));   // narrowed to 8 bytes (the b64 builtin of this hipcc loads one dword)
          const unsigned hv[2] = {hv4.x, hv4.y};
          const uint4 lv4 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r16rs, off, 2 * rT * 16, 0));
          const unsigned lv[2] = {lv4.x, lv4.y};
          rpre[m][n][4 * rg + 0] = __builtin_bit_cast(float, hv[0]);
          rpre[m][n][4 * rg + 1] = __builtin_bit_cast(float, hv[1]);
          rpre[m][n][4 * rg + 2] = __builtin_bit_cast(float, lv[0]);
          rpre[m][n][4 * rg + 3] = __builtin_bit_cast(float, lv[1]);
        }
      }
    return;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row0 = co_w + m * 32 + 4 * lh;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int q = q_w + n * q_step + l31;
      const bool qok = q < p.T_q && q < q_end;
      const unsigned roff = qok ? (unsigned)(row0 * r_rb + (q * p.res_tstride + p.res_toff) * 4) : 0x80000000u;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rpre[m][n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rrs, roff + ((r & 3) + 8 * (r >> 2)) * r_rb, 0, 0));
    }
  }
}

// raw plane words (hi01, hi23, lo01, lo23 of four consecutive rows) -> the four residual values
__device__ __forceinline__ void decode_res16(float w0, float w1, float w2, float w3, float inv_slope, float (&out)[4]) {
  const unsigned h01 = __builtin_bit_cast(unsigned, w0), h23 = __builtin_bit_cast(unsigned, w1);
  const unsigned l01 = __builtin_bit_cast(unsigned, w2), l23 = __builtin_bit_cast(unsigned, w3);
  out[0] = mix_add_halves<false>(h01, l01);          // (float)hi + (float)lo in one v_fma_mix_f32 (common.h)
  out[1] = mix_add_halves<true>(h01, l01);
  out[2] = mix_add_halves<false>(h23, l23);
  out[3] = mix_add_halves<true>(h23, l23);
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = lrelu_undo_min(out[k], inv_slope);
}

// HAS_BN = false compiles the folded-BatchNorm step out of the fast path (32 registers of scale / shift per row tile):
// for kernels that must fit 168 VGPRs and are only dispatched without ch_scale
template <int MT, int NT, bool RPRE = false, bool HAS_BN = true>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[MT][NT], int b, int g, int co_w,
                                              int q_w, int l31, int lh, int q_step = 32, int q_end = 0x7fffffff,
                                              float (*rpre)[NT][16] = nullptr) {
  const int up = p.up;
  if (p.fast_epi) {
    // plain conv (up == 1): every row of this (utterance, group) sits behind one buffer descriptor
    // whose range check masks rows >= rows_g; lanes past T_q get an out-of-range offset.  Loads of
    // a 32x32 sub-tile (residual, accumulator) are issued back to back before the arithmetic, so
    // the epilogue pays one memory round trip per sub-tile instead of one per element.
    const unsigned OOB = 0x80000000u;
    const long long yb = (long long)b * p.y_bs + (long long)(g * p.cout_g) * p.y_cs;
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + yb), 0, (unsigned)(p.rows_g * p.y_cs * 4), 0x00020000);
    const long long rb = (long long)b * p.r_bs + (long long)(g * p.cout_g) * p.r_cs;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.res ? p.res + rb : p.y), 0, p.res ? (unsigned)(p.rows_g * p.r_cs * 4) : 0u, 0x00020000);
    const unsigned chn = (unsigned)(p.rows_g * 4);
    const int chb = g * p.cout_g;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.bias ? p.bias + chb : p.y), 0, p.bias ? chn : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t scs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.ch_scale ? p.ch_scale + chb : p.y), 0, p.ch_scale ? chn : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t shs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.ch_shift ? p.ch_shift + chb : p.y), 0, p.ch_shift ? chn : 0u, 0x00020000);
    const int y_rb = (int)p.y_cs * 4, r_rb = (int)p.r_cs * 4;  // bytes per row
    const int rT = p.res16_T > 0 ? p.res16_T : p.T_q, rO = p.res16_toff;
    const __amdgpu_buffer_rsrc_t r16rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.res16 ? (char*)p.res16 + (long long)b * p.rows_g * rT * 4 : (char*)p.y), 0,
        p.res16 ? (unsigned)(p.rows_g * rT * 4) : 0u, 0x00020000);
    const bool has_res = p.res != nullptr || p.res16 != nullptr;
    const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.y16 ? (char*)p.y16 + (long long)b * p.rows_g * p.T_q * 4 : (char*)p.y), 0,
        p.y16 ? (unsigned)(p.rows_g * p.T_q * 4) : 0u, 0x00020000);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row0 = co_w + m * 32 + 4 * lh;
      float bi[16], sc[16], sh[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        bi[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, row0 * 4 + ((r & 3) + 8 * (r >> 2)) * 4, 0, 0));
      if (HAS_BN && p.ch_scale) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = ((r & 3) + 8 * (r >> 2)) * 4;
          sc[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(scs, row0 * 4 + ro, 0, 0));
          sh[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(shs, row0 * 4 + ro, 0, 0));
        }
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int q = q_w + n * q_step + l31;
        const bool qok = q < p.T_q && q < q_end;
        const unsigned yoff = qok ? (unsigned)(row0 * y_rb + q * 4) : OOB;
        const unsigned roff = qok ? (unsigned)(row0 * r_rb + (q * p.res_tstride + p.res_toff) * 4) : OOB;
        float rv[16], yv[16];
        if (p.res16) {
          float raw[16];
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            if constexpr (RPRE) {
#pragma unroll
              for (int k = 0; k < 4; ++k) raw[4 * rg + k] = rpre[m][n][4 * rg + k];
            } else {
              const int chunk = ((co_w + m * 32) >> 4) + (rg >> 1);
              const unsigned off = (qok && chunk * 16 < p.rows_g) ? (unsigned)(((chunk * 4 + (rg & 1)) * rT + q + rO) * 16 + 8 * lh) : OOB;
              const uint4 hv4 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r16rs, off, 0, 0));   // narrowed to 8 bytes (the b64 builtin of this hipcc loads one dword)
          const unsigned hv[2] = {hv4.x, hv4.y};
              const uint4 lv4 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r16rs, off, 2 * rT * 16, 0));
          const unsigned lv[2] = {lv4.x, lv4.y};
              raw[4 * rg + 0] = __builtin_bit_cast(float, hv[0]);
              raw[4 * rg + 1] = __builtin_bit_cast(float, hv[1]);
              raw[4 * rg + 2] = __builtin_bit_cast(float, lv[0]);
              raw[4 * rg + 3] = __builtin_bit_cast(float, lv[1]);
            }
          }
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            float o[4];
            decode_res16(raw[4 * rg], raw[4 * rg + 1], raw[4 * rg + 2], raw[4 * rg + 3], p.res16_inv, o);
#pragma unroll
            for (int k = 0; k < 4; ++k) rv[4 * rg + k] = o[k];
          }
        } else if (p.res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (RPRE)
              rv[r] = rpre[m][n][r];
            else
              rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rrs, roff + ((r & 3) + 8 * (r >> 2)) * r_rb, 0, 0));
          }
        }
        if (p.accum) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            yv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  yrs, yoff + ((r & 3) + 8 * (r >> 2)) * y_rb, 0, 0));
        }
        // one wave-uniform branch per option around a 16-element pass (not per element: the per-element
        // form cost ~20k cycles of scalar branching per block); the order of operations is the desc's
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(acc[m][n][r], p.w_descale, bi[r]);   // (the product is exact: one rounding, that of the sum)
        if (has_res && !p.res_after) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += p.res_scale * rv[r];
        }
        if (p.relu && p.relu_first) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        }
        if (HAS_BN && p.ch_scale) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v[r] * sc[r] + sh[r];
        }
        if (p.relu && !p.relu_first) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        }
        if (p.gelu) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            v[r] = gelu_fast(v[r]);
          }
        }
        if (has_res && p.res_after) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += p.res_scale * rv[r];
        }
        if (p.accum) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = yv[r] + v[r];
        }
        if (p.accum_div != 0.f) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v[r] / p.accum_div;
        }
        if (!p.no_y && !p.no_store) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), yrs,
                                                  yoff + ((r & 3) + 8 * (r >> 2)) * y_rb, 0, 0);
        }
        if (p.y16) {
          // D layout: rows 8*rg + 4*lh + k (k < 4) of this lane = 8 bytes of the 16-byte unit
          // (chunk = row/16, half = rg & 1) at its column
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int chunk = ((co_w + m * 32) >> 4) + (rg >> 1);
            if (chunk * 16 >= p.rows_g) continue;   // wave-uniform: padding rows of a 32-row tile
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float t = v[4 * rg + k];
              u[k] = lrelu_max(t, p.y16_slope);
            }
            const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
            const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
            const unsigned off = qok ? (unsigned)(((chunk * 4 + (rg & 1)) * p.T_q + q) * 16 + 8 * lh) : OOB;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 hv;
            hv[0] = __builtin_bit_cast(unsigned, h01); hv[1] = __builtin_bit_cast(unsigned, h23);
            __builtin_amdgcn_raw_buffer_store_b64(hv, y16rs, off, 0, 0);
            if (p.y16_f8) {
              // planes 2 / 3 of the chunk: one byte per channel, these four channels = bytes 8*(rg&1) + 4*lh ..
              const float hf0 = (float)h01[0], hf1 = (float)h01[1], hf2 = (float)h23[0], hf3 = (float)h23[1];
              const unsigned x8h = pack_e4m3x4(hf0, hf1, hf2, hf3);
              const unsigned x8l = pack_e4m3x4((u[0] - hf0) * F8_XLO_SCALE, (u[1] - hf1) * F8_XLO_SCALE,
                                               (u[2] - hf2) * F8_XLO_SCALE, (u[3] - hf3) * F8_XLO_SCALE);
              const unsigned off8 = qok ? (unsigned)(((chunk * 4 + 2) * p.T_q + q) * 16 + 8 * (rg & 1) + 4 * lh) : OOB;
              __builtin_amdgcn_raw_buffer_store_b32(x8h, y16rs, off8, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b32(x8l, y16rs, off8, p.T_q * 16, 0);
            } else {
              const auto l01 = split_lo2(h01, u[0], u[1]);
              const auto l23 = split_lo2(h23, u[2], u[3]);
              u32x2 lv;
              lv[0] = __builtin_bit_cast(unsigned, l01); lv[1] = __builtin_bit_cast(unsigned, l23);
              __builtin_amdgcn_raw_buffer_store_b64(lv, y16rs, off, 2 * p.T_q * 16, 0);
            }
          }
        }
      }
    }
    return;
  }
  // general path: polyphase rows (transposed conv) or tensors too large for 32-bit row offsets
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = co_w + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;  // row inside group
      if (row >= p.rows_g) continue;
      int co_l, ph;
      if (up == 1) {
        co_l = row;
        ph = 0;
      } else {
        co_l = row / up;
        ph = row - co_l * up;
      }
      const int co = g * p.cout_g + co_l;
      const float bias = p.bias ? p.bias[co] : 0.f;
      float sc = 1.f, sh = 0.f;
      if (p.ch_scale) {
        sc = p.ch_scale[co];
        sh = p.ch_shift[co];
      }
      float* __restrict__ yrow = p.y + (long long)b * p.y_bs + (long long)co * p.y_cs;
      const float* __restrict__ rrow =
          p.res ? p.res + (long long)b * p.r_bs + (long long)co * p.r_cs : nullptr;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int q = q_w + n * q_step + l31;
        if (q >= p.T_q || q >= q_end) continue;
        const int t = q * up + ph;
        float v = __builtin_fmaf(acc[m][n][r], p.w_descale, bias);
        if (rrow && !p.res_after) v += p.res_scale * rrow[(long long)t * p.res_tstride + p.res_toff];
        if (p.relu && p.relu_first) v = v > 0.f ? v : 0.f;
        if (p.ch_scale) v = v * sc + sh;
        if (p.relu && !p.relu_first) v = v > 0.f ? v : 0.f;
        if (p.gelu) v = gelu_fast(v);
        if (rrow && p.res_after) v += p.res_scale * rrow[(long long)t * p.res_tstride + p.res_toff];
        if (p.accum) v = yrow[t] + v;
        if (p.accum_div != 0.f) v = v / p.accum_div;
        yrow[t] = v;
      }
    }
  }
}


// ---- the same epilogue for kernels on the 16x16x32 MFMA shape (gemm_ring.hip), fast path only
// (up == 1, 31-bit slab offsets, split-f16 planes).  D layout of a 16 x 16 tile: lane (li = lane & 15, lg = lane >> 4)
// holds rows 4 lg + r (r < 4) at column li, i.e. of the wave tile rows co_w + 16 m + 4 lg + r at q_w + 16 n + li: four
// consecutive channels = 8 bytes of the plane unit (chunk co / 16, half lg >> 1) at byte 8 (lg & 1).
// The order of operations is conv_epilogue's.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// acc += a x b on v_mfma_f32_16x16x32_f16 with the accumulator PINNED: destination = the C operand's registers.  Through
// the builtin, hipcc (ROCm 7.2) let the accumulators of the two-steps-per-trip ring loops alternate between two register
// sets — `v_mfma d, a, b, c` with d != c for 302 of the 480 MFMAs of gemm_f16x3_ring16_kernel: a dependent chain that
// does not accumulate in place loses its back-to-back issue, and a second copy of every accumulator stays live.
// The compiler does not see an MFMA in this statement: it still orders it behind the loads of its operands (register
// dependences of inline asm are tracked), but it inserts no wait states for the matrix pipe's result latency — the
// kernel calls mfma16_drain() between its last MFMA and the first vector instruction that reads an accumulator.
__device__ __forceinline__ void mfma16_acc(f32x4& acc, const h8& a, const h8& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#endif
}
// the same for the block-scaled 8-bit product of K = 128 (scale operands: one E8M0 byte per lane = its 32-byte K block)
__device__ __forceinline__ void mfma8_acc(f32x4& acc, const i32x8& a, const i32x8& b, int scale_a, int scale_b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (A = the weights: e4m3; B = the activations: e5m2, blgp 1)
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] blgp:1" : "+v"(acc) : "v"(a), "v"(b), "v"(scale_a), "v"(scale_b));
#endif
}
__device__ __forceinline__ void mfma16_drain() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
}
// the same, TIED to the accumulators it protects: every accumulator passes through an (empty) volatile asm statement behind the wait
// states, so no vector read, copy or spill of one can be scheduled in front of them whatever else the epilogue depends on
template <int MT, int NT>
__device__ __forceinline__ void mfma16_drain(f32x4 (&acc)[MT][NT]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(acc[m][n]));
#endif
}

// residual words of ONE 16-row strip of the wave tile (rows co_r .. co_r + 15): raw plane words (res16) or f32 values
template <int NT>
__device__ __forceinline__ void epilogue16_load_res_row(const ConvArgs& p, f32x4 (&out)[NT], int b, int co_r, int q_w, int li, int lg) {
  const unsigned OOB = 0x80000000u;
  if (p.res16) {
    const int rT = p.res16_T > 0 ? p.res16_T : p.T_q, rO = p.res16_toff;
    const __amdgpu_buffer_rsrc_t r16rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.res16 + (long long)b * p.rows_g * rT * 4), 0, (unsigned)(p.rows_g * rT * 4), 0x00020000);
    const int chunk = co_r >> 4;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int q = q_w + n * 16 + li;
      const unsigned off = (q < p.T_q && chunk * 16 < p.rows_g) ? (unsigned)(((chunk * 4 + (lg >> 1)) * rT + q + rO) * 16 + 8 * (lg & 1)) : OOB;
      // exactly the 8 bytes of this lane's four channels, as two dwords (the b64 load builtin of this hipcc loads one dword, and
      // a 16-byte load at an 8-byte offset would reach past the last unit of the plane image)
      const unsigned h0 = __builtin_amdgcn_raw_buffer_load_b32(r16rs, off, 0, 0), h1 = __builtin_amdgcn_raw_buffer_load_b32(r16rs, off, 4, 0);
      const unsigned l0 = __builtin_amdgcn_raw_buffer_load_b32(r16rs, off, 2 * rT * 16, 0), l1 = __builtin_amdgcn_raw_buffer_load_b32(r16rs, off, 2 * rT * 16 + 4, 0);
      out[n] = f32x4{__builtin_bit_cast(float, h0), __builtin_bit_cast(float, h1), __builtin_bit_cast(float, l0), __builtin_bit_cast(float, l1)};
    }
    return;
  }
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res ? p.res + (long long)b * p.r_bs : p.y), 0, p.res ? (unsigned)(p.rows_g * p.r_cs * 4) : 0u, 0x00020000);
  const int r_rb = (int)p.r_cs * 4;
  const int row0 = co_r + 4 * lg;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int q = q_w + n * 16 + li;
    const unsigned roff = q < p.T_q ? (unsigned)(row0 * r_rb + (q * p.res_tstride + p.res_toff) * 4) : OOB;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, roff + r * r_rb, 0, 0));
  }
}

template <int MT, int NT>
__device__ __forceinline__ void epilogue16_prefetch_res(const ConvArgs& p, f32x4 (&rpre)[MT][NT], int b, int co_w, int q_w,
                                                        int li, int lg) {
#pragma unroll
  for (int m = 0; m < MT; ++m) epilogue16_load_res_row<NT>(p, rpre[m], b, co_w + 16 * m, q_w, li, lg);
}

// HAS_BN = false compiles the folded-BatchNorm step out (32 registers of scale / shift): for kernels that are only dispatched without ch_scale
// ROWRES: the residual is requested strip by strip inside (the strip after the one being finished: 2 x NT registers
// instead of MT x NT); `rpre` is not read
template <int MT, int NT, bool HAS_BN = true, bool ROWRES = false>
__device__ __forceinline__ void conv_epilogue16(const ConvArgs& p, f32x4 (&acc)[MT][NT], f32x4 (&rpre)[MT][NT], int b, int co_w,
                                                int q_w, int li, int lg) {
  const unsigned OOB = 0x80000000u;
  const unsigned chn = (unsigned)(p.rows_g * 4);
  const __amdgpu_buffer_rsrc_t yrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long long)b * p.y_bs), 0, (unsigned)(p.rows_g * p.y_cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.y), 0, p.bias ? chn : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t scs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ch_scale ? p.ch_scale : p.y), 0, p.ch_scale ? chn : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t shs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ch_shift ? p.ch_shift : p.y), 0, p.ch_shift ? chn : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.y16 ? (char*)p.y16 + (long long)b * p.rows_g * p.T_q * 4 : (char*)p.y), 0,
      p.y16 ? (unsigned)(p.rows_g * p.T_q * 4) : 0u, 0x00020000);
  const int y_rb = (int)p.y_cs * 4;
  const bool has_res = p.res != nullptr || p.res16 != nullptr;
  // Every load of the epilogue is requested BEFORE the first store: vmcnt retires in issue order, so a load behind a row's
  // stores returns only once those stores have been acknowledged — with the per-row channel constants loaded row by row
  // (the first form of this epilogue) each row paid a full store round trip (stamps of conv1d_f16x3_ring16_kernel: 15-17 k
  // cycles for 80 outputs per lane).  Channel constants of all MT rows up front; the MRF accumulator of row m + 1 before
  // the stores of row m.
  float bi[MT][4], sc[MT][4], sh[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row0 = co_w + m * 16 + 4 * lg;
#pragma unroll
    for (int r = 0; r < 4; ++r) bi[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (row0 + r) * 4, 0, 0));
  }
  if (HAS_BN && p.ch_scale) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row0 = co_w + m * 16 + 4 * lg;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(scs, (row0 + r) * 4, 0, 0));
        sh[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(shs, (row0 + r) * 4, 0, 0));
      }
    }
  }
  f32x4 yn[NT];
  auto load_accum = [&](int m) __attribute__((always_inline)) {
    const int row0 = co_w + m * 16 + 4 * lg;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int q = q_w + n * 16 + li;
      const unsigned yoff = q < p.T_q ? (unsigned)(row0 * y_rb + q * 4) : OOB;
#pragma unroll
      for (int r = 0; r < 4; ++r) yn[n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, yoff + r * y_rb, 0, 0));
    }
  };
  f32x4 rn[NT];
  if (ROWRES && has_res) epilogue16_load_res_row<NT>(p, rn, b, co_w, q_w, li, lg);
  if (p.accum) load_accum(0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row0 = co_w + m * 16 + 4 * lg;
    const bool rows_ok = co_w + m * 16 < p.rows_g;       // wave-uniform: padding rows of the block's tile
    f32x4 yv[NT], rr[NT];
    if (ROWRES && has_res) {
#pragma unroll
      for (int n = 0; n < NT; ++n) rr[n] = rn[n];
      if (m + 1 < MT) epilogue16_load_res_row<NT>(p, rn, b, co_w + 16 * (m + 1), q_w, li, lg);
    }
    if (p.accum) {
#pragma unroll
      for (int n = 0; n < NT; ++n) yv[n] = yn[n];
      if (m + 1 < MT) load_accum(m + 1);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int q = q_w + n * 16 + li;
      const bool qok = q < p.T_q;
      const unsigned yoff = qok ? (unsigned)(row0 * y_rb + q * 4) : OOB;
      float rv[4], v[4];
      const f32x4 rw = ROWRES ? rr[n] : rpre[m][n];
      if (p.res16) {
        decode_res16(rw[0], rw[1], rw[2], rw[3], p.res16_inv, rv);
      } else if (p.res) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = rw[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[m][n][r], p.w_descale, bi[m][r]);
      if (has_res && !p.res_after) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p.res_scale * rv[r];
      }
      if (p.relu && p.relu_first) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      }
      if (HAS_BN && p.ch_scale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * sc[m][r] + sh[m][r];
      }
      if (p.relu && !p.relu_first) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      }
      if (p.gelu) gelu_fast4(v);
      if (has_res && p.res_after) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p.res_scale * rv[r];
      }
      if (p.accum) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = yv[n][r] + v[r];
      }
      if (p.accum_div != 0.f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] / p.accum_div;
      }
      if (!p.no_y && !p.no_store) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), yrs, yoff + r * y_rb, 0, 0);
      }
      if (p.y16 && rows_ok) {
        const int chunk = (co_w >> 4) + m;
        float u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = lrelu_max(v[r], p.y16_slope);
        const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
        const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
        const auto l01 = split_lo2(h01, u[0], u[1]);
        const auto l23 = split_lo2(h23, u[2], u[3]);
        const unsigned off = qok ? (unsigned)(((chunk * 4 + (lg >> 1)) * p.T_q + q) * 16 + 8 * (lg & 1)) : OOB;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 hv, lv;
        hv[0] = __builtin_bit_cast(unsigned, h01); hv[1] = __builtin_bit_cast(unsigned, h23);
        lv[0] = __builtin_bit_cast(unsigned, l01); lv[1] = __builtin_bit_cast(unsigned, l23);
        __builtin_amdgcn_raw_buffer_store_b64(hv, y16rs, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(lv, y16rs, off, 2 * p.T_q * 16, 0);
      }
    }
  }
}

// what conv_epilogue16 carries
inline bool epilogue16_supports(const ConvArgs& a) {
  return a.fast_epi && !a.y16_f8 && a.up == 1 && !a.poly_planes && a.rows_g % 4 == 0;
}

// 1x1 convolution on split planes through the LDS-DMA ring GEMM (gemm_ring.hip)
int launch_f16x3_ring(const ConvArgs& a, int B, hipStream_t s);
// fused ResBlock1 step for C = 64, 3 taps (pair64.hip)
int launch_pair64(const ConvArgs& a, int B, hipStream_t s);
bool pair64_supports(const ConvArgs& a);
int launch_pair32s(const ConvArgs& a, int B, hipStream_t s);   // pair32s.hip: the 3-tap step at C = 32 as a streaming kernel
bool pair32s_supports(const ConvArgs& a);
void lean_set_balance(int v);   // conv_lean.hip
void pair32s_set_waves(int n);
void pair32w_set(int v);
void pair64w_set(int v);
void pair64_rpre_set(int v);   // pair64.hip: residual words requested a conv ahead of the epilogue
int pair32_debug_stamps(long long* buf);
// three-blocks-per-CU form of the 3 / 7 / 11-tap conv tile on split planes (conv_lean.hip)
int launch_f16x3_lean(const ConvArgs& a, int B, hipStream_t s);
bool lean_supports(const ConvArgs& a);
int launch_f16x3_ring16(const ConvArgs& a, int B, hipStream_t s);   // the same ring on v_mfma_f32_16x16x32_f16
// k-tap convs at C >= 128 as an LDS-DMA ring on the 16x16x32 shape, 256 x 160 / 128 x 320 tiles (conv_ring16.hip)
int launch_f16x3_convring(const ConvArgs& a, int B, hipStream_t s);
int launch_f16x3_convring_ups(const ConvArgs& a, int B, hipStream_t s);      // ConvTranspose1d(stride 4), rows grouped by phase
bool convring_ups_supports(const ConvArgs& a);
int launch_f16x3_convring_multi(const ConvArgs* a, int njobs, int rotate, int B, hipStream_t s);   // up to three convs of one shape in one launch
bool convring_same_shape(const ConvArgs& a, const ConvArgs& b);
bool convring_supports(const ConvArgs& a, int B);
bool convring_wanted(int rows_g, int T_q, int B);
void convring_set(int v);
void convpost_set_quad(int v);  // hifigan.hip: four consecutive outputs per lane in the output stage
void convring_set_blocks(int v);
void convring_set_wr(int v);
int convring_debug_stamps(long long* buf);
bool ring16_supports(const ConvArgs& a);
// the persistent, multi-job form of that ring for plain Linear layers (gemm_walk16.hip)
int launch_f16x3_gemm_walk(const ConvArgs* a, int njobs, int B, hipStream_t s);
bool gemm_walk_supports(const ConvArgs& a);
bool gemm_walk_same_shape(const ConvArgs& a, const ConvArgs& b);
void gemm_walk_set(int v);
bool gemm_walk_wanted(long long tiles);      // a single GEMM: more tiles than CUs (option "gemm_walk" + 2: always)


}  // namespace sat
