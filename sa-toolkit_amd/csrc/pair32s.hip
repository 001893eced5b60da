// One ResBlock1 step of the generator's 32-channel stage at 3 taps,
//     y = x + conv2(lrelu(conv1(lrelu(x), dilation d)))          (satools/satools/hifigan/archi.py:118-122)
// as a streaming kernel on split planes.
//
// At 3 taps the step is HBM-bound (a stage tensor in, one out: 328 MB at 32 x 5 s, 55 us at 6 TB/s) and ran at twice that on
// the general fused step (three blocks per CU, register-staged loads, weights through LDS).  Here, like ups2.hip: one persistent
// 8-wave block per CU walks tiles of TQ output positions; the input planes of the NEXT tile arrive by LDS-DMA (inline asm, counted
// waits) while this one is multiplied; the fragments of both convs (2 x 3 taps x 2 row tiles x (hi, lo)) stay in registers for
// the block's whole walk (v_mfma_f32_16x16x32_f16: K = the 32 input channels of one tap); the inner activation lives in LDS as
// split planes; the residual is decoded from the input image; outputs leave from the accumulators (planes: 8 bytes of a unit per
// lane; f32: four channel rows per lane).
// Arithmetic: split-f16 (lo*hi, hi*lo, hi*hi per K step), accumulation order (tap; all 32 channels) — not the general step's
// (chunk; tap), so results agree with it to f32 rounding of the accumulation, not bit for bit.
#include <algorithm>

#include "conv_common.h"

namespace sat {

namespace {
// NW waves per block: 8 (one block per CU, tiles of 240 output positions) or 4 (two blocks per CU, tiles of 112: the two
// blocks' barriers and DMA waits fall at different times)
template <int NW> struct P32 {
  static constexpr int TQ = NW == 8 ? 240 : 112;       // output positions per tile
  static constexpr int PITCH = NW == 8 ? 256 : 128;    // units per plane row of an image
  static constexpr int ROWS = 8;                       // 2 channel chunks x 4 planes
  static constexpr int IMG = ROWS * PITCH;
};
constexpr int P32_HL = 6;        // the image starts HL positions left of the tile: 1 (second conv) + dilation <= 5 (first conv)
}  // namespace

struct P32Args {
  const void* x16;       // input planes [B][2][4][T][16 B] of lrelu(x, slope)
  void* y16;             // output planes of lrelu(y, y_slope), or null
  float* y;              // f32 output [B][32][T] (y_bs, y_cs), or null
  const void *w1, *w2;   // packed weights (SAT_CONV_F16X3, co_pad 64)
  const float *b1, *b2;
  int T, B, dil;
  long long y_bs, y_cs;
  float d1, d2, slope, inv_slope, y_slope, accum_div;
  int accum;
  int tiles_t, total, per_xcd, nslots;
  long long* dbg;        // diagnostics (sat_pair32_debug_stamps): block 0 of pairw_kernel records cycle counters, [step][stamp][wave]
};

template <bool Y16, bool YF, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) pair32s_kernel(const P32Args p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int TQ = P32<NW>::TQ, PITCH = P32<NW>::PITCH, HL = P32_HL, IMG = P32<NW>::IMG;
  constexpr int NSA = TQ / 16 + 1, NSB = TQ / 16;      // subtiles of the inner activation / of the output
  constexpr int SPS = (Y16 ? 4 : 0) + (YF ? 8 : 0);    // stores a wave issues per output subtile
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, j16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned OOB = 0x80000000u;
  uint4* const T1 = lds4 + 2 * IMG;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_end = min((xcd + 1) * p.per_xcd, p.total);
  int tile = xcd * p.per_xcd + slot;
  if (tile >= tile_end) return;

  auto stage = [&](int tl, uint4* img) __attribute__((always_inline)) {
    const int ub = __builtin_amdgcn_readfirstlane(tl / p.tiles_t);
    const int p0 = (tl - ub * p.tiles_t) * TQ;
    const i32x4 xrs = dma_rsrc((const char*)p.x16 + (long long)ub * 32 * p.T * 4, (unsigned)(32 * p.T * 4));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      constexpr int PPR = PITCH / 64;                   // 64-unit pieces per plane row: 32 (16) pieces over 8 (4) waves
      const int piece = wave + NW * r, row = piece / PPR, c = (piece % PPR) * 64 + lane;
      const int pos = p0 - HL + c;
      const unsigned voff = (pos >= 0 && pos < p.T) ? (unsigned)((row * p.T + pos) * 16) : OOB;   // zero padding of the first conv
      lds_dma16(img + row * PITCH + (piece % PPR) * 64, xrs, voff, 0u);
    }
  };
  stage(tile, lds4);

  // fragments of both convs: [tap][row tile] x (hi, lo); K group g = (chunk g >> 1, half g & 1) of the packing
  h8 a1h[3][2], a1l[3][2], a2h[3][2], a2l[3][2];
  {
    const uint4 *w1 = (const uint4*)p.w1, *w2 = (const uint4*)p.w2;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int u = ((((g >> 1) * 3 + t) * 2) * 2 + (g & 1)) * 64 + 16 * m + j16;
        a1h[t][m] = __builtin_bit_cast(h8, w1[u]);
        a1l[t][m] = __builtin_bit_cast(h8, w1[u + 128]);
        a2h[t][m] = __builtin_bit_cast(h8, w2[u]);
        a2l[t][m] = __builtin_bit_cast(h8, w2[u + 128]);
      }
  }
  float b1[2][4], b2[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      b1[m][k] = p.b1[16 * m + 4 * g + k];
      b2[m][k] = p.b2[16 * m + 4 * g + k];
    }

  int buf = 0;
  bool first = true;
  const int d = p.dil;
  // a K group's plane row: hi rows (chunk * 4 + half), lo rows + 2
  const int krow = (g >> 1) * 4 + (g & 1);
  for (;;) {
    const int b = __builtin_amdgcn_readfirstlane(tile / p.tiles_t);
    const int p0 = (tile - b * p.tiles_t) * TQ;
    const int next = tile + p.nslots;
    const bool more = next < tile_end;
    const uint4* img = lds4 + buf * IMG;
    // this tile's image has landed (its pieces were requested BEFORE the last tile's stores: a counted wait — loads, stores and
    // LDS-DMA retire in issue order); the barrier also says every wave is done with the other image and with T1
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wave == NW - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SPS) : "memory");
    first = false;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (more) stage(next, lds4 + (buf ^ 1) * IMG);

    // ---- first conv: inner activation column u <-> position p0 - 1 + u, image column u + HL - 1 + (tap - 1) d ----
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
      const int s = 2 * wave + i;
      if (s >= NSA) break;
      const uint4* xb = img + krow * PITCH + 16 * s + j16 + (HL - 1) - d;
      h8 bh[3], bl[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        bh[t] = __builtin_bit_cast(h8, xb[t * d]);
        bl[t] = __builtin_bit_cast(h8, xb[2 * PITCH + t * d]);
      }
      const int pos = p0 - 1 + 16 * s + j16;
      const bool inside = pos >= 0 && pos < p.T;          // zero padding of the second conv
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l[t][m], bh[t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h[t][m], bl[t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h[t][m], bh[t], acc, 0, 0, 0);
        }
        float u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = __builtin_fmaf(acc[k], p.d1, b1[m][k]);
          const float a = lrelu_max(v, p.slope);
          u[k] = inside ? a : 0.f;
        }
        const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
        const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
        const auto l01 = split_lo2(h01, u[0], u[1]);
        const auto l23 = split_lo2(h23, u[2], u[3]);
        // channels 16 m + 4 g ..: chunk m, half g >> 1, bytes 8 (g & 1) of the unit
        u32x2* dst = (u32x2*)(T1 + (m * 4 + (g >> 1)) * PITCH + 16 * s + j16) + (g & 1);
        dst[0] = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
        dst[2 * PITCH * 2] = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- second conv: output column o <-> position p0 + o, inner columns o + tap; residual = image column o + HL ----
    const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(Y16 ? (char*)p.y16 + (long long)b * 32 * p.T * 4 : (char*)p.x16), 0, Y16 ? (unsigned)(32 * p.T * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(YF ? (char*)(p.y + (long long)b * p.y_bs) : (char*)p.x16), 0, YF ? (unsigned)(32 * p.y_cs * 4) : 0u, 0x00020000);
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
      const int s = 2 * wave + i;
      if (s >= NSB) break;
      const uint4* tb = T1 + krow * PITCH + 16 * s + j16;
      h8 bh[3], bl[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        bh[t] = __builtin_bit_cast(h8, tb[t]);
        bl[t] = __builtin_bit_cast(h8, tb[2 * PITCH + t]);
      }
      const int pos = p0 + 16 * s + j16;
      const bool ok = pos < p.T;
      float yv[2][4];
      if (YF && p.accum) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            yv[m][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     yrs, ok ? (unsigned)(((16 * m + 4 * g + k) * p.y_cs + pos) * 4) : OOB, 0, 0));
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2l[t][m], bh[t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h[t][m], bl[t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2h[t][m], bh[t], acc, 0, 0, 0);
        }
        // residual: the block input, hi + lo of the image with the leaky-relu undone (conv_common.h decode_res16)
        // (read as four halves: hipcc of ROCm 7.2 miscompiles `bit_cast<half2>(u32x2 element)` here — both halves of the
        // pair come out as element 0, a 4-byte LDS read; DESIGN.md toolchain notes)
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4* src = (const h4*)(img + (m * 4 + (g >> 1)) * PITCH + 16 * s + j16 + HL) + (g & 1);
        const h4 rh = src[0], rl = src[2 * PITCH * 2];
        float rv[4];
        {
          typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
          const u32x2v hw = __builtin_bit_cast(u32x2v, rh), lw = __builtin_bit_cast(u32x2v, rl);      // (hi0 hi1 | hi2 hi3), (lo0 lo1 | lo2 lo3)
          rv[0] = mix_add_halves<false>(hw[0], lw[0]);
          rv[1] = mix_add_halves<true>(hw[0], lw[0]);
          rv[2] = mix_add_halves<false>(hw[1], lw[1]);
          rv[3] = mix_add_halves<true>(hw[1], lw[1]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) rv[k] = lrelu_undo_min(rv[k], p.inv_slope);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] = __builtin_fmaf(acc[k], p.d2, b2[m][k]) + rv[k];
          if (YF && p.accum) v[k] = yv[m][k] + v[k];
        }
        if (p.accum_div != 0.f) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = v[k] / p.accum_div;
        }
        if (YF) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), yrs,
                                                  ok ? (unsigned)(((16 * m + 4 * g + k) * p.y_cs + pos) * 4) : OOB, 0, 0);
        }
        if (Y16) {
          float u[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) u[k] = lrelu_max(v[k], p.y_slope);
          const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
          const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
          const auto l01 = split_lo2(h01, u[0], u[1]);
          const auto l23 = split_lo2(h23, u[2], u[3]);
          const unsigned off = ok ? (unsigned)((((m * 4 + (g >> 1)) * p.T + pos) * 16) + 8 * (g & 1)) : OOB;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)}, y16rs, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)}, y16rs, off, 2 * p.T * 16, 0);
        }
      }
    }
    if (!more) break;
    tile = next;
    buf ^= 1;
  }
}


// ------------------------------------------------------------------------------------------------
// 7 and 11 taps: the fragments of ONE conv (KS x 2 row tiles x (hi, lo) = 112 / 176 VGPRs) are all a wave can keep, so the
// waves specialise — waves 0-3 hold conv1 and produce the inner activation of tile n into one of two LDS buffers while waves
// 4-7 hold conv2 and turn the inner activation of tile n - 1 (the other buffer) into outputs: a two-stage pipeline with ONE
// barrier per tile, a conv1 and a conv2 wave on every SIMD.  LDS then only serves the B fragments (2 reads per 6 MFMAs,
// ~65 % of its bandwidth at full MFMA rate; with the weights passing through LDS the general fused step sits at the LDS
// limit).  Input images by LDS-DMA one tile ahead (issued and awaited by the conv1 waves only: a plain vmcnt(0), they have
// no other memory traffic); the conv2 waves fetch their residual words from the planes in global memory (L2: the image
// of the same tile went by a moment ago), so an image is dead once conv1 has read it and two buffers suffice.
// ------------------------------------------------------------------------------------------------
//
// C = 64 at 3 taps (one conv: 3 taps x 4 row tiles x 2 K steps x (hi, lo) = 192 VGPRs) runs the same pipeline with each role's
// four waves split 2 x 2 over (row half, column half): a wave keeps the fragments of two row tiles (96 VGPRs) and reads the B
// fragments of its columns (both row halves read them: twice the LDS reads, still ~65 % of the LDS at full MFMA rate).
namespace {
template <int C, int KS> struct P32W {
  static constexpr int TQ = C == 32 ? 240 : 112;    // output positions per tile
  static constexpr int HC = (KS - 1) / 2;           // taps left of the centre
  static constexpr int HL = HC * 6;                 // image start left of the tile: HC (conv2) + HC * dilation <= 5 HC (conv1)
  static constexpr int ROWS = C / 4;                // plane rows: C / 16 chunks x 4
  static constexpr int PITCH = C == 32 ? 320 : 128; // units per plane row of an image (TQ + 2 HL <= 300 / 124: five / two DMA pieces)
  static constexpr int IMG = ROWS * PITCH;          // 40 / 32 KB
  static constexpr int T1P = C == 32 ? 256 : 128;   // units per plane row of an inner-activation buffer (TQ + 2 HC <= 250 / 114)
  static constexpr int T1IMG = ROWS * T1P;          // 32 KB
};
}  // namespace

#ifndef P32W_DBG_SPLIT
#define P32W_DBG_SPLIT 0   // 1: the conv2 waves stamp [after MFMAs, after epilogue] of subtiles 0 and 1 instead of the four subtile ends
#endif
#ifndef P32W_SB
#define P32W_SB 1
#endif
constexpr int P32_DBG_STEPS = 8, P32_DBG_STAMPS = 6;
#define P32_STAMP(idx)                                                                                      \
  do {                                                                                                      \
    if (p.dbg && blockIdx.x == 0 && step < P32_DBG_STEPS) {                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      unsigned long long t_;                                                                                \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
      if (lane == 0) p.dbg[(step * P32_DBG_STAMPS + (idx)) * 8 + wave] = (long long)t_;                       \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
  } while (0)

template <int C, int KS, bool Y16, bool YF>
__global__ void __launch_bounds__(512, 1) pairw_kernel(const P32Args p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  using G = P32W<C, KS>;
  constexpr int TQ = G::TQ, HC = G::HC, HL = G::HL, PITCH = G::PITCH, IMG = G::IMG, T1P = G::T1P, T1IMG = G::T1IMG;
  constexpr int NSA = T1P / 16, NSB = TQ / 16;
  constexpr int KST = C / 32;                         // K steps (32 channels) per tap
  constexpr int CG = 4 / (C / 32);                    // column groups of a role's four waves (x C / 32 row halves)
  static_assert(NSA == 4 * CG, "four subtiles per wave");
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, j16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  const int rw = wave & 3;                            // wave index inside its role
  const int mh = C == 32 ? 0 : rw / CG, cg = C == 32 ? rw : rw % CG;   // its row half (two 16-row tiles: 2 mh, 2 mh + 1) and column group
  const unsigned OOB = 0x80000000u;
  uint4* const T1 = lds4 + 2 * IMG;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_end = min((xcd + 1) * p.per_xcd, p.total);
  const int tile0 = xcd * p.per_xcd + slot;
  if (tile0 >= tile_end) return;
  const int n_my = (tile_end - tile0 + p.nslots - 1) / p.nslots;

  // pieces r0 .. r1 - 1 of a tile's image: ROWS rows x PPR pieces of 64 units over the 4 producer waves (NPW = 10 / 8 each)
  constexpr int PPR = PITCH / 64, NPW = G::ROWS * PPR / 4;
  auto stage = [&](int tl, uint4* img, int r0, int r1) __attribute__((always_inline)) {
    const int ub = __builtin_amdgcn_readfirstlane(tl / p.tiles_t);
    const int p0 = (tl - ub * p.tiles_t) * TQ;
    const i32x4 xrs = dma_rsrc((const char*)p.x16 + (long long)ub * C * p.T * 4, (unsigned)(C * p.T * 4));
#pragma unroll
    for (int r = 0; r < NPW; ++r) {
      if (r < r0 || r >= r1) continue;
      const int piece = rw + 4 * r, row = piece / PPR, c = (piece % PPR) * 64 + lane;
      const int pos = p0 - HL + c;
      const unsigned voff = (pos >= 0 && pos < p.T) ? (unsigned)((row * p.T + pos) * 16) : OOB;
      lds_dma16(img + row * PITCH + (piece % PPR) * 64, xrs, voff, 0u);
    }
  };
  if (producer) stage(tile0, lds4, 0, NPW);

  // this role's fragments: [tap][K step][row tile of the wave's half] x (hi, lo); K group g of step ks = (chunk 2 ks + (g >> 1),
  // half g & 1) of the packing
  h8 ah[KS][KST][2], al[KS][KST][2];
  float4* const bias4 = (float4*)(T1 + 2 * T1IMG);   // [2][C / 4] the roles' biases, read back per epilogue (8 VGPRs the 11-tap consumer lacks)
  {
    const uint4* w = (const uint4*)(producer ? p.w1 : p.w2);
    const float* bp = producer ? p.b1 : p.b2;
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int ks = 0; ks < KST; ++ks)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int u = ((((2 * ks + (g >> 1)) * KS + t) * 2) * 2 + (g & 1)) * 64 + 16 * (2 * mh + m) + j16;
          ah[t][ks][m] = __builtin_bit_cast(h8, w[u]);
          al[t][ks][m] = __builtin_bit_cast(h8, w[u + 128]);
        }
    if (rw == 0 && lane < C) ((float*)(bias4 + (producer ? 0 : C / 4)))[lane] = bp[lane];
  }
  const float4* const my_bias = bias4 + (producer ? 0 : C / 4) + 8 * mh + g;      // channels 16 (2 mh + m) + 4 g ..: my_bias[4 m]
  const float descale = producer ? p.d1 : p.d2;
  const int d = p.dil;
  const int krow = (g >> 1) * 4 + (g & 1);           // a K group's plane row: hi rows (chunk * 4 + half), lo rows + 2

  // A role's four 16-column subtiles of a tile.  B fragments of tap t: base[t * step] (hi) and base[lo_off + t * step] (lo),
  // kept PD taps ahead of their MFMAs in a ring of PD register pairs that runs on across the subtiles (the first taps of
  // subtile i + 1 are requested under the last MFMAs of subtile i, so its epilogue and their LDS round trip overlap).  Both
  // loops are fully unrolled: the ring slot (i KS + t) % PD is a compile-time number.  (PD = 1, no real read-ahead, in the
  // 11-tap kernels with an f32 output: registers they do not have; the SIMD's other wave covers.)
  // With two K steps per tap (C = 64) the ring runs over (tap, K step): entry q = t KST + ks at base[t step + ks ks_off].
  constexpr int PD = KS == 11 ? (YF ? 1 : 2) : 3;
  constexpr int KQ = KS * KST;
  auto run_conv = [&](auto base_of, int step_, int ks_off, int lo_off, int ns, auto before, auto epi, auto stamp) __attribute__((always_inline)) {
    h8 ring_h[PD], ring_l[PD];
    const int s0 = 4 * cg;
    auto at = [&](int q) __attribute__((always_inline)) { return KST == 1 ? q * step_ : (q / KST) * step_ + (q % KST) * ks_off; };
    {
      const uint4* b0 = base_of(s0);
#pragma unroll
      for (int q = 0; q < PD; ++q) {
        ring_h[q] = __builtin_bit_cast(h8, b0[at(q)]);
        ring_l[q] = __builtin_bit_cast(h8, b0[lo_off + at(q)]);
      }
    }
    constexpr int UNR = PD > 1 ? 4 : 1;      // (a ring of one needs no compile-time slot: keep the loop rolled, it spares registers)
#pragma clang loop unroll_count(UNR)
    for (int i = 0; i < 4; ++i) {
      const int s_ = s0 + i;
      if (s_ < ns) {
        const uint4 *bs = base_of(s_), *nb = base_of(s_ + 1);
        const bool has_next = i < 3 && s_ + 1 < ns;
        before(i, s_, has_next);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const int slot = (i * KQ + q) % PD, t = q / KST, ks = q % KST;
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t][ks][0], ring_h[slot], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t][ks][1], ring_h[slot], acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][ks][0], ring_l[slot], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][ks][1], ring_l[slot], acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][ks][0], ring_h[slot], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][ks][1], ring_h[slot], acc[1], 0, 0, 0);
          if (P32W_SB) __builtin_amdgcn_sched_barrier(0);
          if (q + PD < KQ) {
            ring_h[slot] = __builtin_bit_cast(h8, bs[at(q + PD)]);
            ring_l[slot] = __builtin_bit_cast(h8, bs[lo_off + at(q + PD)]);
          } else if (has_next) {
            ring_h[slot] = __builtin_bit_cast(h8, nb[at(q + PD - KQ)]);
            ring_l[slot] = __builtin_bit_cast(h8, nb[lo_off + at(q + PD - KQ)]);
          }
          if (P32W_SB) __builtin_amdgcn_sched_barrier(0);
        }
        if (P32W_DBG_SPLIT && !producer) { if (i < 2) stamp(2 + 2 * i); }
        epi(i, s_, acc);
        if (P32W_DBG_SPLIT && !producer) { if (i < 2) stamp(3 + 2 * i); } else stamp(2 + i);
      }
    }
  };

  // conv2 waves: residual words of the lane's four channels per row tile (hi, lo) from the planes in global memory (~1.5-2 us
  // away under this kernel's own traffic), requested one subtile ahead (RD = 2 register sets) where registers allow, and the
  // first subtile's of a tile at the end of the previous step, across the barrier
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int RD = KS == 11 ? 1 : 2;
  constexpr bool XB = (KS == 7 && !YF) || C == 64;      // (the other kernels spill with these registers live across the barrier)
  u32x4 rh[RD][2], rl[RD][2];
  float yv[2][4];
  auto res_load = [&](int set, int tile, int s_) __attribute__((always_inline)) {
    const int b = __builtin_amdgcn_readfirstlane(tile / p.tiles_t);
    const int pos = (tile - b * p.tiles_t) * TQ + 16 * s_ + j16;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x16 + (long long)b * C * p.T * 4), 0, (unsigned)(C * p.T * 4), 0x00020000);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const unsigned off = pos < p.T ? (unsigned)(((((2 * mh + m) * 4 + (g >> 1)) * p.T + pos) * 16) + 8 * (g & 1)) : OOB;
      rh[set][m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));              // (narrowed to 8 bytes)
      rl[set][m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 2 * p.T * 16, 0));
    }
  };
  if (XB && !producer) res_load(0, tile0, 4 * cg);

  for (int step = 0; step <= n_my; ++step) {
    P32_STAMP(0);
    if (producer && step < n_my) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // image of tile `step` (this wave's pieces)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    P32_STAMP(1);
    if (producer) {
      if (step >= n_my) continue;
      const int tile = tile0 + step * p.nslots;
      const int b = __builtin_amdgcn_readfirstlane(tile / p.tiles_t);
      const int p0 = (tile - b * p.tiles_t) * TQ;
      const uint4* img = lds4 + (step & 1) * IMG;
      uint4* t1 = T1 + (step & 1) * T1IMG;
      // inner column u <-> position p0 - HC + u; tap t reads image column u + (HL - HC - HC d) + t d
      const uint4* xb = img + krow * PITCH + j16 + (HL - HC) - HC * d;
      run_conv([&](int s_) { return xb + 16 * s_; }, d, 8 * PITCH, 2 * PITCH, NSA, [](int, int, bool) {},
               [&](int, int s_, f32x4 (&acc)[2]) __attribute__((always_inline)) {
                 const int pos = p0 - HC + 16 * s_ + j16;
                 const bool inside = pos >= 0 && pos < p.T;          // zero padding of the second conv
#pragma unroll
                 for (int m = 0; m < 2; ++m) {
                   const float4 bq = my_bias[4 * m];
                   const float bias_m[4] = {bq.x, bq.y, bq.z, bq.w};
                   float u[4];
#pragma unroll
                   for (int k = 0; k < 4; ++k) {
                     const float v = __builtin_fmaf(acc[m][k], descale, bias_m[k]);
                     const float a = lrelu_max(v, p.slope);
                     u[k] = inside ? a : 0.f;
                   }
                   const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
                   const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
                   const auto l01 = split_lo2(h01, u[0], u[1]);
                   const auto l23 = split_lo2(h23, u[2], u[3]);
                   u32x2* dst = (u32x2*)(t1 + ((2 * mh + m) * 4 + (g >> 1)) * T1P + 16 * s_ + j16) + (g & 1);
                   dst[0] = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
                   dst[2 * T1P * 2] = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
                 }
               },
               [&](int idx) __attribute__((always_inline)) {
                 // the next tile's image, a few pieces after every subtile: ten at once (40 KB per CU) stall the issuing waves
                 // for ~3500 cycles and hold up the conv2 waves' residual loads (cycle stamps, tools/stamp_pair32.py)
                 if (step + 1 < n_my) {
                   const int i = idx - 2;
                   const int r0 = NPW == 10 ? (i == 0 ? 0 : i == 1 ? 3 : i == 2 ? 6 : 8) : 2 * i;
                   const int r1 = NPW == 10 ? (i == 0 ? 3 : i == 1 ? 6 : i == 2 ? 8 : 10) : 2 * i + 2;
                   stage(tile0 + (step + 1) * p.nslots, lds4 + ((step + 1) & 1) * IMG, r0, r1);
                 }
                 P32_STAMP(idx);
               });
    } else {
      if (step == 0) continue;
      const int tile = tile0 + (step - 1) * p.nslots;
      const int b = __builtin_amdgcn_readfirstlane(tile / p.tiles_t);
      const int p0 = (tile - b * p.tiles_t) * TQ;
      const uint4* t1 = T1 + ((step - 1) & 1) * T1IMG;
      const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(Y16 ? (char*)p.y16 + (long long)b * C * p.T * 4 : (char*)p.x16), 0, Y16 ? (unsigned)(C * p.T * 4) : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(YF ? (char*)(p.y + (long long)b * p.y_bs) : (char*)p.x16), 0, YF ? (unsigned)(C * p.y_cs * 4) : 0u, 0x00020000);
      // output column o <-> position p0 + o, inner columns o + tap
      const uint4* tb = t1 + krow * T1P + j16;
      run_conv([&](int s_) { return tb + 16 * s_; }, 1, 8 * T1P, 2 * T1P, NSB,
               [&](int i, int s_, bool has_next) __attribute__((always_inline)) {
                 if (!XB && i == 0) res_load(0, tile, s_);
                 if (RD == 2) {
                   if (has_next) res_load((i + 1) & 1, tile, s_ + 1);
                 } else if (i > 0) {
                   res_load(0, tile, s_);
                 }
                 if (YF && p.accum) {
                   const int pos = p0 + 16 * s_ + j16;
#pragma unroll
                   for (int m = 0; m < 2; ++m)
#pragma unroll
                     for (int k = 0; k < 4; ++k)
                       yv[m][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                yrs, pos < p.T ? (unsigned)(((16 * (2 * mh + m) + 4 * g + k) * p.y_cs + pos) * 4) : OOB, 0, 0));
                 }
               },
               [&](int i, int s_, f32x4 (&acc)[2]) __attribute__((always_inline)) {
                 const int pos = p0 + 16 * s_ + j16;
                 const bool ok = pos < p.T;
                 const int set = RD == 2 ? (i & 1) : 0;
#pragma unroll
                 for (int m = 0; m < 2; ++m) {
                   const float rsum[4] = {mix_add_halves<false>(rh[set][m][0], rl[set][m][0]), mix_add_halves<true>(rh[set][m][0], rl[set][m][0]),
                                          mix_add_halves<false>(rh[set][m][1], rl[set][m][1]), mix_add_halves<true>(rh[set][m][1], rl[set][m][1])};
                   const float4 bq = my_bias[4 * m];
                   const float bias_m[4] = {bq.x, bq.y, bq.z, bq.w};
                   float v[4];
#pragma unroll
                   for (int k = 0; k < 4; ++k) {
                     float r = rsum[k];
                     r = lrelu_undo_min(r, p.inv_slope);
                     v[k] = __builtin_fmaf(acc[m][k], descale, bias_m[k]) + r;
                     if (YF && p.accum) v[k] = yv[m][k] + v[k];
                   }
                   if (p.accum_div != 0.f) {
#pragma unroll
                     for (int k = 0; k < 4; ++k) v[k] = v[k] / p.accum_div;
                   }
                   if (YF) {
#pragma unroll
                     for (int k = 0; k < 4; ++k)
                       __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), yrs,
                                                             ok ? (unsigned)(((16 * (2 * mh + m) + 4 * g + k) * p.y_cs + pos) * 4) : OOB, 0, 0);
                   }
                   if (Y16) {
                     float u[4];
#pragma unroll
                     for (int k = 0; k < 4; ++k) u[k] = lrelu_max(v[k], p.y_slope);
                     const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
                     const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
                     const auto l01 = split_lo2(h01, u[0], u[1]);
                     const auto l23 = split_lo2(h23, u[2], u[3]);
                     const unsigned off = ok ? (unsigned)(((((2 * mh + m) * 4 + (g >> 1)) * p.T + pos) * 16) + 8 * (g & 1)) : OOB;
                     __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)}, y16rs, off, 0, 0);
                     __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)}, y16rs, off, 2 * p.T * 16, 0);
                   }
                 }
               },
               [&](int idx) __attribute__((always_inline)) { P32_STAMP(idx); });
      if (XB && step < n_my) res_load(0, tile0 + step * p.nslots, 4 * cg);      // the next tile's first subtile (set 0 is free again)
    }
  }
}

static long long* g_pair32_dbg = nullptr;
int pair32_debug_stamps(long long* buf) {
  g_pair32_dbg = buf;
  return P32_DBG_STEPS * P32_DBG_STAMPS * 8;
}

template <int C, int KS, bool Y16, bool YF>
static int launch_pairw_t(P32Args& p, hipStream_t s) {
  using G = P32W<C, KS>;
  const size_t lds_bytes = (size_t)(2 * G::IMG + 2 * G::T1IMG + 32) * 16;   // 144 / 128 KB + the bias table
  auto kern = pairw_kernel<C, KS, Y16, YF>;
  static std::atomic<uint64_t> attr_done{0};
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  p.tiles_t = ceil_div(p.T, G::TQ);
  p.total = p.tiles_t * p.B;
  p.per_xcd = ceil_div(p.total, 8);
  p.nslots = std::max(1, std::min(32, p.per_xcd));
  p.dbg = g_pair32_dbg;
  hipLaunchKernelGGL(kern, dim3(8 * p.nslots), dim3(512), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("pairw_kernel");
  return SAT_OK;
}

template <int C, int KS>
static int launch_pairw(P32Args& p, hipStream_t s) {
  if (p.y16 && p.y) return launch_pairw_t<C, KS, true, true>(p, s);
  if (p.y16) return launch_pairw_t<C, KS, true, false>(p, s);
  return launch_pairw_t<C, KS, false, true>(p, s);
}

static int g_pair32s_waves = 8;

template <bool Y16, bool YF, int NW>
static int launch_pair32s_t(P32Args& p, hipStream_t s) {
  const size_t lds_bytes = (size_t)(3 * P32<NW>::IMG + 16) * 16;   // two images, the inner activation, slack for the reads past a row's end
  auto kern = pair32s_kernel<Y16, YF, NW>;
  static std::atomic<uint64_t> attr_done{0};
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  p.tiles_t = ceil_div(p.T, P32<NW>::TQ);
  p.total = p.tiles_t * p.B;
  p.per_xcd = ceil_div(p.total, 8);
  p.nslots = std::max(1, std::min(NW == 8 ? 32 : 64, p.per_xcd));
  hipLaunchKernelGGL(kern, dim3(8 * p.nslots), dim3(64 * NW), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("pair32s_kernel");
  return SAT_OK;
}

static int g_pair32w = 1;   // 7 and 11 taps at C = 32 on the wave-specialised kernel
static int g_pair64w = 0;   // 3 taps at C = 64: level with the general fused step (133-141 against 136-157 us: with 36 MFMAs per wave and
                            // subtile the conv2 waves' epilogue, the same per subtile as at C = 32, bounds the step) — off by default
void pair32w_set(int v) { g_pair32w = v != 0; }
void pair64w_set(int v) { g_pair64w = v != 0; }

// the fused steps these kernels serve: C = 32 (3 / 7 / 11 taps) and C = 64 (3 taps), dilation <= 5, planes in, residual from the planes
bool pair32s_supports(const ConvArgs& a) {
  // (11 taps with an f32 output: the conv2 waves run out of registers — no read-ahead, spills — and measured 212 against 184 us)
  const bool shape = (a.cin_g == 32 && a.rows_g == 32 && (a.ksize == 3 || (g_pair32w && (a.ksize == 7 || (a.ksize == 11 && a.no_y))))) ||
                     (a.cin_g == 64 && a.rows_g == 64 && a.ksize == 3 && g_pair64w);
  return shape && a.dil >= 1 &&
         a.dil <= P32_HL - 1 && a.x16 && a.res16 == a.x16 && !a.res &&
         !a.ch_scale && !a.relu && !a.gelu && a.co_pad == 64 && a.res_scale == 1.f && !a.y16_f8 && (a.y16 || !a.no_y) && !a.no_store;
}

int launch_pair32s(const ConvArgs& a, int B, hipStream_t s) {
  P32Args p{};
  p.x16 = a.x16;
  p.y16 = a.y16;
  p.y = a.no_y ? nullptr : a.y;
  p.w1 = a.w;
  p.w2 = a.w2;
  p.b1 = a.bias1;
  p.b2 = a.bias;
  p.T = a.T_q;
  p.B = B;
  p.dil = a.dil;
  p.y_bs = a.y_bs;
  p.y_cs = a.y_cs;
  p.d1 = a.w_descale1;
  p.d2 = a.w_descale;
  p.slope = a.in_slope;
  p.inv_slope = a.res16_inv;
  p.y_slope = a.y16_slope;
  p.accum = a.accum;
  p.accum_div = a.accum_div;
  if (a.cin_g == 64) return launch_pairw<64, 3>(p, s);
  if (a.ksize == 7) return launch_pairw<32, 7>(p, s);
  if (a.ksize == 11) return launch_pairw<32, 11>(p, s);
  if (g_pair32s_waves == 4) {
    if (p.y16 && p.y) return launch_pair32s_t<true, true, 4>(p, s);
    if (p.y16) return launch_pair32s_t<true, false, 4>(p, s);
    return launch_pair32s_t<false, true, 4>(p, s);
  }
  if (p.y16 && p.y) return launch_pair32s_t<true, true, 8>(p, s);
  if (p.y16) return launch_pair32s_t<true, false, 8>(p, s);
  return launch_pair32s_t<false, true, 8>(p, s);
}

void pair32s_set_waves(int n) { g_pair32s_waves = n == 4 ? 4 : 8; }

}  // namespace sat
