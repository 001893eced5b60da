// k-tap convolution on split planes as an LDS-DMA ring on v_mfma_f32_16x16x32_f16 (the generator's resblock convs at
// C >= 128: reference satools/satools/hifigan/nn.py:96-175, archi.py:82-86).
#include "conv_common.h"

#include <type_traits>

namespace sat {

// ------------------------------------------------------------------------------------------------
// What the register-staged conv tile (conv_lean.hip: 64 rows x 256 positions, three 4-wave blocks per CU) loses, by
// its own diagnostic builds: 13 % to the registers -> ds_write publish of both operands, 14 % to 640 tiles
// on 768 slots at C = 256 / T = 1250 / 32 utterances, and its LDS is as busy as its matrix pipe.  This form:
//   * BOTH operands travel global -> LDS by LDS-DMA (inline asm, counted vmcnt; no staging registers, no ds_write);
//   * one 8-wave block per CU over a tile of (64 WR) rows x (80 WC) positions, WR x WC = 8: 256 x 160 for C >= 256
//     (1250 positions x 32 utterances = 256 tiles = ONE per CU, nothing ragged), 128 x 320 for C = 128; the wave tile
//     is 64 x 80 = 4 x 5 accumulators of 16 x 16, so the weights of a step are fetched once per 160 / 320 positions
//     instead of once per 256 and an activation tile once per 256 / 128 rows instead of once per 64;
//   * the 16x16x32 MFMA shape (holds a ~20 % higher clock than 32x32x16 at the same cycles per FLOP, gemm_ring.hip):
//     K = 32 of one instruction = the same tap of TWO consecutive 16-channel chunks, lane (li, lg) reads the 16-byte
//     unit (chunk lg >> 1, half lg & 1, row / column li) of each operand.
// One step = (chunk pair, tap): 60 MFMAs per wave behind ONE raw s_barrier.
//   W ring, three slots:  [chunk of the pair][hi0 hi1 lo0 lo1][64 WR rows] x 16 B — the packed weights as they lie in
//                         memory; step s + 3 is requested into the slot of step s right behind barrier s (the A fragments
//                         of step s are in registers by then: they were read during step s - 1)
//   X tiles, two slots:   [chunk of the pair][4 planes][XW columns] x 16 B — the planes as they lie in memory, with the
//                         halo of the dilated taps; pair p + 1 is requested behind the barrier of step (p, tap 0)
// Fragments flow: the B pair of column n + 1 is read while column n is multiplied (two pairs live), the A pairs of the
// next step during columns 1..4 (two A sets, alternating by step parity: the loop is instantiated per parity).
// The SIMD partners (waves k and k + 4) issue their DMA pieces at different points of a step (gemm_ring.hip).
// Per accumulator: pairs of chunks ascending, taps ascending, lo*hi, hi*lo, hi*hi — K = 32 inside one instruction
// associates differently from the 32x32x16 tile: agreement to f32 rounding of the accumulation, not bit for bit.
//
// F8 (round 5, SAT_CONV_F16F8R): 2 MFMA units per product instead of 3.  hi * hi stays on the f16 MFMA; the two cross terms
// (2^-11 of the product) of a PAIR of taps go through ONE block-scaled 8-bit v_mfma_scale_f32_16x16x128_f8f6f4 (weights e4m3, activations e5m2: f16's exponent range): K = 128 = lane
// group lg -> (term lg >> 1, chunk lg & 1 of the pair) x 32 bytes = 16 channels of the first tap | 16 channels of the second.
// The rings do not change shape.  A step is still 8 planes of ROWS 16-byte units from the W ring and reads the same X tile:
//   E step (even): W planes (tap of the pair, chunk, half) = hi f16: 40 f16 MFMAs per wave (both taps)
//   O step (odd):  W planes (tap of the pair, term, chunk): e4m3(W_lo * 2^9) | e4m3(W_hi * 2^-2): 20 8-bit MFMAs — the same
//                  640 matrix cycles; the per-lane E8M0 scales undo the powers of two exactly
//   X tile:        [chunk][hi0 hi1 | e5m2(hi) e5m2(lo * 2^10)][XW]: units 0, 1 from the main planes, 2, 3 from the sidecar
// The K dimension is the LINEAR sequence of (chunk pair, tap) elements, L = pp x taps + t, and a pair = elements (2 q, 2 q + 1): the last
// tap of a chunk pair goes with the first tap of the next one (its two B halves then come from the two X slots), so the odd kernel
// sizes of the generator (3, 7, 11) cost no padding.  X tile p + 1 is requested as soon as the last tap of tile p - 1 has been
// multiplied; the first step that touches a tile reads its column 0 behind its own barrier instead of one step ahead.
// ------------------------------------------------------------------------------------------------

#define SAT_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define SAT_WAIT_VM_LGKM0(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n) : "memory")

// diagnostic instantiation (STAMP): waves 0 and 4 of every block record cycle counters (sat_convring_debug_stamps)
constexpr int CR_STAMPS = 8;
__device__ __forceinline__ long long cr_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return (long long)t;
}
__device__ __forceinline__ long long cr_realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return (long long)t;
}

// up to three convolutions of one shape (the three branches of an MRF block: archi.py:82-86) served by one launch.
// What a tile needs of its conv, compact (the K loop and the epilogue copy it into scalar registers once per tile: read field
// by field from a ConvArgs table in the kernel-argument segment, every use was a scalar load with its own round trip)
struct RingJob {
  const void* x16;       // input planes
  const void* w;         // packed split-f16 weights
  const float* bias;
  float* y;              // f32 output (and MRF accumulator), or null
  void* y16;             // output planes of lrelu(y, y16_slope), or null
  const void* res16;     // residual as planes of lrelu(r, 1 / res16_inv), or null
  const void* x8;        // F8: e5m2 sidecar of x16
  void* y8;              // e5m2 sidecar of y16 to write, or null
  long long y_bs, y_cs;  // f32 output strides (elements) — per job: a job may store into a pitched view of its own
  float w_descale, y16_slope, res16_inv, accum_div;
  int ksize, dil, pad_left, accum;
  unsigned w_bytes;
  int nstep;             // F8: steps of a tile, 2 * ceil(chunk pairs x ksize / 2)
  int hi_only;           // the lo units of y16 are not stored
  int no_store;          // y is the MRF sum so far (read: accum) and the new sum is not written back (sat_conv1d_desc.accum_no_store)
};
struct RingArgs {
  RingJob job[3];
  int cin_g, cin_pad, rows_g, co_pad, T_in, T_q;
  int njobs;        // 1..3
  int rotate;       // the order of the jobs rotates with the region (no job reads what another one writes)
  int n_rt, n_ct;   // row tiles, column tiles per utterance
  int total;        // column tiles x utterances
  int n_vb;         // regions, numbered like the blocks of gemm_ring.hip: 8 * n_rt * ceil(total / 8), some of them empty
  int diag;         // diagnostic builds (results are wrong): 2 = no K loop, 4 = no epilogue; with stamps: 8 = no DMA issue, 16 = no fragment reads in the loop
};

// ---- epilogue of a wave tile (MT x NT accumulators of 16 x 16; D: row 4 lg + r, column li): bias, residual rebuilt from
// planes, MRF accumulation, f32 and / or plane stores — the arithmetic of conv_epilogue16 for exactly these options, in its
// order (same bits).  Every load is requested before the stores that would delay it (vmcnt retires in issue order): the
// biases of all rows first, then per 16-row strip the residual words and the accumulator values of the NEXT strip.
// PROF: the option set as compile-time constants for the three launches a ResBlock step makes (round 5: the generic form spent ~7
// wave-uniform branches and ~20 register copies per 16 x 16 subtile on options that never change inside a launch) —
//   0 whatever the job says (run-time flags);
//   1 conv1: planes out (+ sidecar, lo units not stored, when Y8), nothing else;
//   2 conv2 of steps 1, 2: residual from planes, planes out (+ sidecar when Y8);
//   3 conv2 of step 3: residual from planes, f32 out with the MRF accumulation / division and the next stage's planes as the job says.
// Same arithmetic in the same order: the bits of the generic form.
template <int MT, int NT, int PROF = 0, bool Y8 = false>
__device__ __forceinline__ void ring_epilogue(const RingJob& e, const RingArgs& A, f32x4 (&acc)[MT][NT], int b, int co_w, int q_w, int li, int lg) {
  const unsigned OOB = 0x80000000u;
  const int rows_g = A.rows_g, T_q = A.T_q;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(e.y ? e.y + (long long)b * e.y_bs : (float*)e.y16), 0, e.y ? (unsigned)(rows_g * e.y_cs * 4) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)e.bias, 0, (unsigned)(rows_g * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(e.y16 ? (char*)e.y16 + (long long)b * rows_g * T_q * 4 : (char*)e.y), 0, e.y16 ? (unsigned)(rows_g * T_q * 4) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r16rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(e.res16 ? (const char*)e.res16 + (long long)b * rows_g * T_q * 4 : (const char*)e.bias), 0,
      e.res16 ? (unsigned)(rows_g * T_q * 4) : 0u, 0x00020000);
  // e5m2 sidecar of the output planes: [chunk][e5m2(hi) | e5m2(lo * 2^10)][T_q] x 16 bytes (one byte per channel)
  const __amdgpu_buffer_rsrc_t y8rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(e.y8 ? (char*)e.y8 + (long long)b * rows_g * T_q * 2 : (char*)e.bias), 0, e.y8 ? (unsigned)(rows_g * T_q * 2) : 0u, 0x00020000);
  const int y_rb = (int)e.y_cs * 4;
  const bool has_y = PROF == 0 ? e.y != nullptr : PROF == 3;
  const bool has_y16 = (PROF == 0 || PROF == 3) ? e.y16 != nullptr : true;
  const bool has_res = PROF == 0 ? e.res16 != nullptr : PROF >= 2;
  const bool accum = (PROF == 0 || PROF == 3) ? e.accum != 0 : false;
  const bool has_y8 = PROF == 0 ? e.y8 != nullptr : (PROF == 3 ? false : Y8);
  const bool hi_only = PROF == 0 ? e.hi_only != 0 : (PROF == 1 && Y8);
  const bool no_store = (PROF == 0 || PROF == 3) ? e.no_store != 0 : false;
  const float descale = e.w_descale, slope = e.y16_slope, inv = e.res16_inv, div = (PROF == 0 || PROF == 3) ? e.accum_div : 0.f;
  float bi[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row0 = co_w + m * 16 + 4 * lg;
#pragma unroll
    for (int r = 0; r < 4; ++r) bi[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (row0 + r) * 4, 0, 0));
  }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rbuf[2][NT];          // residual words / accumulator values of strip m in set m & 1 (the loop is unrolled: static indices, no copies)
  f32x4 ybuf[2][NT];
  auto load_strip = [&](int m) __attribute__((always_inline)) {
    u32x4 (&rn)[NT] = rbuf[m & 1];
    f32x4 (&yn)[NT] = ybuf[m & 1];
    const int chunk = (co_w >> 4) + m, row0 = co_w + m * 16 + 4 * lg;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int q = q_w + n * 16 + li;
      if (has_res) {
        // one WHOLE 16-byte unit per lane: lanes of even lg the hi unit of (chunk, half lg >> 1), their partners lg ^ 1 the lo
        // unit; the halves are exchanged when the strip is consumed (v_permlane16_swap)
        const unsigned off = (q < T_q && chunk * 16 < rows_g) ? (unsigned)(((chunk * 4 + (lg >> 1) + 2 * (lg & 1)) * T_q + q) * 16) : OOB;
        // (kept as an UNSIGNED vector until the words are used: hipcc of ROCm 7.2 narrows `bit_cast<float x 4>(raw_buffer_load_b128)` to a one-dword load
        // whose value fills all four elements — tools/scratch/b128_bitcast_repro.hip)
        rn[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r16rs, off, 0, 0));
      }
      if (accum) {
        const unsigned yoff = q < T_q ? (unsigned)(row0 * y_rb + q * 4) : OOB;
#pragma unroll
        for (int r = 0; r < 4; ++r) yn[n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, yoff + r * y_rb, 0, 0));
      }
    }
  };
  if (has_res || accum) load_strip(0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row0 = co_w + m * 16 + 4 * lg, chunk = (co_w >> 4) + m;
    const bool rows_ok = co_w + m * 16 < rows_g;       // wave-uniform: padding rows of the block's tile
    u32x4 (&rr)[NT] = rbuf[m & 1];
    f32x4 (&yv)[NT] = ybuf[m & 1];
    if (has_res || accum) {
      if (m + 1 < MT) load_strip(m + 1);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int q = q_w + n * 16 + li;
      const bool qok = q < T_q;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[m][n][r], descale, bi[m][r]);
      if (has_res) {
        float rv[4];
        // (hi01, hi23) of this lane's four channels = words 0, 1 of the hi unit (even lg) or 2, 3 (odd lg), the lo words likewise
        // of the lo unit, which the partner lane holds: swap(u0, u2), swap(u1, u3) hand every lane its (hi, lo) pair
        const auto s0 = __builtin_amdgcn_permlane16_swap(rr[n][0], rr[n][2], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(rr[n][1], rr[n][3], false, false);
        // (the words go to the instruction as they are: bit_cast<half2> of the elements of a 2 x u32 vector is miscompiled by this
        // hipcc — element 0 used for both, tools/hipcc_bitcast_repro.hip)
        // s0 = ((hi0, hi1), (lo0, lo1)) of channels 0, 1; s1 of channels 2, 3: (float)hi + (float)lo in one v_fma_mix_f32 each (common.h)
        rv[0] = mix_add_halves<false>(s0[0], s0[1]);
        rv[1] = mix_add_halves<true>(s0[0], s0[1]);
        rv[2] = mix_add_halves<false>(s1[0], s1[1]);
        rv[3] = mix_add_halves<true>(s1[0], s1[1]);
        // (leaky-relu undone: x > 0 ? x : x * inv with inv >= 1 is min(x, x * inv) — one instruction less per value)
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = lrelu_undo_min(rv[r], inv);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv[r];
      }
      if (accum) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = yv[n][r] + v[r];
      }
      if (div != 0.f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] / div;
      }
      if (has_y && !no_store) {
        const unsigned yoff = qok ? (unsigned)(row0 * y_rb + q * 4) : OOB;
#pragma unroll
        for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), yrs, yoff + r * y_rb, 0, 0);
      }
      if (has_y16 && rows_ok) {
        float u[4];
        // (leaky-relu with 0 < slope <= 1 is max(v, v * slope))
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = __builtin_fmaxf(v[r], v[r] * slope);
        const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
        const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
        const auto l01 = split_lo2(h01, u[0], u[1]);
        const auto l23 = split_lo2(h23, u[2], u[3]);
        // a lane holds 8 bytes of the hi unit and 8 of the lo unit of (chunk, half lg >> 1), its partner lg ^ 1 the other 8 of
        // each: after swap(hi, lo) per word the lanes of even lg hold the whole hi unit and their partners the whole lo unit —
        // ONE 16-byte store per lane instead of two 8-byte ones (the epilogue is bound by the issue of its stores)
        const auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, l01), false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, l23), false, false);
        const u32x4 unit = {s0[0], s1[0], s0[1], s1[1]};
        const unsigned off = (qok && !(hi_only && (lg & 1))) ? (unsigned)(((chunk * 4 + (lg >> 1) + 2 * (lg & 1)) * T_q + q) * 16) : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(unit, y16rs, off, 0, 0);
        if (has_y8) {
          // a function of the plane values (hi, lo as f16): four channels = bytes 4 lg .. 4 lg + 3 of the chunk's unit at q
          const unsigned x8h = pack_e5m2x4((float)h01[0], (float)h01[1], (float)h23[0], (float)h23[1]);
          const unsigned x8l = pack_e5m2_lo_x4(l01, l23);
          const unsigned off8 = qok ? (unsigned)(((chunk * 2) * T_q + q) * 16 + 4 * lg) : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(x8h, y8rs, off8, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(x8l, y8rs, off8, T_q * 16, 0);
        }
      }
    }
  }
}

// ---- the upsampler form: a ConvTranspose1d of stride 4 as a polyphase conv whose rows are ordered (16-channel group, phase, channel) —
// the four 16-row strips of a wave's 64 rows are the four output phases of ONE channel group, strip m belongs at the output times
// 4 q + m.  Planes only (bias, leaky-relu, hi / lo split as in ring_epilogue), no residual, no accumulation.
// A lane holds column q = q_w + 16 n + li of every strip: stored as they are, the 16-byte units of one instruction would lie 64 bytes
// apart (measured: the epilogue alone 44 us for 82 MB).  The four strips are TRANSPOSED inside every quad of lanes first (two
// butterfly stages of v_cndmask with a DPP quad_perm source per dword): store j of lane l then carries phase l & 3 of column
// (l & ~3) + j — the four lanes of a quad write 64 contiguous bytes.
template <int NT>
__device__ __forceinline__ void ring_epilogue_ups(const RingJob& e, const RingArgs& A, f32x4 (&acc)[4][NT], int b, int co_w, int q_w, int li, int lg) {
  const unsigned OOB = 0x80000000u;
  const int rows_g = A.rows_g, T_q = A.T_q;
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)e.bias, 0, (unsigned)(rows_g), 0x00020000);      // rows_g / 4 channels
  const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc((char*)e.y16 + (long long)b * rows_g * T_q * 4, 0, (unsigned)(rows_g * T_q * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t y8rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(e.y8 ? (char*)e.y8 + (long long)b * rows_g * T_q * 2 : (char*)e.bias), 0, e.y8 ? (unsigned)(rows_g * T_q * 2) : 0u, 0x00020000);
  const bool has_y8 = e.y8 != nullptr;
  const float descale = e.w_descale, slope = e.y16_slope;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  float bi[4];                                          // channel (co_w / 64) * 16 + 4 lg + r, whatever the phase
#pragma unroll
  for (int r = 0; r < 4; ++r) bi[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, ((co_w >> 2) + 4 * lg + r) * 4, 0, 0));
  if (co_w >= rows_g) return;                           // wave-uniform: padding rows of the block's tile
  const bool l0 = li & 1, l1 = li & 2;
  // unit index of (channel group, plane of this lane after the hi / lo exchange) at output time 0
  const int plane_base = ((co_w >> 6) * 4 + (lg >> 1) + 2 * (lg & 1)) * (4 * T_q);
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    u32x4 un[4];
    unsigned x8[2][4];                                  // e5m2 sidecar: [e5m2(hi) | e5m2(lo * 2^10)][strip], this lane's four channels
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float u[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = __builtin_fmaf(acc[m][n][r], descale, bi[r]);
        u[r] = lrelu_max(v, slope);
      }
      const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
      const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
      const auto l01 = split_lo2(h01, u[0], u[1]);
      const auto l23 = split_lo2(h23, u[2], u[3]);
      // (ring_epilogue: after swap(hi, lo) per word the lanes of even lg hold the whole hi unit, their partners lg ^ 1 the whole lo unit)
      const auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, l01), false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, l23), false, false);
      un[m] = u32x4{s0[0], s1[0], s0[1], s1[1]};
      if (has_y8) {
        x8[0][m] = pack_e5m2x4((float)h01[0], (float)h01[1], (float)h23[0], (float)h23[1]);
        x8[1][m] = pack_e5m2_lo_x4(l01, l23);
      }
    }
    // 4 x 4 transposition (strip m, lane l of the quad) -> (store j, lane l): first strip bit 0 against lane bit 0, then bit 1 against bit 1
    u32x4 p[4], t[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned ev = un[2 * h][w], od = un[2 * h + 1][w];
        const unsigned ev_x = (unsigned)__builtin_amdgcn_mov_dpp((int)ev, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]: lane ^ 1
        const unsigned od_x = (unsigned)__builtin_amdgcn_mov_dpp((int)od, 0xB1, 0xF, 0xF, true);
        p[2 * h][w] = l0 ? od_x : ev;
        p[2 * h + 1][w] = l0 ? od : ev_x;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned lo_ = p[h][w], hi_ = p[h + 2][w];
        const unsigned lo_x = (unsigned)__builtin_amdgcn_mov_dpp((int)lo_, 0x4E, 0xF, 0xF, true);     // quad_perm [2, 3, 0, 1]: lane ^ 2
        const unsigned hi_x = (unsigned)__builtin_amdgcn_mov_dpp((int)hi_, 0x4E, 0xF, 0xF, true);
        t[h][w] = l1 ? hi_x : lo_;
        t[h + 2][w] = l1 ? hi_ : lo_x;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = q_w + n * 16 + (li & ~3) + j;
      const unsigned off = q < T_q ? (unsigned)((plane_base + 4 * q + (li & 3)) * 16) : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(t[j], y16rs, off, 0, 0);
    }
    if (has_y8) {
      // the sidecar dwords through the same 4 x 4 transposition: store j of lane l = phase l & 3 of column (l & ~3) + j, bytes 4 lg ..
#pragma unroll
      for (int term = 0; term < 2; ++term) {
        unsigned p8[4], t8[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned ev = x8[term][2 * h], od = x8[term][2 * h + 1];
          const unsigned ev_x = (unsigned)__builtin_amdgcn_mov_dpp((int)ev, 0xB1, 0xF, 0xF, true);
          const unsigned od_x = (unsigned)__builtin_amdgcn_mov_dpp((int)od, 0xB1, 0xF, 0xF, true);
          p8[2 * h] = l0 ? od_x : ev;
          p8[2 * h + 1] = l0 ? od : ev_x;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned lo_ = p8[h], hi_ = p8[h + 2];
          const unsigned lo_x = (unsigned)__builtin_amdgcn_mov_dpp((int)lo_, 0x4E, 0xF, 0xF, true);
          const unsigned hi_x = (unsigned)__builtin_amdgcn_mov_dpp((int)hi_, 0x4E, 0xF, 0xF, true);
          t8[h] = l1 ? hi_x : lo_;
          t8[h + 2] = l1 ? hi_ : lo_x;
        }
        const int base8 = ((co_w >> 6) * 2 + term) * (4 * T_q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = q_w + n * 16 + (li & ~3) + j;
          const unsigned off8 = q < T_q ? (unsigned)((base8 + 4 * q + (li & 3)) * 16 + 4 * lg) : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(t8[j], y8rs, off8, 0, 0);
        }
      }
    }
  }
}

// A block walks the tiles (region, job) of its regions vb = blockIdx.x, + gridDim.x, ...: the next tile's first X tile and
// W slots are requested BEFORE the epilogue of the current one (the rings are free behind the barrier that ends a K
// loop), and with `rotate` neighbouring blocks take the jobs in different orders, so that their epilogues — 164 KB of
// output per tile — do not reach HBM as one burst of every CU at once (measured on the one-tile-per-launch form: the
// epilogue of 256 blocks finishing together ran at the HBM write rate, 7.5 us for 42 MB).
// UMASK >= 0: the upsampler form (ring_epilogue<UPS>), three tap slots, bit (slot * 4 + phase) of UMASK = the weights of that slot are all
// zero for that phase: those products (and the reads of their A fragments) are left out AT COMPILE TIME — the loop is unrolled over the
// three slots (x two fragment parities).  (Skipping by a run-time mask put scalar branches around the inline-asm MFMAs: hipcc then spilled
// 124 registers, some of them accumulators stored right behind the MFMA that writes them, which it cannot see inside the asm: wrong values.)
template <int WR, bool STAMP = false, int UMASK = -1, bool F8 = false>
__global__ void __launch_bounds__(512, 2) conv1d_f16x3_ring16_kernel(const RingArgs A, long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  static_assert(!(F8 && UMASK >= 0) && !(F8 && WR == 1), "F8: the plain 256 x 160 / 128 x 320 forms only");
  // WR = 4: 256 x 160, WR = 2: 128 x 320 (wave tile 64 x 80); WR = 1: 64 x 384 (wave tile 64 x 48: the 64-channel stage, whose X
  // tiles hold BOTH chunk pairs of its K)
  constexpr bool UPS = UMASK >= 0;
  constexpr int WC = 8 / WR, MT = 4, NT = WR == 1 ? 3 : 5, ROWS = 64 * WR, COLS = 16 * NT * WC;
  constexpr int XW = WR == 4 ? 224 : WR == 2 ? 384 : 448;       // columns of an X tile (>= COLS + (ksize - 1) * dilation)
  constexpr int XP = (XW + 63) / 64;                     // DMA pieces per X row; the last one half-filled when XW % 64 == 32
  constexpr bool XHALF = XW % 64 != 0;
  constexpr int X_UNITS = 8 * XW, W_UNITS = 8 * ROWS, W0 = 2 * X_UNITS;     // 16-byte units
  constexpr int PW = WR, PX = XP;                        // pieces per wave: of a W slot, of an X tile
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, idx = wave & 3;
  const int wr = WR == 4 ? idx : WR == 2 ? (idx & 1) : 0, wc = WR == 4 ? half : WR == 2 ? ((idx >> 1) + 2 * half) : (idx + 4 * half);
  const int njobs = A.njobs;
  const int nreg = (A.n_vb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // regions of this block
  const int ntiles = nreg * njobs;
  long long st_t0 = 0, st_r0 = 0, st_wait = 0, st_pro = 0, st_loop = 0, st_epi = 0;
  if constexpr (STAMP) st_t0 = cr_clock(), st_r0 = cr_realtime();

  // tile k of this block -> job and coordinates; regions like the blocks of gemm_ring.hip (the row tiles of one
  // (utterance, column tile) share an XCD); an empty region (past the last column tile) has b < 0
  struct Tile { int j, b, co_b, q_b; };
  auto locate = [&](int k) __attribute__((always_inline)) {
    Tile t;
    const int i = __builtin_amdgcn_readfirstlane(k / njobs), jj = k - i * njobs;
    const int vb = (int)blockIdx.x + i * (int)gridDim.x;
    const int xcd = vb & 7, rest = vb >> 3;
    const int rt = __builtin_amdgcn_readfirstlane(rest % A.n_rt);
    const int g = __builtin_amdgcn_readfirstlane((rest / A.n_rt) * 8 + xcd);
    t.j = A.rotate ? __builtin_amdgcn_readfirstlane((jj + vb) % njobs) : jj;
    t.b = g < A.total ? __builtin_amdgcn_readfirstlane(g / A.n_ct) : -1;
    t.co_b = rt * ROWS;
    t.q_b = (g - t.b * A.n_ct) * COLS;
    return t;
  };

  // ---- DMA state of the tile whose operands are being requested
  i32x4 xrs, wrs;
  int KS = 1, dil = 1, NP = 1, NS = 1, seg_bytes = 0, x_chunk_bytes = 0;      // (KS: the taps; NS: steps of a tile)
  int ktaps = 1;
  int d_co = 0, d_q0 = 0, d_copad = 0, d_tin = 0;      // row / first input position of the tile, padded rows, input length
  // this wave's DMA pieces: of every W slot the (chunk of the pair = wave >> 2, segment = wave & 3) row, of every X tile the
  // (chunk of the pair, plane = wave & 3) row
  const int my_w_unit = half * 4 * ROWS + idx * ROWS, my_x_unit = half * 4 * XW + idx * XW;
  auto setup = [&](const Tile& t) __attribute__((always_inline)) {
    const RingJob p = A.job[t.j];
    KS = p.ksize, dil = p.dil, ktaps = p.ksize;
    NP = A.cin_pad / (2 * CI_CHUNK), NS = F8 ? p.nstep : NP * KS;
    seg_bytes = A.co_pad * 16;
    if (F8 && idx >= 2) {
      // this wave's X rows are the sidecar's: units (e5m2(hi), e5m2(lo * 2^10)) of its chunk
      x_chunk_bytes = 2 * A.T_in * 16;
      xrs = dma_rsrc((const char*)p.x8 + (long long)t.b * A.cin_g * A.T_in * 2, (unsigned)(A.cin_g * A.T_in * 2));
    } else {
      x_chunk_bytes = 4 * A.T_in * 16;
      xrs = dma_rsrc((const char*)p.x16 + (long long)t.b * A.cin_g * A.T_in * 4, (unsigned)(A.cin_g * A.T_in * 4));
    }
    wrs = dma_rsrc(p.w, p.w_bytes);
    d_co = t.co_b, d_q0 = t.q_b - p.pad_left, d_copad = A.co_pad, d_tin = A.T_in;
  };
  // per-lane byte offsets of a piece, computed where it is issued (a handful of vector instructions per piece; kept in
  // registers across the K loop they were what the allocator spilled INTO the loop)
  auto voff_w = [&](int j) __attribute__((always_inline)) {
    const int row = d_co + j * 64 + lane;
    return row < d_copad ? (unsigned)(row * 16 + (F8 ? 0 : idx * seg_bytes)) : 0x80000000u;
  };
  auto voff_x = [&](int k) __attribute__((always_inline)) {
    const int xi = d_q0 + k * 64 + lane;
    return (xi >= 0 && xi < d_tin) ? (unsigned)(((F8 ? (idx & 1) : idx) * d_tin + xi) * 16) : 0x80000000u;
  };
  auto issue_w = [&](int pp, int t, int slot) __attribute__((always_inline)) {
    const uint4* dst = lds4 + W0 + slot * W_UNITS + my_w_unit;
    // F8: packed [step][plane = wave][row] (called with pp = 0, t = the step); else [chunk][tap][plane][row]
    const unsigned soff = F8 ? (unsigned)((t * 8 + wave) * seg_bytes) : (unsigned)((((2 * pp + half) * KS + t) * 4) * seg_bytes);
#pragma unroll
    for (int j = 0; j < PW; ++j) lds_dma16(dst + j * 64, wrs, voff_w(j), soff);
  };
  auto issue_x = [&](int pp) __attribute__((always_inline)) {
    const uint4* dst = lds4 + (pp & 1) * X_UNITS + my_x_unit;
    const unsigned soff = (unsigned)((2 * pp + half) * x_chunk_bytes);
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      if (XHALF && k == PX - 1) lds_dma16_lo32(dst + k * 64, xrs, voff_x(k), soff);
      else lds_dma16(dst + k * 64, xrs, voff_x(k), soff);
    }
  };
  // first requests of a tile: X tile of pairs 0 and 1, W of steps 0..2 (taps >= 3: at least three steps)
  auto issue_prologue = [&]() __attribute__((always_inline)) {
    issue_x(0);
    issue_w(0, 0, 0);
    if (NP > 1) issue_x(1);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);
  };

  // fragment addresses (units): lane (li, lg) reads chunk lg >> 1, half lg & 1; lo planes 2 segments further
  // (F8: W plane = 4 (tap of the pair) + lg in both kinds of step; the 8-bit B unit of lane group lg = plane 2 + term of chunk lg & 1)
  const int a_lane = F8 ? lg * ROWS + wr * 64 + li : (lg >> 1) * 4 * ROWS + (lg & 1) * ROWS + wr * 64 + li;
  const int b_lane = (lg >> 1) * 4 * XW + (lg & 1) * XW + wc * (16 * NT) + li;
  const int b_lane8 = (lg & 1) * 4 * XW + (2 + (lg >> 1)) * XW + wc * (16 * NT) + li;
  // E8M0 scales (term 0: W_lo8 . x_hi8, term 1: W_hi8 . x_lo8).  The instruction numbers k = 16 g + j for bytes 0-15 of lane group g and
  // 64 + 16 g + j for bytes 16-31, and takes the scale of its 32-wide block b from lane group b (measured: tools/scratch/probe_mfma_scale.hip):
  // block 0 = first halves of groups 0, 1 (term 0, first tap), block 1 = first halves of groups 2, 3 (term 1), blocks 2, 3 the second tap's
  const int sc_a = (lg & 1) ? F8R_E_WHI : F8R_E_WLO, sc_b = (lg & 1) ? F8_E_XLO : F8_E_XHI;
  h8 fa[2][MT][2], fb[2][2];
  // F8: E steps use fa[0] / fb (taps 0, 1 of the pair), O steps the 32-byte forms
  i32x8 fa8[MT], fb8[2];
  typedef int i32x4v __attribute__((ext_vector_type(4)));
  auto read8 = [&](i32x8& dst, const uint4* p0, const uint4* p1) __attribute__((always_inline)) {
    const i32x4v lo = __builtin_bit_cast(i32x4v, *p0), hi = __builtin_bit_cast(i32x4v, *p1);
    dst = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_a = [&](h8 (&dst)[2], const uint4* wb, int m) __attribute__((always_inline)) {
    dst[0] = __builtin_bit_cast(h8, wb[m * 16]);
    dst[1] = __builtin_bit_cast(h8, wb[m * 16 + (F8 ? 4 : 2) * ROWS]);
  };
  auto read_b = [&](h8 (&dst)[2], const uint4* xb, int n) __attribute__((always_inline)) {
    dst[0] = __builtin_bit_cast(h8, xb[n * 16]);
    dst[1] = __builtin_bit_cast(h8, xb[n * 16 + 2 * XW]);
  };
  f32x4 acc[MT][NT];

  int s = 0, pp = 0, t = 0, ws = 0;          // current step, its pair, tap, W slot
  int p3 = 0, t3 = 0;                        // pair and tap of step s + 3
  auto body = [&](auto cur, auto late_c, auto more_c, auto tap_c) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur)::value;
    constexpr int TC = decltype(tap_c)::value;         // UPS: the tap slot of this step, known at compile time (else -1)
    // strips (= phases) whose weights are zero at this slot / at the next step's
    constexpr unsigned SK = TC >= 0 ? ((unsigned)UMASK >> (4 * TC)) & 15u : 0u, SK1 = TC >= 0 ? ((unsigned)UMASK >> (4 * ((TC + 1) % 3))) & 15u : 0u;
    constexpr bool LATE = decltype(late_c)::value, MORE = decltype(more_c)::value;
    constexpr int N_HAND = LATE ? NT / 2 : 0;     // the DMA issue stands in front of this column's MFMAs
    constexpr int AR = (MT + NT - 2) / (NT - 1);   // rows of the next step's A fragments read per column (columns 1 .. NT - 1)
    if constexpr (MORE) {
      long long w0 = 0;
      if constexpr (STAMP) w0 = cr_clock();
      // barrier s: W of step s + 1 landed (this wave's pieces: all but the youngest step's, and an X tile requested in
      // between), this wave's reads of slot s are back; behind it everybody's are: slot s is free for step s + 3
      if (s + 2 < NS) {
        if (t == 1 && pp >= 1 && pp + 1 < NP) SAT_WAIT_VM_LGKM0(PW + PX);
        else SAT_WAIT_VM_LGKM0(PW);
      } else {
        SAT_WAIT_VM_LGKM0(0);
      }
      asm volatile("s_barrier" ::: "memory");
      if constexpr (STAMP) st_wait += cr_clock() - w0;
    }
    int t1 = t + 1, p1 = pp;
    if (t1 == KS) t1 = 0, p1 = pp + 1;
    const int ws1 = ws == 2 ? 0 : ws + 1;
    const uint4* wnext = lds4 + W0 + ws1 * W_UNITS + a_lane;
    const uint4* xcur = lds4 + (pp & 1) * X_UNITS + b_lane + t * dil;
    const uint4* xnext = lds4 + (p1 & 1) * X_UNITS + b_lane + t1 * dil;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      __builtin_amdgcn_sched_barrier(0);
      // (diagnostic instantiation only: diag 8 = no DMA issue in the loop, 16 = no fragment reads in the loop)
      if (n == N_HAND && !(STAMP && (A.diag & 8))) {
        if (t == 0 && pp >= 1 && pp + 1 < NP) issue_x(pp + 1);
        if (s + 3 < NS) issue_w(p3, t3, ws);
      }
      if (!(STAMP && (A.diag & 16))) {
        if (n + 1 < NT) read_b(fb[(n + 1 + CUR) & 1], xcur, n + 1);
        else if constexpr (MORE) read_b(fb[(NT + CUR) & 1], xnext, 0);
        if constexpr (MORE) {
          if (n >= 1) {
#pragma unroll
            for (int m2 = (n - 1) * AR; m2 < n * AR && m2 < MT; ++m2) {
              if ((SK1 >> m2) & 1u) continue;
              read_a(fa[CUR ^ 1][m2], wnext, m2);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if ((SK >> m) & 1u) continue;
        mfma16_acc(acc[m][n], fa[CUR][m][1], fb[(n + CUR) & 1][0]);
        mfma16_acc(acc[m][n], fa[CUR][m][0], fb[(n + CUR) & 1][1]);
        mfma16_acc(acc[m][n], fa[CUR][m][0], fb[(n + CUR) & 1][0]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ++s, pp = p1, t = t1, ws = ws1;
    if (++t3 == KS) t3 = 0, ++p3;
  };
  // ---- F8: CUR = 0 the E step of a pair of elements (hi * hi of both on the f16 MFMA: 40 per wave), CUR = 1 its O step (both cross
  // terms of both in one 8-bit MFMA of K = 128: 20 per wave, the same matrix cycles).  Same rings, same counted waits.
  int e0p = 0, e0t = 0, e1p = 0, e1t = 1;      // the two elements (chunk pair, tap) of the current pair
  int n0p = 0, n0t = 0, n1p = 0, n1t = 0;      // ... of the next pair (set by the O step)
  int nxt_x = 2;                               // next X tile to request (0 and 1: the prologue)
  bool xprev = false, fresh = false;           // an X tile was requested in the previous step; this E step is the first to touch a tile
  auto body8 = [&](auto cur, auto late_c, auto more_c) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur)::value;
    constexpr bool LATE = decltype(late_c)::value, MORE = decltype(more_c)::value;
    constexpr int N_HAND = LATE ? NT / 2 : 0;
    static_assert(MT == NT - 1 || !F8, "one A row of the next step per column 1 .. NT - 1");
    if constexpr (MORE) {
      long long w0 = 0;
      if constexpr (STAMP) w0 = cr_clock();
      // barrier s: W of step s + 1 landed — everything but the youngest W request, and the X tile requested in between (step s - 1)
      if (s + 2 < NS) {
        if (xprev) SAT_WAIT_VM_LGKM0(PW + PX);
        else SAT_WAIT_VM_LGKM0(PW);
      } else {
        SAT_WAIT_VM_LGKM0(0);
      }
      asm volatile("s_barrier" ::: "memory");
      if constexpr (STAMP) st_wait += cr_clock() - w0;
    }
    xprev = false;
    const int ws1 = ws == 2 ? 0 : ws + 1;
    const uint4* wnext = lds4 + W0 + ws1 * W_UNITS + a_lane;
    // B columns of the two elements: (X slot of its chunk pair) + tap x dilation
    const int o0 = (e0p & 1) * X_UNITS + e0t * dil, o1 = (e1p & 1) * X_UNITS + e1t * dil;
    const uint4* x16a = lds4 + o0 + b_lane;
    const uint4* x16b = lds4 + o1 + b_lane;
    const uint4* x8a = lds4 + o0 + b_lane8;
    const uint4* x8b = lds4 + o1 + b_lane8;
    const uint4* xn16a = x16a;
    const uint4* xn16b = x16b;
    bool fresh_next = false;
    if constexpr (CUR == 0) {
      if (fresh) {                              // the first step on a tile: its column 0 was not read ahead (the tile may have been in flight)
        fb[0][0] = __builtin_bit_cast(h8, x16a[0]);
        fb[0][1] = __builtin_bit_cast(h8, x16b[0]);
      }
    } else {
      // the next pair: two more elements of the linear sequence (past its end: the zero element, on the last element's columns)
      n0p = e1p, n0t = e1t + 1;
      if (n0t == ktaps) n0t = 0, ++n0p;
      n1p = n0p, n1t = n0t + 1;
      if (n1t == ktaps) n1t = 0, ++n1p;
      if (n1p >= NP) n1p = n0p, n1t = n0t;
      if (n0p >= NP) n0p = e1p, n0t = e1t, n1p = e1p, n1t = e1t;      // (behind the last pair: never multiplied)
      fresh_next = n1p > e1p;
      xn16a = lds4 + (n0p & 1) * X_UNITS + n0t * dil + b_lane;
      xn16b = lds4 + (n1p & 1) * X_UNITS + n1t * dil + b_lane;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      __builtin_amdgcn_sched_barrier(0);
      if (n == N_HAND && !(STAMP && (A.diag & 8))) {
        if constexpr (CUR == 0) {
          // X tile nxt_x takes the slot of tile nxt_x - 2: free once every element of that tile has been multiplied, i.e. this pair's
          // first element lies in tile nxt_x - 1 or later (s = 2 q = the linear index of this pair's first element)
          if (nxt_x < NP && s >= (nxt_x - 1) * ktaps) {
            issue_x(nxt_x);
            ++nxt_x;
            xprev = true;
          }
        }
        if (s + 3 < NS) issue_w(0, s + 3, ws);
      }
      if (!(STAMP && (A.diag & 16))) {
        if constexpr (CUR == 0) {
          if (n + 1 < NT) {
            fb[(n + 1) & 1][0] = __builtin_bit_cast(h8, x16a[(n + 1) * 16]);
            fb[(n + 1) & 1][1] = __builtin_bit_cast(h8, x16b[(n + 1) * 16]);
          } else {
            read8(fb8[0], x8a, x8b);                    // column 0 of the O step that always follows
          }
          if (n >= 1) read8(fa8[n - 1], wnext + (n - 1) * 16, wnext + (n - 1) * 16 + 4 * ROWS);
        } else {
          if (n + 1 < NT) {
            read8(fb8[(n + 1) & 1], x8a + (n + 1) * 16, x8b + (n + 1) * 16);
          } else if constexpr (MORE) {
            if (!fresh_next) {
              fb[0][0] = __builtin_bit_cast(h8, xn16a[0]);
              fb[0][1] = __builtin_bit_cast(h8, xn16b[0]);
            }
          }
          if constexpr (MORE) {
            if (n >= 1) read_a(fa[0][n - 1], wnext, n - 1);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if constexpr (CUR == 0) {
          mfma16_acc(acc[m][n], fa[0][m][0], fb[n & 1][0]);
          mfma16_acc(acc[m][n], fa[0][m][1], fb[n & 1][1]);
        } else {
          mfma8_acc(acc[m][n], fa8[m], fb8[n & 1], sc_a, sc_b);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ++s, ws = ws1;
    if constexpr (CUR == 1) e0p = n0p, e0t = n0t, e1p = n1p, e1t = n1t, fresh = fresh_next;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using NOT = std::integral_constant<int, -1>;
  auto loop = [&](auto late_c) __attribute__((always_inline)) {
    using L = decltype(late_c);
    if constexpr (F8) {
      // NS = 2 ceil(NP x taps / 2): E and O steps alternate, the last step is an O step
      while (s + 2 < NS) {
        body8(I0{}, L{}, std::true_type{});
        body8(I1{}, L{}, std::true_type{});
      }
      body8(I0{}, L{}, std::true_type{});
      body8(I1{}, L{}, std::false_type{});
    } else if constexpr (UPS) {
      // three slots x two fragment parities: NS = 3 NP steps, NP even
      auto six = [&](auto more_last) __attribute__((always_inline)) {
        body(I0{}, L{}, std::true_type{}, I0{});
        body(I1{}, L{}, std::true_type{}, I1{});
        body(I0{}, L{}, std::true_type{}, I2{});
        body(I1{}, L{}, std::true_type{}, I0{});
        body(I0{}, L{}, std::true_type{}, I1{});
        body(I1{}, L{}, more_last, I2{});
      };
      while (s + 6 < NS) six(std::true_type{});
      six(std::false_type{});
    } else {
      // (NS = NP x taps is even: convring_supports sends an odd number of steps to the register-staged tile — one loop tail less to
      // instantiate; with three the register allocator shuffled accumulators between the tails and spilled some right behind the MFMA
      // that writes them, tests/test_codegen_invariants.py)
      while (s + 2 < NS) {
        body(I0{}, L{}, std::true_type{}, NOT{});
        body(I1{}, L{}, std::true_type{}, NOT{});
      }
      body(I0{}, L{}, std::true_type{}, NOT{});
      body(I1{}, L{}, std::false_type{}, NOT{});
    }
  };

  // ---- the walk
  int k = 0;
  Tile cur = locate(0);
  while (k < ntiles && cur.b < 0) cur = locate(++k);
  if (k >= ntiles) return;
  setup(cur);
  issue_prologue();
  while (true) {
    long long e0 = 0;
    if constexpr (STAMP) e0 = cr_clock();
    // pair 0 and step 0 landed (everything older than this tile's last three requests: the previous tile's stores too)
    if (NP > 1) SAT_WAIT_VM(PX + 2 * PW);
    else SAT_WAIT_VM(2 * PW);
    asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int m = 0; m < MT; ++m) read_a(fa[0][m], lds4 + W0 + a_lane, m);
    if constexpr (F8) {
      fb[0][0] = __builtin_bit_cast(h8, lds4[b_lane]);                             // taps 0 and 1 of pair 0, column 0
      fb[0][1] = __builtin_bit_cast(h8, lds4[b_lane + (ktaps > 1 ? dil : 0)]);
    } else {
      read_b(fb[0], lds4 + b_lane, 0);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 3" ::: "memory");       // (vector write of the accumulators -> matrix read, were the loop to start at once)
    s = 0, pp = 0, t = 0, ws = 0;
    p3 = KS > 3 ? 0 : 1, t3 = KS > 3 ? 3 : 0;
    e0p = 0, e0t = 0, e1p = 0, e1t = 1, nxt_x = 2, xprev = false, fresh = false;      // (F8; taps >= 3)
    if constexpr (STAMP) st_pro += cr_clock() - e0;
    long long l0 = 0;
    if constexpr (STAMP) l0 = cr_clock();
    if (!(A.diag & 2)) {
      if (wave >= 4) loop(std::true_type{});
      else loop(std::false_type{});
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (STAMP) st_loop += cr_clock() - l0;
    // the tile whose results are in the accumulators; then the next tile's first requests: every wave has read its last
    // fragments behind this barrier, the rings are free
    const Tile done = cur;
    int kn = k + 1;
    Tile nxt = done;
    while (kn < ntiles) {
      nxt = locate(kn);
      if (nxt.b >= 0) break;
      ++kn;
    }
    const bool more = kn < ntiles;
    if (more) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      setup(nxt);
      issue_prologue();
    }
    long long p0 = 0;
    if constexpr (STAMP) p0 = cr_clock();
    mfma16_drain(acc);
    if (!(A.diag & 4)) {
      const RingJob ej = A.job[done.j];
      if constexpr (UPS) ring_epilogue_ups<NT>(ej, A, acc, done.b, done.co_b + wr * 64, done.q_b + wc * (16 * NT), li, lg);
      else {
        // the option set of this job as one of the three ResBlock profiles (wave-uniform: one branch per tile instead of seven per subtile)
        const bool plain = !ej.y && ej.y16 && !ej.accum && ej.accum_div == 0.f && ((ej.y8 != nullptr) == F8);
        const int co_e = done.co_b + wr * 64, q_e = done.q_b + wc * (16 * NT);
        if (plain && !ej.res16 && ((ej.hi_only != 0) == F8)) ring_epilogue<MT, NT, 1, F8>(ej, A, acc, done.b, co_e, q_e, li, lg);
        else if (plain && ej.res16 && !ej.hi_only) ring_epilogue<MT, NT, 2, F8>(ej, A, acc, done.b, co_e, q_e, li, lg);
        else if (ej.y && ej.res16 && !ej.y8) ring_epilogue<MT, NT, 3, F8>(ej, A, acc, done.b, co_e, q_e, li, lg);
        else ring_epilogue<MT, NT>(ej, A, acc, done.b, co_e, q_e, li, lg);
      }
    }
    if constexpr (STAMP) st_epi += cr_clock() - p0;
    if (!more) break;
    k = kn;
    cur = nxt;
  }
  if constexpr (STAMP) {
    if (dbg && (wave & 3) == 0 && lane == 0) {
      long long* d = dbg + ((long long)blockIdx.x * 2 + half) * CR_STAMPS;
      const long long t_end = cr_clock(), r_end = cr_realtime();
      d[0] = st_pro;              // waits for the first operands of the tiles
      d[1] = st_loop;             // K loops
      d[2] = st_wait;             // of them: s_waitcnt + s_barrier at the step heads
      d[3] = st_epi;              // epilogues (issue; the stores drain behind them)
      d[4] = t_end - st_t0;
      d[5] = r_end - st_r0;       // the same span in 100 MHz ticks
      d[6] = st_r0;
      d[7] = ntiles;
    }
  }
}

static int g_convring = 1;
void convring_set(int v) { g_convring = v; }
static int g_convring_blocks = 0;     // option "convring_blocks": blocks of a launch (0 = one per CU); fewer leave CUs to the kernels of other streams
void convring_set_blocks(int v) { g_convring_blocks = v < 0 ? 0 : v / 8 * 8; }
static long long* g_convring_dbg = nullptr;
int convring_debug_stamps(long long* buf) {
  g_convring_dbg = buf;
  return CR_STAMPS;
}

static int cu_count() {
  static std::atomic<int> n{0};
  int v = n.load(std::memory_order_relaxed);
  if (!v) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) v = pr.multiProcessorCount;
    if (v <= 0) v = 256;
    n.store(v, std::memory_order_relaxed);
  }
  return v;
}

// rows > 64 (the 64-row stage keeps the register-staged tile), taps >= 3, channel pairs, the halo inside the X tile — and enough
// tiles for three quarters of the CUs (one 8-wave block per CU: a batch of a few utterances is better served by the small
// tiles of conv_lean.hip).  Option "convring": 0 = off, 1 = on (default), + 32 = whatever the number of tiles
bool convring_supports(const ConvArgs& a, int B) {
  if (!(g_convring & 1) && !a.f8r) return false;
  if (!(a.x16 && !a.f8 && !a.y16_f8 && !a.poly_planes && !a.k1_wrap && a.fast_epi && a.up == 1 && a.stride == 1)) return false;
  if (a.f8r && !a.x8) return false;
  if (a.y8 && !a.y16) return false;
  // (the epilogue evaluates the leaky-relus as max(v, v * slope) and min(r, r / slope): slopes in (0, 1])
  if (a.y16 && !(a.y16_slope > 0.f && a.y16_slope <= 1.f)) return false;
  if (a.res16 && !(a.res16_inv >= 1.f)) return false;
  // (rows <= 64: the kernel has a 64 x 384 layout, WR = 1, whose X tiles hold both chunk pairs of the 64-channel stage — measured SLOWER
  // than the register-staged tile there, 183-189 against 150 us at 11 taps: 6.6 tiles per block with 112 KB of X each; not dispatched)
  if (!epilogue16_supports(a) || a.ksize < 3 || a.cin_g % 32 != 0 || a.rows_g <= 64) return false;
  // what ring_epilogue carries: bias, residual from planes (scale 1, before the activation), accumulation, f32 / plane stores
  if (a.ch_scale || a.relu || a.gelu || a.res || a.res_after || !a.bias || (a.res16 && a.res_scale != 1.f) || (a.accum && a.no_y)) return false;
  const int halo = (a.ksize - 1) * a.dil;
  if (halo > 64) return false;                     // (X tiles: 160 + 64, 320 + 64, 384 + 64 columns)
  // an even number of K steps (chunk pairs x taps; the F8 form's E / O steps always are): the loop has ONE tail
  if (!a.f8r && ((a.cin_pad / (2 * CI_CHUNK)) * a.ksize) % 2 != 0) return false;
  const long long tiles = a.rows_g > 128 ? (long long)ceil_div(a.rows_g, 256) * ceil_div(a.T_q, 160) * B
                          : a.rows_g > 64 ? (long long)ceil_div(a.T_q, 320) * B : (long long)ceil_div(a.T_q, 384) * B;
  // (SAT_CONV_F16F8R weights are packed for this kernel alone: at every number of tiles)
  return a.f8r || (g_convring & 32) || tiles * 4 >= (long long)cu_count() * 3;
}

// what hifigan.hip asks before it picks the SAT_CONV_F16F8R packing of a stage: would the f16x3 conv of this shape run on the ring?
bool convring_wanted(int rows_g, int T_q, int B) {
  if (!(g_convring & 1) || rows_g <= 64) return false;
  const long long tiles = rows_g > 128 ? (long long)ceil_div(rows_g, 256) * ceil_div(T_q, 160) * B : (long long)ceil_div(T_q, 320) * B;
  return (g_convring & 32) || tiles * 4 >= (long long)cu_count() * 3;
}

// jobs of one shape (channels, lengths, batch): what one launch can walk
// (the f32 output strides are per job: RingJob.y_bs / y_cs)
bool convring_same_shape(const ConvArgs& a, const ConvArgs& b) {
  return a.cin_g == b.cin_g && a.cin_pad == b.cin_pad && a.rows_g == b.rows_g && a.co_pad == b.co_pad && a.T_in == b.T_in && a.T_q == b.T_q &&
         a.f8r == b.f8r;
}

// ConvTranspose1d(stride 4) as a polyphase conv whose rows are ordered (16-channel group, phase, channel) — sat_conv1d_desc.up_grouped:
// only this kernel reads that order (256-row tiles: rows > 128), whatever the number of tiles
bool convring_ups_supports(const ConvArgs& a) {
  if (!(a.x16 && a.y16 && a.no_y && !a.f8 && !a.f8r && !a.y16_f8 && !a.k1_wrap && a.up == 4 && a.stride == 1)) return false;
  if (a.ksize != 3 || a.cin_g % 64 != 0 || a.cin_pad != a.cin_g || a.rows_g <= 128 || a.rows_g % 64 != 0 || !a.bias) return false;
  if (a.ch_scale || a.relu || a.gelu || a.res || a.res16 || a.res_after || a.accum) return false;
  return (a.ksize - 1) * a.dil <= 64;
}

template <int WR, int UMASK = -1, bool F8 = false>
static int launch_convring(const ConvArgs* a, int njobs, int rotate, int B, hipStream_t s) {
  constexpr int WC = 8 / WR, NT = WR == 1 ? 3 : 5, ROWS = 64 * WR, COLS = 16 * NT * WC, XW = WR == 4 ? 224 : WR == 2 ? 384 : 448;
  RingArgs A{};
  for (int j = 0; j < njobs; ++j) {
    RingJob& r = A.job[j];
    r.x16 = a[j].x16, r.w = a[j].w, r.bias = a[j].bias;
    r.y = a[j].no_y ? nullptr : a[j].y;
    r.y16 = a[j].y16, r.res16 = a[j].res16;
    r.w_descale = a[j].w_descale, r.y16_slope = a[j].y16_slope, r.res16_inv = a[j].res16_inv, r.accum_div = a[j].accum_div;
    r.ksize = a[j].ksize, r.dil = a[j].dil, r.pad_left = a[j].pad_left, r.accum = a[j].accum;
    r.w_bytes = (unsigned)a[j].w_gs;
    r.y_bs = a[j].y_bs, r.y_cs = a[j].y_cs;
    r.x8 = a[j].x8, r.y8 = a[j].y8, r.hi_only = a[j].y16_hi_only && a[j].y8;
    r.no_store = a[j].no_store;
    r.nstep = F8 ? 2 * (((a[j].cin_pad / (2 * CI_CHUNK)) * a[j].ksize + 1) / 2) : a[j].ksize;
  }
  A.cin_g = a[0].cin_g, A.cin_pad = a[0].cin_pad, A.rows_g = a[0].rows_g, A.co_pad = a[0].co_pad, A.T_in = a[0].T_in, A.T_q = a[0].T_q;
  A.njobs = njobs;
  A.rotate = rotate && njobs > 1;
  A.diag = g_convring & 30;
  A.n_rt = ceil_div(a[0].rows_g, ROWS);
  A.n_ct = ceil_div(a[0].T_q, COLS);
  A.total = A.n_ct * B;
  A.n_vb = 8 * A.n_rt * ceil_div(A.total, 8);
  const size_t lds_bytes = ((size_t)2 * 8 * XW + (size_t)3 * 8 * ROWS) * 16;
  // one block per CU (the LDS of a block is most of a CU's): a block walks regions blockIdx.x, + gridDim.x, ...
  const int grid = std::min(A.n_vb, std::max(8, (g_convring_blocks ? std::min(g_convring_blocks, cu_count()) : cu_count()) / 8 * 8));
  auto kern = conv1d_f16x3_ring16_kernel<WR, false, UMASK, F8>;
  auto kern_st = conv1d_f16x3_ring16_kernel<WR, UMASK < 0, UMASK, F8>;      // (no stamped build of the upsampler form)
  static std::atomic<uint64_t> attr_done{0};      // per device
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    SAT_HIP(hipFuncSetAttribute((const void*)kern_st, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  if (g_convring_dbg) hipLaunchKernelGGL(kern_st, dim3(grid), dim3(512), lds_bytes, s, A, g_convring_dbg);
  else hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, s, A, (long long*)nullptr);
  SAT_LAUNCH_CHECK(F8 ? "conv1d_f16x3_ring16_kernel (F8: 8-bit cross terms)" : "conv1d_f16x3_ring16_kernel");
  return SAT_OK;
}

// option "convring_wr": tile of the convs with more than 128 rows — 0 (default): 256 x 160; 2: 128 x 320 (bit 0: the f16x3 form, bit 1: F8)
static int g_convring_wr = 0;
void convring_set_wr(int v) { g_convring_wr = v; }
int launch_f16x3_convring_multi(const ConvArgs* a, int njobs, int rotate, int B, hipStream_t s) {
  if (a[0].f8r) return a[0].rows_g > 128 && !(g_convring_wr & 2) ? launch_convring<4, -1, true>(a, njobs, rotate, B, s) : launch_convring<2, -1, true>(a, njobs, rotate, B, s);
  return a[0].rows_g > 128 && !(g_convring_wr & 1) ? launch_convring<4>(a, njobs, rotate, B, s) : launch_convring<2>(a, njobs, rotate, B, s);
}

int launch_f16x3_convring(const ConvArgs& a, int B, hipStream_t s) { return launch_f16x3_convring_multi(&a, 1, 0, B, s); }

// the zero pattern of ConvTranspose1d(k 8, stride 4, padding 2) — phases 0, 1: slots 0, 1; phases 2, 3: slots 1, 2 — has its own instantiation;
// any other pattern multiplies everything (the zero weights are in the packed tensor)
constexpr unsigned UPS_MASK_K8 = 0x30cu;
int launch_f16x3_convring_ups(const ConvArgs& a, int B, hipStream_t s) {
  return a.up_zero_taps == UPS_MASK_K8 ? launch_convring<4, (int)UPS_MASK_K8>(&a, 1, 0, B, s) : launch_convring<4, 0>(&a, 1, 0, B, s);
}

}  // namespace sat
