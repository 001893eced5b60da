// LDS-DMA ring GEMM for 1x1 convolutions on split planes (wav2vec2 Linear layers, TDNNF linearA, output affines).
#include "conv_common.h"

#include <type_traits>

#ifndef SAT_RING_ABLATE
#define SAT_RING_ABLATE 0
#endif

namespace sat {

// ------------------------------------------------------------------------------------------------
// 1x1 convolution on split planes, ring form: BOTH operands go global -> LDS by LDS-DMA (`buffer_load ... lds`,
// 16 bytes per lane, no staging registers, no ds_write pass) into a ring of six one-chunk buffers, with the loads of
// four chunks in flight across the barriers: per 16-channel chunk ONE raw s_barrier behind a COUNTED s_waitcnt vmcnt
// (never 0 before the tail; hipcc keeps ds_reads clear of pending LDS-DMA of other buffers as long as all LDS is one
// array — MI355X guide, "Pipelining across barriers"), and the barrier stands between a chunk's fragment reads and
// its MFMAs: a wave arrives with 12 MFMAs' operands already in registers, passes, issues the next chunk's loads and
// fragment reads and only then multiplies — the matrix pipe never waits for an LDS round trip behind a barrier.
// Block = 8 waves as 2 (rows) x 4 (columns) over a 128-row x 256-column tile, wave tile 64 x 64 (A and B fragment
// reuse 2 each: 8 ds_read_b128 per 12 MFMAs), per chunk:
//   A  [hi|lo][half][128 rows]    x 16 B    8 KB — the packed weights as they lie in memory
//   B  [hi|lo][half][256 columns] x 16 B   16 KB — the planes as they lie in memory
// Both images are lane-linear (a wave-instruction = 64 consecutive 16-byte units of one segment), which is what
// LDS-DMA needs and what makes every fragment read a conflict-free ds_read_b128.  6 x 24 KB of LDS, one block per
// CU; 1024-row layers at 249 frames x 32 utterances = 256 blocks = one per CU, 4096 rows = 4 per CU.
// Per-accumulator arithmetic (chunk order, lo*hi, hi*lo, hi*hi) is that of conv1d_f16x3_k1_kernel: same bits.
// Against that kernel (weights through registers + ds_write, activations straight to registers, 128 x 128 tile,
// two blocks per CU): half the L2 -> CU bytes per MFMA, no VGPRs spent on staging, loads four chunks ahead.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 2) gemm_f16x3_ring_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int CO_B = 128, T_B = 256, MT = 2, NT = 2, RING = 6;
  constexpr int A_UNITS = 4 * CO_B, B_UNITS = 4 * T_B, ST_UNITS = A_UNITS + B_UNITS;     // 16-byte units of a chunk
  constexpr int A_PIECES = A_UNITS / 64, PIECES = ST_UNITS / 64, PPW = PIECES / 8;       // 1 KB wave-instructions
  static_assert(PIECES % 8 == 0 && PPW == 3 && RING == 6, "the vmcnt literals below assume three pieces per wave and chunk, six buffers");
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // block -> (co tile, time tile, utterance): the co tiles of one (utterance, time tile) share an XCD (its L2 serves
  // the activations to all of them); division results kept scalar (conv1d_f16x3_k1_kernel)
  const int n_co = p.co_tiles_g, n_tt = p.pp_tiles_t;
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int co_t = __builtin_amdgcn_readfirstlane(rest % n_co);
  const int g = __builtin_amdgcn_readfirstlane((rest / n_co) * 8 + xcd);
  if (g >= p.pp_total) return;
  const int b = __builtin_amdgcn_readfirstlane(g / n_tt);
  const int co_b = co_t * CO_B;
  const int q_b = (g - b * n_tt) * T_B;
  const int nch = p.cin_pad / CI_CHUNK;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x16 + (long long)b * p.cin_g * p.T_in * 4), 0, (unsigned)(p.cin_g * p.T_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;
  const int x_chunk_bytes = 4 * p.T_in * 16;

  // this wave's three pieces of a chunk: per-lane byte offset, scalar step per chunk, LDS unit
  unsigned voff[PPW];
  int sstep[PPW], lunit[PPW];
  bool is_a[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int u = wave * PPW + j;
    is_a[j] = u < A_PIECES;
    if (is_a[j]) {
      const int seg = u >> 1, half = u & 1;
      voff[j] = (unsigned)((co_b + half * 64 + lane) * 16 + seg * seg_bytes);
      sstep[j] = 4 * seg_bytes;
      lunit[j] = seg * CO_B + half * 64;
    } else {
      const int v = u - A_PIECES;
      const int plane = v >> 2, cq = v & 3;
      const int xi = q_b + cq * 64 + lane - p.pad_left;
      voff[j] = (xi >= 0 && xi < p.T_in) ? (unsigned)((plane * p.T_in + xi) * 16) : 0x80000000u;
      sstep[j] = x_chunk_bytes;
      lunit[j] = A_UNITS + plane * T_B + cq * 64;
    }
    // wave-uniform by construction; made PROVABLY scalar, or every LDS-DMA below sits in a waterfall loop over its
    // scalar offset (a v_readfirstlane / s_and_saveexec round per piece)
    sstep[j] = __builtin_amdgcn_readfirstlane(sstep[j]);
    lunit[j] = __builtin_amdgcn_readfirstlane(lunit[j]);
  }
  auto issue = [&](int ch, int ring) {
    uint4* dst = lds4 + ring * ST_UNITS;
    // device pass only: in the host pass this target builtin is a deferred error that silently drops the kernel's
    // host stub (the handle stays an undefined external and the fat binary is not embedded)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < PPW; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a[j] ? wrs : xrs, (__attribute__((address_space(3))) void*)(uintptr_t)(dst + lunit[j]), 16,
                                               voff[j], ch * sstep[j], 0, 0);
#else
    (void)dst; (void)xrs; (void)wrs;
#endif
  };
  h8 fa[2][MT][2], fb[2][NT][2];          // fragments of the chunk being multiplied and of the next one
  auto read_frags = [&](int buf, int ring) {
    const uint4* wb = lds4 + ring * ST_UNITS + lh * CO_B + wm * 64 + l31;
    const uint4* xb = lds4 + ring * ST_UNITS + A_UNITS + lh * T_B + wn * 64 + l31;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      fa[buf][m][0] = __builtin_bit_cast(h8, wb[0 * CO_B + m * 32]);
      fa[buf][m][1] = __builtin_bit_cast(h8, wb[2 * CO_B + m * 32]);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      fb[buf][n][0] = __builtin_bit_cast(h8, xb[0 * T_B + n * 32]);
      fb[buf][n][1] = __builtin_bit_cast(h8, xb[2 * T_B + n * 32]);
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // prologue: chunks 0..4 in flight, chunk 0 landed and visible, its fragments read
#pragma unroll
  for (int c = 0; c < RING - 1; ++c)
    if (c < nch) issue(c, c);
  if (nch > RING - 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // all but the four youngest chunks = chunk 0
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  read_frags(0, 0);
  int ring = 0;                              // buffer of chunk c
  // The two waves that share a SIMD (k and k + 4) run the same program behind the same barrier: left alone they do their
  // loads / fragment reads at the same time and then queue their MFMAs behind each other, and the matrix pipe idles
  // through every hand-over (s_memtime stamps: 1 180 cycles per chunk against 768 of MFMAs, waves 0-3 waiting a third
  // of the loop at the barrier for waves 4-7).  Waves 4-7 therefore take their hand-over work AFTER the first half
  // of the chunk's MFMAs: one partner multiplies while the other issues its DMA and reads (MI355X guide, "Two waves
  // that run the SAME program with one barrier per block: try a stagger").
  // The loop is instantiated per role (LATE) and the last chunk is peeled (MORE), so that which register set a
  // fragment read fills is static everywhere: with runtime conditions around the reads hipcc keeps the two sets apart
  // by copying them (32 v_mov_b64 per chunk and wave).
  auto body = [&](int c, auto cur, auto late_c, auto more_c) {
    constexpr int CUR = decltype(cur)::value;
    constexpr bool LATE = decltype(late_c)::value, MORE = decltype(more_c)::value;
    int nxt = ring;
    if constexpr (MORE) {
      // hand-over point: chunk c + 1 landed (this wave's pieces: all but the three youngest chunks c+2..c+4), this wave's
      // reads of chunk c are back; then everybody's are, and everybody's chunk c + 1 is visible
      if (c + RING - 2 < nch) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      nxt = ring == RING - 1 ? 0 : ring + 1;
    }
    auto handover_work = [&]() {
      if (c + RING - 1 < nch) issue(c + RING - 1, ring == 0 ? RING - 1 : ring - 1);   // buffer of chunk c - 1: every wave has left it
      read_frags(CUR ^ 1, nxt);
    };
    if constexpr (MORE && !LATE) handover_work();
    __builtin_amdgcn_sched_barrier(0);       // the reads above stay ahead of these MFMAs (they feed the NEXT chunk)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m == 1) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE && LATE) handover_work();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[CUR][m][1], fb[CUR][n][0], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[CUR][m][0], fb[CUR][n][1], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[CUR][m][0], fb[CUR][n][0], acc[m][n], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ring = nxt;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto loop = [&](auto late_c) {
    using L = decltype(late_c);
    int c = 0;
    for (; c + 2 < nch; c += 2) {            // two chunks per trip: the fragment register sets alternate statically
      body(c, I0{}, L{}, std::true_type{});
      body(c + 1, I1{}, L{}, std::true_type{});
    }
    if (c + 1 < nch) {                       // an even number of chunks is left: the last one is peeled
      body(c, I0{}, L{}, std::true_type{});
      body(c + 1, I1{}, L{}, std::false_type{});
    } else if (c < nch) {
      body(c, I0{}, L{}, std::false_type{});
    }
  };
  // The two waves that share a SIMD (k and k + 4) run the same program behind the same barrier: waves 4-7 take their
  // hand-over work (DMA issue, fragment reads) AFTER the first half of the chunk's MFMAs, waves 0-3 before them (MI355X
  // guide, "Two waves that run the SAME program with one barrier per block: try a stagger").
  if (wave >= 4) loop(std::true_type{});
  else loop(std::false_type{});
  // the residual of the whole 64 x 64 wave tile is requested in one go (the fragment registers are dead by now): one
  // memory round trip in the exposed epilogue of a one-block-per-CU kernel instead of one per 32 x 32 sub-tile
  float rpre[MT][NT][16];
  if (p.res || p.res16) epilogue_prefetch_res<MT, NT>(p, rpre, b, 0, co_b + wm * 64, q_b + wn * 64, l31, lh);
  conv_epilogue<MT, NT, true>(p, acc, b, 0, co_b + wm * 64, q_b + wn * 64, l31, lh, 32, 0x7fffffff, rpre);
}

// ------------------------------------------------------------------------------------------------
// The same ring on the 16x16x32 MFMA shape.  A bare loop of the three split products holds a ~20 % higher clock on
// v_mfma_f32_16x16x32_f16 than on v_mfma_f32_32x32x16_f16 at the same cycles per FLOP (tools/mfma_rate: 731 against
// 608 useful TFLOP/s; the chip's power management gives less back to the cheaper shape — MI355X guide, DVFS give-back),
// and the LDS images need no change: K = 32 of one instruction = the two halves of TWO consecutive chunks, lane
// (i = lane & 15, g = lane >> 4) reads the 16-byte unit (chunk g >> 1, half g & 1, row / column i) of each operand.
// One step = two chunks: one barrier, six DMA pieces and sixteen ds_read_b128 per wave and 48 MFMAs (768 cycles);
// three two-chunk buffers, two steps of loads in flight across the barriers.  The wave tile stays 64 x 64 = 4 x 4
// accumulators of 16 x 16; D: row 4 g + r, column i.  A sum over K = 32 inside one instruction associates differently
// from two K = 16 instructions: results agree with the 32x32 kernels to f32 rounding of the accumulation (~1e-7
// relative), not bit for bit.
// Epilogue: conv_epilogue16 (everything but e4m3 planes, which stay on the 32x32 kernel, launch_f16x3_ring).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 2) gemm_f16x3_ring16_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int CO_B = 128, T_B = 256, MT = 4, NT = 4, RING = 6;
  constexpr int A_UNITS = 4 * CO_B, B_UNITS = 4 * T_B, ST_UNITS = A_UNITS + B_UNITS;
  constexpr int A_PIECES = A_UNITS / 64, PIECES = ST_UNITS / 64, PPW = PIECES / 8;
  static_assert(PPW == 3 && RING == 6, "the vmcnt literals below assume six pieces per wave and step, three step buffers");
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int n_co = p.co_tiles_g, n_tt = p.pp_tiles_t;
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int co_t = __builtin_amdgcn_readfirstlane(rest % n_co);
  const int g = __builtin_amdgcn_readfirstlane((rest / n_co) * 8 + xcd);
  if (g >= p.pp_total) return;
  const int b = __builtin_amdgcn_readfirstlane(g / n_tt);
  const int co_b = co_t * CO_B;
  const int q_b = (g - b * n_tt) * T_B;
  const int nst = p.cin_pad / (2 * CI_CHUNK);          // steps of two chunks

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x16 + (long long)b * p.cin_g * p.T_in * 4), 0, (unsigned)(p.cin_g * p.T_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;
  const int x_chunk_bytes = 4 * p.T_in * 16;

  // K chunks past p.k1_wrap read the planes again from chunk 0 on, one position later (x_wrap_channels of the desc:
  // the third tap of a stride-2 3-tap conv over phase-split input); voff1 = a B piece's offsets at that shift
  const int nwrap = p.k1_wrap > 0 ? p.k1_wrap : 0x7fffffff;
  unsigned voff[PPW], voff1[PPW];
  int sstep[PPW], lunit[PPW];
  bool is_a[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int u = wave * PPW + j;
    is_a[j] = u < A_PIECES;
    if (is_a[j]) {
      const int seg = u >> 1, half = u & 1;
      voff[j] = voff1[j] = (unsigned)((co_b + half * 64 + lane) * 16 + seg * seg_bytes);
      sstep[j] = 4 * seg_bytes;
      lunit[j] = seg * CO_B + half * 64;
    } else {
      const int v = u - A_PIECES;
      const int plane = v >> 2, cq = v & 3;
      const int xi = q_b + cq * 64 + lane - p.pad_left;
      voff[j] = (xi >= 0 && xi < p.T_in) ? (unsigned)((plane * p.T_in + xi) * 16) : 0x80000000u;
      voff1[j] = (xi + 1 >= 0 && xi + 1 < p.T_in) ? (unsigned)((plane * p.T_in + xi + 1) * 16) : 0x80000000u;
      sstep[j] = x_chunk_bytes;
      lunit[j] = A_UNITS + plane * T_B + cq * 64;
    }
    sstep[j] = __builtin_amdgcn_readfirstlane(sstep[j]);
    lunit[j] = __builtin_amdgcn_readfirstlane(lunit[j]);
  }
  // step st -> chunk buffers 2 * sb, 2 * sb + 1 (sb = st mod 3)
  auto issue = [&](int st, int sb) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint4* dst = lds4 + (2 * sb + h) * ST_UNITS;
      const int ch = 2 * st + h;
      const bool wrapped = ch >= nwrap;                  // wave-uniform (and the same for both chunks of a step: nwrap is even)
      const int ch_b = wrapped ? ch - nwrap : ch;
#pragma unroll
      for (int j = 0; j < PPW; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a[j] ? wrs : xrs, (__attribute__((address_space(3))) void*)(uintptr_t)(dst + lunit[j]), 16,
                                                 wrapped ? voff1[j] : voff[j], (is_a[j] ? ch : ch_b) * sstep[j], 0, 0);
    }
#else
    (void)st; (void)sb; (void)xrs; (void)wrs; (void)nwrap;
#endif
  };
  // B fragments of the step being multiplied and of the next one; A fragments flow: row m's pair is dead after row
  // m's MFMAs, so the next step's rows are read one row behind the multiplication (5 pairs live, not 8)
  h8 fa[MT][2], fb[2][NT][2];
  auto frag_base = [&](int sb) { return lds4 + (2 * sb + (lg >> 1)) * ST_UNITS; };
  auto read_a = [&](h8 (&dst)[2], int sb, int m) __attribute__((always_inline)) {
    const uint4* wb = frag_base(sb) + (lg & 1) * CO_B + wm * 64 + li + m * 16;
    dst[0] = __builtin_bit_cast(h8, wb[0 * CO_B]);
    dst[1] = __builtin_bit_cast(h8, wb[2 * CO_B]);
  };
  auto read_b = [&](int buf, int sb) __attribute__((always_inline)) {
    const uint4* xb = frag_base(sb) + A_UNITS + (lg & 1) * T_B + wn * 64 + li;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      fb[buf][n][0] = __builtin_bit_cast(h8, xb[0 * T_B + n * 16]);
      fb[buf][n][1] = __builtin_bit_cast(h8, xb[2 * T_B + n * 16]);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: steps 0..2 in flight, step 0 landed and visible, its fragments read
#pragma unroll
  for (int st = 0; st < 3; ++st)
    if (st < nst) issue(st, st);
  if (nst > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (nst == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  int sb = 0;                                // buffer of step st
  // Waves k and k + 4 share a SIMD: waves 4-7 take their hand-over work (DMA issue, fragment reads) after the first
  // half of the step's MFMAs, waves 0-3 before them (see gemm_f16x3_ring_kernel).  The loop is instantiated per role
  // and the last step peeled, so that the register set of every fragment read is static.
  auto body = [&](int st, auto cur, auto late_c, auto more_c) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur)::value;
    constexpr bool LATE = decltype(late_c)::value, MORE = decltype(more_c)::value;
    constexpr int M_HAND = LATE ? 2 : 0;     // the hand-over work stands in front of this row's MFMAs
    int nxt = sb;
    if constexpr (MORE) {
      // hand-over: step st + 1 landed (this wave's pieces: all but the youngest step), this wave's reads of step st are
      // back; behind the barrier everybody's are, and step st's buffer is free for step st + 3
      if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if !(SAT_RING_ABLATE & 1)                     // tools/ablate_ring.hip (diagnostic builds: 1 no barrier, 2 no DMA issue, 4 no fragment reads in the loop; results are wrong)
      asm volatile("s_barrier" ::: "memory");
#endif
      nxt = sb == 2 ? 0 : sb + 1;
    }
    h8 an[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MORE) {
        if (m == M_HAND) {
#if !(SAT_RING_ABLATE & 2)
          if (st + 3 < nst) issue(st + 3, sb);
#endif
#if !(SAT_RING_ABLATE & 4)
          read_b(CUR ^ 1, nxt);
#pragma unroll
          for (int k = 0; k <= M_HAND; ++k) read_a(an[k], nxt, k);
#else
          fb[CUR ^ 1][0][0] = fb[CUR][0][0]; fb[CUR ^ 1][0][1] = fb[CUR][0][1]; fb[CUR ^ 1][1][0] = fb[CUR][1][0]; fb[CUR ^ 1][1][1] = fb[CUR][1][1];
          fb[CUR ^ 1][2][0] = fb[CUR][2][0]; fb[CUR ^ 1][2][1] = fb[CUR][2][1]; fb[CUR ^ 1][3][0] = fb[CUR][3][0]; fb[CUR ^ 1][3][1] = fb[CUR][3][1];
          for (int k = 0; k < MT; ++k) an[k][0] = fa[k][0], an[k][1] = fa[k][1];
#endif
        } else if (m > M_HAND) {
#if !(SAT_RING_ABLATE & 4)
          read_a(an[m], nxt, m);             // row m - 1 has been multiplied: a pair of registers is free
#endif
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        mfma16_acc(acc[m][n], fa[m][1], fb[CUR][n][0]);
        mfma16_acc(acc[m][n], fa[m][0], fb[CUR][n][1]);
        mfma16_acc(acc[m][n], fa[m][0], fb[CUR][n][0]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MORE) {
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[m][0] = an[m][0], fa[m][1] = an[m][1];
    }
    sb = nxt;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto loop = [&](auto late_c) __attribute__((always_inline)) {
    using L = decltype(late_c);
    read_b(0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) read_a(fa[m], 0, m);
    int st = 0;
    for (; st + 2 < nst; st += 2) {
      body(st, I0{}, L{}, std::true_type{});
      body(st + 1, I1{}, L{}, std::true_type{});
    }
    if (st + 1 < nst) {
      body(st, I0{}, L{}, std::true_type{});
      body(st + 1, I1{}, L{}, std::false_type{});
    } else if (st < nst) {
      body(st, I0{}, L{}, std::false_type{});
    }
  };
  if (wave >= 4) loop(std::true_type{});
  else loop(std::false_type{});

  // the residual of the whole wave tile is requested in one go (the fragment registers are dead by now): one memory
  // round trip in the exposed epilogue of a one-block-per-CU kernel
  mfma16_drain(acc);
  f32x4 rpre[MT][NT];
  if (p.res || p.res16) epilogue16_prefetch_res<MT, NT>(p, rpre, b, co_b + wm * 64, q_b + wn * 64, li, lg);
  conv_epilogue16<MT, NT>(p, acc, rpre, b, co_b + wm * 64, q_b + wn * 64, li, lg);
}

// options the 16x16 kernel's epilogue carries (the rest stays on the 32x32 kernel)
bool ring16_supports(const ConvArgs& a) { return epilogue16_supports(a) && (a.cin_pad / CI_CHUNK) % 2 == 0 && a.k1_wrap % 2 == 0; }

int launch_f16x3_ring16(const ConvArgs& a, int B, hipStream_t s) {
  ConvArgs p = a;
  p.xw = 256;
  p.co_tiles_g = ceil_div(p.rows_g, 128);
  p.pp_tiles_t = ceil_div(p.T_q, 256);
  p.pp_total = p.pp_tiles_t * B;
  const size_t lds_bytes = (size_t)6 * (4 * 128 + 4 * 256) * 16;
  auto kern = gemm_f16x3_ring16_kernel;
  static std::atomic<uint64_t> attr_done{0};      // per device
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  dim3 grid(8 * p.co_tiles_g * ceil_div(p.pp_total, 8), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("gemm_f16x3_ring16_kernel");
  return SAT_OK;
}

int launch_f16x3_ring(const ConvArgs& a, int B, hipStream_t s) {
  ConvArgs p = a;
  p.xw = 256;
  p.co_tiles_g = ceil_div(p.rows_g, 128);
  p.pp_tiles_t = ceil_div(p.T_q, 256);
  p.pp_total = p.pp_tiles_t * B;
  const size_t lds_bytes = (size_t)6 * (4 * 128 + 4 * 256) * 16;
  auto kern = gemm_f16x3_ring_kernel;
  static std::atomic<uint64_t> attr_done{0};      // per device
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  dim3 grid(8 * p.co_tiles_g * ceil_div(p.pp_total, 8), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("gemm_f16x3_ring_kernel");
  return SAT_OK;
}

}  // namespace sat
