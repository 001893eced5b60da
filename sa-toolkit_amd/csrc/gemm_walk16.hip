// Persistent form of the LDS-DMA ring GEMM on v_mfma_f32_16x16x32_f16 (gemm_ring.hip) for the Linear layers of the wav2vec2
// encoder (reference: torchaudio Wav2Vec2Model behind egs/asr/librispeech/local/chain/tuning/tdnnf_wav2vec2_vq.py:39-56, 289-314).
#include "conv_common.h"

#include <type_traits>

namespace sat {

// ------------------------------------------------------------------------------------------------
// What gemm_f16x3_ring16_kernel pays per TILE outside its K loop (one 8-wave block per CU, one tile per block): a launch or
// block hand-over, the first operands' round trip and an epilogue that nothing covers — ~12-15 us next to a 29 us K loop at
// K = 1024, four times per CU for the 1024 -> 4096 layer, and once per launch for each of q, k, v.  This kernel keeps the loop
// of gemm_f16x3_ring16_kernel (same per-accumulator arithmetic: same bits) and walks tiles with it: a block takes up to
// three GEMMs of one shape per launch (q | k | v) times its regions, and requests the next tile's first three steps — and
// its 128 biases, which travel to LDS like operands — behind the barrier that ends a K loop, BEFORE the epilogue of the
// finished tile.
// Tried and not kept: the epilogue of tile i in pieces (one 16 x 16 accumulator, no global load: bias from LDS) at the heads of
// the steps of tile i + 1.  It needs the finished accumulators in a second register set; with the A fragments flowing in
// place (40 registers instead of 64: what makes this kernel spill-free at 253 VGPRs) the loop still holds 64 accumulators +
// 64 B-fragment registers (the B set of the NEXT step must be complete before the step's buffer is refilled, two steps of
// DMA flight ahead; reading B one column ahead from the buffer in use — conv_ring16.hip does that with its separate X
// tiles — would leave the refill one step of flight): all 64 pending accumulators cost 160 spilled registers, half of
// them still 55, and hipcc reloads them from scratch INSIDE the K loop (a scratch load waits for vmcnt(0): the DMA queue).
// ------------------------------------------------------------------------------------------------
struct WalkJob {
  const void* x16;       // input planes
  const void* w;         // packed split-f16 weights
  const float* bias;
  float* y;              // f32 output, or null
  void* y16;             // output planes, or null
  const float* res;      // f32 residual (added before the activation), or null
  long long y_bs, y_cs, r_bs, r_cs;
  float w_descale, y16_slope, res_scale;
  int gelu;
  unsigned w_bytes;
  int pad_;
};
struct WalkArgs {
  WalkJob job[3];
  int cin_g, cin_pad, rows_g, co_pad, T;
  int njobs;
  int n_rt, n_ct, total, n_vb;
  int diag;         // diagnostic: 2 = no K loop, 4 = no epilogue
};

constexpr int WK_CO = 128, WK_T = 256, WK_MT = 4, WK_NT = 4;
constexpr int WK_A_UNITS = 4 * WK_CO, WK_B_UNITS = 4 * WK_T, WK_ST_UNITS = WK_A_UNITS + WK_B_UNITS;     // 16-byte units of a chunk
constexpr int WK_RING_UNITS = 6 * WK_ST_UNITS;      // three two-chunk buffers
constexpr int WK_BIAS_UNITS = 32;                   // 128 floats per tile, two tiles

// one 16 x 16 accumulator of a finished tile -> memory.  D: row 4 lg + r, column li.  `bl` = the tile's 128 biases in LDS.
__device__ __forceinline__ void walk_epilogue_piece(const WalkJob& e, const WalkArgs& A, const f32x4 acc, const int m, const int n, const int b,
                                                    const int co_w, const int q_w, const int li, const int lg, const float* bl, const bool with_res) {
  const unsigned OOB = 0x80000000u;
  const int rows_g = A.rows_g, T = A.T;
  const int row0 = co_w + m * 16 + 4 * lg, q = q_w + n * 16 + li;
  const bool qok = q < T;
  const f32x4 bi = *(const f32x4*)(bl + (row0 & (WK_CO - 1)));
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(acc[r], e.w_descale, bi[r]);
  if (with_res) {
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(e.res + (long long)b * e.r_bs), 0, (unsigned)(rows_g * e.r_cs * 4), 0x00020000);
    const int r_rb = (int)e.r_cs * 4;
    const unsigned roff = qok ? (unsigned)(row0 * r_rb + q * 4) : OOB;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += e.res_scale * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, roff + r * r_rb, 0, 0));
  }
  if (e.gelu) gelu_fast4(v);
  if (e.y) {
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(e.y + (long long)b * e.y_bs), 0, (unsigned)(rows_g * e.y_cs * 4), 0x00020000);
    const int y_rb = (int)e.y_cs * 4;
    const unsigned yoff = qok ? (unsigned)(row0 * y_rb + q * 4) : OOB;
#pragma unroll
    for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), yrs, yoff + r * y_rb, 0, 0);
  }
  if (e.y16) {
    // (rows past rows_g of the last tile: the descriptor's range check drops them)
    const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)e.y16 + (long long)b * rows_g * T * 4), 0, (unsigned)(rows_g * T * 4), 0x00020000);
    float u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = lrelu_max(v[r], e.y16_slope);
    const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
    const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
    const auto l01 = split_lo2(h01, u[0], u[1]);
    const auto l23 = split_lo2(h23, u[2], u[3]);
    // 16-byte units: lanes of even lg the hi unit, their partners lg ^ 1 the lo unit (conv_ring16.hip)
    const auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, l01), false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, l23), false, false);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 unit = {s0[0], s1[0], s0[1], s1[1]};
    const int chunk = (co_w >> 4) + m;
    const unsigned off = (qok && chunk * 16 < rows_g) ? (unsigned)(((chunk * 4 + (lg >> 1) + 2 * (lg & 1)) * T + q) * 16) : OOB;
    __builtin_amdgcn_raw_buffer_store_b128(unit, y16rs, off, 0, 0);
  }
}

__global__ void __launch_bounds__(512, 2) gemm_f16x3_walk16_kernel(const WalkArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int MT = WK_MT, NT = WK_NT, CO_B = WK_CO, T_B = WK_T, A_UNITS = WK_A_UNITS, ST_UNITS = WK_ST_UNITS;
  constexpr int PPW = 3;                               // DMA pieces per wave and chunk (24 per chunk: 8 of A, 16 of B)
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool late = wave >= 4;                         // waves k and k + 4 share a SIMD
  const int njobs = A.njobs;
  const int nreg = (A.n_vb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ntiles = nreg * njobs;
  const int nst = A.cin_pad / (2 * CI_CHUNK);          // steps of two chunks (even: the loop runs two steps per trip)
  float* const bias_lds = (float*)(lds4 + WK_RING_UNITS);

  struct Tile { int j, b, co_b, q_b; };
  auto locate = [&](int k) __attribute__((always_inline)) {
    Tile t;
    const int i = __builtin_amdgcn_readfirstlane(k / njobs);
    t.j = k - i * njobs;
    const int vb = (int)blockIdx.x + i * (int)gridDim.x;
    const int xcd = vb & 7, rest = vb >> 3;
    const int rt = __builtin_amdgcn_readfirstlane(rest % A.n_rt);
    const int g = __builtin_amdgcn_readfirstlane((rest / A.n_rt) * 8 + xcd);
    t.b = g < A.total ? __builtin_amdgcn_readfirstlane(g / A.n_ct) : -1;
    t.co_b = rt * CO_B;
    t.q_b = (g - t.b * A.n_ct) * T_B;
    return t;
  };

  // ---- DMA state of the tile whose operands are being requested: this wave's three pieces of a chunk (per-lane byte offset,
  // scalar step per chunk, LDS unit: gemm_f16x3_ring16_kernel), re-derived per tile
  i32x4 xrs, wrs, brs;
  const int seg_bytes = A.co_pad * 16, x_chunk_bytes = 4 * A.T * 16;
  unsigned voff[PPW];
  int sstep[PPW], lunit[PPW], d_co = 0;
  bool is_a[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int u = wave * PPW + j;
    is_a[j] = u < A_UNITS / 64;
    if (is_a[j]) {
      sstep[j] = 4 * seg_bytes;
      lunit[j] = (u >> 1) * CO_B + (u & 1) * 64;
    } else {
      const int v = u - A_UNITS / 64;
      sstep[j] = x_chunk_bytes;
      lunit[j] = A_UNITS + (v >> 2) * T_B + (v & 3) * 64;
    }
    sstep[j] = __builtin_amdgcn_readfirstlane(sstep[j]);
    lunit[j] = __builtin_amdgcn_readfirstlane(lunit[j]);
  }
  auto setup = [&](const Tile& t) __attribute__((always_inline)) {
    const WalkJob p = A.job[t.j];
    xrs = dma_rsrc((const char*)p.x16 + (long long)t.b * A.cin_g * A.T * 4, (unsigned)(A.cin_g * A.T * 4));
    wrs = dma_rsrc(p.w, p.w_bytes);
    brs = dma_rsrc(p.bias, (unsigned)(A.rows_g * 4));
    d_co = t.co_b;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int u = wave * PPW + j;
      if (is_a[j]) {
        const int row = t.co_b + (u & 1) * 64 + lane;
        voff[j] = row < A.co_pad ? (unsigned)(row * 16 + (u >> 1) * seg_bytes) : 0x80000000u;
      } else {
        const int v = u - A_UNITS / 64;
        const int xi = t.q_b + (v & 3) * 64 + lane;
        voff[j] = xi < A.T ? (unsigned)(((v >> 2) * A.T + xi) * 16) : 0x80000000u;
      }
    }
  };
  // step st -> chunk buffers 2 sb, 2 sb + 1
  auto issue = [&](int st, int sb) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint4* dst = lds4 + (2 * sb + h) * ST_UNITS;
      const int ch = 2 * st + h;
#pragma unroll
      for (int j = 0; j < PPW; ++j) lds_dma16(dst + lunit[j], is_a[j] ? wrs : xrs, voff[j], (unsigned)(ch * sstep[j]));
    }
  };
  // first requests of a tile: steps 0 .. 2 and (wave 0, lanes 0-31) the 128 biases of its rows -> bias slot `par`
  auto issue_prologue = [&](int par) __attribute__((always_inline)) {
    issue(0, 0);
    if (nst > 1) issue(1, 1);
    if (nst > 2) issue(2, 2);
    if (wave == 0) {
      const int row = d_co + 4 * lane;
      const unsigned voff = row < A.rows_g ? (unsigned)(row * 4) : 0x80000000u;
      lds_dma16_lo32(lds4 + WK_RING_UNITS + par * WK_BIAS_UNITS, brs, voff, 0u);       // (lanes 0-31: 32 x 16 bytes)
    }
  };

  // B fragments of the step being multiplied and of the next one; A fragments flow one row behind (gemm_f16x3_ring16_kernel)
  h8 fa[MT - 1][2], fa3[2][2], fb[2][NT][2];
  auto frag_base = [&](int sb) __attribute__((always_inline)) { return lds4 + (2 * sb + (lg >> 1)) * ST_UNITS; };
  auto read_a = [&](h8 (&dst)[2], int sb, int m) __attribute__((always_inline)) {
    const uint4* wb = frag_base(sb) + (lg & 1) * CO_B + wm * 64 + li + m * 16;
    dst[0] = __builtin_bit_cast(h8, wb[0]);
    dst[1] = __builtin_bit_cast(h8, wb[2 * CO_B]);
  };
  auto read_b = [&](int buf, int sb) __attribute__((always_inline)) {
    const uint4* xb = frag_base(sb) + A_UNITS + (lg & 1) * T_B + wn * 64 + li;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      fb[buf][n][0] = __builtin_bit_cast(h8, xb[n * 16]);
      fb[buf][n][1] = __builtin_bit_cast(h8, xb[2 * T_B + n * 16]);
    }
  };

  f32x4 acc[MT][NT];
  int st = 0, sb = 0;
  auto body = [&](auto cur, auto late_c, auto more_c) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur)::value;
    constexpr bool LATE = decltype(late_c)::value, MORE = decltype(more_c)::value;
    constexpr int M_HAND = LATE ? 2 : 0;     // the hand-over work stands in front of this row's MFMAs
    int nxt = sb;
    if constexpr (MORE) {
      // hand-over: step st + 1 landed (this wave's pieces: all but the youngest step), this wave's reads of step st are back;
      // behind the barrier everybody's are, and step st's buffer is free for step st + 3
      if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      nxt = sb == 2 ? 0 : sb + 1;
    }
    // A fragments flow IN PLACE: row m's pair is dead once row m's MFMAs have been issued (in-order issue: they have read it; the
    // LDS data of a read issued behind them arrives tens of cycles later), so the next step's row m is read into the same
    // registers in front of row m + 1 — row 3's, whose turn would come behind the step's last MFMA, into a spare pair
    // (fa3[CUR ^ 1]) in front of row 3: 40 A registers instead of 64
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MORE) {
        if (m == M_HAND) {
          if (st + 3 < nst) issue(st + 3, sb);
          read_b(CUR ^ 1, nxt);
        }
        if (m >= 1 && m > M_HAND) read_a(fa[m - 1], nxt, m - 1);
        if (m == M_HAND && M_HAND > 0) {
#pragma unroll
          for (int k2 = 0; k2 < M_HAND; ++k2) read_a(fa[k2], nxt, k2);
        }
        if (m == MT - 1) read_a(fa3[CUR ^ 1], nxt, MT - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (m < MT - 1) {
          mfma16_acc(acc[m][n], fa[m][1], fb[CUR][n][0]);
          mfma16_acc(acc[m][n], fa[m][0], fb[CUR][n][1]);
          mfma16_acc(acc[m][n], fa[m][0], fb[CUR][n][0]);
        } else {
          mfma16_acc(acc[m][n], fa3[CUR][1], fb[CUR][n][0]);
          mfma16_acc(acc[m][n], fa3[CUR][0], fb[CUR][n][1]);
          mfma16_acc(acc[m][n], fa3[CUR][0], fb[CUR][n][0]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ++st;
    sb = nxt;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto loop = [&](auto late_c) __attribute__((always_inline)) {
    using L = decltype(late_c);
    while (st + 2 < nst) {
      body(I0{}, L{}, std::true_type{});
      body(I1{}, L{}, std::true_type{});
    }
    body(I0{}, L{}, std::true_type{});       // (an even number of steps: the last one is peeled)
    body(I1{}, L{}, std::false_type{});
  };

  // ---- the walk
  int k = 0;
  Tile cur = locate(0);
  while (k < ntiles && cur.b < 0) cur = locate(++k);
  if (k >= ntiles) return;
  int par = 0;
  setup(cur);
  issue_prologue(par);
  while (true) {
    // steps 0 (.. 2) and the biases requested; step 0 landed: all but the two youngest steps' pieces (and older stores)
    if (nst > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    read_b(0, 0);
#pragma unroll
    for (int m = 0; m < MT - 1; ++m) read_a(fa[m], 0, m);
    read_a(fa3[0], 0, MT - 1);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 3" ::: "memory");
    st = 0, sb = 0;
    if (!(A.diag & 2)) {
      if (late) loop(std::true_type{});
      else loop(std::false_type{});
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const Tile done = cur;
    const int done_par = par;
    int kn = k + 1;
    Tile nxt_t = done;
    while (kn < ntiles) {
      nxt_t = locate(kn);
      if (nxt_t.b >= 0) break;
      ++kn;
    }
    const bool more_tiles = kn < ntiles;
    if (more_tiles) {
      // every wave has read its last fragments (and the biases of the tile before this one): the rings are free
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      par ^= 1;
      setup(nxt_t);
      issue_prologue(par);
    }
    mfma16_drain(acc);
    if (!(A.diag & 4)) {
      const WalkJob e = A.job[done.j];
      const float* bl = bias_lds + done_par * (WK_BIAS_UNITS * 4);
      const int co_w = done.co_b + wm * 64, q_w = done.q_b + wn * 64;
      const bool with_res = e.res != nullptr;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) walk_epilogue_piece(e, A, acc[m][n], m, n, done.b, co_w, q_w, li, lg, bl, with_res);
    }
    if (!more_tiles) break;
    k = kn;
    cur = nxt_t;
  }
}

// option "gemm_walk": 0 = off, 1 = on (default), + 2 = also for single GEMMs with no more tiles than CUs (tests), + 8 / 16 = diagnostic
// builds without the K loop / without the epilogue
static int walk_cu_count() {
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

static int g_walk = 1;
void gemm_walk_set(int v) { g_walk = v; }

// 1x1 convs on split planes whose epilogue this kernel carries: bias, f32 residual before the activation, GELU, f32 and / or
// plane stores (no folded BatchNorm, no ReLU, no accumulation, no residual from planes, no wrapped K)
bool gemm_walk_supports(const ConvArgs& a) {
  if (!(g_walk & 1)) return false;
  if (!(a.x16 && a.ksize == 1 && !a.f8 && !a.y16_f8 && !a.poly_planes && !a.k1_wrap && a.fast_epi && a.up == 1 && a.stride == 1 && a.pad_left == 0)) return false;
  if (!epilogue16_supports(a) || !a.bias || a.ch_scale || a.relu || a.accum || a.accum_div != 0.f || a.res16 || a.res_after) return false;
  if (a.res && (a.res_toff != 0 || a.res_tstride != 1)) return false;
  return a.rows_g >= 128 && a.co_pad % 128 == 0 && (a.cin_pad / CI_CHUNK) % 4 == 0 && a.T_in == a.T_q && a.cin_g % 16 == 0;
}

bool gemm_walk_wanted(long long tiles) { return (g_walk & 2) || tiles > walk_cu_count(); }

bool gemm_walk_same_shape(const ConvArgs& a, const ConvArgs& b) {
  return a.cin_g == b.cin_g && a.cin_pad == b.cin_pad && a.rows_g == b.rows_g && a.co_pad == b.co_pad && a.T_in == b.T_in && a.T_q == b.T_q;
}

int launch_f16x3_gemm_walk(const ConvArgs* a, int njobs, int B, hipStream_t s) {
  WalkArgs A{};
  for (int j = 0; j < njobs; ++j) {
    WalkJob& r = A.job[j];
    r.x16 = a[j].x16, r.w = a[j].w, r.bias = a[j].bias;
    r.y = a[j].no_y ? nullptr : a[j].y;
    r.y16 = a[j].y16, r.res = a[j].res;
    r.y_bs = a[j].y_bs, r.y_cs = a[j].y_cs, r.r_bs = a[j].r_bs, r.r_cs = a[j].r_cs;
    r.w_descale = a[j].w_descale, r.y16_slope = a[j].y16_slope, r.res_scale = a[j].res_scale;
    r.gelu = a[j].gelu;
    r.w_bytes = (unsigned)a[j].w_gs;
  }
  A.cin_g = a[0].cin_g, A.cin_pad = a[0].cin_pad, A.rows_g = a[0].rows_g, A.co_pad = a[0].co_pad, A.T = a[0].T_q;
  A.njobs = njobs;
  A.n_rt = ceil_div(A.rows_g, WK_CO);
  A.n_ct = ceil_div(A.T, WK_T);
  A.total = A.n_ct * B;
  A.n_vb = 8 * A.n_rt * ceil_div(A.total, 8);
  A.diag = (g_walk >> 2) & 6;
  const size_t lds_bytes = ((size_t)WK_RING_UNITS + 2 * WK_BIAS_UNITS) * 16;
  const int grid = std::min(A.n_vb, std::max(8, walk_cu_count() / 8 * 8));
  auto kern = gemm_f16x3_walk16_kernel;
  static std::atomic<uint64_t> attr_done{0};      // per device
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, s, A);
  SAT_LAUNCH_CHECK("gemm_f16x3_walk16_kernel");
  return SAT_OK;
}

}  // namespace sat
