// A whole multi-receptive-field block of the thin generator stages in ONE launch: for C = 16 the three ResBlock1
// branches (kernel sizes 3 / 7 / 11, each three steps x + conv2(lrelu(conv1(lrelu(x))))) with dilations 1 / 3 / 5) and
// their mean — 18 convs — run on a tile that never leaves the CU.  Reference: the `xs` loop of
// CoreHifiGan.forward_resnet, satools/satools/hifigan/archi.py:82-86; ResBlock1.forward, satools/satools/hifigan/nn.py:179-186.
//
// The launches this replaces (nine resblock_pair16 launches per stage) move the stage tensor through HBM nine times and
// read-modify-write the f32 MRF sum twice; here a tile of W = 512 output positions is loaded ONCE with the halo of the
// widest branch (12 (k - 1) / 2 = 60 positions per side at 11 taps: 23 % more columns at the first conv, none at the
// last), the running x of a branch and the intermediate t1 ping-pong between two LDS images, the MRF sum of a lane's
// outputs stays in its registers, and one store of the mean leaves the CU.
//
// Arithmetic: resblock_pair16_kernel's, instruction for instruction — v_mfma_f32_16x16x32_f16 (16 rows = the 16
// channels, K = a pair of taps x 16 channels), three split products lo*hi, hi*lo, hi*hi per tap pair in tap order, the
// intermediate x of a step re-split to hi | lo f16 planes of lrelu(x) and the residual rebuilt from those planes — so
// the result is BIT-IDENTICAL to the nine-launch path (tests/test_hip_parity.py).
//
// Layout of an LDS image: [4 planes][WP columns] 16-byte units (hi ch 0-7 | hi ch 8-15 | lo ch 0-7 | lo ch 8-15), column
// c <-> position t0 - H + c.  Lanes: A[row l & 15][k = 8 (l >> 4) ..], B[k = 8 (l >> 4) ..][column l & 15]; k-group
// g = l >> 4: tap (g >> 1) of the pair, channel half (g & 1); D: column l & 15, rows 4 g + r.
#include <algorithm>

#include "conv_common.h"

namespace sat {

constexpr int MRF_W = 512;        // output positions per tile
constexpr int MRF_THREADS = 512;  // 8 waves: subtile s (columns 16 s .. 16 s + 15 of the tile's fixed grid) belongs to wave s & 7
constexpr int MRF_HP = 64;        // grid column of the tile's first output position (>= the halo of 11 taps, 60; a multiple of 16)
constexpr int MRF_ML = 32;        // spare columns left of column 0: a subtile's taps reach up to 25 columns outside the grid
constexpr int MRF_WP = MRF_ML + 640 + 32;   // columns per plane of an LDS image

struct MrfArgs {
  const void* x16;             // input: planes of lrelu(x, slope), [B][4][T][16 B]
  float* y;                    // mean of the branches as f32 [B][16][T], or null
  void* y16;                   // ... as planes of lrelu(., y16_slope), or null
  const uint4* blob;           // the block's weights as the kernel stages them (mrf_pack_kernel): per conv [tap][hi|lo][half][16 rows]
                               // 16-byte units + 4 units of biases; convs in (branch, step, conv) order
  int T, B;
  float slope, inv_slope, y16_slope, out_div;
  int tiles_t, total, per_xcd, nslots;
  long long* dbg;              // diagnostics (sat_mrf_debug_stamps): block 0 records its waves' cycle counters per phase
};

// time stamps of block 0 (tools/stamp_mrf.py): slot [tile visit][stamp][wave]; 80 stamps per tile, 4 tile visits
constexpr int MRF_DBG_STAMPS = 80, MRF_DBG_TILES = 4;
#define MRF_STAMP()                                                                                     \
  do {                                                                                                  \
    if (p.dbg && blockIdx.x == 0 && dbg_tile < MRF_DBG_TILES) {                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      unsigned long long t_;                                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
      if (lane == 0 && dbg_idx < MRF_DBG_STAMPS) p.dbg[(dbg_tile * MRF_DBG_STAMPS + dbg_idx) * 8 + wave] = (long long)t_; \
      ++dbg_idx;                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                \
    }                                                                                                   \
  } while (0)

template <int NTP>
struct MrfFrags {               // A operand of one conv, resident in registers for its phase
  h8 hi[NTP], lo[NTP];
};

// Weights of the NEXT conv travel to their LDS slot while the current conv runs.  They come from ONE buffer that holds
// every conv of the block as the image the slot takes ([tap][hi|lo][half][16 rows] units = the packing without its
// padding rows, then 4 units of biases): a conv's address is arithmetic on the scalar unit — with one pointer per conv in
// the kernel arguments every phase began with scalar loads, and an s_load shares its counter with the LDS reads.
// Wave w moves the 64-unit pieces w, w + 8 of the image; the piece that holds the biases zero-fills the 60 units
// behind them (range check), which are the phantom tap's.
template <int KS>
__host__ __device__ constexpr int mrf_image_units() { return KS * 64 + 5; }     // taps, 16 biases, (descale, 0, 0, 0)
template <int KS>
__device__ __forceinline__ void mrf_stage_weights(uint4* slot, const uint4* image, int wave, int lane) {
  const i32x4 rs = dma_rsrc(image, (unsigned)(mrf_image_units<KS>() * 16));
#pragma unroll
  for (int r = 0; r < (KS + 1 + 7) / 8; ++r) {
    const int piece = wave + 8 * r;
    if (piece <= KS) lds_dma16(slot + piece * 64, rs, (unsigned)(lane * 16), (unsigned)(piece * 1024));
  }
}
// LDS-DMA instructions mrf_stage_weights issues on this wave
template <int KS>
__device__ __forceinline__ int mrf_stage_count(int wave) { return (KS - wave) / 8 + 1; }

template <int KS, int NTPA>
__device__ __forceinline__ void mrf_read_frags(MrfFrags<NTPA>& f, const uint4* slot, int j16, int g) {
  constexpr int NTP = (KS + 1) / 2;
  static_assert(NTP <= NTPA, "fragment set too small");
  const int gh = g & 1, gt = g >> 1;
  // ONE address register, every read an immediate offset from it: left alone, hipcc hoists a separate address VGPR per
  // read out of the step loop (24 registers for the two slots at 11 taps)
  int bo = gt * 64 + gh * 16 + j16;
  asm volatile("" : "+v"(bo));        // (the OFFSET goes through the asm: a laundered pointer loses its LDS address space -> flat loads)
  const uint4* bp = slot + bo;
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) {
    const uint4 hv = bp[tp * 128], lv = bp[tp * 128 + 32];
    if (2 * tp + 1 < KS) {
      f.hi[tp] = __builtin_bit_cast(h8, hv);
      f.lo[tp] = __builtin_bit_cast(h8, lv);
    } else {
      // last pair of an odd count: the lanes of the phantom tap (they read the 64 units behind the taps) hold zeros
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);
      f.hi[tp] = __builtin_bit_cast(h8, gt ? z : hv);
      f.lo[tp] = __builtin_bit_cast(h8, gt ? z : lv);
    }
  }
}

template <int KS>
__device__ __forceinline__ void mrf_read_bias(float (&bv)[4], const uint4* slot, int g) {
  const uint4 v = slot[KS * 64 + g];
  bv[0] = __builtin_bit_cast(float, v.x); bv[1] = __builtin_bit_cast(float, v.y);
  bv[2] = __builtin_bit_cast(float, v.z); bv[3] = __builtin_bit_cast(float, v.w);
}
template <int KS>
__device__ __forceinline__ float mrf_read_descale(const uint4* slot) {
  return __builtin_bit_cast(float, ((const unsigned*)(slot + KS * 64 + 4))[0]);
}

// packed weights + biases of every conv of the block -> the images mrf16_kernel stages (one block per conv)
struct MrfPackArgs {
  const void* w[18];
  const float* bias[18];
  int ks[18], off[18];       // taps and first unit of conv n's image
  float descale[18];
  int seg_bytes;             // co_pad * 16: bytes between the (tap, hi|lo, half) segments of the packing
};
__global__ void __launch_bounds__(256) mrf_pack_kernel(const MrfPackArgs a, uint4* __restrict__ blob) {
  const int n = blockIdx.x;
  const int nu = a.ks[n] * 64;
  const uint4* w = (const uint4*)a.w[n];
  for (int u = threadIdx.x; u < nu + 5; u += 256)
    blob[a.off[n] + u] = u < nu ? w[((u >> 4) * a.seg_bytes >> 4) + (u & 15)]
                       : u < nu + 4 ? ((const uint4*)a.bias[n])[u - nu]
                                    : make_uint4(__builtin_bit_cast(unsigned, a.descale[n]), 0u, 0u, 0u);
}

// B operand of one subtile (16 columns) for all tap pairs of a conv: read ahead of its MFMAs
template <int NTP>
struct MrfBFrags {
  h8 hi[NTP], lo[NTP];
};

//   src: the image's plane (g & 1) at the subtile's first column + (l & 15) - h * dil (margin included)
template <int KS, int NTPA>
__device__ __forceinline__ void mrf_read_b(MrfBFrags<NTPA>& f, const uint4* src, int WP, int dil, int gt) {
  constexpr int NTP = (KS + 1) / 2;
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) {
    // the phantom tap (A = 0) reads the last real tap's columns: always inside what the phase may read
    const int tap = (2 * tp + 1 < KS) ? 2 * tp + gt : KS - 1;
    const uint4* xt = src + tap * dil;
    f.hi[tp] = __builtin_bit_cast(h8, xt[0]);
    f.lo[tp] = __builtin_bit_cast(h8, xt[2 * WP]);
  }
}

// one 16-column subtile of a conv: three split products per tap pair, in tap order (resblock_pair16_kernel's order)
template <int KS, int NTPA>
__device__ __forceinline__ f32x4 mrf_mfma(const MrfFrags<NTPA>& a, const MrfBFrags<NTPA>& b) {
  constexpr int NTP = (KS + 1) / 2;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tp = 0; tp < NTP; ++tp) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo[tp], b.hi[tp], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[tp], b.lo[tp], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[tp], b.hi[tp], acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ float mrf_lrelu(float v, float slope) {      // 0 < slope < 1: max(v, slope v) == (v > 0 ? v : slope v)
  return __builtin_fmaxf(v, v * slope);
}

// four consecutive channels of one column -> the 8-byte hi and lo words of their plane unit
// (lo = v - hi in one v_fma_mix_f32 each: fma(float(hi), -1, v), exact like the subtraction)
__device__ __forceinline__ void mrf_split4(const float (&v)[4], uint2& hv, uint2& lv) {
  const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[0], v[1]));
  const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2], v[3]));
  const auto l01 = __builtin_amdgcn_cvt_pkrtz(mix_sub_half<false>(v[0], h01), mix_sub_half<true>(v[1], h01));
  const auto l23 = __builtin_amdgcn_cvt_pkrtz(mix_sub_half<false>(v[2], h23), mix_sub_half<true>(v[3], h23));
  hv = make_uint2(h01, h23);
  lv = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

// plane words of four channels -> their values (hi + lo), the leaky-relu undone: min(r, r / slope) for 0 < slope < 1
__device__ __forceinline__ void mrf_decode4(const uint2 hv, const uint2 lv, float inv_slope, float (&out)[4]) {
  out[0] = mix_add_halves<false>(hv.x, lv.x);
  out[1] = mix_add_halves<true>(hv.x, lv.x);
  out[2] = mix_add_halves<false>(hv.y, lv.y);
  out[3] = mix_add_halves<true>(hv.y, lv.y);
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = lrelu_undo_min(out[k], inv_slope);
}

// EXACT: the residual x of steps 2 and 3 stays in f32 registers (closer to the reference's f32 arithmetic than the
// 22-bit value the launch-by-launch path rebuilds from its planes, and 16 vector instructions per subtile cheaper);
// !EXACT rebuilds it from the split values — bit-identical to the launch-by-launch path.
// workgroup barrier for LDS hand-overs only: the waves' LDS accesses are retired (lgkmcnt), the global loads in flight
// (next weights, next tile) and the tile's stores stay in flight across it — __syncthreads() would drain them (vmcnt(0))
__device__ __forceinline__ void mrf_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NB, int K0, int K1, int K2, bool EXACT>
__global__ void __launch_bounds__(MRF_THREADS, 1) mrf16_kernel(const MrfArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int KMAX = NB == 1 ? K0 : (K0 > K1 ? (K0 > K2 ? K0 : K2) : (K1 > K2 ? K1 : K2));
  constexpr int HP = MRF_HP, ML = MRF_ML, WP = MRF_WP;
  static_assert(6 * (KMAX - 1) <= HP, "halo of the widest branch");
  constexpr int NTPA = (KMAX + 1) / 2;
  // images: column c of the fixed 16-column grid <-> position t0 - HP + c, at unit index ML + c of a plane
  uint4* X0 = lds4 + ML;            // the stage input of this tile (all branches start from it)
  uint4* XA = X0 + 4 * WP;          // lrelu(running x of the branch): conv1's operand
  uint4* T1 = XA + 4 * WP;          // lrelu(conv1 + b1): conv2's operand
  uint4* WA = T1 + 4 * WP - ML;     // weights of the next conv1 / conv2, each landing while the other conv runs
  uint4* WB = WA + KMAX * 64 + 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j16 = lane & 15;
  const int g = lane >> 4;
  const int gh = g & 1, gt = g >> 1;
  const unsigned OOB = 0x80000000u;

  // persistent walk (as resblock_pair16_kernel): XCD x owns a contiguous tile range, its blocks take consecutive tiles
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_end = min((xcd + 1) * p.per_xcd, p.total);
  int tile = xcd * p.per_xcd + slot;
  if (tile >= tile_end) return;

  // input staging: 2 x 320 columns per plane (positions t0 - 64 .. t0 + 576), wave = (plane, column half), five
  // 64-column LDS-DMA pieces per wave; columns outside the utterance arrive as zeros (range check).  The NEXT tile's
  // image is requested as soon as this tile's is dead (second step of the last branch) and is only waited for at the
  // end of the tile.
  const int xpl = wave & 3, xhf = wave >> 2;
  auto stage_x = [&](int tl) __attribute__((always_inline)) {
    const int ub = __builtin_amdgcn_readfirstlane(tl / p.tiles_t);
    const int pos0 = (tl - ub * p.tiles_t) * MRF_W - HP + xhf * 320;
    const i32x4 xrs = dma_rsrc((const char*)p.x16 + (long long)ub * 16 * p.T * 4, (unsigned)(16 * p.T * 4));
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int pos = pos0 + lane + 64 * it;
      const unsigned voff = (pos >= 0 && pos < p.T) ? (unsigned)((xpl * p.T + pos) * 16) : OOB;
      lds_dma16(X0 + xpl * WP + xhf * 320 + 64 * it, xrs, voff, 0u);
    }
  };

  MrfFrags<NTPA> fr;            // A operand of the running conv
  float xr[5][4];               // the running x of this lane's columns (subtile wave + 8 i, channels 4 g ..)
  f32x4 held[NB == 1 ? 1 : 2][4];   // outputs of the branches processed first (subtiles of the last phase: slot & 3)

  stage_x(tile);
  // first unit of branch j's six images (convs in (step, conv) order)
  constexpr int OFF0 = 0, OFF1 = NB == 1 ? 0 : 6 * mrf_image_units<K0>(), OFF2 = NB == 1 ? 0 : OFF1 + 6 * mrf_image_units<K1>();
  constexpr int KF = NB == 1 ? K0 : K2;               // the branch processed first
  constexpr int OFFF = NB == 1 ? OFF0 : OFF2;
  mrf_stage_weights<KF>(WA, p.blob + OFFF, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int dbg_tile = 0, dbg_idx = 0;
  for (;;) {
    const int b = __builtin_amdgcn_readfirstlane(tile / p.tiles_t);
    const int t0 = (tile - b * p.tiles_t) * MRF_W;
    const int next = tile + p.nslots;
    const bool more = next < tile_end;
    const bool edge = t0 - HP < 0 || t0 + MRF_W + HP + 16 > p.T;     // some column of the tile lies outside the utterance
    mrf_barrier();
    dbg_idx = 0;
    MRF_STAMP();

    // one ResBlock1 branch on the tile: KS taps, `last` = the branch whose end stores the tile.
    // A phase (one conv over the tile) is a software pipeline inside each wave over its subtiles ("slots" 0 .. 4: subtile
    // wave + 8 slot): per tap pair one chunk = { the 3 MFMAs of slot n | the operand reads of slot n + 1, landing in the
    // registers those MFMAs have just read | one stage of the epilogue of slot n - 1 }, chunks fenced from each other so
    // that the vector work issues in the shadow of the matrix work.  Slot 0 always runs (a wave below the phase's first
    // subtile computes columns nobody reads), slot 4 only where the phase reaches it.
    auto branch = [&](auto ks_tag, auto ksn_tag, const int woff, const int woff_next, const int order, const bool last) __attribute__((always_inline)) {
      constexpr int KS = decltype(ks_tag)::value;
      constexpr int KSN = decltype(ksn_tag)::value;      // kernel size of the conv that follows this branch
      constexpr int h = (KS - 1) / 2;
      constexpr int NTP = (KS + 1) / 2;
      struct Epi { float v[4]; uint2 hv, lv; };
      // the pipeline: `stage(st, e, acc, slot)` = stage st (0 .. 3) of a slot's epilogue
      auto run_phase = [&](const uint4* src, const int dil, const bool a4, auto&& stage) __attribute__((always_inline)) {
        MrfBFrags<NTPA> bf;
        Epi e;
        auto step = [&](const int n, const f32x4 prev, const bool has_prev, const bool has_next) __attribute__((always_inline)) -> f32x4 {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tp = 0; tp < NTP; ++tp) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr.lo[tp], bf.hi[tp], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr.hi[tp], bf.lo[tp], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr.hi[tp], bf.hi[tp], acc, 0, 0, 0);
            if (has_next) {
              const int tap = (2 * tp + 1 < KS) ? 2 * tp + gt : KS - 1;      // (phantom tap: the last real tap's columns)
              const uint4* xt = src + 128 * (n + 1) + tap * dil;
              bf.hi[tp] = __builtin_bit_cast(h8, xt[0]);
              bf.lo[tp] = __builtin_bit_cast(h8, xt[2 * WP]);
            }
            if (has_prev) {
#pragma unroll
              for (int st = 0; st < 4; ++st)
                if ((st * NTP) / 4 == tp) stage(st, e, prev, n - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          return acc;
        };
        auto drain = [&](const f32x4 acc, const int n) __attribute__((always_inline)) {
#pragma unroll
          for (int st = 0; st < 4; ++st) stage(st, e, acc, n);
        };
        mrf_read_b<KS>(bf, src, WP, dil, gt);
        const f32x4 a0 = step(0, f32x4{0.f, 0.f, 0.f, 0.f}, false, true);
        const f32x4 a1 = step(1, a0, true, true);
        const f32x4 a2 = step(2, a1, true, true);
        if (a4) {
          const f32x4 a3 = step(3, a2, true, true);
          const f32x4 a4v = step(4, a3, true, false);
          drain(a4v, 4);
        } else {
          const f32x4 a3 = step(3, a2, true, false);
          drain(a3, 3);
        }
      };
#pragma unroll 1
      for (int i = 0; i < 3; ++i) {
        const int dil = 2 * i + 1;
        const int Rn = i == 0 ? 10 * h : i == 1 ? 6 * h : 0;     // halo still needed after this step
        const uint4* xin = i == 0 ? X0 : XA;
        // ---------------- conv1: t1 = lrelu(conv1(x) + b1) on [t0 - E, t0 + W + E), E = Rn + h ----------------
        {
          const int E = Rn + h;
          const int s_hi = (HP + MRF_W + E - 1) >> 4;
          // every per-lane address of the phase is derived from this copy: left alone, hipcc computes them all ahead of the
          // step loop and keeps ~60 VGPRs of loop invariants (the store offsets of every slot, one address per LDS read)
          int jl = j16;
          asm volatile("" : "+v"(jl));
          float bias1[4];
          mrf_read_frags<KS>(fr, WA, j16, g);
          mrf_read_bias<KS>(bias1, WA, g);
          const float dsc1 = mrf_read_descale<KS>(WA);
          mrf_stage_weights<KS>(WB, p.blob + woff + (2 * i + 1) * mrf_image_units<KS>(), wave, lane);
          const bool xfetch = last && i == 1 && more;         // the input image is dead from the second step of the last branch on
          if (xfetch) stage_x(next);
          MRF_STAMP();
          uint2* const t1w = (uint2*)(T1 + gt * WP + 16 * wave + jl) + gh;
          auto stage = [&](const int st, Epi& e, const f32x4 acc, const int sl) __attribute__((always_inline)) {
            if (st == 0) {
#pragma unroll
              for (int k = 0; k < 4; ++k) e.v[k] = __builtin_fmaf(acc[k], dsc1, bias1[k]);
            } else if (st == 1) {
#pragma unroll
              for (int k = 0; k < 4; ++k) e.v[k] = mrf_lrelu(e.v[k], p.slope);
              if (edge) {                                      // t1 outside the utterance is conv2's zero padding
                const int pos = t0 - HP + 16 * (wave + 8 * sl) + jl;
                const bool inside = pos >= 0 && pos < p.T;
#pragma unroll
                for (int k = 0; k < 4; ++k) e.v[k] = inside ? e.v[k] : 0.f;
              }
            } else if (st == 2) {
              mrf_split4(e.v, e.hv, e.lv);
            } else {
              t1w[(128 * sl) * 2] = e.hv;                     // (uint2 units: 2 per 16-byte unit)
              t1w[(128 * sl + 2 * WP) * 2] = e.lv;
            }
          };
          run_phase(xin + gh * WP + 16 * wave + jl - h * dil, dil, wave + 32 <= s_hi, stage);
          MRF_STAMP();
          // the weights of conv2 have landed (the five pieces of the next tile's image, requested after them, may still fly)
          if (xfetch) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          MRF_STAMP();
          mrf_barrier();
          MRF_STAMP();
        }
        // ---------------- conv2: x' = conv2(t1) + b2 + x on [t0 - Rn, t0 + W + Rn) ----------------
        {
          const int s_hi = (HP + MRF_W + Rn - 1) >> 4;
          int jl = j16;
          asm volatile("" : "+v"(jl));
          float bias2[4];
          mrf_read_frags<KS>(fr, WB, j16, g);
          mrf_read_bias<KS>(bias2, WB, g);
          const float dsc2 = mrf_read_descale<KS>(WB);
          if (i < 2) {
            mrf_stage_weights<KS>(WA, p.blob + woff + (2 * i + 2) * mrf_image_units<KS>(), wave, lane);
          } else {
            mrf_stage_weights<KSN>(WA, p.blob + woff_next, wave, lane);
          }
          if (i == 0) {      // the residual of the first step: the stage input, from its planes
            const uint2* xw = (const uint2*)(X0 + gt * WP + 16 * wave + jl) + gh;
#pragma unroll
            for (int sl = 0; sl < 5; ++sl) mrf_decode4(xw[(128 * sl) * 2], xw[(128 * sl + 2 * WP) * 2], p.inv_slope, xr[sl]);
          }
          MRF_STAMP();
          uint2* const xaw = (uint2*)(XA + gt * WP + 16 * wave + jl) + gh;
          const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(p.y ? (char*)p.y + (long long)b * 16 * p.T * 4 : (char*)p.x16), 0, p.y ? (unsigned)(16 * p.T * 4) : 0u, 0x00020000);
          const __amdgpu_buffer_rsrc_t y16rs = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(p.y16 ? (char*)p.y16 + (long long)b * 16 * p.T * 4 : (char*)p.x16), 0, p.y16 ? (unsigned)(16 * p.T * 4) : 0u, 0x00020000);
          auto stage = [&](const int st, Epi& e, const f32x4 acc, const int sl) __attribute__((always_inline)) {
            if (i < 2) {
              if (st == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) e.v[k] = __builtin_fmaf(acc[k], dsc2, bias2[k]) + xr[sl][k];
              } else if (st == 1) {
                if (edge) {                                    // x' outside the utterance is the next conv1's zero padding
                  const int pos = t0 - HP + 16 * (wave + 8 * sl) + jl;
                  const bool inside = pos >= 0 && pos < p.T;
#pragma unroll
                  for (int k = 0; k < 4; ++k) e.v[k] = inside ? e.v[k] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  if constexpr (EXACT) xr[sl][k] = e.v[k];
                  e.v[k] = mrf_lrelu(e.v[k], p.slope);
                }
              } else if (st == 2) {
                mrf_split4(e.v, e.hv, e.lv);
                if constexpr (!EXACT) mrf_decode4(e.hv, e.lv, p.inv_slope, xr[sl]);
              } else {
                xaw[(128 * sl) * 2] = e.hv;
                xaw[(128 * sl + 2 * WP) * 2] = e.lv;
              }
            } else {
              // the branch's output on the tile proper (subtiles 4 .. 35: slots 0-3 of waves 4-7, slots 1-4 of waves 0-3):
              // held for the sum, or — last branch — summed, divided and stored
              const int si = sl & 3;
              if (st == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) e.v[k] = __builtin_fmaf(acc[k], dsc2, bias2[k]) + xr[sl][k];
              } else if (st == 1) {
                if (!last) {
#pragma unroll
                  for (int k = 0; k < 4; ++k) held[NB == 1 ? 0 : order][si][k] = e.v[k];
                } else {
                  if constexpr (NB == 3) {      // the sum in the reference's order: (rb_0 + rb_1) + rb_2, this branch being rb_0
#pragma unroll
                    for (int k = 0; k < 4; ++k) e.v[k] = (e.v[k] + held[1][si][k]) + held[0][si][k];
                  }
                  if (p.out_div != 0.f) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) e.v[k] = e.v[k] / p.out_div;
                  }
                }
              } else if (last) {
                const int qq = 16 * (wave + 8 * sl) + jl - HP;              // position inside the tile
                const int q = t0 + qq;
                const bool ok = qq >= 0 && qq < MRF_W && q < p.T;
                if (st == 2) {
                  if (p.y) {
                    const unsigned yoff = ok ? (unsigned)((4 * g) * p.T * 4 + q * 4) : OOB;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e.v[k]), yrs, yoff, k * p.T * 4, 0);
                  }
                } else if (p.y16) {
                  float u[4];
#pragma unroll
                  for (int k = 0; k < 4; ++k) u[k] = mrf_lrelu(e.v[k], p.y16_slope);
                  uint2 hv, lv;
                  mrf_split4(u, hv, lv);
                  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                  const unsigned off = ok ? (unsigned)(((0 + gt) * p.T + q) * 16 + 8 * gh) : OOB;
                  __builtin_amdgcn_raw_buffer_store_b64(u32x2{hv.x, hv.y}, y16rs, off, 0, 0);
                  __builtin_amdgcn_raw_buffer_store_b64(u32x2{lv.x, lv.y}, y16rs, off, 2 * p.T * 16, 0);
                }
              }
            }
          };
          run_phase(T1 + gh * WP + 16 * wave + jl - h, 1, wave + 32 <= s_hi, stage);
          MRF_STAMP();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next conv1's weights (and, at the end of the tile, its stores and the next image)
          MRF_STAMP();
          mrf_barrier();
          MRF_STAMP();
        }
      }
    };
    if constexpr (NB == 1) {
      branch(std::integral_constant<int, K0>{}, std::integral_constant<int, K0>{}, OFF0, OFF0, 0, true);
    } else {
      // widest branch first: while it runs (48 + 48 fragment registers at 11 taps) no other branch's output is held
      branch(std::integral_constant<int, K2>{}, std::integral_constant<int, K1>{}, OFF2, OFF1, 0, false);
      branch(std::integral_constant<int, K1>{}, std::integral_constant<int, K0>{}, OFF1, OFF0, 1, false);
      branch(std::integral_constant<int, K0>{}, std::integral_constant<int, K2>{}, OFF0, OFF2, 2, true);
    }
    if (!more) break;
    tile = next;
    ++dbg_tile;
  }
}

template <int NB, int K0, int K1, int K2, bool EXACT>
static int launch_mrf16(const MrfArgs& a, hipStream_t s) {
  MrfArgs p = a;
  constexpr int KMAX = NB == 1 ? K0 : std::max(K0, std::max(K1, K2));
  const size_t lds_bytes = ((size_t)3 * 4 * MRF_WP + 2 * (KMAX * 64 + 64)) * 16;
  auto kern = mrf16_kernel<NB, K0, K1, K2, EXACT>;
  static std::atomic<uint64_t> attr_done{0};
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  p.tiles_t = ceil_div(p.T, MRF_W);
  p.total = p.tiles_t * p.B;
  p.per_xcd = ceil_div(p.total, 8);
  p.nslots = std::max(1, std::min(32, p.per_xcd));      // one block per CU (LDS), 32 CUs per XCD
  dim3 grid(8 * p.nslots, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(MRF_THREADS), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("mrf16_kernel");
  return SAT_OK;
}

template <bool EXACT>
static int dispatch_mrf16(const MrfArgs& a, int n_branches, int k0, hipStream_t s) {
  if (n_branches == 3) return launch_mrf16<3, 3, 7, 11, EXACT>(a, s);
  switch (k0) {
    case 3: return launch_mrf16<1, 3, 0, 0, EXACT>(a, s);
    case 7: return launch_mrf16<1, 7, 0, 0, EXACT>(a, s);
    default: return launch_mrf16<1, 11, 0, 0, EXACT>(a, s);
  }
}

}  // namespace sat

using namespace sat;

static long long* g_mrf_dbg = nullptr;
extern "C" int sat_mrf_debug_stamps(int64_t* buf) {
  g_mrf_dbg = (long long*)buf;
  return MRF_DBG_STAMPS * MRF_DBG_TILES * 8;
}

extern "C" int sat_resblock_mrf_supported(int C, int n_branches, const int* ksize, const int* dilations) {
  if (C != 16 || !ksize || !dilations) return 0;
  if (n_branches == 1) {
    if (ksize[0] != 3 && ksize[0] != 7 && ksize[0] != 11) return 0;
  } else if (n_branches == 3) {
    if (ksize[0] != 3 || ksize[1] != 7 || ksize[2] != 11) return 0;
  } else {
    return 0;
  }
  for (int j = 0; j < n_branches; ++j)
    if (dilations[3 * j] != 1 || dilations[3 * j + 1] != 3 || dilations[3 * j + 2] != 5) return 0;
  return 1;
}

static size_t mrf_scratch_units(int n_branches, const int* ksize) {
  size_t u = 0;
  for (int j = 0; j < n_branches; ++j) u += (size_t)6 * (ksize[j] * 64 + 5);
  return u;
}

extern "C" size_t sat_resblock_mrf_scratch_bytes(int n_branches, const int* ksize) {
  if (n_branches < 1 || n_branches > 3 || !ksize) return 0;
  return mrf_scratch_units(n_branches, ksize) * 16;
}

extern "C" int sat_resblock_mrf_f16x3(const sat_mrf_desc* d, void* stream) {
  SAT_REQUIRE(d && d->x_split && (d->y || d->y_split) && d->scratch, "resblock_mrf: null pointer");
  SAT_REQUIRE(d->B > 0 && d->T > 0, "resblock_mrf: empty shape");
  SAT_REQUIRE(sat_resblock_mrf_supported(d->C, d->n_branches, d->ksize, &d->dilation[0][0]),
              "resblock_mrf: C = 16 with kernel sizes (3, 7, 11) or one of them, dilations (1, 3, 5) only");
  SAT_REQUIRE(d->slope > 0.f && d->slope <= 1.f && (!d->y_split || (d->y_split_slope > 0.f && d->y_split_slope <= 1.f)), "resblock_mrf: slopes must lie in (0, 1]");
  SAT_REQUIRE((long long)d->C * d->T * 4 < (1LL << 31), "resblock_mrf: slab too large for 31-bit offsets");
  SAT_REQUIRE_WORKSPACE(d->scratch_bytes >= sat_resblock_mrf_scratch_bytes(d->n_branches, d->ksize), "resblock_mrf: scratch too small");
  hipStream_t s = (hipStream_t)stream;
  // the block's 6 n_branches convs gathered into the images the kernel stages (130 KB for 3 / 7 / 11 taps; a 3 us launch)
  MrfPackArgs pa{};
  pa.seg_bytes = 64 * 16;                // co_pad = 64 rows per segment (sat_conv1d_packed_dims)
  int n = 0, off = 0;
  for (int j = 0; j < d->n_branches; ++j)
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 2; ++c, ++n) {
        SAT_REQUIRE(d->w[j][i][c] && d->bias[j][i][c], "resblock_mrf: conv (%d, %d, %d) has no weights", j, i, c);
        pa.w[n] = d->w[j][i][c];
        pa.bias[n] = d->bias[j][i][c];
        pa.ks[n] = d->ksize[j];
        pa.off[n] = off;
        pa.descale[n] = d->w_descale[j][i][c] != 0.f ? d->w_descale[j][i][c] : 1.f;
        off += d->ksize[j] * 64 + 5;
      }
  hipLaunchKernelGGL(mrf_pack_kernel, dim3(n), dim3(256), 0, s, pa, (uint4*)d->scratch);
  SAT_LAUNCH_CHECK("mrf_pack_kernel");
  MrfArgs a{};
  a.x16 = d->x_split;
  a.y = d->y;
  a.y16 = d->y_split;
  a.blob = (const uint4*)d->scratch;
  a.T = d->T;
  a.B = d->B;
  a.slope = d->slope;
  a.inv_slope = 1.0f / d->slope;
  a.y16_slope = d->y_split_slope;
  a.out_div = d->out_div;
  a.dbg = g_mrf_dbg;
  return d->residual_from_planes ? dispatch_mrf16<false>(a, d->n_branches, d->ksize[0], s)
                                 : dispatch_mrf16<true>(a, d->n_branches, d->ksize[0], s);
}
