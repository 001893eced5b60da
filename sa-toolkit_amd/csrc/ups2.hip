// The two thin upsamplers of the generator — ConvTranspose1d(C -> C / 2, k = 4, stride 2, padding 1) at C = 64 and 32
// (satools/satools/hifigan/archi.py:47-59, 80-81: the last two stages) — as a streaming kernel on split planes.
//
// These launches are HBM-bound (a stage tensor in, one out: 328 MB at 32 x 5 s, 55 us at 6 TB/s) and ran 2.4 x off that
// floor on the general polyphase conv tile (3 tap slots of which 2 are non-zero per phase, an LDS-transposed epilogue).
// Here: output t = 2 q + r reads two taps,
//     y[2q]     = W[:, :, 1] x[q] + W[:, :, 3] x[q - 1]          y[2q + 1] = W[:, :, 2] x[q] + W[:, :, 0] x[q + 1]
// i.e. in the packed polyphase weights (packing.convtranspose_as_phase_conv: rows co * 2 + r, tap slots delta = -1, 0, +1)
// phase 0 uses slots 0, 1 and phase 1 slots 1, 2.  One persistent 8-wave block per CU walks tiles of TQ input positions:
// the input planes of the NEXT tile arrive by LDS-DMA (inline asm, counted waits) while this one is multiplied; the A
// fragments of every (row tile, phase, tap, K step) stay in registers for the block's whole walk (v_mfma_f32_16x16x32_f16:
// 16 rows = 16 output channels, K = 32 input channels of one tap); a lane's four channels of an output position are 8 bytes
// of its plane unit, stored straight from the registers (no LDS transpose: the two phases of a column are adjacent units).
// Arithmetic: split-f16 (lo*hi, hi*lo, hi*hi per K step), accumulation order (phase; tap; K step) — not the conv tile's
// (chunk; tap), so the planes agree with it to f32 rounding of the accumulation, not bit for bit.
#include <algorithm>

#include "conv_common.h"

namespace sat {

struct Ups2Args {
  const void* x16;        // input planes [B][CIN/16][4][T][16 B] (of lrelu(x, slope): whatever the producer wrote)
  void* y16;              // output planes [B][COUT/16][4][2T][16 B] of lrelu(y, y_slope)
  const void* w;          // packed polyphase weights (SAT_CONV_F16X3 packing, up = 2, 3 tap slots, co_pad rows)
  const float* bias;      // [COUT]
  int T, B, co_pad;
  float descale, y_slope;
  int tiles_t, total, per_xcd, nslots;
};

template <int CIN>
__global__ void __launch_bounds__(512, 1) ups2_kernel(const Ups2Args p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int COUT = CIN / 2, NCH = CIN / 16, MT = COUT / 16, KST = CIN / 32;
  constexpr int TQ = CIN == 32 ? 496 : 240;        // input positions per tile; the image holds q0 - 1 .. q0 + TQ
  constexpr int PITCH = CIN == 32 ? 512 : 256;     // units per plane row of an image (a multiple of 64: whole DMA pieces)
  constexpr int NSUB = TQ / 16;                    // 16-position subtiles per tile
  constexpr int ROWS = NCH * 4;
  constexpr int IMG = ROWS * PITCH;                // 64 KB
  constexpr int PIECES = ROWS * PITCH / 64, PPW = PIECES / 8;
  const int tid = threadIdx.x, lane = tid & 63, j16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned OOB = 0x80000000u;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_end = min((xcd + 1) * p.per_xcd, p.total);
  int tile = xcd * p.per_xcd + slot;
  if (tile >= tile_end) return;

  auto stage = [&](int tl, uint4* img) __attribute__((always_inline)) {
    const int ub = __builtin_amdgcn_readfirstlane(tl / p.tiles_t);
    const int q0 = (tl - ub * p.tiles_t) * TQ;
    const i32x4 xrs = dma_rsrc((const char*)p.x16 + (long long)ub * CIN * p.T * 4, (unsigned)(CIN * p.T * 4));
#pragma unroll
    for (int r = 0; r < PPW; ++r) {
      const int piece = wave + 8 * r, row = piece / (PITCH / 64), c = (piece % (PITCH / 64)) * 64 + lane;
      const int q = q0 - 1 + c;
      const unsigned voff = (q >= 0 && q < p.T && c < TQ + 2) ? (unsigned)((row * p.T + q) * 16) : OOB;
      lds_dma16(img + row * PITCH + (piece % (PITCH / 64)) * 64, xrs, voff, 0u);
    }
  };
  stage(tile, lds4);

  // A fragments: [row tile][phase][tap of the phase][K step] x (hi, lo); unit of (chunk, slot, hi|lo, half, row) in the packing
  h8 ah[MT][2][2][KST], al[MT][2][2][KST];
  {
    const uint4* wu = (const uint4*)p.w;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ks = 0; ks < KST; ++ks) {
            const int chunk = 2 * ks + (g >> 1), half = g & 1, slot_ = r + t;          // phase 0: slots 0, 1; phase 1: slots 1, 2
            const size_t u = (size_t)((chunk * 3 + slot_) * 4 + half) * p.co_pad + (16 * m + j16) * 2 + r;
            ah[m][r][t][ks] = __builtin_bit_cast(h8, wu[u]);
            al[m][r][t][ks] = __builtin_bit_cast(h8, wu[u + 2 * (size_t)p.co_pad]);
          }
  }
  float bias[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int k = 0; k < 4; ++k) bias[m][k] = p.bias[16 * m + 4 * g + k];

  int buf = 0;
  bool first = true;
  constexpr int SPS = MT * 2 * 2;                          // plane stores a wave issues per subtile
  const int my_stores = ((NSUB - wave + 7) / 8) * SPS;     // ... per tile (16, 12 or 8)
  for (;;) {
    const int b = __builtin_amdgcn_readfirstlane(tile / p.tiles_t);
    const int q0 = (tile - b * p.tiles_t) * TQ;
    const int next = tile + p.nslots;
    const bool more = next < tile_end;
    uint4* img = lds4 + buf * IMG;
    // this tile's image has landed everywhere (its pieces were requested BEFORE the last tile's stores, which may still be
    // in flight: a counted wait); the other image is free (its readers passed the barrier of the last round)
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (my_stores == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (my_stores == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    first = false;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (more) stage(next, lds4 + (buf ^ 1) * IMG);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)p.y16 + (long long)b * COUT * (2 * p.T) * 4), 0, (unsigned)(COUT * (2 * p.T) * 4), 0x00020000);
    const int To = 2 * p.T;
    for (int s = wave; s < NSUB; s += 8) {
      // B fragments of the subtile: column c = 1 + 16 s + j16 + delta (the image starts at q0 - 1), K group g -> (chunk, half)
      const uint4* xb = img + ((g >> 1) * 4 + (g & 1)) * PITCH + 16 * s + j16;
      h8 bh[3][KST], bl[3][KST];
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
          bh[d][ks] = __builtin_bit_cast(h8, xb[(2 * ks) * 4 * PITCH + d]);
          bl[d][ks] = __builtin_bit_cast(h8, xb[(2 * ks) * 4 * PITCH + 2 * PITCH + d]);
        }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
              acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m][r][t][ks], bh[r + t][ks], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][r][t][ks], bl[r + t][ks], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][r][t][ks], bh[r + t][ks], acc, 0, 0, 0);
            }
          const int tpos = 2 * (q0 + 16 * s + j16) + r;
          float u[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v = __builtin_fmaf(acc[k], p.descale, bias[m][k]);
            u[k] = __builtin_fmaxf(v, v * p.y_slope);
          }
          const auto h01 = __builtin_amdgcn_cvt_pkrtz(u[0], u[1]);
          const auto h23 = __builtin_amdgcn_cvt_pkrtz(u[2], u[3]);
          const auto l01 = split_lo2(h01, u[0], u[1]);
          const auto l23 = split_lo2(h23, u[2], u[3]);
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          // channels 16 m + 4 g ..: chunk m, half g >> 1, bytes 8 (g & 1) of the unit at time tpos
          const unsigned off = (q0 + 16 * s + j16 < p.T) ? (unsigned)((((m * 4 + (g >> 1)) * To + tpos) * 16) + 8 * (g & 1)) : OOB;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)}, yrs, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)}, yrs, off, 2 * To * 16, 0);
        }
    }
    if (!more) break;
    tile = next;
    buf ^= 1;
  }
}

template <int CIN>
static int launch_ups2(const Ups2Args& a, hipStream_t s) {
  Ups2Args p = a;
  constexpr int TQ = CIN == 32 ? 496 : 240;
  const size_t lds_bytes = (size_t)2 * 64 * 1024;
  auto kern = ups2_kernel<CIN>;
  static std::atomic<uint64_t> attr_done{0};
  int dev;
  if (attr_needed_on_current_device(attr_done, &dev)) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done_on_device(attr_done, dev);
  }
  p.tiles_t = ceil_div(p.T, TQ);
  p.total = p.tiles_t * p.B;
  p.per_xcd = ceil_div(p.total, 8);
  p.nslots = std::max(1, std::min(32, p.per_xcd));
  hipLaunchKernelGGL(kern, dim3(8 * p.nslots), dim3(512), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("ups2_kernel");
  return SAT_OK;
}

}  // namespace sat

using namespace sat;

extern "C" int sat_upsample2_supported(int C_in, int ksize, int stride, int padding) {
  return (C_in == 32 || C_in == 64) && ksize == 4 && stride == 2 && padding == 1;
}

extern "C" int sat_upsample2_f16x3(const void* x_split, const void* w_packed, const float* bias, float w_descale, void* y_split,
                                   float y_split_slope, int B, int C_in, int T, void* stream) {
  SAT_REQUIRE(x_split && w_packed && bias && y_split, "upsample2: null pointer");
  SAT_REQUIRE(B > 0 && T > 0 && sat_upsample2_supported(C_in, 4, 2, 1), "upsample2: C_in 32 or 64 only (k 4, stride 2, padding 1)");
  SAT_REQUIRE(y_split_slope > 0.f && y_split_slope <= 1.f, "upsample2: y_split_slope in (0, 1]");
  SAT_REQUIRE((long long)C_in * T * 4 < (1LL << 31), "upsample2: slab too large for 31-bit offsets");
  Ups2Args a{};
  a.x16 = x_split;
  a.y16 = y_split;
  a.w = w_packed;
  a.bias = bias;
  a.T = T;
  a.B = B;
  a.co_pad = 64;                     // rows C_out * 2 <= 64 (sat_conv1d_packed_dims)
  a.descale = w_descale != 0.f ? w_descale : 1.f;
  a.y_slope = y_split_slope;
  return C_in == 32 ? launch_ups2<32>(a, (hipStream_t)stream) : launch_ups2<64>(a, (hipStream_t)stream);
}
