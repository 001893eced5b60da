// Fused ResBlock1 step for C = 64, 3 taps, split planes end to end (the generator's third stage: 20 000 frames per
// utterance, where the two-launch form is HBM-bound).
#include "conv_common.h"

namespace sat {

// ------------------------------------------------------------------------------------------------
// x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 in one launch: the intermediate activation never leaves the CU.  Two
// launches move 20 bytes per element (x in, t1 out; t1 in, residual in, planes out); this one moves 8 — and at 64
// channels x 20 000 frames x 32 utterances the 3-tap launches ran at 4.3 TB/s, i.e. on the HBM roof.
//   Block = 126 output positions x all 64 channels, four waves.  conv1 (dilation d) computes t1 on the 128 columns
//   [t0 - 1, t0 + 127): wave w owns columns 32 w .. 32 w + 31 for both 32-row tiles, K = 4 chunks x 3 taps; t1 =
//   lrelu(. + b1) is written to LDS as split planes (zero outside the utterance: conv2's own zero padding) OVER the
//   input chunk, which is dead by then; conv2 (dilation 1) reads column o + tap of it for output column o, its last two
//   columns of the 128 are not stored (q_end).  Weights stream through LDS per 16-channel chunk (12 KB), the next
//   chunk's loads in flight in registers during a chunk's MFMAs (conv2's first chunk under conv1's last).
//   LDS 45 KB, < 128 VGPRs: three blocks per CU.
// Order of operations per accumulator (chunk, tap, lo*hi, hi*lo, hi*hi), the t1 split (leaky-relu, round-toward-zero
// pack) and the epilogue are those of the two-launch path on the conv tile: the same bits.
// ------------------------------------------------------------------------------------------------
constexpr int P64_TO = 126;     // output positions per block
constexpr int P64_W1 = 128;     // t1 window: [t0 - 1, t0 + 127)

__global__ void __launch_bounds__(256, 3) resblock_pair64_k3_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  constexpr int KS = 3, NCH = 4, W1 = P64_W1, XWI = 3, XWP = 64 * XWI, W_UNITS = KS * 4 * 64, W_IT = W_UNITS / 256;
  uint4* ldsw = lds4;                     // [KS][4][64]     weight chunk (conv1, then conv2)
  uint4* ldst = lds4 + W_UNITS;           // [NCH][4][W1]    t1, all channels
  uint4* ldsx = ldst;                     // [4][XWP]        input chunk, over the head of the t1 image

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * P64_TO;
  const int xi0 = t0 - 1 - p.pad_left;    // input position of staged column 0 (pad_left = conv1's halo)

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x16 + (long long)b * 64 * p.T_in * 4), 0, (unsigned)(64 * p.T_in * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (unsigned)p.w_gs, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (unsigned)p.w_gs, 0x00020000);
  const int seg_bytes = p.co_pad * 16;

  uint4 xst[XWI], wst[W_IT];
  auto issue_x = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < XWI; ++it) {
      const int xi = xi0 + lane + 64 * it;
      const unsigned voff = (xi >= 0 && xi < p.T_in) ? (unsigned)((wave * p.T_in + xi) * 16) : 0x80000000u;
      xst[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, chunk * 4 * p.T_in * 16, 0));
    }
  };
  auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rs, int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int u = tid + 256 * i;         // unit of the chunk image [tap][part * 2 + half][64 rows]
      wst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (u & 63) * 16 + (u >> 6) * seg_bytes,
                                                                                chunk * (KS * 4) * seg_bytes, 0));
    }
  };
  auto publish_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < XWI; ++it) ldsx[wave * XWP + lane + 64 * it] = xst[it];
  };
  auto publish_w = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) ldsw[tid + 256 * i] = wst[i];
  };
  f32x16 acc[2][1];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;
  };
  // one chunk of MFMAs: B fragments at base + tap * step in an image of plane pitch `pitch`
  auto mfma_chunk = [&](const uint4* base, int pitch, int step) __attribute__((always_inline)) {
    const uint4* wb = ldsw + lh * 64 + l31;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const uint4* xt = base + t * step;
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

  // ---- conv1: four chunks of the input through LDS ----
  zero_acc();
  issue_x(0);
  issue_w(w1rs, 0);
  const uint4* xbase = ldsx + lh * XWP + wave * 32 + l31;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    __syncthreads();                        // every wave is done reading the previous chunk's tiles
    publish_x();
    publish_w();
    __syncthreads();
    if (ch + 1 < NCH) {
      issue_x(ch + 1);
      issue_w(w1rs, ch + 1);
    } else {
      issue_w(w2rs, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(xbase, XWP, p.dil);
  }
  __syncthreads();                          // every wave is done reading the input chunk: t1 goes over it
  // ---- t1 = lrelu(conv1 + b1) -> LDS as conv2's B operand (hi | lo planes of the four chunks), zero outside the utterance ----
  {
    const int col = wave * 32 + l31;
    const int pos = t0 - 1 + col;
    const bool inside = pos >= 0 && pos < p.T_in;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = acc[m][0][4 * rg + k] + p.bias1[32 * m + 8 * rg + 4 * lh + k];
          t = t > 0.f ? t : t * p.in_slope;
          v[k] = inside ? t : 0.f;
        }
        const auto h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
        const auto h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
        const auto l01 = __builtin_amdgcn_cvt_pkrtz(v[0] - (float)h01[0], v[1] - (float)h01[1]);
        const auto l23 = __builtin_amdgcn_cvt_pkrtz(v[2] - (float)h23[0], v[3] - (float)h23[1]);
        // rows 32 m + 8 rg + 4 lh + k: chunk 2 m + rg / 2, half rg & 1, bytes 8 lh .. of the unit
        const int chunk = 2 * m + (rg >> 1);
        ((uint2*)(ldst + ((chunk * 4 + 0 + (rg & 1)) * W1 + col)))[lh] =
            make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
        ((uint2*)(ldst + ((chunk * 4 + 2 + (rg & 1)) * W1 + col)))[lh] =
            make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
      }
  }
  // ---- conv2: t1 from LDS, four chunks of weights ----
  zero_acc();
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    __syncthreads();                        // t1 complete (first pass) / the previous weight chunk is no longer read
    publish_w();
    __syncthreads();
    if (ch + 1 < NCH) issue_w(w2rs, ch + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(ldst + (ch * 4 + lh) * W1 + wave * 32 + l31, W1, 1);
  }
  // residual = the block input (from its planes), bias, MRF accumulate, f32 and / or planes out; columns >= t0 + 126 dropped
  conv_epilogue<2, 1, false, false>(p, acc, b, 0, 0, t0 + wave * 32, l31, lh, 32, t0 + P64_TO);
}

bool pair64_supports(const ConvArgs& a) {
  return a.cin_g == 64 && a.rows_g == 64 && a.ksize == 3 && a.x16 && a.res16 && a.fast_epi && !a.ch_scale && a.co_pad == 64 &&
         P64_W1 + (a.ksize - 1) * a.dil <= 192;
}

int launch_pair64_k3(const ConvArgs& a, int B, hipStream_t s) {
  // (+ 2 units: conv2's last two, unstored, columns read t1 columns 128 and 129)
  const size_t lds_bytes = ((size_t)3 * 4 * 64 + (size_t)4 * 4 * P64_W1 + 2) * 16;
  dim3 grid(ceil_div(a.T_q, P64_TO), 1, B);
  hipLaunchKernelGGL(resblock_pair64_k3_kernel, grid, dim3(256), lds_bytes, s, a);
  SAT_LAUNCH_CHECK("resblock_pair64_k3_kernel");
  return SAT_OK;
}

}  // namespace sat
