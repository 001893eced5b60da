// wav2vec2 support kernels (everything matrix-shaped runs on the fused conv1d kernel):
// first conv layer (1 -> C, k, stride), LayerNorm over channels (+GELU, + even/odd phase split for a
// following stride-2 conv), softmax over keys of the transposed score matrix, per-head transpose of V.
// Reference: torchaudio.models.wav2vec2 (third-party; semantics restated in oracle/wav2vec2.py) as
// configured by egs/asr/librispeech/local/chain/tuning/tdnnf_wav2vec2_vq.py:39-56.
#include "common.h"

namespace sat {

__device__ __forceinline__ float gelu_erf(float v) { return gelu_fast(v); }

// y[b][c][t] = bias[c] + sum_j w[c][j] * x[b][t*stride + j]      (C <= 512, k <= 16)
__global__ void __launch_bounds__(256) w2v2_conv0_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int n,
                                                        int C, int k, int stride, int T) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // w [C][k], bias [C]
  float* wl = lds;
  float* bl = lds + C * k;
  for (int i = threadIdx.x; i < C * k; i += 256) wl[i] = w[i];
  for (int i = threadIdx.x; i < C; i += 256) bl[i] = bias[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float xv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) xv[j] = (j < k) ? x[(size_t)b * n + (size_t)t * stride + j] : 0.f;
  float* yb = y + (size_t)b * C * T + t;
  for (int c = 0; c < C; ++c) {
    float acc = bl[c];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < k) acc = fmaf(wl[c * k + j], xv[j], acc);
    yb[(size_t)c * T] = acc;
  }
}

// LayerNorm over the channel axis of x [B][C][T] (eps 1e-5, biased variance, affine), optional GELU.
// split = 0: y [B][C][T] (row pitch y_pitch);  split = 1: y [B][2C][ceil(T/2)], y[(ph*C + c)][u] = ln(x)[c][2u + ph].
// A block owns 64 consecutive frames of one utterance; its 16 waves each reduce a 1/16 slice of the
// channels (frames of a channel are contiguous, so every access is a coalesced 256-B row piece) and the
// partial sums meet in LDS.  Two-pass mean / variance like torch's CPU kernel.
constexpr int LN_FR = 32;          // frames per block: a wave's load covers two whole 128-byte lines
// Block = 32 frames x SLICES channel slices: a thread reads its (up to) MAXPT consecutive channels sl*MAXPT .. of one
// frame ONCE into registers — all loads in flight at once — and both statistics passes and the output pass run
// from there; the frame's totals are added in slice order.  32 channels per thread keep it at ~90 VGPRs (5 waves
// per SIMD); <16, 32> serves C <= 512 (feature extractor) and <32, 32> C <= 1024 (transformer: 249 frames are 256
// blocks in all, so each one brings 16 waves to hide its own load latency).  Outputs: f32 `y` and / or split planes
// `y16` (satools_hip.h: hi | lo f16 of the value, 16-byte units of 8 channels) — a thread's channels are whole units.
// CONV0: the input is not read but computed — the wav2vec2 feature extractor's first conv (1 -> C channels, ck <= 12
// taps, stride cstride) of the waveform x [B][cn], weights in LDS as rows of 12 floats: y = LN(conv0(wav)) without the
// [B][512][16k] f32 tensor ever reaching HBM.  Same accumulation order as w2v2_conv0_kernel, so the same bits.
// (Tried: 64 frames per block, a wave = one channel slice, the weights as SGPR operands from scalar loads instead of 96 broadcast
// ds_read_b128 per thread — 517 -> 745 us; channel pairs in packed f32 FMAs — hipcc spilled 3 000 registers.)
template <int SLICES, int MAXPT, bool CONV0 = false>
__global__ void __launch_bounds__(LN_FR * SLICES) layernorm_ch_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
    uint4* __restrict__ y16, int C, int T, long long x_bs, long long x_cs, long long y_bs, long long y_cs, int gelu, int split,
    const float* __restrict__ cw = nullptr, const float* __restrict__ cbias = nullptr, int cn = 0, int ck = 0, int cstride = 0) {
  __shared__ float part[SLICES][LN_FR];
  __shared__ float s_mean[LN_FR], s_rstd[LN_FR];
  extern __shared__ __attribute__((aligned(16))) float ln_wl[];     // CONV0: [C][12]
  const int b = blockIdx.y;
  const int tx = threadIdx.x & (LN_FR - 1), sl = threadIdx.x / LN_FR;
  const int t = blockIdx.x * LN_FR + tx;
  const bool ok = t < T;
  const int cb = sl * MAXPT;
  float v[MAXPT];
  float s = 0.f;
  if constexpr (CONV0) {
    for (int i = threadIdx.x; i < C * 12; i += LN_FR * SLICES) {
      const int c = i / 12, j = i - 12 * c;
      ln_wl[i] = j < ck ? cw[c * ck + j] : 0.f;
    }
    __syncthreads();
    float xv[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) xv[j] = (ok && j < ck) ? x[(size_t)b * cn + (size_t)t * cstride + j] : 0.f;
#pragma unroll
    for (int k = 0; k < MAXPT; ++k) {
      v[k] = 0.f;
      if (ok && cb + k < C) {
        const float4 w0 = *(const float4*)&ln_wl[(cb + k) * 12], w1 = *(const float4*)&ln_wl[(cb + k) * 12 + 4],
                     w2 = *(const float4*)&ln_wl[(cb + k) * 12 + 8];
        const float wr[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
        float acc = cbias[cb + k];
#pragma unroll
        for (int j = 0; j < 12; ++j)
          if (j < ck) acc = fmaf(wr[j], xv[j], acc);
        v[k] = acc;
      }
    }
  } else {
    const float* xb = x + (size_t)b * x_bs + (ok ? t : 0);
#pragma unroll
    for (int k = 0; k < MAXPT; ++k) v[k] = (ok && cb + k < C) ? xb[(size_t)(cb + k) * x_cs] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < MAXPT; ++k)
    if (cb + k < C) s += v[k];
  part[sl][tx] = s;
  __syncthreads();
  if (sl == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < SLICES; ++i) tot += part[i][tx];
    s_mean[tx] = tot / (float)C;
  }
  __syncthreads();
  const float mean = s_mean[tx];
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXPT; ++k)
    if (cb + k < C) {
      const float d = v[k] - mean;
      q = fmaf(d, d, q);
    }
  __syncthreads();
  part[sl][tx] = q;
  __syncthreads();
  if (sl == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < SLICES; ++i) tot += part[i][tx];
    s_rstd[tx] = 1.0f / sqrtf(tot / (float)C + 1e-5f);
  }
  __syncthreads();
  if (!ok) return;
  const float rstd = s_rstd[tx];
  const int ph = split ? (t & 1) : 0;
  const int u = split ? (t >> 1) : t;
  const bool tail = split && (T & 1) && t == T - 1;   // odd length: the odd phase's last slot is zero padding
  // output in groups of 8 channels (= one 16-byte unit of the planes): gamma / beta as two float4 loads each
  float* yf = y ? y + (size_t)b * y_bs : nullptr;
  const int Co = split ? 2 * C : C, Tp = split ? (T + 1) / 2 : T;
  uint4* yp = y16 ? y16 + (size_t)b * (Co / 4) * Tp : nullptr;             // Co * Tp * 4 bytes per utterance
#pragma unroll
  for (int g8 = 0; g8 < MAXPT / 8; ++g8) {
    const int c0 = cb + 8 * g8;
    if (c0 >= C) break;
    float gm[8], bt[8], o[8];
    if (c0 + 8 <= C) {
      *(float4*)&gm[0] = *(const float4*)&gamma[c0];
      *(float4*)&gm[4] = *(const float4*)&gamma[c0 + 4];
      *(float4*)&bt[0] = *(const float4*)&beta[c0];
      *(float4*)&bt[4] = *(const float4*)&beta[c0 + 4];
    } else {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        gm[jj] = c0 + jj < C ? gamma[c0 + jj] : 0.f;
        bt[jj] = c0 + jj < C ? beta[c0 + jj] : 0.f;
      }
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) o[jj] = (v[8 * g8 + jj] - mean) * rstd * gm[jj] + bt[jj];
    if (gelu) {                                  // packed f32 arithmetic, two values per instruction (common.h)
      float o0[4] = {o[0], o[1], o[2], o[3]}, o1[4] = {o[4], o[5], o[6], o[7]};
      gelu_fast4(o0);
      gelu_fast4(o1);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) o[jj] = o0[jj], o[4 + jj] = o1[jj];
    }
    if (yf) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        if (c0 + jj < C) {
          yf[(size_t)(ph * C + c0 + jj) * y_cs + u] = o[jj];
          if (tail) yf[(size_t)(C + c0 + jj) * y_cs + u] = 0.f;
        }
    }
    if (yp) {
      unsigned hi[4], lo[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const auto h = __builtin_amdgcn_cvt_pkrtz(o[2 * jj], o[2 * jj + 1]);
        const auto l = __builtin_amdgcn_cvt_pkrtz(o[2 * jj] - (float)h[0], o[2 * jj + 1] - (float)h[1]);
        hi[jj] = __builtin_bit_cast(unsigned, h);
        lo[jj] = __builtin_bit_cast(unsigned, l);
      }
      const int co = ph * C + c0;
      const size_t un = (size_t)((co >> 4) * 4 + ((co >> 3) & 1)) * Tp + u;
      yp[un] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      yp[un + 2 * (size_t)Tp] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      if (tail) {
        const int c1 = C + c0;
        const size_t u1 = (size_t)((c1 >> 4) * 4 + ((c1 >> 3) & 1)) * Tp + u;
        yp[u1] = make_uint4(0, 0, 0, 0);
        yp[u1 + 2 * (size_t)Tp] = make_uint4(0, 0, 0, 0);
      }
    }
  }
}

// in place: st [G*T rows (key j)][pitch] -> softmax over j of scale*s, for every (group, query column).
// Block = 64 query columns x 16 row slices; a column's max and sum meet in LDS.
__global__ void __launch_bounds__(1024) softmax_cols_kernel(float* __restrict__ st, int T, int pitch, float scale) {
  __shared__ float part[16][64];
  __shared__ float s_a[64];
  const int g = blockIdx.y;
  const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int qcol = blockIdx.x * 64 + tx;
  const bool ok = qcol < T;
  float* base = st + (size_t)g * T * pitch + (ok ? qcol : 0);
  float mx = -INFINITY;
  if (ok)
    for (int j = sl; j < T; j += 16) mx = fmaxf(mx, base[(size_t)j * pitch] * scale);
  part[sl][tx] = mx;
  __syncthreads();
  if (sl == 0) {
    float m = part[0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, part[i][tx]);
    s_a[tx] = m;
  }
  __syncthreads();
  mx = s_a[tx];
  float sum = 0.f;
  if (ok)
    for (int j = sl; j < T; j += 16) {
      const float e = expf(base[(size_t)j * pitch] * scale - mx);
      base[(size_t)j * pitch] = e;
      sum += e;
    }
  __syncthreads();
  part[sl][tx] = sum;
  __syncthreads();
  if (sl == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += part[i][tx];
    s_a[tx] = 1.0f / tot;
  }
  __syncthreads();
  if (!ok) return;
  const float inv = s_a[tx];
  for (int j = sl; j < T; j += 16) base[(size_t)j * pitch] *= inv;
}

// ------------------------------------------------------------------------------------------------
// Fused self-attention of one (utterance, head) on the f16 matrix cores in split-f16 arithmetic (hi + lo f16
// operands, lo*hi + hi*lo + hi*hi, f32 accumulate): S^T = K^T Q, softmax over keys, O = V P — the score matrix
// never leaves registers.  hd = 64; keys go in blocks of 256 (one block at T = 249) with a running softmax over
// the blocks for longer utterances.
//   Q, K arrive as split planes (satools_hip.h) straight from their projections' epilogues: a 16-byte unit holds 8
//   channels of one frame, which is an MFMA operand fragment both as A (K: row = key) and as B (Q: column = query).
//   A wave owns 32 queries and ALL keys: S^T[key][query] is 8 accumulator tiles (128 VGPRs), a lane's column is one
//   query, so the softmax is an in-lane reduction plus one exchange between the two lane halves.
//   P feeds the second product from the same registers: MFMA k-slot e of lane half lh is given the key
//   32m + 16hh + 8(e>>2) + 4lh + (e&3) — the key the accumulator register 8hh + e of tile m holds — and V's A
//   fragments (split to f16 once, when the head's V is staged) are laid out in that order, so no data moves.
//   Block = NW waves = 32 NW queries; the head's K planes (64 KB), then its V (split to f16 in fragment order, 66 KB), are
//   staged in LDS.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AT_VU = 33;       // LDS row pitch of the split V image in 16-byte units: rows 4 banks apart -> conflict-free 16-byte reads

__device__ __forceinline__ void split8(const float (&a)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(a[2 * j], a[2 * j + 1]);
    const auto l = __builtin_amdgcn_cvt_pkrtz(a[2 * j] - (float)h[0], a[2 * j + 1] - (float)h[1]);
    hi[2 * j] = h[0]; hi[2 * j + 1] = h[1];
    lo[2 * j] = l[0]; lo[2 * j + 1] = l[1];
  }
}

// NW waves per block = 32 NW queries: 8 (one block per head at 249 frames: K and V are fetched and staged ONCE per head) or 4
// (utterances of up to 128 frames).
//
// Round 3: the kernel was a chain of exposed loads — by its own cycle stamps (tools/bench_attention.py stamps) a head took
// 46.5 k cycles of which 15.4 k waited for K and Q and 9.3 k for V to arrive through registers, one 8-wave block per CU (256
// VGPRs).  Now a block walks HPB heads (and, for utterances longer than 256 frames, their key blocks) as STAGES over two LDS
// regions A / B that swap roles, every operand arriving by LDS-DMA (inline asm, counted waits) one stage ahead:
//   stage s:  K_s in R = A or B;  V_s (f32 rows) in the other region R';  split V image over K_s in R
//     Q fragments <- global | wait K_s | barrier | DMA V_s -> R'          (R' is free: the image of stage s - 1 is done)
//     S = K^T Q, softmax                                                   (V_s lands underneath)
//     wait V_s | barrier | V_s f32 (R') -> hi | lo f16 fragment image (R) | barrier | DMA K_{s+1} -> R'
//     O += V P                                                             (K_{s+1} lands underneath)
//     barrier (the image is no longer read)
// Only the first K of a block is waited for in the open.
// diagnostics (sat_attention_debug_stamps): block (0, 0, 0) records its waves' cycle counters at 7 points of its first stage
#define AT_STAMP(i)                                                                                          \
  do {                                                                                                       \
    if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      unsigned long long t_;                                                                                 \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      if (lane == 0) dbg[(i) * 8 + wave] = (long long)t_;                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
  } while (0)

template <int NW>
__global__ void __launch_bounds__(64 * NW, 1) attention_f16x3_kernel(const uint4* __restrict__ qs, const uint4* __restrict__ ks,
                                                                const float* __restrict__ v, float* __restrict__ o,
                                                                uint4* __restrict__ os, int C, int T, int v_pitch, float scale,
                                                                int hpb, long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) uint4 at_lds[];
  constexpr int REG_UNITS = 2 * 64 * AT_VU;      // one region: the split V image (66 KB) >= K planes (16 x 256 units) = V f32 rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z;
  constexpr int NTHR = 64 * NW;
  const int q = blockIdx.x * (32 * NW) + wave * 32 + l31;     // this lane's query column
  const bool qok = q < T;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  const size_t ub = (size_t)b * (C / 4) * T;                 // units per utterance: C * T * 4 bytes
  const float scale2 = scale * 1.44269504088896340736f;      // softmax through the hardware exp2
  const int nkb = (T + 255) >> 8;                            // key blocks of 256 per head
  const int nst = hpb * nkb;                                 // stages of this block
  const int h0 = blockIdx.y * hpb;
  const i32x4 krs = dma_rsrc((const char*)ks + ub * 16, (unsigned)((size_t)C * T * 4));

  // LDS-DMA of a stage's K planes: 16 plane rows x 256 keys = 64 pieces of 64 units, 64 / NW per wave; keys >= T arrive as zeros
  auto dma_k = [&](uint4* reg, int hh, int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 64 / NW; ++r) {
      const int piece = wave + NW * r, row = piece >> 2, col = (piece & 3) * 64 + lane;
      const unsigned voff = k0 + col < T ? (unsigned)((((4 * hh + (row >> 2)) * 4 + (row & 3)) * T + k0 + col) * 16) : 0x80000000u;
      lds_dma16(reg + row * 256 + (piece & 3) * 64, krs, voff, 0u);
    }
  };
  // ... and of its V rows (f32 [64 head dims][256 keys]): a piece = one head dim; columns >= v_pitch arrive as zeros, columns in
  // [T, v_pitch) carry whatever the caller's buffer holds and are masked when the image is built
  auto dma_v = [&](uint4* reg, int hh, int k0) __attribute__((always_inline)) {
    const i32x4 vrs = dma_rsrc(v + ((size_t)b * C + (size_t)hh * 64) * v_pitch, (unsigned)(64 * v_pitch * 4));
#pragma unroll
    for (int r = 0; r < 64 / NW; ++r) {
      const int d = wave + NW * r, jg = k0 + lane * 4;
      lds_dma16(reg + d * 64, vrs, jg < v_pitch ? (unsigned)((d * v_pitch + jg) * 4) : 0x80000000u, 0u);
    }
  };

  f32x16 oa[2];
  float m_run = -INFINITY, l_run = 0.f;          // running maximum and sum of this lane's query column
  AT_STAMP(0);
  dma_k(at_lds, h0, 0);
  for (int s = 0; s < nst; ++s) {
    const int hh = h0 + s / nkb, k0 = (s % nkb) * 256;
    uint4* R = at_lds + (s & 1) * REG_UNITS;
    uint4* Rp = at_lds + ((s & 1) ^ 1) * REG_UNITS;
    if (k0 == 0) {
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[m2][r] = 0.f;
      m_run = -INFINITY;
      l_run = 0.f;
    }
    // Q fragments of the wave's 32 queries (re-read per stage from L2: holding them across the V product would spill)
    h8 qh[4], ql[4];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
      const size_t u = ub + (size_t)((4 * hh + cl) * 4 + lh) * T + q;
      qh[cl] = __builtin_bit_cast(h8, qok ? qs[u] : zero4);
      ql[cl] = __builtin_bit_cast(h8, qok ? qs[u + 2 * (size_t)T] : zero4);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // K_s has landed everywhere (and Q; hipcc does not count the DMAs)
    dma_v(Rp, hh, k0);
    if (s == 0) AT_STAMP(1);

    // ---- S^T = K^T Q: 8 key tiles x (4 chunks of 16 channels) ----
    f32x16 st[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[m][r] = 0.f;
    {
      // K fragments of tile i + 1 are read ahead of tile i's MFMAs (second register pair, fenced: left alone the
      // scheduler sinks each read to its first use and every three MFMAs start on an LDS round trip)
      h8 ka[2][2];
      auto ldk = [&](int buf, int i) __attribute__((always_inline)) {
        const int cl = i >> 3, m = i & 7;
        ka[buf][0] = __builtin_bit_cast(h8, R[(cl * 4 + 0 + lh) * 256 + 32 * m + l31]);
        ka[buf][1] = __builtin_bit_cast(h8, R[(cl * 4 + 2 + lh) * 256 + 32 * m + l31]);
      };
      ldk(0, 0);
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int cl = i >> 3, m = i & 7;
        if (i + 1 < 32) ldk((i + 1) & 1, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        st[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[i & 1][1], qh[cl], st[m], 0, 0, 0);
        st[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[i & 1][0], ql[cl], st[m], 0, 0, 0);
        st[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[i & 1][0], qh[cl], st[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (s == 0) AT_STAMP(2);
    // ---- softmax over keys: register r of tile m is key k0 + 32m + 8(r>>2) + 4lh + (r&3) ----
    // exp(scale s - max) = exp2(fma(s, scale log2 e, -max')) with the maximum taken on the raw scores (scale > 0): one max,
    // one fma, one v_exp_f32 and one add per score; only a tile that reaches past T masks its keys
    float mr = -INFINITY;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (k0 + 32 * m + 32 > T) {           // (wave-uniform)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jk = k0 + 32 * m + 8 * (r >> 2) + 4 * lh + (r & 3);
          st[m][r] = jk < T ? st[m][r] : -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) mr = fmaxf(mr, st[m][r]);
    }
    mr = fmaxf(mr, __shfl_xor(mr, 32));
    const float mx = fmaxf(m_run, mr * scale2);               // running maximum in units of log2
    const float alpha = __builtin_amdgcn_exp2f(m_run - mx);   // 0 on the first block (m_run = -inf); v_exp_f32, 1 ulp
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[m][r], scale2, -mx));     // masked keys: exp2(-inf) = 0
        st[m][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32);
    l_run = l_run * alpha + sum;
    m_run = mx;
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int r = 0; r < 16; ++r) oa[m2][r] *= alpha;
    if (s == 0) AT_STAMP(3);

    // ---- V_s: f32 rows in R' -> split to hi | lo f16 ONCE, laid out as the A fragments of the second product over K_s in R:
    // unit [part][d][16 k-steps x 2 lane halves] (row pitch AT_VU units: 4 banks apart, conflict-free 16-byte reads down a
    // column of d) holding the keys 16 ks + 4 lh + {0..3} and 16 ks + 8 + 4 lh + {0..3} — the k-slot order of the P
    // registers.  Keys >= T zero.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // V_s has landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... everywhere, and every wave is done with K_s
    if (s == 0) AT_STAMP(4);
#pragma unroll
    for (int i8 = 0; i8 < 4096 / NTHR; ++i8) {
      const int u = tid + NTHR * i8;             // float4 index: row d = u / 64, keys k0 + 4 (u % 64) ..
      const int d = u >> 6, j4 = (u & 63) * 4, jg = k0 + j4;
      float4 w = __builtin_bit_cast(float4, Rp[u]);
      if (jg + 0 >= T) w.x = 0.f;
      if (jg + 1 >= T) w.y = 0.f;
      if (jg + 2 >= T) w.z = 0.f;
      if (jg + 3 >= T) w.w = 0.f;
      const auto h01 = __builtin_amdgcn_cvt_pkrtz(w.x, w.y);
      const auto h23 = __builtin_amdgcn_cvt_pkrtz(w.z, w.w);
      const auto l01 = split_lo2(h01, w.x, w.y);
      const auto l23 = split_lo2(h23, w.z, w.w);
      // keys j4 .. j4 + 3 of the block: k-step j4 / 16, lane half (j4 / 4) & 1, first or second 8 bytes of the unit
      uint2* dst = (uint2*)(R + d * AT_VU + (j4 >> 4) * 2 + ((j4 >> 2) & 1)) + ((j4 >> 3) & 1);
      dst[0] = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
      dst[2 * 64 * AT_VU] = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the image is complete, the f32 rows are no longer read
    if (s + 1 < nst) dma_k(Rp, h0 + (s + 1) / nkb, ((s + 1) % nkb) * 256);
    if (s == 0) AT_STAMP(5);

    // ---- O += V P: 2 row tiles (64 head dims) x 16 k-steps of 16 keys ----
    {
      h8 va[2][2][2];                       // [buffer][row tile][hi | lo]: the next k-step's V fragments are read ahead
      auto ldv = [&](int buf, int kst) __attribute__((always_inline)) {
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          const uint4* vu = R + (32 * m2 + l31) * AT_VU + kst * 2 + lh;
          va[buf][m2][0] = __builtin_bit_cast(h8, vu[0]);
          va[buf][m2][1] = __builtin_bit_cast(h8, vu[64 * AT_VU]);
        }
      };
      ldv(0, 0);
#pragma unroll
      for (int kst = 0; kst < 16; ++kst) {  // k-step = keys 16 kst .. 16 kst + 15 = registers 8 (kst & 1) .. of tile kst / 2
        const int m = kst >> 1, hh2 = kst & 1;
        if (kst + 1 < 16) ldv((kst + 1) & 1, kst + 1);
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = st[m][8 * hh2 + e];
        h8 b_hi, b_lo;
        split8(pv, b_hi, b_lo);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          oa[m2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[kst & 1][m2][1], b_hi, oa[m2], 0, 0, 0);
          oa[m2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[kst & 1][m2][0], b_lo, oa[m2], 0, 0, 0);
          oa[m2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[kst & 1][m2][0], b_hi, oa[m2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (s == 0) AT_STAMP(6);
    if (k0 + 256 >= T && qok) {
      // ---- the head is complete: store; register r of tile m2 is head dim 32 m2 + 8 (r>>2) + 4 lh + (r&3) ----
      const float inv = 1.0f / l_run;
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float w[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = oa[m2][4 * g + k] * inv;
          const int c = hh * 64 + 32 * m2 + 8 * g;               // first channel of the 8-channel unit (this lane: + 4 lh)
          if (o) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[((size_t)b * C + c + 4 * lh + k) * T + q] = w[k];
          }
          if (os) {
            const auto h01 = __builtin_amdgcn_cvt_pkrtz(w[0], w[1]);
            const auto h23 = __builtin_amdgcn_cvt_pkrtz(w[2], w[3]);
            const auto l01 = split_lo2(h01, w[0], w[1]);
            const auto l23 = split_lo2(h23, w[2], w[3]);
            const size_t un = ub + (size_t)((c >> 4) * 4 + ((c >> 3) & 1)) * T + q;
            ((uint2*)(os + un))[lh] = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            ((uint2*)(os + un + 2 * (size_t)T))[lh] = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the image of this stage is no longer read: V_{s+1} may land on it
  }
}

// v [B][H*D][pitch] -> vt [B*H][jpad][D] (packed-weight layout [ci = key j][co = d]); rows j >= T are zero
__global__ void __launch_bounds__(256) transpose_heads_kernel(const float* __restrict__ v, float* __restrict__ vt, int D,
                                                             int T, int pitch, int jpad) {
  __shared__ float tile[64][65];
  const int g = blockIdx.y;           // (b, h)
  const int j0 = blockIdx.x * 64;
  const float* src = v + (size_t)g * D * pitch;
  float* dst = vt + (size_t)g * jpad * D;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int d = ty; d < D; d += 4) {
    const int j = j0 + tx;
    tile[d][tx] = (j < T) ? src[(size_t)d * pitch + j] : 0.f;
  }
  __syncthreads();
  for (int jj = ty; jj < 64; jj += 4) {
    const int j = j0 + jj;
    if (j < jpad && tx < D) dst[(size_t)j * D + tx] = tile[tx][jj];
  }
}

}  // namespace sat

using namespace sat;

extern "C" int sat_w2v2_conv0_f32(const float* x, const float* w, const float* bias, float* y, int B, int n, int C,
                                  int k, int stride, void* stream) {
  SAT_REQUIRE(x && w && bias && y, "w2v2_conv0: null pointer");
  SAT_REQUIRE(B > 0 && C > 0 && C <= 1024 && k >= 1 && k <= 16 && stride >= 1 && n >= k, "w2v2_conv0: unsupported shape");
  const int T = (n - k) / stride + 1;
  const size_t lds = ((size_t)C * k + C) * sizeof(float);
  dim3 grid(ceil_div(T, 256), B);
  hipLaunchKernelGGL(w2v2_conv0_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, y, n, C, k, stride, T);
  SAT_LAUNCH_CHECK("w2v2_conv0_kernel");
  return SAT_OK;
}

static int layernorm_launch(const float* x, const float* gamma, const float* beta, float* y, void* y_split, int B, int C, int T,
                            int64_t x_bstride, int64_t x_cstride, int64_t y_bstride, int64_t y_cstride, int gelu,
                            int split_phases, void* stream) {
  SAT_REQUIRE(x && gamma && beta && (y || y_split), "layernorm_channels: null pointer");
  SAT_REQUIRE(B > 0 && C > 0 && T > 0, "layernorm_channels: empty shape");
  SAT_REQUIRE(C <= 1024, "layernorm_channels: at most %d channels", 1024);
  SAT_REQUIRE(!y_split || C % 16 == 0, "layernorm_channels: split planes need a multiple of 16 channels (got %d)", C);
  dim3 grid(ceil_div(T, LN_FR), B);
  const bool wide = C > 512;
  auto kern = wide ? layernorm_ch_kernel<32, 32, false> : layernorm_ch_kernel<16, 32, false>;
  hipLaunchKernelGGL(kern, grid, dim3(LN_FR * (wide ? 32 : 16)), 0, (hipStream_t)stream, x, gamma, beta, y, (uint4*)y_split, C, T,
                     (long long)x_bstride, (long long)x_cstride, (long long)y_bstride, (long long)y_cstride, gelu,
                     split_phases, (const float*)nullptr, (const float*)nullptr, 0, 0, 0);
  SAT_LAUNCH_CHECK("layernorm_ch_kernel");
  return SAT_OK;
}

extern "C" int sat_w2v2_conv0_layernorm_f32(const float* wav, const float* w, const float* bias, const float* gamma,
                                            const float* beta, float* y, void* y_split, int B, int n, int C, int k, int stride,
                                            int64_t y_bstride, int64_t y_cstride, int gelu, int split_phases, void* stream) {
  SAT_REQUIRE(wav && w && bias && gamma && beta && (y || y_split), "w2v2_conv0_layernorm: null pointer");
  SAT_REQUIRE(B > 0 && C > 0 && C <= 512 && k >= 1 && k <= 12 && stride >= 1 && n >= k, "w2v2_conv0_layernorm: unsupported shape");
  SAT_REQUIRE(!y_split || C % 16 == 0, "w2v2_conv0_layernorm: split planes need a multiple of 16 channels (got %d)", C);
  const int T = (n - k) / stride + 1;
  dim3 grid(ceil_div(T, LN_FR), B);
  auto kern = layernorm_ch_kernel<16, 32, true>;
  hipLaunchKernelGGL(kern, grid, dim3(LN_FR * 16), (size_t)C * 12 * sizeof(float), (hipStream_t)stream, wav, gamma, beta, y,
                     (uint4*)y_split, C, T, 0LL, 0LL, (long long)y_bstride, (long long)y_cstride, gelu, split_phases, w, bias, n, k,
                     stride);
  SAT_LAUNCH_CHECK("layernorm_ch_kernel<conv0>");
  return SAT_OK;
}

extern "C" int sat_layernorm_channels_f32(const float* x, const float* gamma, const float* beta, float* y, int B, int C,
                                          int T, int64_t x_bstride, int64_t x_cstride, int64_t y_bstride,
                                          int64_t y_cstride, int gelu, int split_phases, void* stream) {
  SAT_REQUIRE(y, "layernorm_channels: null pointer");
  return layernorm_launch(x, gamma, beta, y, nullptr, B, C, T, x_bstride, x_cstride, y_bstride, y_cstride, gelu, split_phases, stream);
}

extern "C" int sat_layernorm_channels_planes_f32(const float* x, const float* gamma, const float* beta, float* y, void* y_split,
                                                 int B, int C, int T, int64_t x_bstride, int64_t x_cstride, int64_t y_bstride,
                                                 int64_t y_cstride, int gelu, int split_phases, void* stream) {
  SAT_REQUIRE(y_split, "layernorm_channels_planes: null pointer");
  return layernorm_launch(x, gamma, beta, y, y_split, B, C, T, x_bstride, x_cstride, y_bstride, y_cstride, gelu, split_phases, stream);
}

extern "C" int sat_softmax_columns_f32(float* st, int G, int T, int pitch, float scale, void* stream) {
  SAT_REQUIRE(st && G > 0 && T > 0 && pitch >= T, "softmax_columns: bad arguments");
  dim3 grid(ceil_div(T, 64), G);
  hipLaunchKernelGGL(softmax_cols_kernel, grid, dim3(1024), 0, (hipStream_t)stream, st, T, pitch, scale);
  SAT_LAUNCH_CHECK("softmax_cols_kernel");
  return SAT_OK;
}

extern "C" int sat_transpose_heads_f32(const float* v, float* vt, int G, int D, int T, int pitch, int jpad, void* stream) {
  SAT_REQUIRE(v && vt && G > 0 && D > 0 && D <= 64 && T > 0 && pitch >= T && jpad >= T, "transpose_heads: bad arguments");
  dim3 grid(ceil_div(jpad, 64), G);
  hipLaunchKernelGGL(transpose_heads_kernel, grid, dim3(256), 0, (hipStream_t)stream, v, vt, D, T, pitch, jpad);
  SAT_LAUNCH_CHECK("transpose_heads_kernel");
  return SAT_OK;
}

static long long* g_attention_dbg = nullptr;
extern "C" int sat_attention_debug_stamps(int64_t* buf) {
  g_attention_dbg = (long long*)buf;
  return 7 * 8;
}

extern "C" int sat_attention_f16x3(const void* q_split, const void* k_split, const float* v, float* o, void* o_split, int B,
                                   int heads, int head_dim, int T, int v_pitch, float scale, void* stream) {
  SAT_REQUIRE(q_split && k_split && v && (o || o_split), "attention: null pointer");
  SAT_REQUIRE(B > 0 && heads > 0 && T > 0, "attention: empty shape");
  SAT_REQUIRE(head_dim == 64, "attention: head dimension 64 only (got %d)", head_dim);
  SAT_REQUIRE(v_pitch >= T && v_pitch % 4 == 0, "attention: v needs a row pitch >= T that is a multiple of 4 floats (got %d)", v_pitch);
  // the running maximum is taken on the raw scores and the scale applied afterwards: a positive scale only
  SAT_REQUIRE(scale > 0.f, "attention: the score scale must be positive (got %g)", (double)scale);
  // the K planes of an utterance (heads * head_dim channels x T x 4 B) and the V rows of a head (64 x v_pitch x 4 B) sit behind
  // 32-bit buffer descriptors
  SAT_REQUIRE((long long)heads * head_dim * T * 16 < (1LL << 31) && (long long)64 * v_pitch * 4 < (1LL << 31),
              "attention: %d heads x %d frames (v pitch %d) do not fit 31-bit buffer offsets", heads, T, v_pitch);
  const size_t lds_bytes = (size_t)2 * 2 * 64 * AT_VU * 16;  // two regions of a split V image [hi|lo][64][33 units] >= K planes = V f32 rows (64 KB)
  const int nw = T > 128 ? 8 : 4;
  auto kern = nw == 8 ? attention_f16x3_kernel<8> : attention_f16x3_kernel<4>;
  {
    // per device and per instantiation, once (the attribute call costs a driver round trip on each of the 24 layers)
    static std::atomic<uint64_t> attr_done[2] = {{0}, {0}};
    int dev;
    if (attr_needed_on_current_device(attr_done[nw == 8], &dev)) {
      SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      attr_done_on_device(attr_done[nw == 8], dev);
    }
  }
  const int hpb = heads % 2 == 0 ? 2 : 1;      // heads a block walks: the second one's K arrives under the first one's V product
  dim3 grid(ceil_div(T, 32 * nw), heads / hpb, B);
  hipLaunchKernelGGL(kern, grid, dim3(64 * nw), lds_bytes, (hipStream_t)stream, (const uint4*)q_split,
                     (const uint4*)k_split, v, o, (uint4*)o_split, heads * head_dim, T, v_pitch, scale, hpb, g_attention_dbg);
  SAT_LAUNCH_CHECK("attention_f16x3_kernel");
  return SAT_OK;
}
