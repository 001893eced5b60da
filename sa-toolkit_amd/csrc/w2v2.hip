// wav2vec2 support kernels (everything matrix-shaped runs on the fused conv1d kernel):
// first conv layer (1 -> C, k, stride), LayerNorm over channels (+GELU, + even/odd phase split for a
// following stride-2 conv), softmax over keys of the transposed score matrix, per-head transpose of V.
// Reference: torchaudio.models.wav2vec2 (third-party; semantics restated in oracle/wav2vec2.py) as
// configured by egs/asr/librispeech/local/chain/tuning/tdnnf_wav2vec2_vq.py:39-56.
#include "common.h"

namespace sat {

__device__ __forceinline__ float gelu_erf(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }

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
constexpr int LN_SLICES = 16;
constexpr int LN_FR = 16;          // frames per block
constexpr int LN_MAXPT = 64;       // channels per thread kept in registers: C <= 16 * 64
// Block = 16 frames x 16 channel slices (256 threads, 512 blocks at B = 32, T = 249): a thread reads its 64
// channels of one frame ONCE into registers and both statistics passes and the output pass run from there.
// Slice c = sl, sl + 16, ... and the slice-ordered total are the summation order of the 64-frame version.
__global__ void __launch_bounds__(LN_FR * LN_SLICES) layernorm_ch_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
    int C, int T, long long x_bs, long long x_cs, long long y_bs, long long y_cs, int gelu, int split) {
  __shared__ float part[LN_SLICES][LN_FR];
  __shared__ float s_mean[LN_FR], s_rstd[LN_FR];
  const int b = blockIdx.y;
  const int tx = threadIdx.x & (LN_FR - 1), sl = threadIdx.x / LN_FR;
  const int t = blockIdx.x * LN_FR + tx;
  const bool ok = t < T;
  const float* xb = x + (size_t)b * x_bs + (ok ? t : 0);
  float v[LN_MAXPT];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXPT; ++k) {
    const int c = sl + k * LN_SLICES;
    v[k] = (ok && c < C) ? xb[(size_t)c * x_cs] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < LN_MAXPT; ++k)
    if (sl + k * LN_SLICES < C) s += v[k];
  part[sl][tx] = s;
  __syncthreads();
  if (sl == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < LN_SLICES; ++i) tot += part[i][tx];
    s_mean[tx] = tot / (float)C;
  }
  __syncthreads();
  const float mean = s_mean[tx];
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXPT; ++k)
    if (sl + k * LN_SLICES < C) {
      const float d = v[k] - mean;
      q = fmaf(d, d, q);
    }
  __syncthreads();
  part[sl][tx] = q;
  __syncthreads();
  if (sl == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < LN_SLICES; ++i) tot += part[i][tx];
    s_rstd[tx] = 1.0f / sqrtf(tot / (float)C + 1e-5f);
  }
  __syncthreads();
  if (!ok) return;
  const float rstd = s_rstd[tx];
  float* yb = y + (size_t)b * y_bs;
  const int ph = split ? (t & 1) : 0;
  const int u = split ? (t >> 1) : t;
  const bool tail = split && (T & 1) && t == T - 1;   // odd length: the odd phase's last slot is zero padding
#pragma unroll
  for (int k = 0; k < LN_MAXPT; ++k) {
    const int c = sl + k * LN_SLICES;
    if (c < C) {
      float o = (v[k] - mean) * rstd * gamma[c] + beta[c];
      if (gelu) o = gelu_erf(o);
      yb[(size_t)(ph * C + c) * y_cs + u] = o;
      if (tail) yb[(size_t)(C + c) * y_cs + u] = 0.f;
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

extern "C" int sat_layernorm_channels_f32(const float* x, const float* gamma, const float* beta, float* y, int B, int C,
                                          int T, int64_t x_bstride, int64_t x_cstride, int64_t y_bstride,
                                          int64_t y_cstride, int gelu, int split_phases, void* stream) {
  SAT_REQUIRE(x && gamma && beta && y, "layernorm_channels: null pointer");
  SAT_REQUIRE(B > 0 && C > 0 && T > 0, "layernorm_channels: empty shape");
  SAT_REQUIRE(C <= LN_SLICES * LN_MAXPT, "layernorm_channels: at most %d channels", LN_SLICES * LN_MAXPT);
  dim3 grid(ceil_div(T, LN_FR), B);
  hipLaunchKernelGGL(layernorm_ch_kernel, grid, dim3(LN_FR * LN_SLICES), 0, (hipStream_t)stream, x, gamma, beta, y, C, T,
                     (long long)x_bstride, (long long)x_cstride, (long long)y_bstride, (long long)y_cstride, gelu,
                     split_phases);
  SAT_LAUNCH_CHECK("layernorm_ch_kernel");
  return SAT_OK;
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
