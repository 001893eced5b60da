// Kernels of the x-vector extractor (ECAPA-TDNN) that are not convolutions.
// Reference: satools/satools/sidekit/preprocessor.py:164-236 (MelSpecFrontEnd: PreEmphasis -> torchaudio
// MelSpectrogram(n_fft 1024, win 400, hop 160, 90-7600 Hz, 80 mel, power 2) + 1e-6 -> log -> InstanceNorm1d),
// satools/satools/augmentation.py:219-244 (PreEmphasis), sidekit/nn.py:75-154 (Res2Net sums, SE gate),
// sidekit/pooling.py:141-155 (AttentiveStatsPool), egs/asv/voxceleb/local/tuning/ecapa_tdnn.py:74-76 (L2 norm).
#include "common.h"

namespace sat {

constexpr int XV_NFFT = 1024;
constexpr int XV_WIN = 400;
constexpr int XV_HOP = 160;
constexpr int XV_FPB = 4;   // frames per block, one wave each

__device__ __forceinline__ float xv_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float xv_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// pre-emphasised sample i of the utterance: x[i] - coef * x[i-1], x[-1] := x[1] (reflect pad of one sample)
__device__ __forceinline__ float preemph(const float* __restrict__ x, int n, int i, float coef) {
  const float prev = i > 0 ? x[i - 1] : x[n > 1 ? 1 : 0];
  return x[i] - coef * prev;
}

// One wave per frame: 400 windowed samples of the pre-emphasised, centre-reflect-padded signal in the middle of
// a 1024-point frame -> radix-2 FFT in LDS -> power -> sparse mel dot products -> log(. + 1e-6).
// out [B][n_mel][frames], frames = 1 + n / hop (torch.stft, center=True).
__global__ void __launch_bounds__(64 * XV_FPB)
melspec_logmel_kernel(const float* __restrict__ wav, float* __restrict__ out, const float* __restrict__ window,
                      const float* __restrict__ fb, const int* __restrict__ fb_lo, const int* __restrict__ fb_hi, int n,
                      int frames, int n_mel, float coef) {
  __shared__ float s_re[XV_FPB][XV_NFFT];
  __shared__ float s_im[XV_FPB][XV_NFFT];
  __shared__ float s_twr[XV_NFFT / 2], s_twi[XV_NFFT / 2];
  __shared__ float s_pw[XV_FPB][XV_NFFT / 2 + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int f = blockIdx.x * XV_FPB + wv;
  const bool live = f < frames;
  const float* x = wav + (size_t)b * n;
  for (int k = threadIdx.x; k < XV_NFFT / 2; k += blockDim.x) {
    float s, c;
    sincospif(-(float)k / (float)(XV_NFFT / 2), &s, &c);
    s_twr[k] = c;
    s_twi[k] = s;
  }
  float* re = s_re[wv];
  float* im = s_im[wv];
  constexpr int off = (XV_NFFT - XV_WIN) / 2;   // the window sits in the middle of the n_fft frame (torch.stft)
#pragma unroll
  for (int k = 0; k < XV_NFFT / 64; ++k) {
    const int j = lane + 64 * k;
    float v = 0.f;
    if (live && j >= off && j < off + XV_WIN) {
      int s = f * XV_HOP + j - XV_NFFT / 2;       // index into the signal, reflect-padded by n_fft/2 on both sides
      if (s < 0) s = -s;
      if (s >= n) s = 2 * (n - 1) - s;
      s = s < 0 ? 0 : (s >= n ? n - 1 : s);
      v = preemph(x, n, s, coef) * window[j - off];
    }
    re[(int)(__brev((unsigned)j) >> 22)] = v;   // bit-reversed scatter (10 bits)
    im[j] = 0.f;
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 1; s <= 10; ++s) {
    const int half = 1 << (s - 1);
    const int tstep = XV_NFFT >> s;
#pragma unroll
    for (int k = 0; k < XV_NFFT / 2 / 64; ++k) {
      const int i = lane + 64 * k;
      const int pos = i & (half - 1);
      const int a = ((i >> (s - 1)) << s) + pos;
      const int c = a + half;
      const float wr = s_twr[pos * tstep], wi = s_twi[pos * tstep];
      const float xr = re[c], xi = im[c];
      const float tr = xr * wr - xi * wi;
      const float ti = xr * wi + xi * wr;
      const float ur = re[a], ui = im[a];
      re[a] = ur + tr;
      im[a] = ui + ti;
      re[c] = ur - tr;
      im[c] = ui - ti;
    }
    __syncthreads();
  }
  for (int k = lane; k <= XV_NFFT / 2; k += 64) s_pw[wv][k] = re[k] * re[k] + im[k] * im[k];   // |X|^2 (power = 2)
  __syncthreads();
  if (!live) return;
  for (int m = lane; m < n_mel; m += 64) {
    const float* row = fb + (size_t)m * (XV_NFFT / 2 + 1);
    float acc = 0.f;
    for (int k = fb_lo[m]; k < fb_hi[m]; ++k) acc = fmaf(s_pw[wv][k], row[k], acc);
    out[((size_t)b * n_mel + m) * frames + f] = logf(acc + 1e-6f);
  }
}

// InstanceNorm1d(affine=False, eps): per row of [R][T], (x - mean) / sqrt(biased var + eps); one wave per row
__global__ void __launch_bounds__(256) instnorm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int T,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* xr = x + (size_t)r * T;
  float s = 0.f;
  for (int t = lane; t < T; t += 64) s += xr[t];
  const float mean = xv_wave_sum(s) / (float)T;
  float q = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float d = xr[t] - mean;
    q = fmaf(d, d, q);
  }
  const float rstd = 1.0f / sqrtf(xv_wave_sum(q) / (float)T + eps);
  for (int t = lane; t < T; t += 64) y[(size_t)r * T + t] = (xr[t] - mean) * rstd;
}

// mean over time of every row of [R][T]; one wave per row (SE_Connect's x.mean(dim=2), sidekit/nn.py:132)
__global__ void __launch_bounds__(256) row_mean_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int T) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float s = 0.f;
  for (int t = lane; t < T; t += 64) s += x[(size_t)r * T + t];
  s = xv_wave_sum(s);
  if (lane == 0) y[r] = s / (float)T;
}

// y = a + b (+ c): channel slices of [B][C][T] tensors (batch / channel strides, T contiguous)
__global__ void __launch_bounds__(256) add3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ c, float* __restrict__ y, int C, int T,
                                                   long long a_bs, long long a_cs, long long b_bs, long long b_cs,
                                                   long long c_bs, long long c_cs, long long y_bs, long long y_cs) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int ch = blockIdx.y, bt = blockIdx.z;
  if (t >= T) return;
  float v = a[bt * a_bs + ch * a_cs + t] + b[bt * b_bs + ch * b_cs + t];
  if (c) v = v + c[bt * c_bs + ch * c_cs + t];
  y[bt * y_bs + ch * y_cs + t] = v;
}

// SE gate and the block's skip connections: y = z * sigmoid(g[b][c]) + s1 (+ s2 (+ s3)), added left to right like
// `layer(x) + out1 + out2 + out3` (sidekit/archi.py:183-185)
__global__ void __launch_bounds__(256) se_gate_add_kernel(const float* __restrict__ z, const float* __restrict__ g,
                                                          const float* __restrict__ s1, const float* __restrict__ s2,
                                                          const float* __restrict__ s3, float* __restrict__ y, int C, int T,
                                                          long long y_bs, long long y_cs) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int ch = blockIdx.y, bt = blockIdx.z;
  if (t >= T) return;
  const size_t i = ((size_t)bt * C + ch) * T + t;
  const float gate = 1.0f / (1.0f + expf(-g[(size_t)bt * C + ch]));
  float v = z[i] * gate;
  if (s1) v = v + s1[i];
  if (s2) v = v + s2[i];
  if (s3) v = v + s3[i];
  y[bt * y_bs + ch * y_cs + t] = v;
}

__global__ void __launch_bounds__(256) tanh_kernel(float* __restrict__ x, size_t n) {
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i < n) x[i] = tanhf(x[i]);
}

// AttentiveStatsPool tail: per (b, c) row  w = softmax_t(logits), mean = sum w x, std = sqrt(max(sum w x^2 - mean^2, 1e-9))
// out [B][2C]: means then stds; one wave per row
__global__ void __launch_bounds__(256) attentive_stats_kernel(const float* __restrict__ x, const float* __restrict__ logits,
                                                              float* __restrict__ out, int B, int C, int T) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B * C) return;
  const float* xr = x + (size_t)r * T;
  const float* lr = logits + (size_t)r * T;
  float mx = -INFINITY;
  for (int t = lane; t < T; t += 64) mx = fmaxf(mx, lr[t]);
  mx = xv_wave_max(mx);
  float se = 0.f;
  for (int t = lane; t < T; t += 64) se += expf(lr[t] - mx);
  se = xv_wave_sum(se);
  float m1 = 0.f, m2 = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float w = expf(lr[t] - mx) / se;
    const float v = xr[t];
    m1 += w * v;
    m2 += w * (v * v);
  }
  m1 = xv_wave_sum(m1);
  m2 = xv_wave_sum(m2);
  if (lane == 0) {
    const int b = r / C, c = r - b * C;
    const float var = m2 - m1 * m1;
    out[(size_t)b * 2 * C + c] = m1;
    out[(size_t)b * 2 * C + C + c] = sqrtf(var > 1e-9f ? var : 1e-9f);
  }
}

// F.normalize(x, dim=1): rows of [R][D] divided by max(||row||, 1e-12); one wave per row
__global__ void __launch_bounds__(256) l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int D) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s = fmaf(x[(size_t)r * D + d], x[(size_t)r * D + d], s);
  const float nrm = fmaxf(sqrtf(xv_wave_sum(s)), 1e-12f);
  for (int d = lane; d < D; d += 64) y[(size_t)r * D + d] = x[(size_t)r * D + d] / nrm;
}

// ---- Linear layers on pooled vectors (the squeeze-excitation MLP, the embedding layer): y[b][o] = epi(w[o] . x[b]).
// One wave per output row and group of NB batch rows (the weight row is read once for the group); the conv tile at T = 1 spent
// 141 us per layer on a K loop of 32 dependent chunk round trips for one useful column.
template <int NB, bool VEC>
__global__ void __launch_bounds__(256) linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ y,
                                                          int B, int Cin, int Cout, int relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o = blockIdx.x * 4 + wave, b0 = blockIdx.y * NB;
  if (o >= Cout) return;
  const int nb = min(NB, B - b0);
  float acc[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) acc[k] = 0.f;
  const float* wr = w + (size_t)o * Cin;
  if (VEC) {
    for (int i = 4 * lane; i < Cin; i += 256) {
      const float4 wv = *(const float4*)(wr + i);
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const float4 xv = *(const float4*)(x + (size_t)(b0 + (k < nb ? k : 0)) * Cin + i);
        acc[k] = __builtin_fmaf(wv.x, xv.x, acc[k]);
        acc[k] = __builtin_fmaf(wv.y, xv.y, acc[k]);
        acc[k] = __builtin_fmaf(wv.z, xv.z, acc[k]);
        acc[k] = __builtin_fmaf(wv.w, xv.w, acc[k]);
      }
    }
  } else {
    for (int i = lane; i < Cin; i += 64) {
      const float wv = wr[i];
#pragma unroll
      for (int k = 0; k < NB; ++k) acc[k] = __builtin_fmaf(wv, x[(size_t)(b0 + (k < nb ? k : 0)) * Cin + i], acc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < NB; ++k)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
  if (lane == 0) {
    const float bi = bias ? bias[o] : 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if (k >= nb) break;
      float v = acc[k] + bi;
      if (relu) v = fmaxf(v, 0.f);
      if (scale) v = v * scale[o] + (shift ? shift[o] : 0.f);
      y[(size_t)(b0 + k) * Cout + o] = v;
    }
  }
}

// ---- Res2Conv1dReluBn (sidekit/nn.py:74-110) of 64-channel pieces as ONE launch: sp_i = bn_i(relu(conv_i(sp_{i-1} + x_i))), i = 0 .. nums - 1,
// the last piece copied.  Launched conv by conv the chain is 7 x (add3 + a 64 -> 64 three-tap conv of 31 us, latency-bound) per block of the
// net.  Here a block keeps a window of W = 64 NT columns of ONE utterance in LDS — the piece's input (x_i + previous output) and its output,
// 32 columns of halo on either side (the receptive field of the chain grows by one dilation per piece: 7 x 4 <= 32) — and walks the pieces:
// exact f32 on v_mfma_f32_32x32x2_f32 (A = the piece's weights [tap][ci][co] staged in LDS, B = input columns shifted by the tap), four waves =
// two 32-row tiles x two halves of the window.  Only the W - 64 centre columns are stored; fringe columns compute on zeros and are dropped.
constexpr int R2_MARG = 4, R2_HALO = 32;
typedef float r2_f32x16 __attribute__((ext_vector_type(16)));

template <int NT>
__global__ void __launch_bounds__(256) res2_chain_kernel(const float* __restrict__ y, float* __restrict__ z, const float* __restrict__ w,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int C, int T, int nums,
                                                         int dil) {
  extern __shared__ __attribute__((aligned(16))) float r2_lds[];
  constexpr int W = 64 * NT, WP = W + 2 * R2_MARG, TT = W - 2 * R2_HALO;
  float* bufA = r2_lds;                   // [64][WP] input of the current piece
  float* bufB = bufA + 64 * WP;           // [64][WP] its output (zero outside the utterance)
  float* wl = bufB + 64 * WP;             // [3][64][64] weights of the current piece
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * TT, t_lo = t0 - R2_HALO;
  const float* yb = y + (size_t)b * C * T;
  float* zb = z + (size_t)b * C * T;
  const int mt = wave >> 1, n0 = (wave & 1) * NT;      // this wave: rows 32 mt .., column tiles n0 .. n0 + NT - 1

  for (int i = tid; i < 64 * 2 * R2_MARG; i += 256) {  // the margins the shifted fragment reads touch: zero, never written again
    const int c = i / (2 * R2_MARG), m = i % (2 * R2_MARG);
    const int col = m < R2_MARG ? m : W + m;
    bufA[c * WP + col] = 0.f;
  }
  {
    float y0[64 * W / 256];
#pragma unroll
    for (int k = 0; k < 64 * W / 256; ++k) {
      const int i = tid + 256 * k, c = i / W, j = i - c * W, t = t_lo + j;
      y0[k] = (t >= 0 && t < T) ? yb[(size_t)c * T + t] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 64 * W / 256; ++k) {
      const int i = tid + 256 * k, c = i / W, j = i - c * W;
      bufA[c * WP + R2_MARG + j] = y0[k];
    }
  }
  constexpr int NIN = 64 * W / 256;                    // input elements per thread
  for (int p = 0; p < nums; ++p) {
    // (every global load of the piece is in flight before anything waits for one: a block is one wave per SIMD, nothing else hides them.
    // Requesting the weights one piece ahead, to wait in registers behind the MFMAs: 135 us against 116, not kept)
    float4 wv[12];
    const float4* w4 = (const float4*)(w + (size_t)p * 3 * 64 * 64);
#pragma unroll
    for (int k = 0; k < 12; ++k) wv[k] = w4[tid + 256 * k];
    float yn[NIN];                                     // piece p + 1 of y, added to this piece's output below
    const bool more = p + 1 < nums;
    {
      const float* yp = yb + (size_t)(p + 1) * 64 * T;
#pragma unroll
      for (int k = 0; k < NIN; ++k) {
        const int i = tid + 256 * k, c = i / W, j = i - c * W, t = t_lo + j;
        yn[k] = (more && t >= 0 && t < T) ? yp[(size_t)c * T + t] : 0.f;
      }
    }
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
      sc[r] = scale[p * 64 + co], sh[r] = shift[p * 64 + co];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) ((float4*)wl)[tid + 256 * k] = wv[k];
    __syncthreads();
    r2_f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // 96 steps (tap, channel pair), four at a time: their fragment reads are issued together, so that one LDS round trip stands in front
    // of 4 NT MFMAs instead of NT
    const float* xa = bufA + R2_MARG + 32 * n0 + l31 + lh * WP;
    const float* wa = wl + 32 * mt + l31 + lh * 64;
    for (int tap = 0; tap < 3; ++tap) {
      const float* xt = xa + (tap - 1) * dil;
      const float* wt = wa + tap * 64 * 64;
#pragma unroll 2
      for (int ci = 0; ci < 64; ci += 8) {
        float a4[4], b4[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a4[u] = wt[(ci + 2 * u) * 64];
#pragma unroll
          for (int n = 0; n < NT; ++n) b4[u][n] = xt[(ci + 2 * u) * WP + 32 * n];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], b4[u][n], acc[n], 0, 0, 0);
      }
    }
    // relu, BatchNorm affine; the centre columns to memory, every column (zero outside the utterance) to bufB for the next piece
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int j = 32 * (n0 + n) + l31, t = t_lo + j;
      const bool inside = t >= 0 && t < T, keep = inside && j >= R2_HALO && j < R2_HALO + TT;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = fmaxf(acc[n][r], 0.f) * sc[r] + sh[r];
        bufB[co * WP + R2_MARG + j] = inside ? v : 0.f;
        if (keep) zb[(size_t)(p * 64 + co) * T + t] = v;
      }
    }
    __syncthreads();
    if (more) {
#pragma unroll
      for (int k = 0; k < NIN; ++k) {
        const int i = tid + 256 * k, c = i / W, j = i - c * W;
        bufA[c * WP + R2_MARG + j] = yn[k] + bufB[c * WP + R2_MARG + j];
      }
    }
  }
  // the piece the chain leaves untouched
  const float* yl = yb + (size_t)nums * 64 * T;
  float* zl = zb + (size_t)nums * 64 * T;
  for (int i = tid; i < 64 * TT; i += 256) {
    const int c = i / TT, t = t0 + (i - c * TT);
    if (t < T) zl[(size_t)c * T + t] = yl[(size_t)c * T + t];
  }
}

}  // namespace sat

using namespace sat;

extern "C" int sat_melspec_logmel_f32(const float* wav, float* out, const float* window, const float* fb,
                                      const int32_t* fb_lo, const int32_t* fb_hi, int B, int n, int n_mel, float coef,
                                      void* stream) {
  SAT_REQUIRE(wav && out && window && fb && fb_lo && fb_hi, "melspec_logmel: null pointer");
  SAT_REQUIRE(B > 0 && n > XV_NFFT / 2 && n_mel > 0, "melspec_logmel: utterance of %d samples is shorter than the reflect padding", n);
  const int frames = 1 + n / XV_HOP;
  dim3 grid(ceil_div(frames, XV_FPB), B);
  hipLaunchKernelGGL(melspec_logmel_kernel, grid, dim3(64 * XV_FPB), 0, (hipStream_t)stream, wav, out, window, fb, fb_lo, fb_hi,
                     n, frames, n_mel, coef);
  SAT_LAUNCH_CHECK("melspec_logmel_kernel");
  return SAT_OK;
}

extern "C" int sat_instnorm_rows_f32(const float* x, float* y, int R, int T, float eps, void* stream) {
  SAT_REQUIRE(x && y && R > 0 && T > 0, "instnorm_rows: bad arguments");
  hipLaunchKernelGGL(instnorm_rows_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, (hipStream_t)stream, x, y, R, T, eps);
  SAT_LAUNCH_CHECK("instnorm_rows_kernel");
  return SAT_OK;
}

extern "C" int sat_row_mean_f32(const float* x, float* y, int R, int T, void* stream) {
  SAT_REQUIRE(x && y && R > 0 && T > 0, "row_mean: bad arguments");
  hipLaunchKernelGGL(row_mean_rows_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, (hipStream_t)stream, x, y, R, T);
  SAT_LAUNCH_CHECK("row_mean_rows_kernel");
  return SAT_OK;
}

extern "C" int sat_add3_f32(const float* a, const float* b, const float* c, float* y, int B, int C, int T, int64_t a_bs,
                            int64_t a_cs, int64_t b_bs, int64_t b_cs, int64_t c_bs, int64_t c_cs, int64_t y_bs, int64_t y_cs,
                            void* stream) {
  SAT_REQUIRE(a && b && y && B > 0 && C > 0 && T > 0 && B < 65536 && C < 65536, "add3: bad arguments");
  hipLaunchKernelGGL(add3_kernel, dim3(ceil_div(T, 256), C, B), dim3(256), 0, (hipStream_t)stream, a, b, c, y, C, T,
                     (long long)a_bs, (long long)a_cs, (long long)b_bs, (long long)b_cs, (long long)c_bs, (long long)c_cs,
                     (long long)y_bs, (long long)y_cs);
  SAT_LAUNCH_CHECK("add3_kernel");
  return SAT_OK;
}

extern "C" int sat_se_gate_add_f32(const float* z, const float* gate_logits, const float* s1, const float* s2,
                                   const float* s3, float* y, int B, int C, int T, int64_t y_bs, int64_t y_cs, void* stream) {
  SAT_REQUIRE(z && gate_logits && y && B > 0 && C > 0 && T > 0 && B < 65536 && C < 65536, "se_gate_add: bad arguments");
  hipLaunchKernelGGL(se_gate_add_kernel, dim3(ceil_div(T, 256), C, B), dim3(256), 0, (hipStream_t)stream, z, gate_logits, s1, s2,
                     s3, y, C, T, (long long)y_bs, (long long)y_cs);
  SAT_LAUNCH_CHECK("se_gate_add_kernel");
  return SAT_OK;
}

extern "C" int sat_tanh_inplace_f32(float* x, size_t n, void* stream) {
  SAT_REQUIRE(x && n > 0, "tanh: bad arguments");
  hipLaunchKernelGGL(tanh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n);
  SAT_LAUNCH_CHECK("tanh_kernel");
  return SAT_OK;
}

extern "C" int sat_attentive_stats_f32(const float* x, const float* logits, float* out, int B, int C, int T, void* stream) {
  SAT_REQUIRE(x && logits && out && B > 0 && C > 0 && T > 0, "attentive_stats: bad arguments");
  hipLaunchKernelGGL(attentive_stats_kernel, dim3(ceil_div(B * C, 4)), dim3(256), 0, (hipStream_t)stream, x, logits, out, B, C, T);
  SAT_LAUNCH_CHECK("attentive_stats_kernel");
  return SAT_OK;
}

extern "C" int sat_l2norm_rows_f32(const float* x, float* y, int R, int D, void* stream) {
  SAT_REQUIRE(x && y && R > 0 && D > 0, "l2norm_rows: bad arguments");
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, (hipStream_t)stream, x, y, R, D);
  SAT_LAUNCH_CHECK("l2norm_rows_kernel");
  return SAT_OK;
}

extern "C" int sat_linear_rows_f32(const float* x, const float* w, const float* bias, const float* ch_scale, const float* ch_shift,
                                   int relu, float* y, int B, int Cin, int Cout, void* stream) {
  SAT_REQUIRE(x && w && y && x != y && B > 0 && Cin > 0 && Cout > 0 && B < 65536 * 8, "linear_rows: bad arguments");
  SAT_REQUIRE(ch_scale || !ch_shift, "linear_rows: ch_shift without ch_scale");
  constexpr int NB = 8;
  const bool vec = Cin % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0;
  const dim3 grid(ceil_div(Cout, 4), ceil_div(B, NB));
  if (vec) hipLaunchKernelGGL((linear_rows_kernel<NB, true>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, ch_scale, ch_shift, y, B, Cin, Cout, relu);
  else hipLaunchKernelGGL((linear_rows_kernel<NB, false>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, ch_scale, ch_shift, y, B, Cin, Cout, relu);
  SAT_LAUNCH_CHECK("linear_rows_kernel");
  return SAT_OK;
}

extern "C" int sat_res2_chain_f32(const float* y, float* z, const float* w, const float* scale, const float* shift, int B, int C, int T,
                                  int nums, int dilation, void* stream) {
  SAT_REQUIRE(y && z && y != z && w && scale && shift && B > 0 && T > 0 && B < 65536, "res2_chain: bad arguments");
  SAT_REQUIRE(nums >= 1 && C == (nums + 1) * 64, "res2_chain: pieces of 64 channels, C = (nums + 1) * 64");
  SAT_REQUIRE(dilation >= 1 && dilation <= R2_MARG && dilation * nums <= R2_HALO, "res2_chain: dilation x pieces exceeds the staged halo");
  // 128-column centres when that fills the CUs, else 64
  const bool wide = (long long)B * ceil_div(T, 128) >= 256;
  if (wide) {
    constexpr int NT = 3;
    const size_t lds = ((size_t)2 * 64 * (64 * NT + 2 * R2_MARG) + 3 * 64 * 64) * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};      // per device
    int dev;
    if (attr_needed_on_current_device(attr_done, &dev)) {
      SAT_HIP(hipFuncSetAttribute((const void*)res2_chain_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done_on_device(attr_done, dev);
    }
    hipLaunchKernelGGL(res2_chain_kernel<NT>, dim3(ceil_div(T, 64 * NT - 2 * R2_HALO), B), dim3(256), lds, (hipStream_t)stream, y, z, w, scale, shift, C, T, nums, dilation);
  } else {
    constexpr int NT = 2;
    const size_t lds = ((size_t)2 * 64 * (64 * NT + 2 * R2_MARG) + 3 * 64 * 64) * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};      // per device
    int dev;
    if (attr_needed_on_current_device(attr_done, &dev)) {
      SAT_HIP(hipFuncSetAttribute((const void*)res2_chain_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done_on_device(attr_done, dev);
    }
    hipLaunchKernelGGL(res2_chain_kernel<NT>, dim3(ceil_div(T, 64 * NT - 2 * R2_HALO), B), dim3(256), lds, (hipStream_t)stream, y, z, w, scale, shift, C, T, nums, dilation);
  }
  SAT_LAUNCH_CHECK("res2_chain_kernel");
  return SAT_OK;
}
