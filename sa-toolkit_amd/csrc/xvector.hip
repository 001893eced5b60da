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
