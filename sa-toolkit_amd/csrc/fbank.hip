// Kaldi-compatible log-mel filterbank front end, fused per frame, plus per-utterance mean
// normalisation and edge-replicate padding.
//
// Reference: satools/satools/kaldifeature.py:461-593 (fbank) with _get_strided :80-122
// (snip_edges=False framing: left reflection of the first 120 samples, whole reversed waveform
// appended on the right), _get_window :200-264 (remove DC, pre-emphasis 0.97 with replicate,
// povey window, zero pad 400 -> 512), rfft, |.|^2, mel (80 x 257), log(max(., 1e-6));
// UttCMVN() cmvn.py:157-165; pad_input tdnnf_vq.py:228-234.
//
// One wave per frame: 400 samples -> LDS -> 512-point radix-2 FFT in LDS -> power -> sparse mel
// dot products -> log.  Output is channel-major [B][n_mel][frames] so the TDNNF stack (a valid
// conv over frames) consumes it directly.
#include "common.h"

namespace sat {

constexpr int FB_NFFT = 512;
constexpr int FB_WIN = 400;
constexpr int FB_SHIFT = 160;
constexpr int FB_FRAMES_PER_BLOCK = 4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// padded-signal sample i of _get_strided(snip_edges=False): [w[119..0], w[0..n-1], w[n-1..0]]
__device__ __forceinline__ float framed_sample(const float* __restrict__ w, int n, int i) {
  constexpr int pad = FB_WIN / 2 - FB_SHIFT / 2;  // 120
  int idx;
  if (i < pad)
    idx = pad - 1 - i;
  else if (i < pad + n)
    idx = i - pad;
  else
    idx = n - 1 - (i - pad - n);
  idx = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
  return w[idx];
}

__global__ void __launch_bounds__(64 * FB_FRAMES_PER_BLOCK)
fbank_frames_kernel(const float* __restrict__ wav, float* __restrict__ raw,
                    const float* __restrict__ window, const float* __restrict__ mel,
                    const int* __restrict__ mel_lo, const int* __restrict__ mel_hi, int n, int m,
                    float scale, int n_mel) {
  __shared__ float s_re[FB_FRAMES_PER_BLOCK][FB_NFFT];
  __shared__ float s_im[FB_FRAMES_PER_BLOCK][FB_NFFT];
  __shared__ float s_twr[FB_NFFT / 2], s_twi[FB_NFFT / 2];
  __shared__ float s_pw[FB_FRAMES_PER_BLOCK][FB_NFFT / 2 + 1];

  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int f = blockIdx.x * FB_FRAMES_PER_BLOCK + wv;
  const bool live = f < m;
  const float* w = wav + (size_t)b * n;

  for (int k = threadIdx.x; k < FB_NFFT / 2; k += blockDim.x) {
    float s, c;
    sincospif(-(float)k / (float)(FB_NFFT / 2), &s, &c);
    s_twr[k] = c;
    s_twi[k] = s;
  }

  float* re = s_re[wv];
  float* im = s_im[wv];
  // load the frame (scaled), compute its mean
  float xv[7];
  float part = 0.f;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int j = lane + 64 * k;
    float v = 0.f;
    if (live && j < FB_WIN) v = framed_sample(w, n, f * FB_SHIFT + j) * scale;
    xv[k] = v;
    part += v;
  }
  const float mean = wave_sum(part) / (float)FB_WIN;
  // x - mean, staged so that the pre-emphasis can read the left neighbour
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int j = lane + 64 * k;
    if (j < FB_WIN) re[j] = xv[k] - mean;
  }
  __syncthreads();
  float yv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int j = lane + 64 * k;
    float v = 0.f;
    if (j < FB_WIN) {
      const float cur = re[j];
      const float prev = re[j > 0 ? j - 1 : 0];
      v = (cur - 0.97f * prev) * window[j];
    }
    yv[k] = v;
  }
  __syncthreads();
  // bit-reversed scatter (9 bits), imaginary part zero
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int j = lane + 64 * k;
    const int r = (int)(__brev((unsigned)j) >> 23);
    re[r] = yv[k];
    im[r] = 0.f;
  }
  __syncthreads();
  // 9 radix-2 DIT stages, 256 butterflies each (4 per lane)
#pragma unroll 1
  for (int s = 1; s <= 9; ++s) {
    const int half = 1 << (s - 1);
    const int tstep = FB_NFFT >> s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
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
  // power spectrum, bins 0..256: reference takes abs() then pow(2)
  for (int k = lane; k <= FB_NFFT / 2; k += 64) {
    const float a = sqrtf(re[k] * re[k] + im[k] * im[k]);
    s_pw[wv][k] = a * a;
  }
  __syncthreads();
  if (!live) return;
  for (int mb = lane; mb < n_mel; mb += 64) {
    const float* mrow = mel + (size_t)mb * (FB_NFFT / 2 + 1);
    float acc = 0.f;
    const int lo = mel_lo[mb], hi = mel_hi[mb];
    for (int k = lo; k < hi; ++k) acc = fmaf(s_pw[wv][k], mrow[k], acc);
    raw[((size_t)b * n_mel + mb) * m + f] = logf(fmaxf(acc, 1e-6f));
  }
}

// mean over frames of one (utterance, mel channel) row
__global__ void __launch_bounds__(256) row_mean_kernel(const float* __restrict__ raw, float* __restrict__ means, int m) {
  __shared__ float s_part[4];
  const size_t row = blockIdx.x;
  const float* src = raw + row * m;
  float part = 0.f;
  for (int t = threadIdx.x; t < m; t += 256) part += src[t];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) means[row] = (s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)m;
}

// Right padding of the reference's pad_input (tdnnf_vq.py:228-234): the last frames of the N
// utterances are tiled as one sequence and cut into N pieces, so right-pad frame p of utterance b
// comes from utterance (b*pad + p) mod N.  `interleave` selects that behaviour (plain replicate
// otherwise).
__device__ __forceinline__ int right_pad_source(int b, int p, int pad, int B, int interleave) {
  return interleave ? (int)(((long long)b * pad + p) % B) : b;
}

// one block per (utterance, mel channel): subtract the mean over frames, write with padding
__global__ void __launch_bounds__(256) cmvn_pad_kernel(const float* __restrict__ raw, const float* __restrict__ means,
                                                       float* __restrict__ out, int m, int pad, int do_cmvn,
                                                       int n_ch, int B) {
  const int row = blockIdx.x;
  const int b = row / n_ch, c = row - b * n_ch;
  const float* src = raw + (size_t)row * m;
  float* dst = out + (size_t)row * (size_t)(m + 2 * pad);
  const float mean = do_cmvn ? means[row] : 0.f;
  for (int t = threadIdx.x; t < m + pad; t += 256) {
    const int s = t < pad ? 0 : t - pad;
    dst[t] = src[s] - mean;
  }
  for (int p = threadIdx.x; p < pad; p += 256) {
    const int srow = right_pad_source(b, p, pad, B, 1) * n_ch + c;
    const float mu = do_cmvn ? means[srow] : 0.f;
    dst[pad + m + p] = raw[(size_t)srow * m + (m - 1)] - mu;
  }
}

__global__ void __launch_bounds__(256) pad_replicate_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int T, int left, int right, int C, int B,
                                                            int interleave) {
  const int row = blockIdx.y;
  const int b = row / C, c = row - b * C;
  const int To = left + T + right;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= To) return;
  float v;
  if (t < left + T) {
    const int s = t < left ? 0 : t - left;
    v = x[(size_t)row * T + s];
  } else {
    const int srow = right_pad_source(b, t - left - T, right, B, interleave) * C + c;
    v = x[(size_t)srow * T + (T - 1)];
  }
  y[(size_t)row * To + t] = v;
}

}  // namespace sat

using namespace sat;

extern "C" size_t sat_fbank_workspace_bytes(int B, int n) {
  if (B <= 0 || n <= 0) return 0;
  const size_t m = (size_t)(n + FB_SHIFT / 2) / FB_SHIFT;
  return align_up((size_t)B * 128 * (m + 1) * sizeof(float), 256);  // up to 128 mel channels: rows + row means
}

extern "C" int sat_fbank_cmvn_pad_f32(const float* wav, float* feats, const float* window, const float* mel,
                                      const int32_t* mel_lo, const int32_t* mel_hi, void* workspace,
                                      size_t workspace_bytes, int B, int n, float scale, int n_mel, int pad,
                                      int do_cmvn, void* stream) {
  SAT_REQUIRE(wav && feats && window && mel && mel_lo && mel_hi && workspace, "fbank: null pointer");
  SAT_REQUIRE(B > 0 && n_mel > 0 && n_mel <= 128 && pad >= 0, "fbank: bad sizes");
  // reference asserts 2 <= window_size <= len(waveform) (kaldifeature.py:189-191)
  SAT_REQUIRE(n >= FB_WIN, "fbank: choose a window size %d that is [2, %d]", FB_WIN, n);
  const int m = (n + FB_SHIFT / 2) / FB_SHIFT;
  SAT_REQUIRE_WORKSPACE(workspace_bytes >= (size_t)B * n_mel * (m + 1) * sizeof(float), "fbank: workspace too small");
  float* raw = (float*)workspace;
  float* means = raw + (size_t)B * n_mel * m;
  dim3 grid(ceil_div(m, FB_FRAMES_PER_BLOCK), B);
  hipLaunchKernelGGL(fbank_frames_kernel, grid, dim3(64 * FB_FRAMES_PER_BLOCK), 0, (hipStream_t)stream, wav, raw,
                     window, mel, mel_lo, mel_hi, n, m, scale, n_mel);
  SAT_LAUNCH_CHECK("fbank_frames_kernel");
  if (do_cmvn) {
    hipLaunchKernelGGL(row_mean_kernel, dim3(B * n_mel), dim3(256), 0, (hipStream_t)stream, raw, means, m);
    SAT_LAUNCH_CHECK("row_mean_kernel");
  }
  hipLaunchKernelGGL(cmvn_pad_kernel, dim3(B * n_mel), dim3(256), 0, (hipStream_t)stream, raw, means, feats, m, pad,
                     do_cmvn, n_mel, B);
  SAT_LAUNCH_CHECK("cmvn_pad_kernel");
  return SAT_OK;
}

extern "C" int sat_pad_replicate_f32(const float* x, float* y, int B, int C, int T, int left, int right,
                                     int interleave_right, void* stream) {
  SAT_REQUIRE(x && y && B > 0 && C > 0 && T > 0 && left >= 0 && right >= 0, "pad_replicate: bad arguments");
  dim3 grid(ceil_div(left + T + right, 256), B * C);
  hipLaunchKernelGGL(pad_replicate_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, T, left, right, C, B, interleave_right);
  SAT_LAUNCH_CHECK("pad_replicate_kernel");
  return SAT_OK;
}
