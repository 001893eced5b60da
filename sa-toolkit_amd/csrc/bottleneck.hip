// Bottleneck vector quantiser, F0 normalisation / transformation and generator-input assembly.
//
// Reference: VectorQuantizerEMA.forward (eval), satools/satools/chain/nn.py:402-476;
// UttCMVN(var_norm=True, keep_zeros=True), satools/satools/cmvn.py:143-155;
// quantize_f0 / awgn_f0, satools/satools/hifigan/nn.py:28-62;
// Net._forward input assembly, egs/vc/libritts/local/tuning/hifigan.py:83-97.
#include "common.h"

namespace sat {

constexpr int VQ_MAX_CODES = 64;

// One thread per frame; frames of a channel are contiguous so the wave reads z coalesced.
// The codebook sits transposed in LDS ([d][code]) and is read as a broadcast.
// Arithmetic follows the reference formula and association exactly:
//   dist[e] = (sum_d x_d^2 + sum_d e_d^2) - 2 * (x . e)      all f32, first minimum wins.
// Block = 64 frames x 4 code groups (one wave per group): every (frame, code) distance is still one thread's
// d-ordered fma chain, the groups only share the work of a frame; the arg-min is combined in code order with
// a strict '<', i.e. the first minimum wins as in torch.argmin.
template <int NC, bool TIE>
__global__ void __launch_bounds__(256) vq_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                 float* __restrict__ q, int* __restrict__ idx_out,
                                                 float* __restrict__ dist_out, int D, int T, int n_codes,
                                                 const float* __restrict__ pair_dist, float tie_scale, int* __restrict__ tie_count) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [D][NC] codebook^T, then [NC] norms, then [4][64] (d, idx, d2, idx2)
  constexpr int G = 4, CPG = NC / G;
  float* et = lds;
  float* ee = lds + (size_t)D * NC;
  float* bd = ee + NC;               // [G][64] best distance of the group
  int* bi = (int*)(bd + G * 64);     // [G][64] its code
  float* bd2 = (float*)(bi + G * 64);  // [G][64] runner-up of the group (TIE only)
  int* bi2 = (int*)(bd2 + G * 64);
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  // the codebook [code][d] is read as it lies in memory (consecutive threads, consecutive words: the transposed gather of rounds 1-5 —
  // thread i fetching cb[(i % NC) * D + i / NC] — cost one latency round trip per 4-byte load, 48 of them in sequence: most of this
  // kernel's 93 us), eight loads in flight per thread, and transposed by the LDS stores
  const int total = n_codes * D;
  for (int base = threadIdx.x; base < total; base += 256 * 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + 256 * k;
      v[k] = i < total ? cb[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + 256 * k;
      if (i < total) {
        const int e = i / D, d = i - e * D;
        et[d * NC + e] = v[k];
      }
    }
  }
  for (int i = threadIdx.x; i < D * (NC - n_codes); i += 256) {      // padding codes: zeros
    const int d = i / (NC - n_codes), e = n_codes + (i - d * (NC - n_codes));
    et[d * NC + e] = 0.f;
  }
  __syncthreads();
  if (threadIdx.x < NC) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) {
      const float v = et[d * NC + threadIdx.x];
      s += v * v;
    }
    ee[threadIdx.x] = s;
  }
  __syncthreads();
  const bool live = t < T;
  const float* zb = z + (size_t)b * D * T + (live ? t : 0);
  float dot[CPG];
#pragma unroll
  for (int e = 0; e < CPG; ++e) dot[e] = 0.f;
  float xx = 0.f;
  const int e0 = grp * CPG;
  // per accumulator the same d-ordered chain as ever; the frame's values are fetched eight rows ahead and the codes of a group four
  // at a time (16-byte LDS broadcasts: e0 and NC are multiples of 4)
  static_assert(CPG % 4 == 0 && NC % 4 == 0, "four codes per LDS read");
  int d0 = 0;
  for (; d0 + 8 <= D; d0 += 8) {
    float xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = zb[(size_t)(d0 + k) * T];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float x = xv[k];
      xx += x * x;
      const float4* er = (const float4*)(et + (d0 + k) * NC + e0);
#pragma unroll
      for (int e4 = 0; e4 < CPG / 4; ++e4) {
        const float4 c = er[e4];
        dot[4 * e4 + 0] = fmaf(x, c.x, dot[4 * e4 + 0]);
        dot[4 * e4 + 1] = fmaf(x, c.y, dot[4 * e4 + 1]);
        dot[4 * e4 + 2] = fmaf(x, c.z, dot[4 * e4 + 2]);
        dot[4 * e4 + 3] = fmaf(x, c.w, dot[4 * e4 + 3]);
      }
    }
  }
  for (int d = d0; d < D; ++d) {
    const float x = zb[(size_t)d * T];
    xx += x * x;
    const float* er = et + d * NC + e0;
#pragma unroll
    for (int e = 0; e < CPG; ++e) dot[e] = fmaf(x, er[e], dot[e]);
  }
  int best = -1, second = -1;
  float bestd = 0.f, secondd = 0.f;
#pragma unroll
  for (int e = 0; e < CPG; ++e) {
    if (e0 + e < n_codes) {
      const float dd = (xx + ee[e0 + e]) - 2.f * dot[e];
      if (dist_out && live) dist_out[((size_t)b * T + t) * n_codes + e0 + e] = dd;
      if (best < 0 || dd < bestd) {
        if (TIE) second = best, secondd = bestd;
        bestd = dd;
        best = e0 + e;
      } else if (TIE && (second < 0 || dd < secondd)) {
        second = e0 + e, secondd = dd;
      }
    }
  }
  bd[grp * 64 + lane] = bestd;
  bi[grp * 64 + lane] = best;
  if (TIE) bd2[grp * 64 + lane] = secondd, bi2[grp * 64 + lane] = second;
  __syncthreads();
  best = bi[lane];
  bestd = bd[lane];
  if (TIE) second = bi2[lane], secondd = bd2[lane];
#pragma unroll
  for (int g = 1; g < G; ++g) {
    const int cand = bi[g * 64 + lane];
    const float cd = bd[g * 64 + lane];
    if (cand >= 0 && cd < bestd) {
      if (TIE) {          // the displaced best, or the group's own runner-up, is the runner-up so far
        const int c2 = bi2[g * 64 + lane];
        const float d2 = bd2[g * 64 + lane];
        if (c2 >= 0 && d2 < bestd) second = c2, secondd = d2;
        else second = best, secondd = bestd;
      }
      bestd = cd;
      best = cand;
    } else if (TIE && cand >= 0 && (second < 0 || cd < secondd)) {
      second = cand, secondd = cd;
    }
  }
  if (TIE && live && grp == 0 && second >= 0) {
    // near-tie of the two best codes: their gap is inside what the arithmetic's feature error (tie_scale * |z_t| per unit of
    // code distance: asrbn.py calibrates it) can move the two distances against each other -> the utterance is decided again on the exact kernels
    const float gap = secondd - bestd;
    if (!(gap > tie_scale * sqrtf(xx) * pair_dist[best * n_codes + second])) {
      // count, first and last near-tie frame of the utterance: tie_count [3][B] (the host decides the frames in between again)
      const int B = (int)gridDim.y;
      atomicAdd(tie_count + b, 1);
      atomicMin(tie_count + B + b, t);
      atomicMax(tie_count + 2 * B + b, t);
    }
  }
  if (!live) return;
  if (grp == 0) idx_out[(size_t)b * T + t] = best;
  float* qb = q + (size_t)b * D * T + t;
  for (int d = grp; d < D; d += G) {
    const float x = zb[(size_t)d * T];
    const float e = et[d * NC + best];
    qb[(size_t)d * T] = x + (e - x);  // `inputs + (quantized - inputs)` (chain/nn.py:459)
  }
}

// ---- TDNNF layer with subsampling_factor 1.5 (chain/nn.py:267-304): the reference unfolds the FLATTENED [T*D] input
// with window D and step int(1.5*D), so window k starts at frame floor(1.5k) and, for odd k, half a frame in:
// channels D/2..D-1 of that frame followed by channels 0..D/2-1 of the next.  The bypass (add_padd) adds frames
// 0, 1, 3, 4, 6, 7, ... (the first int(T/1.5) of them), zero past that.  x [B][D][T] -> win, byp [B][D][Tq],
// Tq = (2(T-1))/3 + 1; the layer is then a 1x1 conv on `win` with `byp` as its residual.
__global__ void __launch_bounds__(256) tdnnf_unfold15_kernel(const float* __restrict__ x, float* __restrict__ win,
                                                             float* __restrict__ byp, int D, int T, int Tq) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (k >= Tq) return;
  const float* xb = x + (size_t)b * D * T;
  const int f0 = (3 * k) / 2;
  int f = f0, ch = c;
  if (k & 1) {
    if (c < D / 2) ch = c + D / 2;
    else { ch = c - D / 2; f = f0 + 1; }
  }
  const size_t o = ((size_t)b * D + c) * Tq + k;
  win[o] = xb[(size_t)ch * T + f];
  byp[o] = k < (2 * T) / 3 ? xb[(size_t)c * T + f0] : 0.f;
}

// log_softmax over the channel axis of x [B][C][T], in place (F.log_softmax(xent_out, dim=2) of the reference's
// [N, T, C] output, tdnnf_vq.py:283).  A block owns 16 consecutive frames; its 16 channel lanes per frame each
// keep a running (max, sum of exp) over every 16th channel, merged through LDS; a second pass subtracts.
__global__ void __launch_bounds__(256) log_softmax_channels_kernel(float* __restrict__ x, int C, int T) {
  __shared__ float sm[16][17], ss[16][17];
  const int ft = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + ft;
  const int b = blockIdx.y;
  float* xb = x + (size_t)b * C * T + t;
  float mx = -INFINITY, s = 0.f;
  if (t < T) {
    for (int c = cl; c < C; c += 16) {
      const float v = xb[(size_t)c * T];
      if (v > mx) { s = s * expf(mx - v) + 1.f; mx = v; }
      else s += expf(v - mx);
    }
  }
  sm[cl][ft] = mx;
  ss[cl][ft] = s;
  __syncthreads();
  float m = -INFINITY;
  for (int k = 0; k < 16; ++k) m = fmaxf(m, sm[k][ft]);
  float tot = 0.f;
  for (int k = 0; k < 16; ++k) tot += ss[k][ft] > 0.f ? ss[k][ft] * expf(sm[k][ft] - m) : 0.f;
  const float lse = m + logf(tot);
  if (t < T)
    for (int c = cl; c < C; c += 16) xb[(size_t)c * T] -= lse;
}

// ---- F0 statistics over the voiced (non-zero) entries of the whole batch ----
__global__ void __launch_bounds__(1024) f0_stats_kernel(const float* __restrict__ f0, int n, float* __restrict__ stats) {
  __shared__ float s_a[16];
  __shared__ float s_b[16];
  __shared__ float s_mean;
  const int tid = threadIdx.x;
  float sum = 0.f, cnt = 0.f;
  for (int i = tid; i < n; i += 1024) {
    const float v = f0[i];
    if (v != 0.f) {
      sum += v;
      cnt += 1.f;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_xor(sum, off, 64);
    cnt += __shfl_xor(cnt, off, 64);
  }
  if ((tid & 63) == 0) {
    s_a[tid >> 6] = sum;
    s_b[tid >> 6] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    float S = 0.f, C = 0.f;
    for (int i = 0; i < 16; ++i) {
      S += s_a[i];
      C += s_b[i];
    }
    s_mean = S / C;  // NaN for an all-unvoiced batch, as torch's mean of an empty tensor
    s_b[0] = C;
  }
  __syncthreads();
  const float mean = s_mean;
  const float C = s_b[0];
  __syncthreads();
  float ss = 0.f;
  for (int i = tid; i < n; i += 1024) {
    const float v = f0[i];
    if (v != 0.f) {
      const float d = v - mean;
      ss += d * d;
    }
  }
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
  if ((tid & 63) == 0) s_a[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) {
    float S = 0.f;
    for (int i = 0; i < 16; ++i) S += s_a[i];
    const float var = S / (C - 1.f);  // unbiased, NaN when one voiced value (torch.var)
    stats[0] = mean;
    stats[1] = sqrtf(var + 1e-6f);
  }
}

__global__ void __launch_bounds__(256) f0_apply_kernel(float* __restrict__ f0, int n, const float* __restrict__ stats,
                                                       int quant_bins, const float* __restrict__ noise) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = f0[i];
  if (v != 0.f) {
    v = v - stats[0];
    v = v / stats[1];
  }
  if (quant_bins > 0) {
    if (v != 0.f) v = rintf(v * (float)quant_bins) / (float)quant_bins;  // torch.round: half to even
  }
  if (noise) {
    if (v != 0.f) v = v + noise[i];  // positions that are 0 after quantisation stay 0
  }
  f0[i] = v;
}

// mean reversion of a normalised F0 track (hifigan/nn.py:64-90): out = (1 - alpha) * f0 + alpha * avg, avg[t] = the
// n-tap moving average over f0[t - n/2 .. t - n/2 + n - 1] (zeros outside), summed as torch's CPU conv1d sums it:
// an FMA chain in tap order with every tap = 1/n (measured on the biquad FIR: oracle/biquad.py); the blend is
// multiply, multiply, add (three separate torch kernels in the reference)
__global__ void __launch_bounds__(256) f0_mean_reversion_kernel(const float* __restrict__ f0, float* __restrict__ out, int T,
                                                                float one_minus_alpha, float alpha, float w, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int t0 = t - n / 2;
  float acc = 0.f;
  for (int k = 0; k < n; ++k) {
    const int i = t0 + k;
    const float x = (i >= 0 && i < T) ? f0[i] : 0.f;
    acc = k == 0 ? w * x : __builtin_fmaf(w, x, acc);
  }
  const float a = one_minus_alpha * f0[t];
  const float b = alpha * acc;
  out[t] = a + b;
}

// x[b] = [ bn[b] ; nearest-interpolated f0[b] ; spk[b] (the one-hot row as f32) broadcast over T ]
__global__ void __launch_bounds__(256) assemble_kernel(const float* __restrict__ bn, const float* __restrict__ f0,
                                                       const float* __restrict__ spk, float* __restrict__ x, int C_bn,
                                                       int T, int T_f0, int n_spk) {
  const int b = blockIdx.z;
  const int c = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int C = C_bn + 1 + n_spk;
  float v;
  if (c < C_bn) {
    v = bn[((size_t)b * C_bn + c) * T + t];
  } else if (c == C_bn) {
    // F.interpolate(mode='nearest'): src = min(floor(dst * (in/out)), in-1), scale in f32
    const float scale = (float)T_f0 / (float)T;
    int s = (int)floorf((float)t * scale);
    s = s < T_f0 - 1 ? s : T_f0 - 1;
    v = f0[(size_t)b * T_f0 + s];
  } else {
    v = spk[(size_t)b * n_spk + (c - C_bn - 1)];  // nearest interpolation of a length-1 axis = broadcast
  }
  x[((size_t)b * C + c) * T + t] = v;
}


// ---- PCM16 <-> f32 (the data plane of the batch job: what torchaudio.load / torchaudio.save(bits_per_sample=16) do to the samples)
template <bool VEC>
__global__ void __launch_bounds__(256) pcm16_from_f32_kernel(const float* __restrict__ x, short* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  auto cv = [](float v) { return (short)fminf(fmaxf(rintf(v * 32768.f), -32768.f), 32767.f); };
  if (VEC && i + 3 < n) {
    const float4 v = *(const float4*)(x + i);
    const short s0 = cv(v.x), s1 = cv(v.y), s2 = cv(v.z), s3 = cv(v.w);
    *(uint2*)(y + i) = make_uint2((unsigned)(unsigned short)s0 | ((unsigned)(unsigned short)s1 << 16),
                                  (unsigned)(unsigned short)s2 | ((unsigned)(unsigned short)s3 << 16));
  } else {
    for (long long j = i; j < n && j < i + 4; ++j) y[j] = cv(x[j]);
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256) pcm16_to_f32_kernel(const short* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (VEC && i + 3 < n) {
    const uint2 v = *(const uint2*)(x + i);
    *(float4*)(y + i) = make_float4((float)(short)(v.x & 0xffffu) * (1.f / 32768.f), (float)(short)(v.x >> 16) * (1.f / 32768.f),
                                    (float)(short)(v.y & 0xffffu) * (1.f / 32768.f), (float)(short)(v.y >> 16) * (1.f / 32768.f));
  } else {
    for (long long j = i; j < n && j < i + 4; ++j) y[j] = (float)x[j] * (1.f / 32768.f);
  }
}

}  // namespace sat

using namespace sat;

static int vq_launch(const float* z, const float* codebook, float* q, int32_t* idx, float* dist, const float* pair_dist, float tie_scale,
                     int32_t* tie_count, int B, int D, int T, int n_codes, void* stream) {
  SAT_REQUIRE(z && codebook && q && idx, "vq: null pointer");
  SAT_REQUIRE(B > 0 && D > 0 && T > 0 && n_codes > 0 && n_codes <= VQ_MAX_CODES, "vq: unsupported sizes (n_codes <= %d)",
              VQ_MAX_CODES);
  dim3 grid(ceil_div(T, 64), B);
  const bool tie = tie_count != nullptr;
  const int nc = n_codes <= 48 ? 48 : 64;
  const size_t lds = ((size_t)D * nc + nc + 1024) * sizeof(float);
  SAT_REQUIRE(lds <= 160 * 1024, "vq: codebook does not fit LDS");
  auto kern = nc == 48 ? (tie ? vq_kernel<48, true> : vq_kernel<48, false>) : (tie ? vq_kernel<64, true> : vq_kernel<64, false>);
  if (lds > 64 * 1024) SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, z, codebook, q, idx, dist, D, T, n_codes, pair_dist, tie_scale, tie_count);
  SAT_LAUNCH_CHECK("vq_kernel");
  return SAT_OK;
}

extern "C" int sat_vq_argmin_gather_f32(const float* z, const float* codebook, float* q, int32_t* idx, float* dist,
                                        int B, int D, int T, int n_codes, void* stream) {
  return vq_launch(z, codebook, q, idx, dist, nullptr, 0.f, nullptr, B, D, T, n_codes, stream);
}

extern "C" int sat_vq_argmin_gather_tie_f32(const float* z, const float* codebook, float* q, int32_t* idx, float* dist,
                                            const float* pair_dist, float tie_scale, int32_t* tie_count,
                                            int B, int D, int T, int n_codes, void* stream) {
  SAT_REQUIRE(pair_dist && tie_count && tie_scale >= 0.f, "vq(tie): pair_dist [n_codes][n_codes], tie_count [3][B] and a tie_scale >= 0");
  return vq_launch(z, codebook, q, idx, dist, pair_dist, tie_scale, tie_count, B, D, T, n_codes, stream);
}

extern "C" int sat_f0_stats_f32(const float* f0, int n, float* stats, void* stream) {
  SAT_REQUIRE(f0 && stats && n > 0, "f0_stats: bad arguments");
  hipLaunchKernelGGL(f0_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, f0, n, stats);
  SAT_LAUNCH_CHECK("f0_stats_kernel");
  return SAT_OK;
}

extern "C" int sat_f0_apply_f32(float* f0, int n, const float* stats, int quant_bins, const float* noise,
                                void* stream) {
  SAT_REQUIRE(f0 && stats && n > 0 && quant_bins >= 0, "f0_apply: bad arguments");
  hipLaunchKernelGGL(f0_apply_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, f0, n, stats,
                     quant_bins, noise);
  SAT_LAUNCH_CHECK("f0_apply_kernel");
  return SAT_OK;
}

extern "C" int sat_f0_mean_reversion_f32(const float* f0, float* out, int T, float alpha, int n, void* stream) {
  SAT_REQUIRE(f0 && out && f0 != out && T > 0 && n > 0, "f0_mean_reversion: bad arguments");
  // (1 - alpha) is taken in double like the reference's Python float, then rounded to f32 as torch's scalar multiply does
  const float oma = (float)(1.0 - (double)alpha);
  const float w = 1.0f / (float)n;     // torch.ones(n) / n in f32
  hipLaunchKernelGGL(f0_mean_reversion_kernel, dim3(ceil_div(T, 256)), dim3(256), 0, (hipStream_t)stream, f0, out, T, oma,
                     alpha, w, n);
  SAT_LAUNCH_CHECK("f0_mean_reversion_kernel");
  return SAT_OK;
}

extern "C" int sat_assemble_input_f32(const float* bn, const float* f0, const float* spk, float* x, int B,
                                      int C_bn, int T, int T_f0, int n_spk, void* stream) {
  SAT_REQUIRE(bn && f0 && (spk || n_spk == 0) && x, "assemble_input: null pointer");
  SAT_REQUIRE(B > 0 && C_bn > 0 && T > 0 && T_f0 > 0 && n_spk >= 0, "assemble_input: bad sizes");
  dim3 grid(ceil_div(T, 256), C_bn + 1 + n_spk, B);
  hipLaunchKernelGGL(assemble_kernel, grid, dim3(256), 0, (hipStream_t)stream, bn, f0, spk, x, C_bn, T, T_f0, n_spk);
  SAT_LAUNCH_CHECK("assemble_kernel");
  return SAT_OK;
}

extern "C" int sat_tdnnf_unfold15_f32(const float* x, float* win, float* byp, int B, int D, int T, void* stream) {
  SAT_REQUIRE(x && win && byp && B > 0 && D > 0 && D % 2 == 0 && T > 0 && B < 65536 && D < 65536, "tdnnf_unfold15: bad arguments");
  const int Tq = (2 * (T - 1)) / 3 + 1;
  hipLaunchKernelGGL(tdnnf_unfold15_kernel, dim3(ceil_div(Tq, 256), D, B), dim3(256), 0, (hipStream_t)stream, x, win, byp, D, T, Tq);
  SAT_LAUNCH_CHECK("tdnnf_unfold15_kernel");
  return SAT_OK;
}

extern "C" int sat_log_softmax_channels_f32(float* x, int B, int C, int T, void* stream) {
  SAT_REQUIRE(x && B > 0 && C > 0 && T > 0, "log_softmax_channels: bad arguments");
  hipLaunchKernelGGL(log_softmax_channels_kernel, dim3(ceil_div(T, 16), B), dim3(256), 0, (hipStream_t)stream, x, C, T);
  SAT_LAUNCH_CHECK("log_softmax_channels_kernel");
  return SAT_OK;
}

extern "C" int sat_pcm16_from_f32(const float* x, int16_t* y, long long n, void* stream) {
  SAT_REQUIRE(x && y && n > 0 && n < (1ll << 40), "pcm16_from_f32: bad arguments");
  const bool vec = ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0);
  const dim3 grid((unsigned)((n + 1023) / 1024));
  if (vec) hipLaunchKernelGGL(pcm16_from_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (short*)y, n);
  else hipLaunchKernelGGL(pcm16_from_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (short*)y, n);
  SAT_LAUNCH_CHECK("pcm16_from_f32_kernel");
  return SAT_OK;
}

extern "C" int sat_pcm16_to_f32(const int16_t* x, float* y, long long n, void* stream) {
  SAT_REQUIRE(x && y && n > 0 && n < (1ll << 40), "pcm16_to_f32: bad arguments");
  const bool vec = ((uintptr_t)y % 16 == 0) && ((uintptr_t)x % 8 == 0);
  const dim3 grid((unsigned)((n + 1023) / 1024));
  if (vec) hipLaunchKernelGGL(pcm16_to_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const short*)x, y, n);
  else hipLaunchKernelGGL(pcm16_to_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const short*)x, y, n);
  SAT_LAUNCH_CHECK("pcm16_to_f32_kernel");
  return SAT_OK;
}
