// Fused conv1d as an implicit GEMM on the gfx950 f32 matrix cores.
//
//   GEMM view:  M = output rows (C_out*up, per group), N = time positions, K = C_in/g * ksize
//   MFMA:       v_mfma_f32_32x32x2_f32 — A[i][k] from lane (i = l&31, k = l>>5),
//               B[k][j] from lane (k = l>>5, j = l&31), D: col = l&31, row = (r&3)+8(r>>2)+4(l>>5)
//               (cdna_hip_programming.md §3).  k = a PAIR of input channels at one tap.
//   B operand:  the input tile [16 channels][tile width + halo] staged in LDS; a fragment read is
//               32 consecutive floats of one channel row per half-wave -> conflict-free ds_read_b32.
//   A operand:  packed weights w[g][ci][tap][co] (co fastest) read straight from L2: one coalesced
//               128-B row per half-wave; all blocks share the same few MB of weights.
//   Epilogue:   bias, residual / bypass, folded BatchNorm, ReLU, MRF accumulation, polyphase store.
//
// Replaces (reference): torch Conv1d/ConvTranspose1d calls of hifigan/archi.py:77-91 and
// hifigan/nn.py:179-186; unfold+matmul/addmm of chain/nn.py:267-292 + BatchNorm/ReLU :338-347.
#include "common.h"

namespace sat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CI_CHUNK = 16;  // input channels staged per K-chunk (8 MFMA k-pairs)

struct ConvArgs {
  const float* x;
  const float* w;
  float* y;
  const float* bias;
  const float* res;
  const float* ch_scale;
  const float* ch_shift;
  long long x_bs, x_cs, y_bs, y_cs, r_bs, r_cs;
  long long w_gs;  // packed weight elements per group
  int cin_g, T_in, rows_g, cout_g, T_q;
  int ksize, dil, stride, pad_left, up;
  int cin_pad, co_pad, xw, co_tiles_g;
  int in_lrelu, relu, accum;
  float in_slope, accum_div, res_scale;
  int res_toff, res_tstride;
};

template <int MT, int NT, int WM, int WN, int KS, bool STRIDE1>
__global__ void __launch_bounds__(256, 2) conv1d_mfma_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CO_B = 32 * MT * WM;
  constexpr int T_B = 32 * NT * WN;
  static_assert(WM * WN == 4, "4 waves per block");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = wave / WM;
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  const int b = blockIdx.z;
  const int g = blockIdx.y / p.co_tiles_g;
  const int cot = blockIdx.y - g * p.co_tiles_g;
  const int co_w = cot * CO_B + wm * (32 * MT);  // first row of this wave inside the group
  const int q_b = blockIdx.x * T_B;              // first output position of the block
  const int q_w = q_b + wn * (32 * NT);
  const bool wave_active = co_w < p.rows_g;  // co_pad is a multiple of 64 >= rows_g, so the wave's rows stay inside the packed weights

  const int ks = KS > 0 ? KS : p.ksize;
  const int st = STRIDE1 ? 1 : p.stride;
  const int XW = p.xw;
  const int xi0 = q_b * st - p.pad_left;

  const float* __restrict__ xg = p.x + (long long)b * p.x_bs + (long long)(g * p.cin_g) * p.x_cs;
  const float* __restrict__ wg = p.w + (long long)g * p.w_gs + co_w + l31;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const long long w_ci_stride = (long long)ks * p.co_pad;  // elements between input channels

  for (int c0 = 0; c0 < p.cin_pad; c0 += CI_CHUNK) {
    __syncthreads();
    // ---- stage the input tile: 16 channels x XW columns, pre-activation fused ----
#pragma unroll
    for (int rr = 0; rr < CI_CHUNK / 4; ++rr) {
      const int r = wave + rr * 4;
      const int ci = c0 + r;
      const bool cok = ci < p.cin_g;
      const float* __restrict__ src = xg + (long long)ci * p.x_cs;
      float* dst = lds + r * XW;
      for (int col = lane; col < XW; col += 64) {
        const int xi = xi0 + col;
        float v = 0.f;
        if (cok && xi >= 0 && xi < p.T_in) {
          v = src[xi];
          if (p.in_lrelu) v = v > 0.f ? v : v * p.in_slope;
        }
        dst[col] = v;
      }
    }
    __syncthreads();
    if (!wave_active) continue;

    const float* __restrict__ wc = wg + (long long)(c0 + lh) * w_ci_stride;
    const float* xrow = lds + lh * XW + (wn * (32 * NT) + l31) * st;

    if constexpr (KS > 0) {
      float a_cur[KS][MT];
#pragma unroll
      for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[t][m] = wc[(long long)t * p.co_pad + m * 32];
#pragma unroll
      for (int pr = 0; pr < CI_CHUNK / 2; ++pr) {
        float a_nxt[KS][MT];
        if (pr + 1 < CI_CHUNK / 2) {
          const float* __restrict__ wn_ = wc + (long long)(2 * (pr + 1)) * w_ci_stride;
#pragma unroll
          for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) a_nxt[t][m] = wn_[(long long)t * p.co_pad + m * 32];
        }
        const float* xp = xrow + (2 * pr) * XW;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          float bf[NT];
          const float* xt = xp + t * p.dil;
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = xt[n * 32 * st];
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t][m], bf[n], acc[m][n], 0, 0, 0);
        }
        if (pr + 1 < CI_CHUNK / 2) {
#pragma unroll
          for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) a_cur[t][m] = a_nxt[t][m];
        }
      }
    } else {
      // runtime tap count (k = 2, 10, 128 ...): taps looped, one k-pair of channels at a time
#pragma unroll 1
      for (int pr = 0; pr < CI_CHUNK / 2; ++pr) {
        const float* __restrict__ wp = wc + (long long)(2 * pr) * w_ci_stride;
        const float* xp = xrow + (2 * pr) * XW;
        float a_nx[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a_nx[m] = wp[m * 32];
#pragma unroll 2
        for (int t = 0; t < ks; ++t) {
          float a[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) a[m] = a_nx[m];
          if (t + 1 < ks) {
#pragma unroll
            for (int m = 0; m < MT; ++m) a_nx[m] = wp[(long long)(t + 1) * p.co_pad + m * 32];
          }
          float bf[NT];
          const float* xt = xp + t * p.dil;
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = xt[n * 32 * st];
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bf[n], acc[m][n], 0, 0, 0);
        }
      }
    }
  }

  if (!wave_active) return;

  // ---- epilogue ----
  const int up = p.up;
  const int T_out = p.T_q * up;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = co_w + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;  // row inside group
      if (row >= p.rows_g) continue;
      int co_l, ph;
      if (up == 1) {
        co_l = row;
        ph = 0;
      } else {
        co_l = row / up;
        ph = row - co_l * up;
      }
      const int co = g * p.cout_g + co_l;
      const float bias = p.bias ? p.bias[co] : 0.f;
      float sc = 1.f, sh = 0.f;
      if (p.ch_scale) {
        sc = p.ch_scale[co];
        sh = p.ch_shift[co];
      }
      float* __restrict__ yrow = p.y + (long long)b * p.y_bs + (long long)co * p.y_cs;
      const float* __restrict__ rrow =
          p.res ? p.res + (long long)b * p.r_bs + (long long)co * p.r_cs : nullptr;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int q = q_w + n * 32 + l31;
        if (q >= p.T_q) continue;
        const int t = q * up + ph;
        float v = acc[m][n][r] + bias;
        if (rrow) v += p.res_scale * rrow[(long long)t * p.res_tstride + p.res_toff];
        if (p.ch_scale) v = v * sc + sh;
        if (p.relu) v = v > 0.f ? v : 0.f;
        if (p.accum) v = yrow[t] + v;
        if (p.accum_div != 0.f) v = v / p.accum_div;
        yrow[t] = v;
      }
    }
  }
  (void)T_out;
}

template <int MT, int NT, int WM, int WN, int KS, bool S1>
static int launch_cfg(const ConvArgs& a, int B, int groups, hipStream_t s) {
  constexpr int CO_B = 32 * MT * WM;
  constexpr int T_B = 32 * NT * WN;
  ConvArgs p = a;
  p.xw = (T_B - 1) * p.stride + (p.ksize - 1) * p.dil + 1;
  p.co_tiles_g = ceil_div(p.rows_g, CO_B);
  const size_t lds_bytes = (size_t)CI_CHUNK * p.xw * sizeof(float);
  auto kern = conv1d_mfma_kernel<MT, NT, WM, WN, KS, S1>;
  if (lds_bytes > 160 * 1024) {
    set_error("conv1d: input tile of %zu bytes does not fit LDS (stride %d, ksize %d, dilation %d)",
              lds_bytes, p.stride, p.ksize, p.dil);
    return SAT_ERR_INVALID;
  }
  if (lds_bytes > 64 * 1024) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes));
  }
  dim3 grid(ceil_div(p.T_q, T_B), p.co_tiles_g * groups, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, p);
  SAT_LAUNCH_CHECK("conv1d_mfma_kernel");
  return SAT_OK;
}

template <int MT, int NT, int WM, int WN>
static int launch_ks(const ConvArgs& a, int B, int groups, hipStream_t s) {
  if (a.stride == 1) {
    switch (a.ksize) {
      case 1: return launch_cfg<MT, NT, WM, WN, 1, true>(a, B, groups, s);
      case 3: return launch_cfg<MT, NT, WM, WN, 3, true>(a, B, groups, s);
      case 7: return launch_cfg<MT, NT, WM, WN, 7, true>(a, B, groups, s);
      case 11: return launch_cfg<MT, NT, WM, WN, 11, true>(a, B, groups, s);
      default: return launch_cfg<MT, NT, WM, WN, 0, true>(a, B, groups, s);
    }
  }
  return launch_cfg<MT, NT, WM, WN, 0, false>(a, B, groups, s);
}

}  // namespace sat

using namespace sat;

extern "C" int sat_conv1d_packed_dims(int C_in, int C_out, int up, int groups, int* cin_pad,
                                      int* co_pad) {
  SAT_REQUIRE(C_in > 0 && C_out > 0 && up > 0 && groups > 0, "conv1d_packed_dims: bad sizes");
  SAT_REQUIRE(C_in % groups == 0 && C_out % groups == 0, "conv1d_packed_dims: groups must divide channels");
  if (cin_pad) *cin_pad = round_up(C_in / groups, CI_CHUNK);
  if (co_pad) *co_pad = round_up(C_out / groups * up, 64);
  return SAT_OK;
}

extern "C" int sat_conv1d_f32(const sat_conv1d_desc* d, const float* x, const float* w_packed,
                              float* y, void* stream) {
  SAT_REQUIRE(d && x && w_packed && y, "conv1d: null pointer");
  SAT_REQUIRE(d->B > 0 && d->C_in > 0 && d->C_out > 0 && d->T_in > 0 && d->T_q > 0, "conv1d: empty shape");
  SAT_REQUIRE(d->ksize >= 1 && d->dilation >= 1 && d->stride >= 1 && d->up >= 1 && d->groups >= 1,
              "conv1d: bad ksize/dilation/stride/up/groups");
  SAT_REQUIRE(d->C_in % d->groups == 0 && d->C_out % d->groups == 0, "conv1d: groups must divide channels");
  SAT_REQUIRE(d->up == 1 || d->stride == 1, "conv1d: polyphase output requires stride 1");
  ConvArgs a{};
  a.x = x;
  a.w = w_packed;
  a.y = y;
  a.bias = d->bias;
  a.res = d->res;
  a.ch_scale = d->ch_scale;
  a.ch_shift = d->ch_shift;
  SAT_REQUIRE((d->ch_scale == nullptr) == (d->ch_shift == nullptr), "conv1d: ch_scale and ch_shift go together");
  a.x_bs = d->x_bstride;
  a.x_cs = d->x_cstride;
  a.y_bs = d->y_bstride;
  a.y_cs = d->y_cstride;
  a.r_bs = d->res_bstride;
  a.r_cs = d->res_cstride;
  a.cin_g = d->C_in / d->groups;
  a.cout_g = d->C_out / d->groups;
  a.rows_g = a.cout_g * d->up;
  a.T_in = d->T_in;
  a.T_q = d->T_q;
  a.ksize = d->ksize;
  a.dil = d->dilation;
  a.stride = d->stride;
  a.pad_left = d->pad_left;
  a.up = d->up;
  a.cin_pad = round_up(a.cin_g, CI_CHUNK);
  a.co_pad = round_up(a.rows_g, 64);
  a.w_gs = (long long)a.cin_pad * a.ksize * a.co_pad;
  a.in_lrelu = d->in_lrelu;
  a.in_slope = d->in_slope;
  a.relu = d->relu;
  a.accum = d->accum;
  a.accum_div = d->accum_div;
  a.res_scale = d->res_scale;
  a.res_toff = d->res_toff;
  a.res_tstride = d->res_tstride > 0 ? d->res_tstride : 1;
  hipStream_t s = (hipStream_t)stream;
  // tile shape by output rows per group: wide-in-time tiles for thin layers
  if (a.rows_g > 64) return launch_ks<2, 2, 2, 2>(a, d->B, d->groups, s);   // 128 rows x 128 positions
  if (a.rows_g > 32) return launch_ks<2, 2, 1, 4>(a, d->B, d->groups, s);   //  64 rows x 256 positions
  return launch_ks<1, 4, 1, 4>(a, d->B, d->groups, s);                       //  32 rows x 512 positions
}
