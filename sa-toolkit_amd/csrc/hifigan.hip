// HiFi-GAN generator forward on gfx950: orchestration over the fused conv1d kernel plus the
// streaming output stage.  Reference: CoreHifiGan.forward_resnet, satools/satools/hifigan/
// archi.py:77-91; ResBlock1.forward, satools/satools/hifigan/nn.py:179-186.
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace sat {

bool convring_wanted(int rows_g, int T_q, int B);      // conv_ring16.hip: would a k-tap conv of this shape run on the LDS-DMA ring?

// polyphase view of ConvTranspose1d(k, stride u, padding pad): output t = q*u + r reads input
// positions s = q + delta with tap j = r + pad - u*delta, 0 <= j < k.  Returns the tap window
// [dmin, dmax] over all phases.
static void phase_window(int k, int u, int pad, int* dmin, int* dmax) {
  int lo = 1 << 30, hi = -(1 << 30);
  for (int r = 0; r < u; ++r) {
    for (int d = -k; d <= k; ++d) {
      const int j = r + pad - u * d;
      if (j >= 0 && j < k) {
        if (d < lo) lo = d;
        if (d > hi) hi = d;
      }
    }
  }
  *dmin = lo;
  *dmax = hi;
}

// ---- output stage: leaky_relu(0.01) -> ReflectionPad1d((1,0)) -> Conv1d(C,1,7,pad=3) -> tanh ----
// HBM-streaming kernel: each block produces 1024 output samples of one utterance from a
// [C][1024+6] LDS tile.  Padded signal p[i] (i in [0,T]) = lrelu(x[i-1]) for i>=1, p[0] = lrelu(x[1]).
static int g_convpost_quad = 1;
void convpost_set_quad(int v) { g_convpost_quad = v != 0; }
constexpr int POST_TILE = 1024;
constexpr int POST_MAXC = 64;

template <int C, bool QUAD = false>
__global__ void __launch_bounds__(256) convpost_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ y, int C_rt, int T) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int W = POST_TILE + 6;
  constexpr int NCOL = (W + 255) / 256;
  const int Cn = C > 0 ? C : C_rt;
  float* wl = lds + (size_t)Cn * W;  // weights [C][7]
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * POST_TILE;  // first output index of the tile
  const int To = T + 1;
  for (int i = threadIdx.x; i < Cn * 7; i += 256) wl[i] = w[i];
  if constexpr (C > 0) {
    // every global load of the tile is issued before the first LDS store (80 in flight per lane at C = 16);
    // rows behind one buffer descriptor, columns outside the padded signal read as 0 by the range check
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (size_t)b * C * T), 0, (unsigned)((size_t)C * T * 4), 0x00020000);
    float v[C][NCOL];
#pragma unroll
    for (int k = 0; k < NCOL; ++k) {
      const int col = threadIdx.x + 256 * k;
      const int i = t0 - 3 + col;          // index into the reflection-padded signal
      const int xi = i == 0 ? 1 : i - 1;
      const unsigned voff = (col < W && i >= 0 && i < To) ? (unsigned)(xi * 4) : 0x80000000u;
#pragma unroll
      for (int c = 0; c < C; ++c)
        v[c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, voff, c * T * 4, 0));
    }
#pragma unroll
    for (int k = 0; k < NCOL; ++k) {
      const int col = threadIdx.x + 256 * k;
      if (col < W) {
#pragma unroll
        for (int c = 0; c < C; ++c) lds[c * W + col] = v[c][k] > 0.f ? v[c][k] : v[c][k] * 0.01f;
      }
    }
  } else {
    const float* xb = x + (size_t)b * Cn * T;
    for (int c = 0; c < Cn; ++c) {
      const float* xr = xb + (size_t)c * T;
      for (int col = threadIdx.x; col < W; col += 256) {
        const int i = t0 - 3 + col;
        float v = 0.f;
        if (i >= 0 && i < To) {
          const int xi = i == 0 ? 1 : i - 1;
          v = xr[xi];
          v = v > 0.f ? v : v * 0.01f;
        }
        lds[c * W + col] = v;
      }
    }
  }
  __syncthreads();
  const float bv = bias[0];
  if constexpr (QUAD && C > 0 && W % 4 == 2) {
    // four CONSECUTIVE outputs per lane: their ten columns of a channel lie inside three aligned 16-byte LDS reads (rows are W = 1030
    // floats apart: even rows are aligned at column lt, odd rows two columns earlier), the weights come through the scalar unit; per
    // output the products in the order (channel, tap) of the loop below — the same bits
    const int lt = 4 * threadIdx.x;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int sh = (c & 1) * 2;                  // floats between the aligned window and column lt
      const float4* rp = reinterpret_cast<const float4*>(lds + c * W + lt - sh);
      float pv[12];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4 f = rp[q];
        pv[4 * q] = f.x; pv[4 * q + 1] = f.y; pv[4 * q + 2] = f.z; pv[4 * q + 3] = f.w;
      }
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[o] = fmaf(w[c * 7 + j], pv[sh + o + j], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int t = t0 + lt + o;
      if (t < To) y[(size_t)b * To + t] = tanhf(acc[o] + bv);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < POST_TILE / 256; ++k) {
    const int lt = threadIdx.x + k * 256;
    const int t = t0 + lt;
    if (t >= To) continue;
    float acc = 0.f;
#pragma unroll 4
    for (int c = 0; c < Cn; ++c) {
      const float* row = lds + c * W + lt;
      const float* wr = wl + c * 7;
#pragma unroll
      for (int j = 0; j < 7; ++j) acc = fmaf(wr[j], row[j], acc);
    }
    y[(size_t)b * To + t] = tanhf(acc + bv);
  }
}

// planes S[b][c/16][4][t][16 B]: units 0, 1 of a chunk are the hi halves.  out[0] += values past `limit` or non-finite, out[1] = max bits of |hi|
__global__ void __launch_bounds__(256) planes_range_kernel(const uint4* __restrict__ x, long long n_rows, int T, float limit,
                                                           unsigned long long* __restrict__ out) {
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  unsigned long long cnt = 0;
  float mx = 0.f;
  const long long total = n_rows * T;          // rows = (utterance, chunk, hi unit)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / T;
    const int t = (int)(i - row * T);
    const long long chunk = row >> 1, unit = row & 1;
    const h8v v = __builtin_bit_cast(h8v, x[(chunk * 4 + unit) * T + t]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float a = fabsf((float)v[k]);
      if (!(a <= limit)) ++cnt;
      if (a == a && a > mx) mx = a;
    }
  }
  __shared__ unsigned long long sc[256];
  __shared__ float sm[256];
  sc[threadIdx.x] = cnt, sm[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sc[threadIdx.x] += sc[threadIdx.x + s];
      sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (sc[0]) atomicAdd(out, sc[0]);
    atomicMax(out + 1, (unsigned long long)__float_as_uint(sm[0]));
  }
}

constexpr float SAT_RANGE_LIMIT = 57344.f;

static int planes_range_probe(const void* planes, int B, int C, int T, unsigned long long* out, void* stream) {
  const long long rows = (long long)B * (C / 16) * 2;
  const int grid = (int)std::min<long long>(2048, (rows * T + 255) / 256);
  hipLaunchKernelGGL(planes_range_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)planes, rows, T, SAT_RANGE_LIMIT, out);
  SAT_LAUNCH_CHECK("planes_range_kernel");
  return SAT_OK;
}

}  // namespace sat

using namespace sat;

// side streams of one caller stream: the resblock branches of a stage (kernel sizes 3 / 7 / 11) are
// independent until the MRF sum, so they run on separate HIP streams and fill each other's tails
// (a C=256 conv is 640 blocks on 512 block slots); events keep the order of the sum
struct hifigan_side {
  hipStream_t s[2] = {nullptr, nullptr};
  hipEvent_t fork = nullptr, acc[3] = {nullptr, nullptr, nullptr};
};

struct sat_hifigan {
  std::mutex mu;
  std::map<void*, hifigan_side> sides;   // keyed by the caller's stream
  int branch_streams = 0;   // leading stages whose three resblock branches run on separate streams (single-job latency)
  int in_ch = 0, c0 = 0;
  std::vector<int> up_rates, up_kernels, rb_kernels, rb_dil;
  struct Conv {
    const void* w = nullptr;
    const float* bias = nullptr;
    int mode = SAT_CONV_F32;
    float descale = 1.f;      // power-of-two descale of the packed weights (sat_conv1d_desc.w_descale)
    const void* w8 = nullptr; // second packing for SAT_CONV_F16F8R (same layer scale), or null
  };
  std::vector<Conv> convs;
  int fuse_pairs = 1;
  int fuse_pair64 = 3;       // bit mask: the 3- (1) / 7- (2) / 11-tap (4) ResBlock steps of the C = 64 stage as one launch each (pair64.hip);
                             // 11 taps measured slower fused (its recomputed halo and short blocks cost more than the traffic saved)
  int mrf_exact = 0;         // fused MRF block: 0 = residuals of steps 2 / 3 rebuilt from the 22-bit planes like the launch-by-launch path (bit-identical
                             // to it, and measured 5 % faster: fewer live registers); 1 = kept in f32 registers
  int ups_ring = 0;          // the stride-4 upsamplers' packed rows are grouped by phase (sat_conv1d_desc.up_grouped): set by the packer's side
  int ups2 = 1;              // the thin upsamplers (C_in = 64, 32; k 4, stride 2) on the streaming kernel of ups2.hip
  int f8_stages = 0;         // bit i: the ResBlock convs of stage i as SAT_CONV_F16F8R (8-bit cross terms on the ring kernel) where their second packing is
                             // installed and the ring kernel serves the batch
  int multi_branch = 1;      // thick stages (C > 64): the i-th conv of all MRF branches as one sat_conv1d_multi_f32 call (one launch where the ring kernel serves them)
  int fuse_mrf = 1;          // a whole MRF block (all branches, all steps, the mean) as one launch where mrf.hip supports the stage (C = 16)
  int split_acts = 1;
  int planes_residual = 1;
  int skip_dead_sum = 1;     // the f32 MRF mean of a stage that is not the last is read by nobody (the next upsampler takes its planes): the last
                             // branch's launch does not store it (sat_conv1d_desc.accum_no_store) — 0.45 GB of writes per forward of 32 x 5 s
  int force_f8 = 0;          // SAT_CONV_F16F8R stages at EVERY batch size (a calibration batch is too small for the ring kernel's default dispatch:
                             // check_precision sets it on its own handle's forward instead of touching the process-wide "convring" option)
  mutable std::atomic<int> last_f8_stages{0};   // bit i: stage i of the LAST forward ran its ResBlock convs with 8-bit cross terms
  unsigned long long* range_probe = nullptr;    // device words [n_ups][2] (caller's): per stage, plane values past the 8-bit operand's range | max |hi| bits
  int n_ups() const { return (int)up_rates.size(); }
  int n_rbk() const { return (int)rb_kernels.size(); }
  int id_up(int i) const { return 1 + i; }
  int id_rb(int stage, int j, int pair, int which) const {
    return 1 + n_ups() + ((stage * n_rbk() + j) * 3 + pair) * 2 + which;
  }
  int id_post() const { return 1 + n_ups() + n_ups() * n_rbk() * 6; }
};

extern "C" int sat_convtranspose_phase_dims(int k, int u, int pad, int* ksize, int* pad_left) {
  SAT_REQUIRE(k > 0 && u > 0 && pad >= 0, "convtranspose_phase_dims: bad arguments");
  int lo, hi;
  phase_window(k, u, pad, &lo, &hi);
  if (ksize) *ksize = hi - lo + 1;
  if (pad_left) *pad_left = -lo;
  return SAT_OK;
}

extern "C" int sat_hifigan_create(sat_hifigan** out, int in_channels, int initial_channels, int n_ups,
                                  const int* up_rates, const int* up_kernels, int n_rb_kernels,
                                  const int* rb_kernels, const int* rb_dilations) {
  SAT_REQUIRE(out && up_rates && up_kernels && rb_kernels && rb_dilations, "hifigan_create: null pointer");
  SAT_REQUIRE(in_channels > 0 && initial_channels > 0 && n_ups > 0 && n_rb_kernels > 0, "hifigan_create: bad sizes");
  SAT_REQUIRE((initial_channels >> n_ups) >= 1 && (initial_channels >> n_ups) <= POST_MAXC,
              "hifigan_create: unsupported channel progression");
  auto* h = new sat_hifigan();
  h->in_ch = in_channels;
  h->c0 = initial_channels;
  h->up_rates.assign(up_rates, up_rates + n_ups);
  h->up_kernels.assign(up_kernels, up_kernels + n_ups);
  h->rb_kernels.assign(rb_kernels, rb_kernels + n_rb_kernels);
  h->rb_dil.assign(rb_dilations, rb_dilations + 3 * n_rb_kernels);
  for (int i = 0; i < n_ups; ++i) {
    if ((up_kernels[i] - up_rates[i]) % 2 != 0 || up_kernels[i] < up_rates[i]) {
      delete h;
      set_error("hifigan_create: upsample kernel %d / rate %d not supported", up_kernels[i], up_rates[i]);
      return SAT_ERR_INVALID;
    }
  }
  h->convs.resize(h->id_post() + 1);
  *out = h;
  return SAT_OK;
}

extern "C" int sat_hifigan_num_convs(const sat_hifigan* h) { return h ? (int)h->convs.size() : SAT_ERR_INVALID; }

extern "C" int sat_hifigan_set_conv(sat_hifigan* h, int conv_id, const void* w_packed, const float* bias, int mode) {
  SAT_REQUIRE(h && conv_id >= 0 && conv_id < (int)h->convs.size() && w_packed && bias, "hifigan_set_conv: bad arguments");
  SAT_REQUIRE(mode == SAT_CONV_F32 || mode == SAT_CONV_F16X3 || mode == SAT_CONV_F16F8, "hifigan_set_conv: unknown mode");
  SAT_REQUIRE(conv_id != h->id_post() || mode == SAT_CONV_F32, "hifigan_set_conv: the output stage is f32 only");
  h->convs[conv_id].w = w_packed;
  h->convs[conv_id].bias = bias;
  h->convs[conv_id].mode = mode;
  h->convs[conv_id].descale = 1.f;
  h->convs[conv_id].w8 = nullptr;
  return SAT_OK;
}

extern "C" int sat_hifigan_set_conv_f8r(sat_hifigan* h, int conv_id, const void* w_packed_f8r) {
  SAT_REQUIRE(h && conv_id > h->n_ups() && conv_id < h->id_post(), "hifigan_set_conv_f8r: only the ResBlock convs carry a SAT_CONV_F16F8R packing");
  SAT_REQUIRE(!w_packed_f8r || h->convs[conv_id].mode == SAT_CONV_F16X3, "hifigan_set_conv_f8r: next to a SAT_CONV_F16X3 packing only");
  h->convs[conv_id].w8 = w_packed_f8r;
  return SAT_OK;
}

extern "C" int sat_hifigan_set_conv_descale(sat_hifigan* h, int conv_id, float w_descale) {
  SAT_REQUIRE(h && conv_id >= 0 && conv_id < (int)h->convs.size() && w_descale > 0.f, "hifigan_set_conv_descale: bad arguments");
  SAT_REQUIRE(conv_id != h->id_post() && (h->convs[conv_id].mode != SAT_CONV_F32 || w_descale == 1.f),
              "hifigan_set_conv_descale: only split-f16 convs carry a descale");
  h->convs[conv_id].descale = w_descale;
  return SAT_OK;
}

static size_t hifigan_max_elems(const sat_hifigan* h, int B, int T) {
  size_t mx = (size_t)h->c0 * T;
  int C = h->c0;
  size_t Tc = T;
  for (int i = 0; i < h->n_ups(); ++i) {
    C /= 2;
    Tc *= h->up_rates[i];
    if ((size_t)C * Tc > mx) mx = (size_t)C * Tc;
  }
  return mx * B;
}

// every conv but the output stage on the split-f16 kernel: activations travel as split planes
static bool hifigan_split_acts(const sat_hifigan* h) {
  if (!h->split_acts) return false;
  // conv_pre stages f32 input itself (split-f16 kernel); every later conv reads split planes, all in one mode
  if (h->convs[0].mode != SAT_CONV_F16X3) return false;
  for (int i = 1; i < h->id_post(); ++i)
    if (h->convs[i].mode != h->convs[1].mode || h->convs[i].mode == SAT_CONV_F32) return false;
  int C = h->c0;
  for (int i = 0; i < h->n_ups(); ++i) C /= 2;
  return C % 16 == 0 && h->c0 % 16 == 0;
}

constexpr int WS_SLOTS = 20;   // H (f32 + planes), MRF sum, 2 stage-input planes, 3 branches x (T1, RA, RB) x (f32 + planes)

constexpr size_t WS_MRF_SCRATCH = 256 * 1024;   // behind the slots: the gathered weights of a fused MRF block (mrf.hip)

extern "C" size_t sat_hifigan_workspace_bytes(const sat_hifigan* h, int B, int T) {
  if (!h || B <= 0 || T <= 0) return 0;
  return WS_SLOTS * align_up(hifigan_max_elems(h, B, T) * sizeof(float), 256) + WS_MRF_SCRATCH;
}

extern "C" void sat_hifigan_destroy(sat_hifigan* h) {
  if (!h) return;
  for (auto& kv : h->sides) {
    for (auto st : kv.second.s) if (st) (void)hipStreamDestroy(st);
    if (kv.second.fork) (void)hipEventDestroy(kv.second.fork);
    for (auto e : kv.second.acc) if (e) (void)hipEventDestroy(e);
  }
  delete h;
}

static int hifigan_get_side(sat_hifigan* h, void* stream, hifigan_side** out) {
  std::lock_guard<std::mutex> lock(h->mu);
  auto it = h->sides.find(stream);
  if (it == h->sides.end()) {
    hifigan_side sd;
    for (auto& st : sd.s) SAT_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    SAT_HIP(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
    for (auto& e : sd.acc) SAT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    it = h->sides.emplace(stream, sd).first;
  }
  *out = &it->second;
  return SAT_OK;
}

extern "C" int sat_hifigan_set_option(sat_hifigan* h, const char* name, int value) {
  SAT_REQUIRE(h && name, "hifigan_set_option: null pointer");
  if (std::string(name) == "fuse_pairs") { h->fuse_pairs = value; return SAT_OK; }
  if (std::string(name) == "fuse_pair64") { h->fuse_pair64 = value; return SAT_OK; }
  if (std::string(name) == "fuse_mrf") { h->fuse_mrf = value; return SAT_OK; }
  if (std::string(name) == "multi_branch") { h->multi_branch = value; return SAT_OK; }
  if (std::string(name) == "f8_stages") { h->f8_stages = value; return SAT_OK; }
  if (std::string(name) == "mrf_exact") { h->mrf_exact = value; return SAT_OK; }
  if (std::string(name) == "ups2") { h->ups2 = value; return SAT_OK; }
  if (std::string(name) == "ups_ring") { h->ups_ring = value; return SAT_OK; }
  if (std::string(name) == "split_acts") { h->split_acts = value; return SAT_OK; }
  if (std::string(name) == "planes_residual") { h->planes_residual = value; return SAT_OK; }
  if (std::string(name) == "branch_streams") { h->branch_streams = value; return SAT_OK; }
  if (std::string(name) == "force_f8") { h->force_f8 = value; return SAT_OK; }
  if (std::string(name) == "skip_dead_sum") { h->skip_dead_sum = value; return SAT_OK; }
  set_error("hifigan_set_option: unknown option %s", name);
  return SAT_ERR_INVALID;
}

extern "C" int sat_hifigan_get_option(const sat_hifigan* h, const char* name, int* value) {
  SAT_REQUIRE(h && name && value, "hifigan_get_option: null pointer");
  const std::string n(name);
  if (n == "last_f8_stages") { *value = h->last_f8_stages.load(std::memory_order_relaxed); return SAT_OK; }
  if (n == "fuse_pairs") { *value = h->fuse_pairs; return SAT_OK; }
  if (n == "fuse_pair64") { *value = h->fuse_pair64; return SAT_OK; }
  if (n == "fuse_mrf") { *value = h->fuse_mrf; return SAT_OK; }
  if (n == "multi_branch") { *value = h->multi_branch; return SAT_OK; }
  if (n == "f8_stages") { *value = h->f8_stages; return SAT_OK; }
  if (n == "mrf_exact") { *value = h->mrf_exact; return SAT_OK; }
  if (n == "ups2") { *value = h->ups2; return SAT_OK; }
  if (n == "ups_ring") { *value = h->ups_ring; return SAT_OK; }
  if (n == "split_acts") { *value = h->split_acts; return SAT_OK; }
  if (n == "planes_residual") { *value = h->planes_residual; return SAT_OK; }
  if (n == "branch_streams") { *value = h->branch_streams; return SAT_OK; }
  if (n == "force_f8") { *value = h->force_f8; return SAT_OK; }
  if (n == "skip_dead_sum") { *value = h->skip_dead_sum; return SAT_OK; }
  set_error("hifigan_get_option: unknown option %s", name);
  return SAT_ERR_INVALID;
}

// Range probe of the split planes a forward writes (diagnostic, off unless a buffer is installed): per stage i the words
// buf[2 i] += number of hi values whose magnitude is past `SAT_RANGE_LIMIT` (57 344, the largest e5m2 = what the 8-bit sidecar saturates at;
// f16 itself ends at 65 504) or not finite, buf[2 i + 1] = max over the bit patterns of |hi| as f32.  `buf` = 2 * n_ups device words the
// caller zeroes; nullptr switches the probe off.  Probed: the ResBlock input of every stage, and in the thick stages every inner
// activation and step output that exists as planes.
extern "C" int sat_hifigan_set_range_probe(sat_hifigan* h, uint64_t* buf) {
  SAT_REQUIRE(h, "hifigan_set_range_probe: null handle");
  h->range_probe = (unsigned long long*)buf;
  return SAT_OK;
}

extern "C" int sat_hifigan_convpost_f32(const float* x, const float* w, const float* bias, float* y, int B,
                                        int C, int T, void* stream) {
  SAT_REQUIRE(x && w && bias && y, "convpost: null pointer");
  SAT_REQUIRE(B > 0 && C > 0 && C <= POST_MAXC && T >= 2, "convpost: unsupported shape B=%d C=%d T=%d", B, C, T);
  const size_t lds = ((size_t)C * (POST_TILE + 6) + (size_t)C * 7) * sizeof(float);
  // C = 16 (the reference generator's last stage): batched tile loads; the accumulation order over (c, j) is
  // the same in both instantiations
  auto kern = (C == 16 && (size_t)C * T * 4 < (1ull << 31)) ? (g_convpost_quad ? convpost_kernel<16, true> : convpost_kernel<16>) : convpost_kernel<0>;
  if (lds > 64 * 1024) {
    SAT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  dim3 grid(ceil_div(T + 1, POST_TILE), B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, y, C, T);
  SAT_LAUNCH_CHECK("convpost_kernel");
  return SAT_OK;
}

extern "C" int sat_hifigan_forward_f32(const sat_hifigan* h, const float* x, float* y, void* workspace,
                                       size_t workspace_bytes, int B, int T, void* stream) {
  SAT_REQUIRE(h && x && y && workspace, "hifigan_forward: null pointer");
  SAT_REQUIRE(B > 0 && T > 0, "hifigan_forward: empty batch");
  for (size_t i = 0; i < h->convs.size(); ++i)
    SAT_REQUIRE(h->convs[i].w && h->convs[i].bias, "hifigan_forward: conv %zu has no weights", i);
  SAT_REQUIRE_WORKSPACE(workspace_bytes >= sat_hifigan_workspace_bytes(h, B, T), "hifigan_forward: workspace too small");
  const size_t slot = align_up(hifigan_max_elems(h, B, T) * sizeof(float), 256);
  float* buf[6];
  for (int i = 0; i < 6; ++i) buf[i] = (float*)((char*)workspace + i * slot);
  float* X = buf[0];    // stage input
  float* H = buf[1];    // upsampled
  float* T1 = buf[2];   // inner activation of a resblock pair
  float* RA = buf[3];
  float* RB = buf[4];
  float* ACC = buf[5];  // MRF sum -> next stage input

  auto base_desc = [&](int Cin, int Cout, int Tin, int Tq, int up) {
    sat_conv1d_desc d{};
    d.B = B;
    d.C_in = Cin;
    d.T_in = Tin;
    d.C_out = Cout;
    d.T_q = Tq;
    d.ksize = 1;
    d.dilation = 1;
    d.stride = 1;
    d.groups = 1;
    d.up = up;
    d.x_cstride = Tin;
    d.x_bstride = (int64_t)Cin * Tin;
    d.y_cstride = (int64_t)Tq * up;
    d.y_bstride = (int64_t)Cout * Tq * up;
    d.res_tstride = 1;
    return d;
  };

  if (hifigan_split_acts(h)) {
    // ---- split-plane pipeline: every producer writes the hi|lo f16 planes of leaky_relu(y, 0.1) the
    // consumer will multiply with, so inputs are staged with 16-byte copies and converted once ----
    const int cmode = h->convs[1].mode;                                // SAT_CONV_F16X3 or SAT_CONV_F16F8
    const int yfmt = cmode == SAT_CONV_F16F8 ? 2 : 1;                  // plane format every consumer reads
    char* ws = (char*)workspace;
    float* Hf = (float*)(ws + 0 * slot);   void* Hs = ws + 1 * slot;    // upsampled x (f32 for the split pass, planes)
    float* ACCf = (float*)(ws + 2 * slot);                             // MRF sum
    void* XS = ws + 3 * slot;                                          // stage input planes
    void* XSn = ws + 4 * slot;
    // per resblock branch: planes of the inner activation and of the two ping-pong outputs (+ their f32
    // twins, only written when residuals are not rebuilt from planes)
    auto br = [&](int j, int which) { return ws + (size_t)(5 + j * 5 + which) * slot; };
    const bool planes_res_all = cmode == SAT_CONV_F16X3 && h->planes_residual;
    hifigan_side* side = nullptr;
    const int nk0 = h->n_rbk();
    if (planes_res_all && h->branch_streams && nk0 == 3) {
      int st = hifigan_get_side(const_cast<sat_hifigan*>(h), stream, &side);
      if (st != SAT_OK) return st;
    }
    {
      sat_conv1d_desc d = base_desc(h->in_ch, h->c0, T, T, 1);
      d.ksize = 7;
      d.pad_left = 3;
      d.bias = h->convs[0].bias;
      d.w_descale = h->convs[0].descale;
      d.mode = SAT_CONV_F16X3;
      d.y_split = XS;
      d.y_split_slope = 0.1f;
      d.y_split_format = yfmt;
      d.no_y = 1;
      int s = sat_conv1d_f32(&d, x, h->convs[0].w, nullptr, stream);
      if (s != SAT_OK) return s;
    }
    int C = h->c0, Tc = T;
    const int nk = h->n_rbk();
    int f8_mask = 0;
    for (int i = 0; i < h->n_ups(); ++i) {
      const int u = h->up_rates[i], k = h->up_kernels[i];
      const int Cn = C / 2, Tn = Tc * u;
      const bool last_stage = i == h->n_ups() - 1;
      // this stage's ResBlock convs with 8-bit cross terms (SAT_CONV_F16F8R): the one-launch-per-conv path of the thick stages, every
      // conv's second packing installed, and a batch the ring kernel would serve anyway (small batches keep the f16x3 tiles)
      bool use_f8 = ((h->f8_stages >> i) & 1) && h->multi_branch && planes_res_all && !(side && i < h->branch_streams) && nk >= 2 && nk <= 3 &&
                    Cn > 64 && Cn % 32 == 0 && (h->force_f8 || convring_wanted(Cn, Tn, B));
      for (int j = 0; j < nk && use_f8; ++j)
        for (int pair = 0; pair < 3; ++pair)
          use_f8 = use_f8 && h->convs[h->id_rb(i, j, pair, 0)].w8 && h->convs[h->id_rb(i, j, pair, 1)].w8 &&
                   (h->rb_kernels[j] - 1) * h->rb_dil[j * 3 + pair] <= 64 && h->rb_kernels[j] >= 3;
      // ... and every descriptor of the stage asked of the ring kernel itself (convring_supports has more conditions than
      // convring_wanted — 31-bit output slabs for the fast epilogue, an even step count, ...): a stage one of whose convs it would
      // refuse keeps the f16x3 packing, whose dispatch has the register-staged tiles to fall back on (round-5 advisor item)
      for (int j = 0; j < nk && use_f8; ++j)
        for (int pair = 0; pair < 3 && use_f8; ++pair)
          for (int which = 0; which < 2 && use_f8; ++which) {
            const int rk = h->rb_kernels[j], dil = which ? 1 : h->rb_dil[j * 3 + pair];
            const auto& cv = h->convs[h->id_rb(i, j, pair, which)];
            sat_conv1d_desc d = base_desc(Cn, Cn, Tn, Tn, 1);
            d.ksize = rk, d.dilation = dil, d.pad_left = (rk * dil - dil) / 2;
            d.bias = cv.bias, d.w_descale = cv.descale, d.mode = SAT_CONV_F16F8R;
            d.x_split = ws, d.x_split8 = ws, d.y_split = ws, d.y_split8 = ws, d.y_split_slope = 0.1f;   // (placeholders: nothing is launched)
            d.no_y = 1;
            if (which) {
              d.in_lrelu = 1, d.in_slope = 0.1f, d.res_split = ws, d.res_split_slope = 0.1f, d.res_scale = 1.f;
              if (pair == 2) d.no_y = 0, d.accum = j > 0, d.accum_div = j == nk - 1 ? (float)nk : 0.f, d.y_split8 = nullptr, d.y_split = (j == nk - 1 && !last_stage) ? ws : nullptr;
            } else {
              d.y_split_hi_only = 1;
            }
            use_f8 = sat_conv1d_f8r_supported(&d) != 0;
          }
      if (use_f8) f8_mask |= 1 << i;
      // 8-bit (e5m2) sidecars (half a slot each: 2 bytes per element) in the slots of the f32 twins this pipeline does not write
      void* Hs8 = ws + 0 * slot;
      auto br8 = [&](int j, int which) { return ws + (size_t)(5 + j * 5 + (which < 2 ? 1 : 3)) * slot + (which == 1 ? slot / 2 : 0); };   // 0 T1, 1 RA, 2 RB
      {
        int lo, hi;
        phase_window(k, u, (k - u) / 2, &lo, &hi);
        sat_conv1d_desc d = base_desc(C, Cn, Tc, Tc, u);
        d.ksize = hi - lo + 1;
        d.pad_left = -lo;
        d.bias = h->convs[h->id_up(i)].bias;
        d.w_descale = h->convs[h->id_up(i)].descale;
        d.mode = cmode;
        d.x_split = XS;
        // rates 2 and 4: the transposed conv writes the split planes itself (LDS-transposed epilogue); other
        // rates (5: a block's 64 rows are not whole channel groups) store f32 and split in a streaming pass
        const int co_b = Cn * u > 32 ? 64 : 32;
        const bool direct = cmode == SAT_CONV_F16X3 && h->planes_residual && co_b % (8 * u) == 0;
        int s;
        if (h->ups2 && cmode == SAT_CONV_F16X3 && h->planes_residual && sat_upsample2_supported(C, k, u, (k - u) / 2)) {
          s = sat_upsample2_f16x3(XS, h->convs[h->id_up(i)].w, h->convs[h->id_up(i)].bias, h->convs[h->id_up(i)].descale, Hs, 0.1f, B, C, Tc, stream);
          if (s != SAT_OK) return s;
        } else if (h->ups_ring && sat_upsample_grouped_supported(C, Cn, k, u, (k - u) / 2)) {
          // rows grouped by phase (the packer consulted the same rule): the LDS-DMA ring, zero tap slots skipped (conv_ring16.hip)
          if (!(cmode == SAT_CONV_F16X3 && h->planes_residual)) {
            set_error("hifigan: option ups_ring (packed rows of the stride-4 upsamplers grouped by phase) needs split-f16 weights and the split-plane pipeline");
            return SAT_ERR_INVALID;
          }
          d.y_split = Hs;
          d.y_split_slope = 0.1f;
          d.no_y = 1;
          d.up_grouped = 1;
          d.up_zero_taps = sat_convtranspose_zero_taps(k, u, (k - u) / 2);
          if (use_f8) d.y_split8 = Hs8;        // the ring's upsampler epilogue writes the sidecar next to the planes
          s = sat_conv1d_f32(&d, nullptr, h->convs[h->id_up(i)].w, nullptr, stream);
          if (s != SAT_OK) return s;
        } else if (direct) {
          d.y_split = Hs;
          d.y_split_slope = 0.1f;
          d.no_y = 1;
          s = sat_conv1d_f32(&d, nullptr, h->convs[h->id_up(i)].w, nullptr, stream);
          if (s != SAT_OK) return s;
        } else {
          s = sat_conv1d_f32(&d, nullptr, h->convs[h->id_up(i)].w, Hf, stream);
          if (s != SAT_OK) return s;
          s = sat_act_split_f32(Hf, Hs, B, Cn, Tn, 0.1f, cmode == SAT_CONV_F16F8 ? SAT_SPLIT_F8 : SAT_SPLIT_F16, stream);
          if (s != SAT_OK) return s;
        }
        if (use_f8 && !d.y_split8) {
          s = sat_planes_f8_sidecar(Hs, Hs8, B, Cn, Tn, stream);      // (Hf is dead behind the split pass)
          if (s != SAT_OK) return s;
        }
        if (h->range_probe) {
          s = planes_range_probe(Hs, B, Cn, Tn, h->range_probe + 2 * i, stream);
          if (s != SAT_OK) return s;
        }
      }
      if (h->fuse_mrf && planes_res_all && h->fuse_pairs && nk <= 3 &&
          sat_resblock_mrf_supported(Cn, nk, h->rb_kernels.data(), h->rb_dil.data()) &&
          sat_resblock_mrf_scratch_bytes(nk, h->rb_kernels.data()) <= WS_MRF_SCRATCH) {
        // the whole MRF block of this stage in one launch (mrf.hip): same bits as the loop below
        sat_mrf_desc m{};
        m.B = B; m.C = Cn; m.T = Tn; m.n_branches = nk;
        for (int j = 0; j < nk; ++j) {
          m.ksize[j] = h->rb_kernels[j];
          for (int pair = 0; pair < 3; ++pair) {
            m.dilation[j][pair] = h->rb_dil[j * 3 + pair];
            for (int which = 0; which < 2; ++which) {
              m.w[j][pair][which] = h->convs[h->id_rb(i, j, pair, which)].w;
              m.bias[j][pair][which] = h->convs[h->id_rb(i, j, pair, which)].bias;
              m.w_descale[j][pair][which] = h->convs[h->id_rb(i, j, pair, which)].descale;
            }
          }
        }
        m.slope = 0.1f;
        m.x_split = Hs;
        m.y = ACCf;
        m.y_split = last_stage ? nullptr : XSn;
        m.y_split_slope = 0.1f;
        m.out_div = (float)nk;
        m.residual_from_planes = !h->mrf_exact;
        m.scratch = ws + (size_t)WS_SLOTS * slot;
        m.scratch_bytes = WS_MRF_SCRATCH;
        int s = sat_resblock_mrf_f16x3(&m, stream);
        if (s != SAT_OK) return s;
        void* t = XS;
        XS = XSn;
        XSn = t;
        C = Cn;
        Tc = Tn;
        continue;
      }
      // The thick stages (no fused ResBlock step: C > 64): the i-th conv of ALL branches in one launch
      // (sat_conv1d_multi_f32: the LDS-DMA ring kernel walks the tiles of the three kernel sizes; the MRF sum is
      // accumulated branch by branch inside a block, in the order of the loop below) — 6 launches per stage instead of 18.
      {
        const bool planes_res = cmode == SAT_CONV_F16X3 && h->planes_residual;
        const bool fan = side && i < h->branch_streams;
        if (h->multi_branch && planes_res && !fan && nk >= 2 && nk <= 3 && Cn > 64) {
          const float* xnull[3] = {nullptr, nullptr, nullptr};
          const void* rs[3] = {Hs, Hs, Hs};
          const void* rs8[3] = {Hs8, Hs8, Hs8};
          for (int pair = 0; pair < 3; ++pair) {
            sat_conv1d_desc d1[3], d2[3];
            const void* w1[3];
            const void* w2[3];
            float* y1[3] = {nullptr, nullptr, nullptr};
            float* y2[3];
            void* dst_s[3];
            void* dst8[3] = {nullptr, nullptr, nullptr};
            for (int j = 0; j < nk; ++j) {
              const int rk = h->rb_kernels[j], dil = h->rb_dil[j * 3 + pair];
              const auto& cv1 = h->convs[h->id_rb(i, j, pair, 0)];
              const auto& cv2 = h->convs[h->id_rb(i, j, pair, 1)];
              void* T1s = br(j, 0);
              void* RAs = br(j, 2);
              void* RBs = br(j, 4);
              d1[j] = base_desc(Cn, Cn, Tn, Tn, 1);
              d1[j].ksize = rk;
              d1[j].dilation = dil;
              d1[j].pad_left = (rk * dil - dil) / 2;
              d1[j].bias = cv1.bias;
              d1[j].w_descale = cv1.descale;
              d1[j].mode = cmode;
              d1[j].x_split = rs[j];
              d1[j].y_split = T1s;
              d1[j].y_split_slope = 0.1f;
              d1[j].no_y = 1;
              w1[j] = cv1.w;
              if (use_f8) {
                d1[j].mode = SAT_CONV_F16F8R;
                d1[j].x_split8 = rs8[j];
                d1[j].y_split8 = br8(j, 0);
                d1[j].y_split_hi_only = 1;          // the inner activation is only ever a matrix operand
                w1[j] = cv1.w8;
              }
              d2[j] = base_desc(Cn, Cn, Tn, Tn, 1);
              d2[j].ksize = rk;
              d2[j].dilation = 1;
              d2[j].pad_left = (rk - 1) / 2;
              d2[j].in_lrelu = 1;
              d2[j].in_slope = 0.1f;
              d2[j].bias = cv2.bias;
              d2[j].w_descale = cv2.descale;
              d2[j].mode = cmode;
              d2[j].res_split = rs[j];
              d2[j].res_split_slope = 0.1f;
              d2[j].res_scale = 1.f;
              d2[j].y_split_slope = 0.1f;
              d2[j].x_split = T1s;
              if (pair < 2) {
                dst_s[j] = (rs[j] == RAs) ? RBs : RAs;
                d2[j].no_y = 1;
                y2[j] = nullptr;
              } else {
                y2[j] = ACCf;
                d2[j].accum = j > 0;
                d2[j].accum_div = (j == nk - 1) ? (float)nk : 0.f;
                dst_s[j] = (j == nk - 1 && !last_stage) ? XSn : nullptr;
                // the mean only leaves this stage as planes (the next upsampler's input): its f32 form is not written
                d2[j].accum_no_store = h->skip_dead_sum && j > 0 && dst_s[j] != nullptr;
              }
              d2[j].y_split = dst_s[j];
              w2[j] = cv2.w;
              if (use_f8) {
                d2[j].mode = SAT_CONV_F16F8R;
                d2[j].x_split8 = br8(j, 0);
                if (pair < 2) dst8[j] = (rs8[j] == br8(j, 1)) ? br8(j, 2) : br8(j, 1), d2[j].y_split8 = dst8[j];
                w2[j] = cv2.w8;
              }
            }
            int s = sat_conv1d_multi_f32(d1, xnull, w1, y1, nk, stream);
            if (s != SAT_OK) return s;
            s = sat_conv1d_multi_f32(d2, xnull, w2, y2, nk, stream);
            if (s != SAT_OK) return s;
            for (int j = 0; j < nk && h->range_probe; ++j) {
              s = planes_range_probe(br(j, 0), B, Cn, Tn, h->range_probe + 2 * i, stream);
              if (s == SAT_OK && dst_s[j]) s = planes_range_probe(dst_s[j], B, Cn, Tn, h->range_probe + 2 * i, stream);
              if (s != SAT_OK) return s;
            }
            for (int j = 0; j < nk; ++j) rs[j] = dst_s[j], rs8[j] = dst8[j];
          }
          void* t = XS;
          XS = XSn;
          XSn = t;
          C = Cn;
          Tc = Tn;
          continue;
        }
      }
      if (side && i < h->branch_streams) {
        SAT_HIP(hipEventRecord(side->fork, (hipStream_t)stream));
        for (auto st : side->s) SAT_HIP(hipStreamWaitEvent(st, side->fork, 0));
      }
      for (int j = 0; j < nk; ++j) {
        const int rk = h->rb_kernels[j];
        // branch j on its own stream (the last, longest branch on the caller's); sum order kept by events
        const bool fan = side && i < h->branch_streams;   // option value = number of leading stages fanned out
        void* stream_j = fan && j < 2 ? (void*)side->s[j] : stream;
        void* T1s = br(j, 0);
        float* RAf = (float*)br(j, 1);  void* RAs = br(j, 2);
        float* RBf = (float*)br(j, 3);  void* RBs = br(j, 4);
        const float* rf = Hf;
        const void* rs = Hs;
        for (int pair = 0; pair < 3; ++pair) {
          const int dil = h->rb_dil[j * 3 + pair];
          const auto& cv1 = h->convs[h->id_rb(i, j, pair, 0)];
          const auto& cv2 = h->convs[h->id_rb(i, j, pair, 1)];
          sat_conv1d_desc d2 = base_desc(Cn, Cn, Tn, Tn, 1);
          d2.ksize = rk;
          d2.dilation = 1;
          d2.pad_left = (rk - 1) / 2;
          d2.in_lrelu = 1;
          d2.in_slope = 0.1f;
          d2.bias = cv2.bias;
          d2.w_descale = cv2.descale;
          d2.mode = cmode;
          // the residual is the pair's input: from its split planes (hi + lo, leaky-relu undone) when the
          // format carries both halves, so no f32 copy of the activations is written inside a resblock
          const bool planes_res = cmode == SAT_CONV_F16X3 && h->planes_residual;
          if (planes_res) {
            d2.res_split = rs;
            d2.res_split_slope = 0.1f;
          } else {
            d2.res = rf;
            d2.res_cstride = Tn;
            d2.res_bstride = (int64_t)Cn * Tn;
          }
          d2.res_scale = 1.f;
          d2.y_split_slope = 0.1f;
          float* dstf;
          void* dsts;
          if (pair < 2) {
            dstf = (rf == RAf) ? RBf : RAf;
            dsts = (rs == RAs) ? RBs : RAs;
            d2.no_y = planes_res;
          } else {
            dstf = ACCf;
            d2.accum = j > 0;
            d2.accum_div = (j == nk - 1) ? (float)nk : 0.f;
            dsts = (j == nk - 1 && !last_stage) ? XSn : nullptr;   // the next stage's input planes
            d2.accum_no_store = h->skip_dead_sum && planes_res && j > 0 && dsts != nullptr;      // (see the thick stages above)
          }
          d2.y_split = dsts;
          int s;
          // the MRF sum is read-modify-write on ACC: branch j's last kernel waits for branch j-1's
          auto wait_prev_sum = [&]() -> int {
            if (fan && pair == 2 && j > 0) SAT_HIP(hipStreamWaitEvent((hipStream_t)stream_j, side->acc[j - 1], 0));
            return SAT_OK;
          };
          if ((Cn <= 32 || (Cn == 64 && planes_res && (h->fuse_pair64 & (rk == 3 ? 1 : rk == 7 ? 2 : 4)))) && h->fuse_pairs && cmode == SAT_CONV_F16X3) {
            sat_conv1d_desc df = d2;
            df.dilation = dil;
            df.x_split = rs;
            s = wait_prev_sum();
            if (s != SAT_OK) return s;
            s = sat_resblock_pair_scaled_f16x3(&df, planes_res ? nullptr : rf, cv1.w, cv1.bias, cv1.descale, cv2.w, dstf, stream_j);
            if (s != SAT_OK) return s;
          } else {
            sat_conv1d_desc d1 = base_desc(Cn, Cn, Tn, Tn, 1);
            d1.ksize = rk;
            d1.dilation = dil;
            d1.pad_left = (rk * dil - dil) / 2;
            d1.bias = cv1.bias;
            d1.w_descale = cv1.descale;
            d1.mode = cmode;
            d1.x_split = rs;
            d1.y_split = T1s;
            d1.y_split_slope = 0.1f;
            d1.no_y = 1;
            s = sat_conv1d_f32(&d1, nullptr, cv1.w, nullptr, stream_j);
            if (s != SAT_OK) return s;
            d2.x_split = T1s;
            s = wait_prev_sum();
            if (s != SAT_OK) return s;
            s = sat_conv1d_f32(&d2, nullptr, cv2.w, dstf, stream_j);
            if (s != SAT_OK) return s;
          }
          if (fan && pair == 2) SAT_HIP(hipEventRecord(side->acc[j], (hipStream_t)stream_j));
          rf = dstf;
          rs = dsts;
        }
      }
      void* t = XS;
      XS = XSn;
      XSn = t;
      C = Cn;
      Tc = Tn;
    }
    h->last_f8_stages.store(f8_mask, std::memory_order_relaxed);
    return sat_hifigan_convpost_f32(ACCf, (const float*)h->convs[h->id_post()].w, h->convs[h->id_post()].bias, y, B, C, Tc, stream);
  }
  h->last_f8_stages.store(0, std::memory_order_relaxed);

  // conv_pre (archi.py:78)
  {
    sat_conv1d_desc d = base_desc(h->in_ch, h->c0, T, T, 1);
    d.ksize = 7;
    d.pad_left = 3;
    d.bias = h->convs[0].bias;
    d.w_descale = h->convs[0].descale;
    d.mode = h->convs[0].mode;
    int s = sat_conv1d_f32(&d, x, h->convs[0].w, X, stream);
    if (s != SAT_OK) return s;
  }
  int C = h->c0;
  int Tc = T;
  const int nk = h->n_rbk();
  for (int i = 0; i < h->n_ups(); ++i) {
    const int u = h->up_rates[i], k = h->up_kernels[i];
    const int Cn = C / 2, Tn = Tc * u;
    // x = leaky_relu(x, 0.1); x = ups[i](x)   (archi.py:80-81)
    {
      int lo, hi;
      phase_window(k, u, (k - u) / 2, &lo, &hi);
      if (h->ups_ring && sat_upsample_grouped_supported(C, Cn, k, u, (k - u) / 2)) {
        set_error("hifigan: option ups_ring (packed rows of the stride-4 upsamplers grouped by phase) needs the split-plane pipeline (split_acts)");
        return SAT_ERR_INVALID;
      }
      sat_conv1d_desc d = base_desc(C, Cn, Tc, Tc, u);
      d.ksize = hi - lo + 1;
      d.pad_left = -lo;
      d.in_lrelu = 1;
      d.in_slope = 0.1f;
      d.bias = h->convs[h->id_up(i)].bias;
      d.w_descale = h->convs[h->id_up(i)].descale;
      d.mode = h->convs[h->id_up(i)].mode;
      int s = sat_conv1d_f32(&d, X, h->convs[h->id_up(i)].w, H, stream);
      if (s != SAT_OK) return s;
    }
    // xs = sum_j resblock_j(x); x = xs / num_kernels   (archi.py:82-86)
    for (int j = 0; j < nk; ++j) {
      const int rk = h->rb_kernels[j];
      const float* r = H;
      for (int pair = 0; pair < 3; ++pair) {
        const int dil = h->rb_dil[j * 3 + pair];
        const auto& cv1 = h->convs[h->id_rb(i, j, pair, 0)];
        const auto& cv2 = h->convs[h->id_rb(i, j, pair, 1)];
        float* dst;
        // x = c2(leaky_relu(xt, 0.1)) + x with xt = c1(leaky_relu(x, 0.1))
        sat_conv1d_desc d2 = base_desc(Cn, Cn, Tn, Tn, 1);
        d2.ksize = rk;
        d2.dilation = 1;
        d2.pad_left = (rk - 1) / 2;
        d2.in_lrelu = 1;
        d2.in_slope = 0.1f;
        d2.bias = cv2.bias;
        d2.w_descale = cv2.descale;
        d2.mode = cv2.mode;
        d2.res = r;
        d2.res_scale = 1.f;
        d2.res_cstride = Tn;
        d2.res_bstride = (int64_t)Cn * Tn;
        if (pair < 2) {
          dst = (r == RA) ? RB : RA;
        } else {
          dst = ACC;  // xs += resblock(x); the last one also divides by num_kernels
          d2.accum = j > 0;
          d2.accum_div = (j == nk - 1) ? (float)nk : 0.f;
        }
        int s;
        if (Cn <= 32 && Cn % 16 == 0 && cv1.mode == SAT_CONV_F16X3 && cv2.mode == SAT_CONV_F16X3 && h->fuse_pairs) {
          // thin stages sit on the HBM roofline: one fused kernel, the intermediate stays in LDS
          sat_conv1d_desc df = d2;
          df.dilation = dil;
          s = sat_resblock_pair_scaled_f16x3(&df, r, cv1.w, cv1.bias, cv1.descale, cv2.w, dst, stream);
          if (s != SAT_OK) return s;
        } else {
          sat_conv1d_desc d1 = base_desc(Cn, Cn, Tn, Tn, 1);
          d1.ksize = rk;
          d1.dilation = dil;
          d1.pad_left = (rk * dil - dil) / 2;
          d1.in_lrelu = 1;
          d1.in_slope = 0.1f;
          d1.bias = cv1.bias;
          d1.w_descale = cv1.descale;
          d1.mode = cv1.mode;
          s = sat_conv1d_f32(&d1, r, cv1.w, T1, stream);
          if (s != SAT_OK) return s;
          s = sat_conv1d_f32(&d2, T1, cv2.w, dst, stream);
          if (s != SAT_OK) return s;
        }
        r = dst;
      }
    }
    float* t = X;
    X = ACC;
    ACC = t;
    C = Cn;
    Tc = Tn;
  }
  // x = leaky_relu(x); reflection_pad; conv_post; tanh   (archi.py:87-90)
  return sat_hifigan_convpost_f32(X, (const float*)h->convs[h->id_post()].w, h->convs[h->id_post()].bias, y, B, C, Tc, stream);
}
