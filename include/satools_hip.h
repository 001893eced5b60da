/*
 * satools_hip.h — C ABI of libsatools_hip.so: the MI355X (gfx950) kernels under the
 * SA-toolkit `anonymize` / `model.convert()` hot path.
 *
 * The reference (deep-privacy/SA-toolkit) has no FFI on this path: the boundary is a Python
 * object (`Net.convert/get_bn/get_f0`, egs/vc/libritts/local/tuning/hifigan.py:58-128).  This
 * library sits UNDER that object: the Python host in `sa-toolkit_amd/` mirrors the reference
 * interface and calls these entry points through ctypes.  Each entry point below cites the
 * reference code whose arithmetic it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative sat_status on failure and never throws;
 *     `sat_last_error()` returns a thread-local message for the last failing call;
 *   - all pointers are caller-owned DEVICE pointers unless a parameter is documented as host;
 *     nothing is allocated behind the caller's back (workspaces are sized by a query call and
 *     passed in);
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream); all work is
 *     enqueued asynchronously on it, no entry point synchronises;
 *   - activations are f32, channel-major `[B][C][T]` with T contiguous ("frames of a
 *     channel are contiguous"), the layout the generator consumes
 *     (hifigan.py:94-97 concatenates along dim 1 of [B,C,T]).
 *   - calls are re-entrant and thread-safe for distinct streams/workspaces.
 */
#ifndef SATOOLS_HIP_H
#define SATOOLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAT_ABI_VERSION 8

typedef enum {
  SAT_OK = 0,
  SAT_ERR_INVALID = -1,     /* bad argument / unsupported shape */
  SAT_ERR_HIP = -2,         /* a HIP runtime call failed */
  SAT_ERR_WORKSPACE = -3,   /* workspace too small */
  SAT_ERR_NO_DEVICE = -4    /* no gfx950 device visible */
} sat_status;

int sat_abi_version(void);
const char* sat_last_error(void);
/* kernel family (with its template arguments) of the last launch the calling thread made through this library, e.g.
 * "conv1d_f16x3_planes_lean_kernel<KS = 11, TG = 6>"; "" before the first launch.  Diagnostics: which kernel the
 * dispatch picked for a shape. */
const char* sat_last_dispatch_name(void);
/* name of device 0 and its CU count; SAT_ERR_NO_DEVICE without a GPU. `name` is a host buffer. */
int sat_device_info(char* name, int name_len, int* cu_count);
/* diagnostic (tools/clock_probe.py): one wave records n pairs (shader cycle counter, 100 MHz wall counter) every
 * period_us into samples[2 n] while the caller runs other work on other streams: the clock the chip holds under it */
int sat_clock_probe(int64_t* samples, int n, int period_us, void* stream);
/* diagnostic (tools/stamp_mrf.py): with a device buffer set, block 0 of every sat_resblock_mrf_f16x3 launch records the
 * shader cycle counter of its 8 waves at 4 points of each conv phase (start, operands read, matrix + epilogue work done,
 * next weights committed, after the barrier) for its first 4 tiles: buf[tile][80 stamps][8 waves].  NULL switches it
 * off.  Returns the number of int64 entries the buffer must hold.  Not thread-safe; never set on the product path. */
int sat_mrf_debug_stamps(int64_t* buf);
/* the same for sat_attention_f16x3 (tools/bench_attention.py stamps): block (0, 0, 0), 7 points of the first key block x 8 waves */
int sat_attention_debug_stamps(int64_t* buf);
/* the same for block 0 of the wave-specialised ResBlock step at C = 32 (csrc/pair32s.hip: pairw_kernel): [8 steps][6 stamps][8 waves] */
int sat_pair32_debug_stamps(int64_t* buf);
/* the same for the LDS-DMA ring conv (csrc/conv_ring16.hip): waves 0 and 4 of EVERY block record [prologue, K loop, waits at the
 * step heads, epilogue, whole kernel] in shader cycles, the kernel's span and start in 100 MHz ticks, the step count:
 * buf[block][2][8].  Returns 8 (entries per wave record). */
int sat_convring_debug_stamps(int64_t* buf);

/* ------------------------------------------------------------------------------------------
 * Fused 1-D convolution as an implicit GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32;
 * exact f32, k-ordered fma chain).  One kernel family serves
 *   - Conv1d / dilated Conv1d of the HiFi-GAN generator  (hifigan/archi.py:40-42, nn.py:96-175)
 *   - ConvTranspose1d, as a polyphase conv with `up` output phases (archi.py:47-59)
 *   - TDNNF linearB (unfold + matmul == valid conv over frames, chain/nn.py:267-278) and linearA
 *   - wav2vec2 conv feature extractor / linear layers / grouped positional conv.
 *
 *   y[b, co, q*up + r] = epilogue( sum_{ci,j} W[(co*up + r), ci, j] *
 *                                   pre(x[b, ci, q*stride + j*dilation - pad_left]) )
 *   pre(v)      = in_lrelu ? (v > 0 ? v : v*in_slope) : v ;   out-of-range x reads as 0
 *   epilogue(v) : v += bias[co]; if(res) v += res_scale*res[b,co,(q*up+r)*res_tstride+res_toff];
 *                 if(ch_scale) v = v*ch_scale[co] + ch_shift[co];  if(relu) v = max(v,0);
 *                 if(gelu) v = 0.5*v*(1+erf(v/sqrt 2));   (res_after_act moves the residual add here)
 *                 if(accum) v = y_old + v;  if(accum_div != 0) v = v / accum_div
 *
 * Weights are pre-packed by the host (sa-toolkit_amd/packing.py) as
 *   w_packed[g][cin_pad][ksize][co_pad],  co fastest, cin_pad = roundup(C_in/groups, 16),
 *   co_pad = roundup(rows_per_group, 64), rows = C_out*up, zero filled.
 * ------------------------------------------------------------------------------------------ */
/* arithmetic of the matrix products:
 *   SAT_CONV_F32   v_mfma_f32_32x32x2_f32, exact f32 (bit-for-bit a k-ordered fma chain)
 *   SAT_CONV_F16X3 operands split as hi + lo f16, products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with
 *                  f32 accumulation.  hi + lo carries 22 significand bits (~2^-21 relative per product) while lo is a
 *                  normal f16, i.e. for |x| above ~0.125 (lo >= 2^-14); below that lo is subnormal and the split is an
 *                  ABSOLUTE error of 2^-25; operands must stay inside the f16 range (|x| < 65504: larger values are
 *                  not supported, they would saturate hi).  Generator and encoder activations are O(1) - O(30).
 *                  weights packed as w16[g][cin_pad/16][ksize][hi|lo][channel half][co_pad][8] f16; stride 1;
 *                  ksize in {1, 2, 3, 7, 11} with f32 input, {1, 3, 7, 11} with split-plane input (x_split);
 *                  operands must lie inside the f16 range (|x| < 65504). */
#define SAT_CONV_F32 0
#define SAT_CONV_F16X3 1
/*   SAT_CONV_F16F8 hi*hi on the f16 MFMA; the two cross terms hi*lo + lo*hi (2^-11 of the product) on the
 *                  block-scaled e4m3 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the f16 rate per K) with
 *                  exact power-of-two scales: ~2^-15 relative per product, measured 3e-6 RMS on the
 *                  generator's waveform against 1.5e-7 for SAT_CONV_F16X3 (bar 1e-4).  Input must be split
 *                  planes in SAT_SPLIT_F8 format; weights |w| < 7 packed by pack_conv_weight_f16f8:
 *                  w[g][cin_pad/16][ksize][4][co_pad][16 B], segments hi f16 ch 0-7 | hi f16 ch 8-15 |
 *                  e4m3(lo * 2^16) 16 ch | e4m3(hi * 2^6) 16 ch. */
#define SAT_CONV_F16F8 2
/*   SAT_CONV_F16F8R (round 5) the same decomposition inside the LDS-DMA ring kernel (csrc/conv_ring16.hip; k-tap convs with
 *                  C_out > 64, C_in % 32 == 0, the generator's ResBlock epilogues): hi*hi on v_mfma_f32_16x16x32_f16, both cross
 *                  terms of a PAIR of taps in one block-scaled 8-bit v_mfma_scale_f32_16x16x128_f8f6f4 (weights e4m3, activations e5m2) (K = 128 = 2 terms x 2 taps
 *                  x 32 channels) — 2 MFMA units per product instead of 3.  The main planes stay SAT_SPLIT_F16 (the residual is
 *                  still rebuilt from hi + lo: 22 bits); the 8-bit activation operands travel in a SIDECAR next to them,
 *                    S8[b][c/16][2 units][t][16 B] = e5m2(hi) of 16 channels | e5m2(lo * 2^10) of 16 channels
 *                  (hi, lo = the f16 values of the main planes, round to nearest even, saturating at 57344: a pure function
 *                  of the planes; e5m2 = f16's exponent range, so activations of any f16 magnitude keep their cross terms),
 *                  read through x_split8 and written by the producer's epilogue through y_split8
 *                  (or by sat_planes_f8_sidecar).  Weights: packing.pack_conv_weight_f16f8r,
 *                    w[2 ceil(C_in/32 x k / 2) steps][8 planes][co_pad][16 B] over the LINEAR sequence of (32-channel pair pp, tap t),
 *                    L = pp k + t; elements (2 q, 2 q + 1) form a pair (the last tap of a channel pair goes with the first tap of the
 *                    next one: an odd k costs no padding).  Per pair an f16 step (planes = (element of the pair, chunk of its channel
 *                    pair, channel half): hi f16 of 8 channels) and an 8-bit step (planes = (element, term, chunk): term 0
 *                    e4m3(lo * 2^9), term 1 e4m3(hi * 2^-2) of 16 channels), hi / lo of w * 2^e with the SAT_CONV_F16X3 layer scale
 *                    (largest |w| * 2^e in [2^9, 2^10)).
 *                  Served by that kernel only (sat_conv1d_f8r_supported), at every batch size. */
#define SAT_CONV_F16F8R 3
/* split-plane formats: per 16-channel chunk and position four 16-byte units
 *   SAT_SPLIT_F16: hi f16 ch 0-7 | hi f16 ch 8-15 | lo f16 ch 0-7 | lo f16 ch 8-15
 *   SAT_SPLIT_F8 : hi f16 ch 0-7 | hi f16 ch 8-15 | e4m3(hi) 16 ch | e4m3(lo * 2^10) 16 ch */
#define SAT_SPLIT_F16 0
#define SAT_SPLIT_F8 1

typedef struct {
  int32_t B, C_in, T_in;       /* input  [B][C_in][T_in]  */
  int32_t C_out, T_q;          /* output [B][C_out][T_q*up]; T_q = positions per phase */
  int32_t ksize, dilation, stride, pad_left;
  int32_t mode;                /* SAT_CONV_F32 (exact f32 MFMA) or SAT_CONV_F16X3 (split-f16, see below) */
  int32_t groups;              /* C_in and C_out divisible by groups */
  int32_t up;                  /* 1 for a plain conv */
  int32_t in_lrelu;  float in_slope;
  int32_t relu;
  int32_t gelu;                /* erf GELU after the affine/ReLU steps */
  int32_t res_after_act;       /* add the residual AFTER relu/gelu (x + gelu(conv(x)), wav2vec2 pos-conv) */
  int32_t accum;     float accum_div;
  float   res_scale; int32_t res_toff, res_tstride;
  int64_t x_bstride, x_cstride;      /* element strides of x  */
  int64_t y_bstride, y_cstride;      /* element strides of y  */
  int64_t res_bstride, res_cstride;  /* element strides of res */
  const float* bias;       /* [C_out] or NULL */
  const float* res;        /* residual source or NULL */
  const float* ch_scale;   /* [C_out] or NULL (BatchNorm1d eval folded: 1/sqrt(var+eps)) */
  const float* ch_shift;   /* [C_out] or NULL (-mean/sqrt(var+eps)) */
  /* split-plane activations (SAT_CONV_F16X3, groups 1, up 1, channel counts multiples of 16): a
   * tensor [B][C][T] stored as S[b][c/16][4 units][t][16 B] (SAT_SPLIT_F16: [hi|lo][(c/8)&1][t][c%8] f16)
   * — the exact B-operand layout of the split-f16 kernels, same number of bytes as the f32 tensor.  A producer writes
   * split(lrelu(y, y_split_slope)) next to (or instead of) y; the consumer then stages its input
   * with one 16-byte load per 8 channels and no conversion (`x` and in_lrelu are ignored). */
  const void* x_split;     /* input as split planes of pre(x), or NULL */
  void* y_split;           /* also write the output as split planes, or NULL */
  float y_split_slope;     /* leaky-relu slope applied before splitting; 1 = none */
  int32_t no_y;            /* skip the f32 store of y (y may be NULL) */
  const void* res_split;   /* residual taken from SAT_SPLIT_F16 planes of lrelu(r, res_split_slope) instead of `res`
                              (r = hi + lo, the leaky-relu undone on the fly: 22 significand bits of r); with it
                              a chain of layers needs no f32 copy of its activations.  up 1, groups 1 */
  float res_split_slope;
  int32_t y_split_format;  /* 0 = the format this mode reads (F16X3: SAT_SPLIT_F16, F16F8: SAT_SPLIT_F8),
                              1 = SAT_SPLIT_F16, 2 = SAT_SPLIT_F8 */
  int32_t relu_first;      /* apply the ReLU BEFORE ch_scale / ch_shift (conv -> relu -> BatchNorm, ECAPA-TDNN) */
  int32_t x_wrap_channels; /* 0, or (1x1 conv on split planes, SAT_CONV_F16X3): x_split holds only this many channels and
                              input channel c >= x_wrap_channels reads channel c - x_wrap_channels one position LATER:
                              y[t] = W[:, :Cw] x[:, t] + W[:, Cw:] x[:C_in - Cw, t + 1].  A stride-2 3-tap conv over
                              [even | odd] phase-split input is this with Cw = 2 C, C_in = 3 C (no zero taps, one GEMM).
                              Multiples of 32; C_out % 128 == 0; positions t + 1 >= T_in read as zero */
  float w_descale;         /* SAT_CONV_F16X3 / F16F8: the packed weights are w * 2^e (packing.py picks e per layer so that the
                              largest |w| sits near 2^10: lo = f16(w - hi) is then a NORMAL f16 for every weight within
                              2^-13 of the largest, i.e. hi + lo carries 22 significand bits whatever the layer's scale) and
                              the accumulator is multiplied by w_descale = 2^-e before the bias — exact.  0 means 1 */
  int32_t up_grouped;      /* up = 4, SAT_CONV_F16X3, planes in and planes out (x_split, y_split, no_y): the rows of the polyphase weight
                              are ordered (16-channel group, phase, channel) — row (c / 16 * 4 + r) * 16 + c % 16 produces phase r of
                              channel c — instead of c * up + r (packing.convtranspose_as_phase_conv(..., grouped=True)): the four
                              16-row strips of a wave tile of the LDS-DMA ring (conv_ring16.hip) are then the four output phases of
                              one channel group and go to the planes without a transposition.  C_out % 16 == 0, C_out * 4 > 128,
                              C_in % 64 == 0, three tap slots (sat_upsample_grouped_supported); served by that kernel only */
  uint32_t up_zero_taps;   /* with up_grouped: bit (slot * 4 + r) set = the weights of tap slot `slot` are all zero for phase r
                              (sat_convtranspose_zero_taps: a k = 8, stride-4 layer has 2 taps per phase in 3 slots, mask 0x30c — the
                              pattern the kernel has an instantiation without those products for).  A promise about the WEIGHTS: a
                              mask naming non-zero weights gives wrong results.  0 (or another pattern) = multiply everything */
  const void* x_split8;    /* SAT_CONV_F16F8R: the e4m3 sidecar of x_split (see the mode) */
  void* y_split8;          /* also write the e4m3 sidecar of y_split (k-tap convs and up_grouped upsamplers served by the LDS-DMA ring
                              kernel, SAT_CONV_F16X3 or SAT_CONV_F16F8R), or NULL */
  int32_t y_split_hi_only; /* with y_split8: do not store the lo units of y_split (a tensor that is only ever an F16F8R operand) */
  int32_t accum_no_store;  /* ABI 7, with accum and y_split: y is READ (the running MRF sum, hifigan/archi.py:82-86) but the new sum is not
                              written back — the last branch of a stage whose successor only takes the planes of the mean (y_split):
                              nobody reads that f32 tensor again (164 MB per launch at the 64- and 32-channel stages) */
} sat_conv1d_desc;

/* 1 when sat_conv1d_f32 / sat_conv1d_multi_f32 serve this SAT_CONV_F16F8R descriptor (the LDS-DMA ring kernel's shapes and epilogues), else 0 */
int sat_conv1d_f8r_supported(const sat_conv1d_desc* d);
/* the 8-bit (e5m2) sidecar of SAT_SPLIT_F16 planes [B][C][T] (C % 16 == 0): x_split8[b][c/16][2][t][16 B] = e5m2(hi) | e5m2(lo * 2^10) */
int sat_planes_f8_sidecar(const void* x_split, void* x_split8, int B, int C, int T, void* stream);

/* the shapes sat_conv1d_desc.up_grouped serves (1 / 0), and the zero (tap slot, phase) pairs of ConvTranspose1d(k, stride u, padding
 * pad) seen as a polyphase conv (satools/satools/hifigan/archi.py:47-59) as the mask sat_conv1d_desc.up_zero_taps takes (u <= 4) */
int sat_upsample_grouped_supported(int C_in, int C_out, int ksize, int stride, int padding);
uint32_t sat_convtranspose_zero_taps(int ksize, int stride, int padding);

int sat_conv1d_f32(const sat_conv1d_desc* d, const float* x, const void* w_packed, float* y,
                   void* stream);

/* One TDNNF layer (ABI 8): TDNNFBatchNorm.forward of the reference, satools/satools/chain/nn.py:306-347 — the factorised linear layer
 * (linearB: context_len frames x feat_dim -> bottleneck_dim, chain/nn.py:267-278; linearA: bottleneck_dim -> out_dim),
 * + bypass_scale * x[t + identity_lidx] (chain/nn.py:233-247, 279-292), BatchNorm1d in eval mode (folded: bn_scale, bn_shift), ReLU
 * (chain/nn.py:336-347) — as the two launches sat_conv1d_f32 would make,
 * behind one descriptor (subsampling factor 1; the strided and the 1.5-frame layers stay on sat_conv1d_f32 / sat_tdnnf_unfold15_f32).
 * Weights: packing of the conv view of the two Linear layers ([bottleneck][feat][context_len], [out][bottleneck][1]) for `mode`.
 * T_q = T_in - (context_len - 1).  On split planes (SAT_CONV_F16X3 with x_split / z_split / y_split, channel counts multiples of 16)
 * the bottleneck only exists as planes (z may be NULL).  Planes only: with x = NULL (and x_split) the bypass is rebuilt from the input
 * planes (hi + lo: 22 significand bits of x), with y = NULL (and y_split) the f32 output is not stored — a chain of layers then moves
 * one tensor per layer through HBM instead of two. */
typedef struct {
  int32_t B, feat_dim, bottleneck_dim, out_dim, T_in, context_len;
  int32_t mode;                    /* SAT_CONV_F32 or SAT_CONV_F16X3 */
  float bypass_scale;              /* 0 = no bypass (else out_dim == feat_dim and x is needed) */
  float wB_descale, wA_descale;    /* sat_conv1d_desc.w_descale of the two packings (0 = 1) */
  const float* x;                  /* [B][feat_dim][T_in], or NULL (then x_split: input AND bypass come from the planes) */
  const void* x_split;             /* SAT_SPLIT_F16 planes of x, or NULL */
  const void *wB_packed, *wA_packed;
  const float *bB, *bA;            /* biases or NULL */
  const float *bn_scale, *bn_shift;/* [out_dim] or NULL */
  float* y;                        /* [B][out_dim][T_q], or NULL (then y_split alone) */
  void* y_split;                   /* also planes of y, or NULL */
  float* z;                        /* [B][bottleneck_dim][T_q] scratch, or NULL when z_split serves */
  void* z_split;                   /* planes scratch of the bottleneck, or NULL */
} sat_tdnnf_layer_desc;
int sat_tdnnf_layer_f32(const sat_tdnnf_layer_desc* d, void* stream);
/* n = 1..3 convolutions d[0..n-1] (x[j], w_packed[j], y[j] as for sat_conv1d_f32) that do not depend on each other's
 * results within the call — the three branches of a multi-receptive-field block (reference hifigan/archi.py:82-86: kernel
 * sizes 3 / 7 / 11 on the same stage) — or that depend on them only through y[j] accumulated in index order (d[j].accum on the
 * same y: job j + 1 adds to what job j stored).  Where all of them are split-plane k-tap convs of one shape that the LDS-DMA
 * ring kernel serves (csrc/conv_ring16.hip: C_out > 64, C_in % 32 == 0, ksize >= 3, option "convring"), ONE launch walks the
 * tiles of all jobs: a block finishes job 0, 1, 2 of its region in turn (in a per-block rotated order when no job reads what
 * another writes), requesting the next tile's operands before the epilogue of the current one.  Otherwise: n calls of
 * sat_conv1d_f32 in index order.  Results are those of the n single calls.  Buffers of different jobs must be the SAME (equal base:
 * kept in index order inside the launch) or DISJOINT; byte ranges that overlap at different bases are detected and served by the
 * single calls. */
int sat_conv1d_multi_f32(const sat_conv1d_desc* d, const float* const* x, const void* const* w_packed, float* const* y, int n,
                         void* stream);
/* One fused ResBlock1 step of the thin generator stages (C = 16 or 32; C = 64 with split planes end to end: x_split in,
 * res_split == x_split), split-f16 arithmetic:
 *   y = conv2(lrelu(conv1(lrelu(x)) + bias1)) + d->bias + x      (hifigan/nn.py:179-186)
 * conv1 = (ksize, d->dilation), conv2 = (ksize, 1), both 'same' padded, slope d->in_slope; the
 * intermediate never leaves LDS.  d->res must be x (or d->res_split == d->x_split); d->accum /
 * accum_div as in sat_conv1d_f32 (MRF sum); x_split / y_split / no_y as in sat_conv1d_f32.
 * Weights: SAT_CONV_F16X3 packing. */
int sat_resblock_pair_f16x3(const sat_conv1d_desc* d, const float* x, const void* w1_packed,
                            const float* bias1, const void* w2_packed, float* y, void* stream);
/* the same with the first conv's weights packed as w1 * 2^e1: w1_descale = 2^-e1 (d->w_descale is the second conv's) */
int sat_resblock_pair_scaled_f16x3(const sat_conv1d_desc* d, const float* x, const void* w1_packed,
                                   const float* bias1, float w1_descale, const void* w2_packed, float* y, void* stream);
/* A whole multi-receptive-field block of a thin generator stage in ONE launch (csrc/mrf.hip): n_branches ResBlock1
 * branches (hifigan/nn.py:93-187; three steps x = x + conv2(lrelu(conv1(lrelu(x)))) each, conv1 dilated 1 / 3 / 5)
 * on the same input, summed in branch order and divided by out_div (hifigan/archi.py:82-86: xs / num_kernels):
 *   y = ((rb_0(x) + rb_1(x)) + rb_2(x)) / out_div          (out_div == 0: no division)
 * The tile, the running x of a branch, the intermediates and the sum never leave the CU; arithmetic and rounding are
 * those of sat_resblock_pair_f16x3 with split planes end to end (x_split in, residual from planes), so the result is
 * bit-identical to that sequence of launches.  Supported (sat_resblock_mrf_supported): C = 16, kernel sizes (3, 7, 11)
 * with n_branches = 3 or any one of them with n_branches = 1, dilations (1, 3, 5).  Weights: SAT_CONV_F16X3 packing. */
typedef struct {
  int32_t B, C, T;
  int32_t n_branches;
  int32_t ksize[3];
  int32_t dilation[3][3];
  const void* w[3][3][2];        /* [branch][step][conv1 | conv2] packed weights */
  const float* bias[3][3][2];
  float w_descale[3][3][2];      /* 2^-e of each conv's packed weights (sat_conv1d_desc.w_descale); 0 means 1 */
  float slope;                   /* leaky-relu slope of every conv input (and of x_split) */
  const void* x_split;           /* input: SAT_SPLIT_F16 planes of lrelu(x, slope) */
  float* y;                      /* f32 output [B][C][T], or NULL */
  void* y_split;                 /* output as planes of lrelu(y, y_split_slope), or NULL */
  float y_split_slope;
  float out_div;
  void* scratch;                 /* device scratch of sat_resblock_mrf_scratch_bytes(): the launch gathers the block's weights
                                    and biases into it first (caller-owned, one per stream in flight) */
  size_t scratch_bytes;
  int32_t residual_from_planes;  /* 1: the residual of steps 2 and 3 is rebuilt from the 22-bit split of the step before, as the
                                    launch-by-launch path does (bit-identical to it); 0: it stays in f32 registers */
} sat_mrf_desc;
int sat_resblock_mrf_supported(int C, int n_branches, const int* ksize, const int* dilations /* [n_branches][3] */);
size_t sat_resblock_mrf_scratch_bytes(int n_branches, const int* ksize);
int sat_resblock_mrf_f16x3(const sat_mrf_desc* d, void* stream);
/* The thin upsamplers of the generator, ConvTranspose1d(C_in -> C_in / 2, k = 4, stride 2, padding 1) with C_in = 32 or 64
 * (hifigan/archi.py:47-59, 80-81), as one streaming launch on split planes (csrc/ups2.hip): x_split [B][C_in][T] planes in,
 * y_split [B][C_in/2][2T] planes of lrelu(y, y_split_slope) out; w_packed = the SAT_CONV_F16X3 packing of the polyphase
 * conv (packing.convtranspose_as_phase_conv + pack_conv_weight_f16x3(up = 2)), w_descale its layer scale.  Same split-f16
 * arithmetic as sat_conv1d_f32 with up = 2, another accumulation order (agrees to f32 rounding). */
int sat_upsample2_supported(int C_in, int ksize, int stride, int padding);
int sat_upsample2_f16x3(const void* x_split, const void* w_packed, const float* bias, float w_descale, void* y_split,
                        float y_split_slope, int B, int C_in, int T, void* stream);
/* process-wide switches of the conv dispatch (A/B measurements): "k1_gemm" sends 1x1 convs on split planes through
 * 0 = the conv tile, 1 = the 128 x 128 GEMM kernel, 2 = the LDS-DMA ring GEMM (32x32x16 MFMA shape) where its
 * 256-column tiles fit, 3 (default) = the ring GEMM on the 16x16x32 shape (results of 3 agree with 0-2 to f32 rounding
 * of the accumulation, 0-2 agree bit for bit); "lean3" / "lean7" / "lean11" (default 1) run 3- / 7- / 11-tap convs on
 * split planes without folded BatchNorm through the three-blocks-per-CU form of the conv tile (same bits as 0); "lean_balance"
 * (0 / 1 default / 2) dispatches the ragged end of every row of that tile as 128-column half tiles after the full ones (same
 * bits); "pair32s" (default 1), "pair32w" (1), "pair64w" (0) send sat_resblock_pair_scaled_f16x3 at C = 32 with 3 taps,
 * C = 32 with 7 / 11 taps and C = 64 with 3 taps (planes in, residual from the planes, dilation <= 5) through the streaming /
 * wave-specialised kernels of csrc/pair32s.hip (agree with the general fused step to f32 rounding of the accumulation);
 * "pair32s_waves" (8 / 4) picks that file's block shape at 3 taps.  Round 4: "convring" (0 / 1 default; + 32: whatever the number
 * of tiles) sends k-tap convs on split planes with C_out > 64, C_in % 32 == 0, k >= 3 and the generator's ResBlock epilogues (bias,
 * residual from planes, MRF accumulation; no folded BatchNorm / ReLU / GELU) through the LDS-DMA ring kernel on the 16x16x32 MFMA
 * shape (csrc/conv_ring16.hip: 256 x 160 / 128 x 320 tiles, one 8-wave block per CU) when they fill three quarters of the CUs —
 * K = 32 per instruction associates differently from the register-staged tiles: agreement to f32 rounding of the accumulation;
 * "gemm_walk" (0 / 1 default; + 2: also single GEMMs with no more tiles than CUs) sends 1x1 convs of k1_gemm = 3 with a plain
 * Linear epilogue (bias, f32 residual, GELU, f32 / plane stores) through the persistent form of that ring (csrc/gemm_walk16.hip)
 * when a launch holds more tiles than CUs or several GEMMs (sat_conv1d_multi_f32) — the bits of k1_gemm = 3; "convring_blocks" (0 default = one block per CU; else the
 * blocks of a ring-conv launch, a multiple of 8: CUs left to the kernels of other streams, an A/B switch); "trim_halo" (0 / 1
 * default): the fused ResBlock steps at C = 64 / 32 load only the columns of their staged input window that conv1 reads (same bits).
 * Unknown names return SAT_ERR_INVALID. */
int sat_conv_set_option(const char* name, int value);
/* f32 [B][C][T] -> split planes of lrelu(x, slope) in `format` (SAT_SPLIT_*); C % 16 == 0 */
int sat_act_split_f32(const float* x, void* x_split, int B, int C, int T, float slope, int format, void* stream);
/* cin_pad / co_pad the packed layout must use for this shape (host-side helper, no GPU needed) */
int sat_conv1d_packed_dims(int C_in, int C_out, int up, int groups, int* cin_pad, int* co_pad);
/* Polyphase view of ConvTranspose1d(k, stride u, padding pad): output t = q*u + r reads input
 * s = q + delta with tap j = r + pad - u*delta.  Returns the width of the delta window over all
 * phases (the `ksize` of the equivalent conv with `up = u`) and its left padding (-delta_min). */
int sat_convtranspose_phase_dims(int k, int u, int pad, int* ksize, int* pad_left);

/* ------------------------------------------------------------------------------------------
 * HiFi-GAN generator (CoreHifiGan.forward_resnet, hifigan/archi.py:77-91; ResBlock1,
 * hifigan/nn.py:93-187).  The handle only stores the architecture and the device pointers of
 * the packed weights (owned by the caller, which must keep them alive).
 * Conv ids: 0 = conv_pre; 1..n_ups = ups[i]; then resblock convs in the order
 * resblocks[rb].convs1[0], convs2[0], convs1[1], convs2[1], convs1[2], convs2[2];
 * last = conv_post (plain [C][7] weights, not packed).
 * ------------------------------------------------------------------------------------------ */
typedef struct sat_hifigan sat_hifigan;
int sat_hifigan_create(sat_hifigan** out, int in_channels, int initial_channels, int n_ups,
                       const int* up_rates, const int* up_kernels, int n_rb_kernels,
                       const int* rb_kernels, const int* rb_dilations /* [n_rb_kernels][3] */);
int sat_hifigan_num_convs(const sat_hifigan* h);
int sat_hifigan_set_conv(sat_hifigan* h, int conv_id, const void* w_packed, const float* bias, int mode);
/* power-of-two descale of a conv's packed weights (sat_conv1d_desc.w_descale); 1 after sat_hifigan_set_conv */
int sat_hifigan_set_conv_descale(sat_hifigan* h, int conv_id, float w_descale);
/* a second packing of a ResBlock conv for SAT_CONV_F16F8R (packing.pack_conv_weight_f16f8r; same layer scale w_descale as its
 * SAT_CONV_F16X3 packing): used by the stages option "f8_stages" names when the ring kernel serves the batch, else the
 * SAT_CONV_F16X3 packing of sat_hifigan_set_conv is (small batches).  NULL removes it. */
int sat_hifigan_set_conv_f8r(sat_hifigan* h, int conv_id, const void* w_packed_f8r);
size_t sat_hifigan_workspace_bytes(const sat_hifigan* h, int B, int T);
/* x [B][in_channels][T] -> y [B][1][T*prod(up_rates)+1]  (tanh output, archi.py:87-90) */
int sat_hifigan_forward_f32(const sat_hifigan* h, const float* x, float* y, void* workspace,
                            size_t workspace_bytes, int B, int T, void* stream);
void sat_hifigan_destroy(sat_hifigan* h);
/* options: "fuse_pairs" (default 1): run the conv pairs of stages with C <= 32 as one fused kernel; "fuse_pair64" (bit mask,
 * default 3), "fuse_mrf" (default 1: whole MRF block of the C = 16 stage in one launch), "mrf_exact" (default 0),
 * "ups2" (default 1: the two thin upsamplers on sat_upsample2_f16x3), "multi_branch" (default 1: on the stages with C > 64 the i-th
 * conv of all MRF branches as ONE sat_conv1d_multi_f32 call — 6 launches per stage instead of 18, same bits), "split_acts",
 * "planes_residual", "branch_streams"; "f8_stages" (round 5, default 0): bit i set = the ResBlock convs of stage i run as
 * SAT_CONV_F16F8R where sat_hifigan_set_conv_f8r installed their packing and the ring kernel serves the batch */
int sat_hifigan_set_option(sat_hifigan* h, const char* name, int value);
/* ABI 7.  "force_f8" (default 0): the SAT_CONV_F16F8R stages run that way at EVERY batch size (a calibration batch of one utterance is too
 * small for the ring kernel's default dispatch; the load-time guard `Net.check_precision`, which stands where the reference's
 * infer_helper.load_model, satools/satools/infer_helper.py:10-59, hands out a model, sets it on its own handle).
 * sat_hifigan_get_option reads an option back; read-only "last_f8_stages": bit i = stage i of the handle's LAST forward ran its
 * ResBlock convs with 8-bit cross terms (a batch too small for the ring kernel runs them in SAT_CONV_F16X3: the caller can see which). */
int sat_hifigan_get_option(const sat_hifigan* h, const char* name, int* value);
/* ABI 7.  Range probe of the split planes a forward writes (diagnostic of the guard; off unless installed): `buf` = 2 * n_ups device
 * words, zeroed by the caller; per stage i, buf[2 i] += number of hi halves past 57 344 (the largest e5m2: where the 8-bit sidecar of
 * SAT_CONV_F16F8R saturates; f16 itself ends at 65 504) or not finite, buf[2 i + 1] = max |hi| as the bit pattern of an f32.  Probed:
 * the ResBlock input of every stage and, on the stages with C > 64, every inner activation and step output.  nullptr = off. */
int sat_hifigan_set_range_probe(sat_hifigan* h, uint64_t* buf);

/* final stage alone: leaky_relu(0.01) -> ReflectionPad1d((1,0)) -> Conv1d(C,1,7,pad 3) -> tanh
 * (archi.py:87-90).  x [B][C][T] -> y [B][1][T+1];  w [C][7], bias [1]. */
int sat_hifigan_convpost_f32(const float* x, const float* w, const float* bias, float* y, int B,
                             int C, int T, void* stream);

/* ------------------------------------------------------------------------------------------
 * Kaldi-compatible fbank (satools/satools/kaldifeature.py:461-593, called as
 * fbank(x*32768, num_mel_bins=80, snip_edges=False) at tdnnf_vq.py:243-244) fused with the
 * per-utterance mean normalisation UttCMVN() (cmvn.py:157-165) and the replicate padding of
 * pad_input (tdnnf_vq.py:228-234).
 *   wav [B][n]  ->  feats [B][n_mel][pad + m + pad],  m = (n + shift/2) / shift
 * `window` [400] and `mel` [n_mel][257] are device tables built by the host with the
 * reference's own formulas (povey window; get_mel_banks); `mel_lo/mel_hi` [n_mel] bound the
 * non-zero bins of each triangular filter; `scale` is the 32768 factor.
 * workspace: sat_fbank_workspace_bytes(B, n).
 * ------------------------------------------------------------------------------------------ */
size_t sat_fbank_workspace_bytes(int B, int n);
int sat_fbank_cmvn_pad_f32(const float* wav, float* feats, const float* window, const float* mel,
                           const int32_t* mel_lo, const int32_t* mel_hi, void* workspace,
                           size_t workspace_bytes, int B, int n, float scale, int n_mel, int pad,
                           int do_cmvn, void* stream);

/* ------------------------------------------------------------------------------------------
 * Vector quantiser of the bottleneck layer (VectorQuantizerEMA.forward eval branch,
 * chain/nn.py:402-476): d = (sum x^2 + sum e^2) - 2 x.e^T in f32, first-minimum argmin,
 * output x + (e[idx] - x).   z [B][D][T] (channel-major) , codebook [n_codes][D]
 *   -> q [B][D][T], idx [B][T] (int32), optional dist [B][T][n_codes], optional margin [B][T]
 * ------------------------------------------------------------------------------------------ */
int sat_vq_argmin_gather_f32(const float* z, const float* codebook, float* q, int32_t* idx,
                             float* dist, int B, int D, int T, int n_codes, void* stream);
/* ABI 7.  The same, and tie_count[0][b] += the frames of utterance b whose two best codes a, a' are a NEAR-TIE of this arithmetic:
 * d[a'] - d[a] <= tie_scale * |z_t| * pair_dist[a][a'] (pair_dist [n_codes][n_codes] = |e_a - e_a'|; tie_scale = 2 K sigma_rel / sqrt(D),
 * K standard deviations of the calibrated per-component feature error of the split-f16 extractor against its exact-f32 twin).  The
 * host re-decides flagged utterances on the exact-f32 kernels (asrbn.py), so that the indices of the default arithmetic are the
 * exact ones (chain/nn.py:424-459 is index work).  tie_count [3][B] int32: row 0 the counts (zeroed by the caller), row 1 the first and
 * row 2 the last near-tie frame of the utterance (the caller sets them to INT32_MAX / -1): the frames in between are what is decided again. */
int sat_vq_argmin_gather_tie_f32(const float* z, const float* codebook, float* q, int32_t* idx, float* dist,
                                 const float* pair_dist, float tie_scale, int32_t* tie_count,
                                 int B, int D, int T, int n_codes, void* stream);

/* pad frames at both ends: x [B][C][T] -> y [B][C][left+T+right].  Left = first frame replicated.
 * Right: interleave_right = 0 -> last frame replicated (F.pad(...,"replicate"),
 * tdnnf_wav2vec2_vq.py:299); interleave_right = 1 -> the reference's pad_input
 * (tdnnf_vq.py:228-234), whose right side tiles the last frames of ALL utterances of the batch
 * as one sequence: frame p of utterance b is the last frame of utterance (b*right + p) mod B.
 * sat_fbank_cmvn_pad_f32 applies the same pad_input rule. */
int sat_pad_replicate_f32(const float* x, float* y, int B, int C, int T, int left, int right,
                          int interleave_right, void* stream);

/* ASR half of the bottleneck net (Net.forward, tdnnf_vq.py:259-284).
 *   sat_tdnnf_unfold15_f32: the input windows and the bypass of a TDNNF layer with subsampling_factor 1.5
 *     (chain/nn.py:267-304: `unfold` of the flattened [T*D] input with step int(1.5*D), so every other window
 *     straddles two frames; add_padd).  x [B][D][T] -> win, byp [B][D][Tq], Tq = (2(T-1))/3 + 1; the layer is a
 *     1x1 sat_conv1d_f32 on `win` with `byp` as residual (res_scale = bypass_scale).
 *   sat_log_softmax_channels_f32: in place over C of x [B][C][T] (F.log_softmax(xent_out, dim=2)). */
int sat_tdnnf_unfold15_f32(const float* x, float* win, float* byp, int B, int D, int T, void* stream);
int sat_log_softmax_channels_f32(float* x, int B, int C, int T, void* stream);

/* ------------------------------------------------------------------------------------------
 * Generator input assembly (Net._forward, hifigan.py:83-97): F0 normalisation statistics are
 * batch-coupled (UttCMVN(var_norm=True, keep_zeros=True), cmvn.py:143-155), so the mean/std
 * reduction and the per-element transform are separate calls.
 *   sat_f0_stats_f32:  f0 [n] -> stats[2] = {mean, std} over the non-zero entries
 *                      (std = sqrt(unbiased var + 1e-6))
 *   sat_f0_apply_f32:  in place: voiced -> (v - mean)/std, zeros stay 0; optional quantisation
 *                      round(v*q)/q (half-to-even) keeping zeros (hifigan/nn.py:28-40);
 *                      optional additive noise (host-drawn, nn.py:42-62) re-zeroing positions
 *                      that are 0 after quantisation.
 *   sat_assemble_input_f32: x[b] = [ bn[b] (C_bn x T) ; nearest-interp f0[b] (1 x T_f0 -> T) ;
 *                      spk[b] (n_spk values, the one-hot row as f32) broadcast over T ]
 *                      (F.interpolate nearest + torch.cat, hifigan.py:91-97)
 * ------------------------------------------------------------------------------------------ */
int sat_f0_stats_f32(const float* f0, int n, float* stats, void* stream);
int sat_f0_apply_f32(float* f0, int n, const float* stats, int quant_bins, const float* noise,
                     void* stream);
/* mean-reversion F0 transformation, one track [T] (hifigan/nn.py:64-90 `moving_average_f0` + `mean_reverv_f0`;
 * dispatched at egs/vc/libritts/local/tuning/hifigan.py:79-80): out = (1 - alpha) f0 + alpha movavg_n(f0), window
 * f0[t - n/2 .. t - n/2 + n - 1] with zeros outside.  out != f0.  The reference handles a batch of 1 only. */
int sat_f0_mean_reversion_f32(const float* f0, float* out, int T, float alpha, int n, void* stream);
int sat_assemble_input_f32(const float* bn, const float* f0, const float* spk, float* x,
                           int B, int C_bn, int T, int T_f0, int n_spk, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sample formats of the batch job's data plane (ABI 6).  The reference reads its utterances with torchaudio.load
 * (satools/satools/utils/kaldi.py:113-125: 16-bit PCM normalised to [-1, 1) = s / 32768) and writes the anonymized ones with
 * torchaudio.save(..., encoding='PCM_S', bits_per_sample=16) (satools/satools/bin/pipeline.py:159).  Doing both conversions
 * on the device halves the bytes that cross PCIe in either direction and takes them off the host threads.
 *   sat_pcm16_to_f32:   y[i] = x[i] / 32768 (exact)
 *   sat_pcm16_from_f32: y[i] = clamp(rint(x[i] * 32768), -32768, 32767), round-half-even — the bits of
 *                       numpy.clip(numpy.rint(float64(x) * 32768), -32768, 32767) (sa-toolkit_amd/pipeline.py: save_pcm16)
 * n elements, contiguous; x != y.
 * ------------------------------------------------------------------------------------------ */
int sat_pcm16_to_f32(const int16_t* x, float* y, long long n, void* stream);
int sat_pcm16_from_f32(const float* x, int16_t* y, long long n, void* stream);

/* ------------------------------------------------------------------------------------------
 * YAAPT pitch tracker (satools/satools/hifigan/yaapt.py:795-951), whole batch per call.
 * The plan carries every integer / scalar constant the reference derives from its option dict
 * (yaapt.py:815-886, :156-157, :190-204, :396-406, :686-688); the host computes it
 * (sa-toolkit_amd/f0.py) so that the rounding of those derivations is done once, in Python, the
 * way the reference does it.  lp / hp = biquad constants {b0/a0, b1/a0, b2/a0, a0, a1/a0, a2/a0}
 * (torchaudio's _lfilter normalises b and a by a0 first; FIR = fma chain in tap order, see csrc/yaapt.hip).
 *   wav [B][n]  ->  f0 [B][nframes] (Hz, 0 = unvoiced);  status [B] (device int32):
 *   0 ok, 1 = no voiced frame (the reference raises RuntimeError there), 2 = NCCF window
 *   invalid (the reference's assert N > 0).
 * hann [frame_size], kaiser [2*frame_size], twiddle [4096][2] = exp(-2*pi*i*k/8192) are device
 * tables built by the host.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t n, pad, L, Lz, nfft, frame_size, frame_jump, nframes;
  int32_t nl_lo, nl_hi;
  int32_t nframe_size, half_wl, wl, max_shc, min_shc, nharm, maxpeaks;
  int32_t pk_center, pk_min_lag, pk_max_lag;
  int32_t tda_len, tda_nframes, maxcands, nccf_center, median_value;
  float fs, delta;
  float nlfer_thresh1, nlfer_thresh2;
  float shc_thresh1, inv_shc_thresh1, shc_thresh2, f0_double, f0_half, merit_extra;
  float dp5_k1, f0_min, f0_max, spec_pitch_min_std;
  float nccf_thresh1, nccf_thresh2, merit_boost1, merit_pivot;
  float dp_w1, dp_w2, dp_w3, dp_w4;
  float lp[6], hp[6];
} sat_yaapt_plan;

size_t sat_yaapt_workspace_bytes(const sat_yaapt_plan* plan, int B);
int sat_yaapt_f32(const sat_yaapt_plan* plan, const float* wav, float* f0, int32_t* status,
                  const float* hann, const float* kaiser, const float* twiddle, void* workspace,
                  size_t workspace_bytes, int B, void* stream);
/* Ragged batch: zero-padded utterances [B][plan->n] tracked at their OWN lengths in one launch sequence — what
 * the reference's data loader does one utterance at a time (bin/pipeline.py:35-41: get_f0 per utterance, tracks
 * zero-padded by the collate).  `plan` is the plan of the LONGEST utterance (buffer strides, launch grids);
 * utt_dims [B][4] (device int32) = {samples, padded length L, frames, tda frames} of every utterance, computed
 * by the host with the same derivations as the plan.  f0 [B][plan->nframes], zero past an utterance's frames. */
int sat_yaapt_ragged_f32(const sat_yaapt_plan* plan, const float* wav, const int32_t* utt_dims, float* f0,
                         int32_t* status, const float* hann, const float* kaiser, const float* twiddle,
                         void* workspace, size_t workspace_bytes, int B, void* stream);

/* ------------------------------------------------------------------------------------------
 * wav2vec2 support (torchaudio.models.wav2vec2 as configured by tdnnf_wav2vec2_vq.py:39-56; third
 * party to the reference, restated in oracle/wav2vec2.py).  All matrix-shaped work (conv feature
 * extractor after layer 0, Linear layers, grouped positional conv, attention products) goes through
 * sat_conv1d_f32; these are the remaining pieces.
 *   sat_w2v2_conv0_f32:        y[b][c][t] = bias[c] + sum_j w[c][j] x[b][t*stride + j]
 *   sat_layernorm_channels_f32: LayerNorm over C of x[B][C][T] (eps 1e-5), optional erf-GELU; with
 *                              split_phases the output is [B][2C][ceil(T/2)] = even | odd time phases,
 *                              which turns the next stride-2 conv into a stride-1 conv
 *   sat_layernorm_channels_planes_f32: the same, writing the result as SAT_SPLIT_F16 planes `y_split` (C % 16 == 0;
 *                              [2C or C channels][ceil(T/2) or T frames]) for a following split-f16 conv, and as
 *                              f32 `y` too unless `y` is null
 *   sat_softmax_columns_f32:   in place on S^T [G*T keys][pitch]: softmax over keys of scale*s per query
 *   sat_transpose_heads_f32:   v [G][D][pitch] -> vt [G][jpad][D], rows >= T zero (packed-weight layout)
 * ------------------------------------------------------------------------------------------ */
int sat_w2v2_conv0_f32(const float* x, const float* w, const float* bias, float* y, int B, int n, int C,
                       int k, int stride, void* stream);
int sat_layernorm_channels_f32(const float* x, const float* gamma, const float* beta, float* y, int B,
                               int C, int T, int64_t x_bstride, int64_t x_cstride, int64_t y_bstride,
                               int64_t y_cstride, int gelu, int split_phases, void* stream);
int sat_layernorm_channels_planes_f32(const float* x, const float* gamma, const float* beta, float* y, void* y_split,
                                      int B, int C, int T, int64_t x_bstride, int64_t x_cstride, int64_t y_bstride,
                                      int64_t y_cstride, int gelu, int split_phases, void* stream);
/* conv layer 0 fused with its LayerNorm (+GELU, + phase split): y / y_split = LN(conv0(wav)) as the two entry points
 * above would give it, bit for bit, without the [B][C][T0] f32 tensor in between (C <= 512, k <= 12). */
int sat_w2v2_conv0_layernorm_f32(const float* wav, const float* w, const float* bias, const float* gamma,
                                 const float* beta, float* y, void* y_split, int B, int n, int C, int k, int stride,
                                 int64_t y_bstride, int64_t y_cstride, int gelu, int split_phases, void* stream);
int sat_softmax_columns_f32(float* st, int G, int T, int pitch, float scale, void* stream);
/* Fused self-attention of the wav2vec2 encoder layers (torchaudio SelfAttention as configured by
 * tdnnf_wav2vec2_vq.py:39-56; 16 heads x 64) in split-f16 arithmetic, scores kept in registers:
 *   o[b][h*64 + d][t] = sum_j softmax_j(scale * sum_c q[b][h*64+c][t] k[b][h*64+c][j]) * v[b][h*64+d][j]
 * q_split / k_split: SAT_SPLIT_F16 planes [heads*64 channels][T frames] (the projections' y_split); v: f32
 * [B][heads*64][v_pitch >= T, a multiple of 4]; o (f32 [B][heads*64][T]) and / or o_split (planes), either may be null. */
int sat_attention_f16x3(const void* q_split, const void* k_split, const float* v, float* o, void* o_split, int B,
                        int heads, int head_dim, int T, int v_pitch, float scale, void* stream);
int sat_transpose_heads_f32(const float* v, float* vt, int G, int D, int T, int pitch, int jpad, void* stream);

/* ------------------------------------------------------------------------------------------
 * x-vector extractor (ECAPA-TDNN; egs/asv/voxceleb/local/tuning/ecapa_tdnn.py:18-81).  Every Conv1d / Linear goes
 * through sat_conv1d_f32 (relu_first: ReLU before the folded BatchNorm, sidekit/nn.py:106-118); these are the rest.
 *   sat_melspec_logmel_f32   pre-emphasis (augmentation.py:219-244) + torchaudio MelSpectrogram(n_fft 1024, win 400,
 *                            hop 160, center/reflect, power 2) + 1e-6 + log  (sidekit/preprocessor.py:223-232):
 *                            wav [B][n] -> out [B][n_mel][1 + n/160]; window [400]; fb [n_mel][513] with the non-zero
 *                            range [fb_lo, fb_hi) of every filter (host tables)
 *   sat_instnorm_rows_f32    InstanceNorm1d over time (preprocessor.py:233): rows of [R][T]
 *   sat_row_mean_f32         mean over time of [R][T] rows (SE_Connect, sidekit/nn.py:132)
 *   sat_add3_f32             y = a + b (+ c) on channel slices [B][C][T] (Res2Net partial sums nn.py:99-101,
 *                            block inputs archi.py:183-185)
 *   sat_se_gate_add_f32      y = z * sigmoid(g[b][c]) + s1 + s2 + s3 (SE gate nn.py:133-136 + skip connections)
 *   sat_tanh_inplace_f32, sat_attentive_stats_f32   AttentiveStatsPool (sidekit/pooling.py:148-155): softmax over time
 *                            of `logits`, weighted mean and std of x -> out [B][2C]
 *   sat_l2norm_rows_f32      F.normalize(x, dim=1) (ecapa_tdnn.py:76)
 *   sat_res2_chain_f32       Res2Conv1dReluBn (sidekit/nn.py:74-110; ABI 6) on pieces of 64 channels in ONE launch: piece i of z =
 *                            bn_i(relu(conv_i(piece i of y + piece i - 1 of z))) for i < nums (three taps, `dilation`, zero padding, no bias;
 *                            the BatchNorm in eval as scale / shift [nums][64]), the last piece copied.  y, z [B][(nums + 1) 64][T], y != z;
 *                            w [nums][3 taps][64 ci][64 co] (Conv1d.weight permuted (2, 1, 0)); dilation <= 4, dilation * nums <= 32.
 *                            Exact f32 on the f32 MFMA
 *   sat_linear_rows_f32      nn.Linear on pooled vectors (ABI 6): y[b][o] = (relu?)(w[o] . x[b] + bias[o]) (* ch_scale[o] + ch_shift[o]) —
 *                            SE_Connect.linear1 / linear2 (sidekit/nn.py:133-139), before_speaker_embedding lin + bn2 in eval
 *                            (ecapa_tdnn.py:40-43, :77).  x [B][Cin], w [Cout][Cin] row-major as the checkpoint holds it, y [B][Cout];
 *                            bias / ch_scale / ch_shift may be null.  f32 FMAs, lane-strided partial sums reduced across the wave
 * ------------------------------------------------------------------------------------------ */
int sat_melspec_logmel_f32(const float* wav, float* out, const float* window, const float* fb, const int32_t* fb_lo,
                           const int32_t* fb_hi, int B, int n, int n_mel, float coef, void* stream);
int sat_instnorm_rows_f32(const float* x, float* y, int R, int T, float eps, void* stream);
int sat_row_mean_f32(const float* x, float* y, int R, int T, void* stream);
int sat_add3_f32(const float* a, const float* b, const float* c, float* y, int B, int C, int T, int64_t a_bs, int64_t a_cs,
                 int64_t b_bs, int64_t b_cs, int64_t c_bs, int64_t c_cs, int64_t y_bs, int64_t y_cs, void* stream);
int sat_se_gate_add_f32(const float* z, const float* gate_logits, const float* s1, const float* s2, const float* s3,
                        float* y, int B, int C, int T, int64_t y_bs, int64_t y_cs, void* stream);
int sat_tanh_inplace_f32(float* x, size_t n, void* stream);
int sat_attentive_stats_f32(const float* x, const float* logits, float* out, int B, int C, int T, void* stream);
int sat_l2norm_rows_f32(const float* x, float* y, int R, int D, void* stream);
int sat_res2_chain_f32(const float* y, float* z, const float* w, const float* scale, const float* shift, int B, int C, int T, int nums,
                       int dilation, void* stream);
int sat_linear_rows_f32(const float* x, const float* w, const float* bias, const float* ch_scale, const float* ch_shift, int relu,
                        float* y, int B, int Cin, int Cout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SATOOLS_HIP_H */
