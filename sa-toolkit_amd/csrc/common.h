// Shared host-side helpers for libsatools_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/satools_hip.h"

namespace sat {

void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SAT_ERR_HIP;
  }
  return SAT_OK;
}

#define SAT_HIP(call)                                   \
  do {                                                  \
    int _s = ::sat::check_hip((call), #call);           \
    if (_s != SAT_OK) return _s;                        \
  } while (0)

// every launcher passes the name of the kernel family it has just launched: kept (with the launcher's own signature, which
// spells its template arguments) as this thread's last dispatch, for callers that report WHICH kernel served a shape
// (bench.py's roofline.dominant_kernel.name; sat_last_dispatch_name)
void note_dispatch(const char* kernel, const char* launcher);
#define SAT_LAUNCH_CHECK(name)                          \
  do {                                                  \
    ::sat::note_dispatch(name, __PRETTY_FUNCTION__);    \
    SAT_HIP(hipGetLastError());                         \
  } while (0)

#define SAT_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::sat::set_error(__VA_ARGS__);    \
      return SAT_ERR_INVALID;           \
    }                                   \
  } while (0)

// a caller-owned workspace smaller than sat_*_workspace_bytes() asked for: its own status code (satools_hip.h)
#define SAT_REQUIRE_WORKSPACE(cond, ...)  \
  do {                                    \
    if (!(cond)) {                        \
      ::sat::set_error(__VA_ARGS__);      \
      return SAT_ERR_WORKSPACE;           \
    }                                     \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: a launcher that raises it remembers which
// devices it has done so on.  `done` = one bit per device ordinal; the bit is published only after the attribute
// calls have returned (they are idempotent, so two host threads racing here both make them).
inline bool attr_needed_on_current_device(const std::atomic<uint64_t>& done, int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  *dev_out = dev;
  return dev >= 64 || !(done.load(std::memory_order_acquire) & (1ull << dev));
}
inline void attr_done_on_device(std::atomic<uint64_t>& done, int dev) {
  if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
}

// erf for the GELU epilogues (torch.nn.functional.gelu, exact form): branch-free, ~20 VALU instead of ocml erff's
// ~50 with branches (32 M GELUs per wav2vec2 FFN launch were ~30 us of exposed epilogue).  |x| >= 0.6: Abramowitz-Stegun
// 7.1.26, 1 - (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p |x|), on v_rcp_f32 / v_exp_f32; below: the odd Taylor
// polynomial through x^11 (the A-S form cancels there).  Max abs error 2.6e-7 (two ulp of 1.0; measured against
// float64 erf on 4 M points of [-6, 6]), GELU max abs error 3.6e-7 at |x| ~ 3-4, i.e. ~1e-7 relative.
#if defined(__HIPCC__)
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float q = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  q = __builtin_fmaf(q, t, 1.421413741f);
  q = __builtin_fmaf(q, t, -0.284496736f);
  q = __builtin_fmaf(q, t, 0.254829592f);
  q = q * t;
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(x2 * -1.4426950408889634f);
  const float big = __builtin_copysignf(__builtin_fmaf(-q, e, 1.0f), x);
  float s = __builtin_fmaf(-0.0008548327023450853f, x2, 0.005223977625442188f);
  s = __builtin_fmaf(s, x2, -0.026866170645131252f);
  s = __builtin_fmaf(s, x2, 0.11283791670955126f);
  s = __builtin_fmaf(s, x2, -0.37612638903183754f);
  s = __builtin_fmaf(s, x2, 1.1283791670955126f);
  s = s * x;
  return ax < 0.6f ? s : big;
}
__device__ __forceinline__ float gelu_fast(float v) { return v * 0.5f * (1.0f + erf_fast(v * 0.70710678118654752440f)); }

// GELU of four values at once for the exposed epilogues of the GEMM kernels (32 M GELUs per 1024 -> 4096 launch: 37 of its
// 204 us with gelu_fast): the Abramowitz-Stegun branch alone — GELU multiplies 1 + erf by x / 2, so the RELATIVE accuracy of
// erf near 0, which the Taylor branch of erf_fast buys, is not needed (max abs error of the result 4.7e-7 at |x| ~ 3 against
// float64 on 4 M points of [-8, 8]; gelu_fast: 3.6e-7) — in packed f32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: two values per
// instruction; the reciprocal and the exponential stay scalar).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_pair(const f32x2_t x) {
  const f32x2_t z = x * 0.70710678118654752440f;
  const f32x2_t az = __builtin_elementwise_abs(z);
  const f32x2_t d = __builtin_elementwise_fma(az, f32x2_t{0.3275911f, 0.3275911f}, f32x2_t{1.0f, 1.0f});
  const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2_t q = __builtin_elementwise_fma(t, f32x2_t{1.061405429f, 1.061405429f}, f32x2_t{-1.453152027f, -1.453152027f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{1.421413741f, 1.421413741f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{-0.284496736f, -0.284496736f});
  q = __builtin_elementwise_fma(q, t, f32x2_t{0.254829592f, 0.254829592f});
  q = q * t;
  const f32x2_t a = (z * z) * -1.4426950408889634f;
  const f32x2_t e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const f32x2_t m = __builtin_elementwise_fma(-q, e, f32x2_t{1.0f, 1.0f});                  // erf(|z|)
  const f32x2_t er = {__builtin_copysignf(m[0], z[0]), __builtin_copysignf(m[1], z[1])};
  return (x * 0.5f) * (er + 1.0f);
}
__device__ __forceinline__ void gelu_fast4(float (&v)[4]) {
  const f32x2_t a = gelu_pair(f32x2_t{v[0], v[1]}), b = gelu_pair(f32x2_t{v[2], v[3]});
  v[0] = a[0], v[1] = a[1], v[2] = b[0], v[3] = b[1];
}

// ---- f16 halves <-> f32 in ONE instruction (v_fma_mix_f32: an fma whose operands may be f16 halves of a register, widened exactly).
// hipcc folds fma(fpext(h), c, x) into it only when f32 denormals are flushed (not this build's mode): the split / decode steps of every
// epilogue compiled to v_cvt_f32_f16 + v_add / v_sub — two or three instructions per value where one does, with the SAME result:
// the f16 -> f32 widening is exact, so fma(h, 1, l) is the once-rounded sum (float)h + (float)l and fma(h, -1, v) the once-rounded
// difference v - (float)h (tools/scratch/probe_fma_mix.hip: all 2^32 pairs of halves, and 2^28 (half, float) pairs, bit for bit).
// `w` = a register holding two halves; HI selects the upper one.
template <bool HI>
__device__ __forceinline__ float mix_add_halves(unsigned h, unsigned l) {      // (float)half(h) + (float)half(l)
  float r;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
#else
  r = 0.f;
#endif
  return r;
}
template <bool HI>
__device__ __forceinline__ float mix_sub_half(float v, unsigned h) {           // v - (float)half(h)
  float r;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
#else
  r = 0.f;
#endif
  return r;
}

// min(r, r * inv) — the leaky-relu undone for 0 < slope <= 1 (inv = 1 / slope >= 1) — on a value that came out of the asm above: through
// __builtin_fminf hipcc first canonicalises an operand it did not compute itself (v_max_f32 x, x, x: one more instruction per value)
__device__ __forceinline__ float lrelu_undo_min(float r, float inv) {
  const float m = r * inv;
  float o;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_min_f32 %0, %1, %2" : "=v"(o) : "v"(r), "v"(m));
#else
  o = r < m ? r : m;
#endif
  return o;
}

// max(v, v * slope) == (v > 0 ? v : v * slope) for 0 <= slope <= 1, bit for bit (signed zeros included: -0 * slope = -0 = max(-0, -0);
// NaN stays NaN) — two instructions instead of multiply, compare, select with the wait of the select on the compare's mask; the hosts
// of every kernel that uses it refuse slopes outside [0, 1].  Through the asm like lrelu_undo_min: no canonicalising copy.
__device__ __forceinline__ float lrelu_max(float v, float slope) {
  const float m = v * slope;
  float o;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(v), "v"(m));
#else
  o = v > m ? v : m;
#endif
  return o;
}

// lo halves of a split: f16(a - hi0) | f16(b - hi1) for the packed pair `h` = (hi0, hi1) (what v_cvt_pkrtz returned)
template <class H2>
__device__ __forceinline__ auto split_lo2(const H2 h, float a, float b) {
  const unsigned hw = __builtin_bit_cast(unsigned, h);
  return __builtin_amdgcn_cvt_pkrtz(mix_sub_half<false>(a, hw), mix_sub_half<true>(b, hw));
}

// ---- asynchronous global -> LDS copies (LDS-DMA: buffer_load_dwordx4 ... lds, 64 lanes x 16 bytes = 1 KB contiguous in
// LDS per instruction, no staging registers), issued through inline asm: with the builtin, hipcc waits for the copy
// before the next ds_read it cannot prove disjoint (the whole phase), and it knows nothing of these, so the kernel counts
// them itself: every wait for them below is an explicit s_waitcnt vmcnt.  M0 carries the LDS byte address (saved and
// restored: hipcc owns M0).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 dma_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void lds_dma16(const uint4* lds_dst, const i32x4 rs, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);      // wave-uniform by construction
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(la), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
#endif
}
// the same for lanes 0-31 only (a half-filled piece: 512 bytes)
__device__ __forceinline__ void lds_dma16_lo32(const uint4* lds_dst, const i32x4 rs, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
  unsigned keep;
  unsigned long long ex;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %2\n\ts_mov_b64 exec, 0xffffffff\n\ts_nop 0\n\t"
      "buffer_load_dwordx4 %3, %4, %5 offen lds\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(ex)
      : "s"(la), "v"(voff), "s"(rs), "s"(soff)
      : "memory");
#endif
}

#endif

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace sat
