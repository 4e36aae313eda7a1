// Shared helpers for libnsc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../include/nsc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));   // accumulator of v_mfma_f32_32x32x2_f32
typedef int i32x4_t __attribute__((ext_vector_type(4)));     // payload type of the 16-byte raw buffer stores

#define NSC_LRELU_ALPHA 0.2f

// Profiling / A-B switches read from the environment exist only in a -DNSC_PROBES build (`make PROBES=1`).  The
// shipped library has no environment-dependent behaviour: the macros fold to their defaults at compile time.
#ifdef NSC_PROBES
#include <cstdlib>
#define NSC_PROBE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define NSC_PROBE_SET(name) (getenv(name) != nullptr)
#else
#define NSC_PROBE_INT(name, dflt) (dflt)
#define NSC_PROBE_SET(name) (false)
#endif

void nsc_set_error(const char* fmt, ...);

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also drains vmcnt, i.e. every wave waits for the
// write acknowledgements of all its global stores (and for its prefetch loads) at every phase boundary - in the
// persistent block kernels that was 3 store round trips per tile.  Use where the phases communicate through LDS only.
// "Every VMEM load issued so far has landed" as an s_waitcnt the COMPILER sees (vmcnt(0), expcnt / lgkmcnt untouched).
// Put once between the prologue of a persistent kernel and its tile loop when MFMA operands live in registers that the
// prologue loaded: otherwise the waitcnt pass must guard their first use INSIDE the loop with counted vmcnt waits, and
// because vmcnt retires in order those waits also cover the next tile's prefetch - MFMAs stalled on HBM latency every tile.
__device__ __forceinline__ void nsc_wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }
__device__ __forceinline__ void nsc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define NSC_REQUIRE(cond, code, ...)          \
  do {                                        \
    if (!(cond)) {                            \
      nsc_set_error(__VA_ARGS__);             \
      return (code);                          \
    }                                         \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: asked for once per (kernel instantiation,
// device) - a process that touches a second GPU sets it there too - and a failure is not remembered (the next launch asks again).
#include <atomic>
inline hipError_t nsc_smem_attr(const void* kern, int bytes, std::atomic<unsigned long long>& done) {
  int dv = 0;
  hipError_t e = hipGetDevice(&dv);
  if (e != hipSuccess) return e;
  const bool cached = dv >= 0 && dv < 64;
  if (cached && ((done.load(std::memory_order_relaxed) >> dv) & 1ull)) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && cached) done.fetch_or(1ull << dv, std::memory_order_relaxed);
  return e;
}
#define NSC_SMEM_ATTR(kern, bytes)                                             \
  ([&]() -> hipError_t {                                                       \
    static std::atomic<unsigned long long> done_{0};                           \
    return nsc_smem_attr((const void*)(kern), (int)(bytes), done_);            \
  })()

#define NSC_CHECK_LAUNCH(name)                                                     \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      nsc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
      return NSC_ERR_LAUNCH;                                                       \
    }                                                                              \
  } while (0)

// tanh as the reference's own arithmetic computes it: TensorFlow 1.x evaluates tanh through Eigen's float kernel (Eigen 3.3
// MathFunctionsImpl.h, generic_fast_tanh_float - a dependency absent from /root/reference; restated from its published form):
// clamp to [-9, 9], then the rational x P(x^2) / Q(x^2) with the 7 + 4 coefficients below.  ~3 ulp over the whole line, and
// RELATIVE accuracy near 0 (a 1 - 2 / (exp(2x) + 1) form has 2e-7 absolute error there: the followers' residual inputs are
// small, and the gradients of a follower step came out 1e-2 off).  14 VALU instructions against ~40 of the library tanhf -
// whose time next to an fp32 MFMA stream is not hidden (fp32 MFMA runs on the vector ALUs).  The division is v_rcp + mul; the
// result is clamped to [-1, 1] (the quotient can exceed 1 by an ulp; 1 - th^2 must not go negative); NaN stays NaN.
__device__ __forceinline__ float nsc_tanh(float x0) {
  const float x = __builtin_amdgcn_fmed3f(x0, -9.f, 9.f);
  const float x2 = x * x;
  float p = fmaf(x2, -2.76076847742355e-16f, 2.00018790482477e-13f);
  p = fmaf(x2, p, -8.60467152213735e-11f);
  p = fmaf(x2, p, 5.12229709037114e-08f);
  p = fmaf(x2, p, 1.48572235717979e-05f);
  p = fmaf(x2, p, 6.37261928875436e-04f);
  p = fmaf(x2, p, 4.89352455891786e-03f);
  p *= x;
  float q = fmaf(x2, 1.19825839466702e-06f, 1.18534705686654e-04f);
  q = fmaf(x2, q, 2.26843463243900e-03f);
  q = fmaf(x2, q, 4.89352518554385e-03f);
  const float r = __builtin_amdgcn_fmed3f(p * __builtin_amdgcn_rcpf(q), -1.f, 1.f);
  return x0 != x0 ? x0 : r;                 // the clamps would turn a NaN into -1: a diverged run must stay visibly NaN
}

__device__ __forceinline__ float nsc_apply_act(float v, int act) {
  if (act == NSC_ACT_TANH) return nsc_tanh(v);
  if (act == NSC_ACT_LRELU) return v > 0.f ? v : NSC_LRELU_ALPHA * v;
  return v;
}
__device__ __forceinline__ float nsc_act_grad_from_out(float out, int mode) {
  if (mode == 1) return out > 0.f ? 1.f : NSC_LRELU_ALPHA;  // lrelu'(z) from y = lrelu(z): sign(y) == sign(z)
  if (mode == 2) return 1.f - out * out;                    // tanh'
  return 1.f;
}
// Sum over the 64 lanes, returned to every lane.  DPP moves inside the VALU (quad permutes, row rotates, then the two
// row broadcasts of gfx9) instead of six ds_bpermute round trips through the LDS pipe: kernels that end in a dozen of
// these per wave (depthwise / quantizer gradients) were bound by exactly that.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float nsc_dpp_add(float v) {
  const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, m);      // rows outside ROW_MASK receive 0
}
__device__ __forceinline__ float wave_sum(float v) {
  v = nsc_dpp_add<0xb1>(v);          // quad_perm [1,0,3,2]
  v = nsc_dpp_add<0x4e>(v);          // quad_perm [2,3,0,1]   -> quad sums
  v = nsc_dpp_add<0x124>(v);         // row_ror 4
  v = nsc_dpp_add<0x128>(v);         // row_ror 8             -> row (16-lane) sums in every lane
  v = nsc_dpp_add<0x142, 0xa>(v);    // row_bcast 15 into rows 1, 3
  v = nsc_dpp_add<0x143, 0xc>(v);    // row_bcast 31 into rows 2, 3 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Masked load WITHOUT a branch: hipcc turns `ok ? p[i] : 0.f` into a branch around the load plus a vmcnt(0) wait per
// element, which serialises every weight fetch (guide: "register or load select" trap).  Load unconditionally from a
// clamped (always valid) index and select on the VALUE instead.
__device__ __forceinline__ float nsc_ldm(const float* __restrict__ p, int idx, bool ok) {
  const float v = p[ok ? idx : 0];
  return ok ? v : 0.f;
}
// Stage rows of a [rows, Tin] tensor into LDS: xs[r*ldx + j] = src[r*Tin + u0 + j] for j < width, zero outside the
// tensor / for pad rows (r >= rows_valid) / for j >= width.  One wave per row, lanes along time (coalesced 256-B reads).
// in_up: virtual zero-upsampled-by-2 source (u even -> src[u/2], odd -> 0), virtual length Tvirt = 2*Tin.
template <int NW = 4, int U = 8>   // waves in the workgroup, rows in flight per wave
__device__ __forceinline__ void nsc_stage_rows(float* __restrict__ xs, int ldx, int rows_total, int rows_valid, int width,
                                               const float* __restrict__ src, int Tin, int u0, int Tvirt, int in_up,
                                               int wave, int lane, int jspan = -1) {
  // U rows per batch: all U loads are issued before the first LDS store, so one memory round trip covers U rows
  // (a load->store pair per loop iteration left every row waiting on its own latency).  Loads are branch-free:
  // clamped address + select on the value.
  // jspan: columns to write (default: the whole row stride ldx, zero-filling [width, ldx)); pass `width` when the
  // columns beyond it are never read, to skip the extra pass.
  if (jspan < 0) jspan = ldx;
  for (int jb = 0; jb < jspan; jb += 64) {
    const int j = jb + lane;
    const int u = u0 + j;
    const bool cok = j < width && u >= 0 && u < Tvirt && (!in_up || !(u & 1));
    const int uoff = cok ? (in_up ? (u >> 1) : u) : 0;
    for (int r0 = wave * U; r0 < rows_total; r0 += NW * U) {
      float v[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int r = r0 + q;
        const bool ok = cok && r < rows_valid;
        const float ld = src[(long)(ok ? r : 0) * Tin + uoff];
        v[q] = ok ? ld : 0.f;
      }
      if (j < ldx) {
#pragma unroll
        for (int q = 0; q < U; ++q)
          if (r0 + q < rows_total) xs[(r0 + q) * ldx + j] = v[q];
      }
    }
  }
}
// ---- fp32 -> three bf16 pieces (hi + lo + lo2 = x exactly: 3 x 8 significand bits; round to nearest even) ----
// Used by the split-operand block kernels (block_split.hip) and by the gather that builds their parameter images (misc.hip).
typedef __bf16 nsc_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned nsc_cvt2(float a, float b) {          // packed: low half = bf16(a), high half = bf16(b)
  const nsc_bf2 v = {(__bf16)a, (__bf16)b};                               // v_cvt_pk_bf16_f32; a NaN stays a NaN
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void nsc_split2(float a, float b, unsigned (&p)[3]) {
  p[0] = nsc_cvt2(a, b);
  float ra = a - __builtin_bit_cast(float, p[0] << 16), rb = b - __builtin_bit_cast(float, p[0] & 0xffff0000u);
  p[1] = nsc_cvt2(ra, rb);
  ra -= __builtin_bit_cast(float, p[1] << 16);
  rb -= __builtin_bit_cast(float, p[1] & 0xffff0000u);
  p[2] = nsc_cvt2(ra, rb);
}
static inline int nsc_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
