// Device helpers of the soft-to-hard quantizer shared by quant.hip and the fused quantizer epilogue of the C -> 1 conv (conv.hip).
#pragma once
#include "nsc_common.h"
#define QEPS 1e-20f

// Cross-lane exchange inside a row of 16 lanes with DPP (VALU data path, no LDS crossbar like ds_bpermute):
// xor 1 / xor 2 = quad permutes, "xor 4" = row_half_mirror (lane i <-> 7-i inside each 8 lanes), "xor 8" = row_mirror
// (i <-> 15-i).  After the quad steps every lane of a quad holds the quad total, so the mirrors are valid partners for
// the symmetric reductions (sum, max) used here.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int LPC>
__device__ __forceinline__ float grp_sum(float v) {
  if constexpr (LPC >= 2) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (LPC >= 4) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (LPC >= 8) v += dpp_f<0x141>(v);   // row_half_mirror
  if constexpr (LPC >= 16) v += dpp_f<0x140>(v);  // row_mirror
  if constexpr (LPC >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (LPC >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
// max with a DPP source operand in ONE instruction.  fmaxf(v, dpp_f(v)) costs five (mov, nop, mov_dpp, a canonicalising
// v_max x,x that llvm.maxnum needs for signalling NaNs, max); the values here are never NaN.
#define NSC_MAX_DPP(v, ctrl)                                                                                   \
  asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "0"(v))
template <int LPC>
__device__ __forceinline__ float grp_max(float v) {
  if constexpr (LPC >= 2) NSC_MAX_DPP(v, "quad_perm:[1,0,3,2]");
  if constexpr (LPC >= 4) NSC_MAX_DPP(v, "quad_perm:[2,3,0,1]");
  if constexpr (LPC >= 8) NSC_MAX_DPP(v, "row_half_mirror");
  if constexpr (LPC >= 16) NSC_MAX_DPP(v, "row_mirror");
  if constexpr (LPC >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  if constexpr (LPC >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

#define NSC_MIN_DPP(v, ctrl)                                                                                   \
  asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v) : "0"(v))
template <int LPC>
__device__ __forceinline__ float grp_min8(float v) {      // LPC <= 8 (the wave-per-frame kernel)
  if constexpr (LPC >= 2) NSC_MIN_DPP(v, "quad_perm:[1,0,3,2]");
  if constexpr (LPC >= 4) NSC_MIN_DPP(v, "quad_perm:[2,3,0,1]");
  if constexpr (LPC >= 8) NSC_MIN_DPP(v, "row_half_mirror");
  return v;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Computes this lane's p values for code c.  Returns d_k = |c-b_k| in dist[], p in p[].
template <int LPC, int ITER>
__device__ __forceinline__ void softmax_bins(float c, float alpha, const float (&bv)[ITER][4], const bool (&ok)[ITER][4],
                                             float (&dist)[ITER][4], float (&p)[ITER][4]) {
  // everything in the log2 domain: z = (alpha log2 e) d, p ~ 2^(z - max z): one fma + v_exp_f32 per bin.  |z - m| <= ~30
  // wherever p matters, so folding log2 e into alpha costs < 2e-6 rel.
  const float a2 = alpha * 1.4426950408889634f;
  float z[ITER][4];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dist[i][j] = fabsf(c - bv[i][j]);
      z[i][j] = ok[i][j] ? a2 * dist[i][j] : -INFINITY;
      m = fmaxf(m, z[i][j]);
    }
  m = grp_max<LPC>(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      p[i][j] = ok[i][j] ? __builtin_amdgcn_exp2f(z[i][j] - m) : 0.f;
      s += p[i][j];
    }
  s = grp_sum<LPC>(s);
  const float inv = __builtin_amdgcn_rcpf(s);     // v_rcp_f32 (1 ulp); __frcp_rn expands to the 11-instruction IEEE division
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) p[i][j] *= inv;
}

// lowest-index argmax over the LPC-lane group (tf.nn.top_k tie rule)
template <int LPC>
__device__ __forceinline__ int grp_argmax(float best, int idx) {
#pragma unroll
  for (int o = LPC / 2; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
  }
  return idx;
}

