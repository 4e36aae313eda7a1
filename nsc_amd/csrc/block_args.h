// Argument blocks and small helpers shared by the gated-block kernels (block.hip: v1 / v2, block3.hip: v3).
#pragma once
#include "nsc_common.h"

#define NARROW 20
#define K15 15
#define K9 9

struct BlockArgs {
  int B, C, T, dil, flat;
  const float *x, *w1, *b1, *wl, *bl, *wr, *br, *w9, *b9;
  float *out, *h_out, *lin_out, *th_out, *g_out;  // *_out optional: saved for the unfused backward
  int Cin;   // input channels: C (residual block) or 1 (the first decoder block: x [B,1,T] is broadcast into the residual add)
  const float* img;   // nullable: kernel-ready image of all eight parameter tensors (nsc_gated_block_image_index, which = 0)
};

struct BlockDgradArgs {
  int B, C, T, dil, in_act;
  const float *x, *h, *lin, *th, *dy;
  const float *wt1, *wtl, *wtr, *wt9;
  float *dx, *da, *dz1;
  float* dgate;   // where the second half of da goes and the rows per frame of both halves: da + 20 T / 40 for the joint
  int da_rows;    // [B,40,T] tensor the block weight-gradient kernel reads; a separate [B,20,T] tensor / 20 for per-conv wgrads
  const float* img;   // nullable: kernel-ready image of the four flipped / transposed kernels (which = 1)
};

// ---- weight-gradient kernels of a gated block (block.hip: exact fp32; block_split.hip: split operands on the bf16 matrix cores) ----
struct BlockWgradArgs {
  int B, C, T, dil;
  const float *x, *h, *g, *dy, *da, *dz1;   // da [B,40,T] = dlin | dgate
  const float* wt1;                          // nullable: flipped/transposed 1x1 kernel [20][C]; with dx enables the fused
  float* dx;                                 //   data gradient dx = (W1^T dz1 + dy) * act'(x)
  int in_act;
  float *dw1, *db1, *dwl, *dbl, *dwr, *dbr, *dw9, *db9;   // atomics into the gradients, or (slab != 0) plain stores into
  int ntiles, tiles_per_frame;                             // this workgroup's private partial slab (same relative layout)
  long slab_stride;                                        // floats between consecutive workgroups' slabs (0 = atomics)
  int skip;   // timing-only probe (NSC_WG_SKIP): 1 D1, 2 wgrad MFMA loop, 4 flush, 8 staging loads, 16 staging stores
  int Cin;    // channels of x (= C, or 1 for the first block of a decoder stage: dW1 is then [1,20] and x one row)
};

#define NSC_WG_MAXJ 12
struct BlockWgradBatch {
  BlockWgradArgs a[NSC_WG_MAXJ];
  int wg0[NSC_WG_MAXJ + 1];
  int njobs;
};

// the split-operand form of gated_block_wgrad_batch_kernel<RT9> on the same job table (block_split.hip); ok_shapes: every job has
// T % 4 == 0 and 16-byte aligned tensors (else the caller launches the exact kernel)
bool nsc_block_wgrad_split_ok(const BlockWgradBatch& t);
int nsc_launch_block_wgrad_split(const BlockWgradBatch& t, int rt9, int nwg, hipStream_t st);

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
