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

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
