// upsample.hip - the decoder's up-sampling stage as ONE kernel per direction (gfx950).
// Replaces, for the shapes the codec uses (C = 100 or 50 channels, K = 9):
//   forward   SeparableConv1D(C, 9, 'same') -> leaky-relu -> sub-pixel shuffle [B,C,T] -> [B,C/2,2T]
//             (nn_core_operator.py:17-21 conv1d_depth, neural_speech_coding_module.py:168-181 _up_sampling_mod)
//   backward  un-shuffle -> pointwise^T -> depthwise^T (the data path of tf.gradients through the same three ops)
// The unfused path was depthwise (10 us) + 1x1 conv (29 us: every workgroup re-fetched all 175 weight fragments per wave,
// 46 MB of L2 reads per launch, 15 us with NO staging and NO MFMAs) [+ un-shuffle 12 us + depthwise^T 10 us backward] for
// 13 MB in and 13 MB out.  Here a workgroup owns 64 time steps of one frame: the x tile is staged once, the depthwise stage
// runs LDS -> LDS on the VALU, its result is the B operand of the pointwise MFMAs (A = this wave's row tile of the
// pointwise kernel, NK registers), and the epilogue leaves through LDS as whole shuffled 256-B rows.  64 KB of LDS and
// <= 128 registers: two workgroups share a CU, so one's loads / stores overlap the other's MFMAs.
#include "nsc_common.h"
// the kernels' whole-row stores are nontemporal (up-sampling class 0.103 -> 0.096 ms per step, profiles/r04h_store_flavours.txt)
#define UP_ST(p_, v_) __builtin_nontemporal_store((float)(v_), p_)
#include <algorithm>
#include <type_traits>

namespace {
constexpr int UTT = 64;    // time steps per tile
constexpr int ULD = 80;    // LDS row stride: == 16 (mod 32), lanes 0-15 / 16-31 of a B fragment read disjoint banks
constexpr int ULO = 68;    // row stride of the output tile
constexpr int UK = 9;      // depthwise taps (SAME: 4 | 4)

__device__ __forceinline__ f32x4 up_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float up_bld(const __amdgpu_buffer_rsrc_t& r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
constexpr int UP_OOB = 0x7ffffff0;   // past every descriptor: the hardware bounds check returns 0

struct UpFwdArgs {
  const float *x, *wd, *wp, *bias;
  float *dwo, *y;
  int B, C, T, act;
};
struct UpBwdArgs {
  const float *dz, *wd, *wp;
  float *dzp, *ddw, *dx;
  int B, C, T;
};

// RT row tiles of 16 output channels, NK k-steps of 4 input channels (C = 100: 7, 25;  C = 50: 4, 13).
// RT == 7: wave w < 7 owns row tile w and every column tile; RT == 4: wave = (row tile w & 3, column half w >> 2).
template <int RT, int NK>
__global__ __launch_bounds__(512, 4) void upsample_fwd_kernel(UpFwdArgs a, int tpf, int skip) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int C4 = 4 * NK, NQ = (C4 + 7) / 8;
  float* xs = sm;                // [C4][ULD]  x on [t0 - 4, t0 + 68); later the output tile os [C][ULO]
  float* dws = xs + C4 * ULD;    // [C4][ULD]  depthwise output on [t0, t0 + 64): B operand of the pointwise MFMAs
  const int C = a.C, T = a.T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x / tpf, t0 = (blockIdx.x - b * tpf) * UTT;

  // ---- x tile: wave w rows w, w + 8, ...; lanes along time (64 + 8 columns); out-of-frame columns read 0 ----
  const __amdgpu_buffer_rsrc_t sx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((long)a.B * C * T * 4), 0x00020000);
  float px[NQ][2];
  {
    const int ta = t0 - 4 + lane, tb = t0 + 60 + lane;
    const int va = (ta >= 0 && ta < T) ? ta * 4 : UP_OOB;
    const int vb = (lane < 8 && tb < T) ? tb * 4 : UP_OOB;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int so = (b * C + min(wave + 8 * q, C - 1)) * T * 4;
      px[q][0] = (skip & 1) ? 1.f : up_bld(sx, va, so);
      px[q][1] = (skip & 1) ? 1.f : up_bld(sx, vb, so);
    }
  }
  // ---- this wave's fragments of the pointwise kernel wp [ci][co] (A: row = co, k = ci) and its bias ----
  const int rt = RT == 7 ? (wave < 7 ? wave : 6) : (wave & 3);
  const int cb = RT == 7 ? 0 : (wave >> 2) * 32;
  constexpr int NC = RT == 7 ? 4 : 2;
  float ar[NK], br[4];
#pragma unroll
  for (int u = 0; u < NK; ++u) ar[u] = (skip & 16) ? 0.5f : a.wp[min(4 * u + kq, C - 1) * C + min(16 * rt + l15, C - 1)];   // ci >= C: zero B rows
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) br[reg] = a.bias ? a.bias[min(16 * rt + 4 * kq + reg, C - 1)] : 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int r = wave + 8 * q;
    if (r < C4) {
      xs[r * ULD + lane] = r < C ? px[q][0] : 0.f;
      if (lane < 8) xs[r * ULD + 64 + lane] = r < C ? px[q][1] : 0.f;
    }
  }
  nsc_lds_barrier();
  // ---- depthwise stage, LDS -> LDS (+ the copy the backward pass needs) ----
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int r = wave + 8 * q;
    if (r < C4 && !(skip & 2)) {
      // the row's nine taps are wave-uniform: scalar loads (s_load through the constant cache), no LDS table
      const float* wr = a.wd + min(r, C - 1);
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < UK; ++k) v = fmaf(xs[r * ULD + lane + k], wr[k * C], v);
      dws[r * ULD + lane] = v;                        // rows >= C: x is zero
      if (a.dwo && r < C && t0 + lane < T) UP_ST(a.dwo + ((long)b * C + r) * T + t0 + lane, v);
    }
  }
  nsc_lds_barrier();
  // ---- pointwise conv on the matrix pipe ----
  float* os = xs;                                      // the x tile is dead
  if (RT != 7 || wave < 7) {
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* bb = dws + kq * ULD + cb + l15;
    if (!(skip & 4))
#pragma unroll
    for (int u = 0; u < NK; ++u)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = up_mfma(ar[u], bb[4 * u * ULD + 16 * c], acc[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = 16 * rt + 4 * kq + reg;
        if (co < C) {
          float v = acc[c][reg] + br[reg];
          if (a.act == NSC_ACT_LRELU) v = v > 0.f ? v : NSC_LRELU_ALPHA * v;
          os[co * ULO + cb + 16 * c + l15] = v;
        }
      }
  }
  nsc_lds_barrier();
  // ---- sub-pixel shuffle on the way out: output row oc interleaves channels 2 oc, 2 oc + 1 in time (256-B lines) ----
  const int Ch = C >> 1;
  for (int oc = wave; oc < Ch; oc += 8) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int tl2 = hb * 64 + lane, par = tl2 & 1, tl = tl2 >> 1;
      const int t2 = 2 * t0 + tl2;
      if (t2 < 2 * T && !(skip & 8)) UP_ST(a.y + ((long)b * Ch + oc) * (2L * T) + t2, os[(2 * oc + par) * ULO + tl]);
    }
  }
}

// Backward data path: dz [B, C/2, 2T] (gradient w.r.t. the PRE-activation of the shuffled output) ->
//   dzp [B,C,T]  = dz un-shuffled                      (the pointwise weight gradient reads it)
//   ddw [B,C,T]  = wp^T-applied: ddw[ci] = sum_co wp[ci][co] dzp[co]     (the depthwise weight gradient reads it)
//   dx  [B,C,T]  = depthwise^T: dx[c][t] = sum_k wd[k][c] ddw[c][t - k + 4]
// A tile needs ddw on [t0 - 4, t0 + 68): the pointwise stage runs on five column tiles (80 columns) instead of four.
template <int RT, int NK>
__global__ __launch_bounds__(512, 4) void upsample_bwd_kernel(UpBwdArgs a, int tpf) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int C4 = 4 * NK, NQ = (C4 + 7) / 8, NQH = (C4 / 2 + 7) / 8;
  float* dzs = sm;               // [C4][ULD]  dzp on [t0 - 4, t0 + 76): rows = co (B operand)
  float* dds = dzs + C4 * ULD;   // [C4][ULD]  ddw on the same columns: rows = ci
  const int C = a.C, T = a.T, Ch = C >> 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x / tpf, t0 = (blockIdx.x - b * tpf) * UTT;

  // ---- dz tile, shuffled layout: row oc holds positions p = 2 t + parity; the tile spans p in [2 (t0 - 4), 2 (t0 + 76)) ----
  const __amdgpu_buffer_rsrc_t sz =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, (unsigned)((long)a.B * C * T * 4), 0x00020000);
  float pz[NQH][3];
  {
    const int p0 = 2 * (t0 - 4);
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      const int i = lane + 64 * h, p = p0 + i;
      const int vo = (i < 2 * ULD && p >= 0 && p < 2 * T) ? p * 4 : UP_OOB;
#pragma unroll
      for (int q = 0; q < NQH; ++q) pz[q][h] = up_bld(sz, vo, (b * Ch + min(wave + 8 * q, Ch - 1)) * (2 * T) * 4);
    }
  }
  const int rt = RT == 7 ? (wave < 7 ? wave : 6) : (wave & 3);
  float ar[NK];
#pragma unroll
  for (int u = 0; u < NK; ++u) ar[u] = a.wp[min(16 * rt + l15, C - 1) * C + min(4 * u + kq, C - 1)];   // A: row = ci, k = co
  for (int e = tid; e < (C4 - C) * ULD; e += 512) dzs[C * ULD + e] = 0.f;      // k rows >= C multiply zeros
#pragma unroll
  for (int q = 0; q < NQH; ++q) {
    const int oc = wave + 8 * q;
    if (oc < Ch) {
#pragma unroll
      for (int h = 0; h < 3; ++h) {
        const int i = lane + 64 * h;
        if (i < 2 * ULD) dzs[(2 * oc + (i & 1)) * ULD + (i >> 1)] = pz[q][h];
      }
    }
  }
  nsc_lds_barrier();
  // ---- un-shuffled copy out (this tile's own 64 columns) ----
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int r = wave + 8 * q;
    if (r < C && t0 + lane < T) UP_ST(a.dzp + ((long)b * C + r) * T + t0 + lane, dzs[r * ULD + 4 + lane]);
  }
  // ---- pointwise^T on the matrix pipe: 80 columns ----
  auto mma = [&](auto nc_tag, int ctb) {
    constexpr int NCC = decltype(nc_tag)::value;
    f32x4 acc[NCC];
#pragma unroll
    for (int c = 0; c < NCC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* bb = dzs + kq * ULD + 16 * ctb + l15;
#pragma unroll
    for (int u = 0; u < NK; ++u)
#pragma unroll
      for (int c = 0; c < NCC; ++c) acc[c] = up_mfma(ar[u], bb[4 * u * ULD + 16 * c], acc[c]);
#pragma unroll
    for (int c = 0; c < NCC; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ci = 16 * rt + 4 * kq + reg;
        if (ci < C) dds[ci * ULD + 16 * (ctb + c) + l15] = acc[c][reg];
      }
  };
  if (RT == 7) {
    if (wave < 7) mma(std::integral_constant<int, 5>{}, 0);
  } else if (wave < 4) {
    mma(std::integral_constant<int, 3>{}, 0);      // column tiles 0..2
  } else {
    mma(std::integral_constant<int, 2>{}, 3);      // column tiles 3, 4
  }
  nsc_lds_barrier();
  // ---- ddw copy out + depthwise^T ----
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int r = wave + 8 * q;
    if (r < C && t0 + lane < T) {
      const float* dr = dds + r * ULD + lane;
      const float* wr = a.wd + r;                      // wave-uniform taps: scalar loads
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < UK; ++k) v = fmaf(dr[8 - k], wr[k * C], v);
      const long o = ((long)b * C + r) * T + t0 + lane;
      UP_ST(a.ddw + o, dr[4]);
      UP_ST(a.dx + o, v);
    }
  }
}

template <int RT, int NK>
int launch_up_fwd(const UpFwdArgs& a, hipStream_t st) {
  const size_t smem = (size_t)2 * 4 * NK * ULD * sizeof(float);
  auto kern = upsample_fwd_kernel<RT, NK>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "upsample_fwd: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a.T, UTT);
  static const int skip = NSC_PROBE_INT("NSC_UP_SKIP", 0);   // timing probe (PROBES build only)
  hipLaunchKernelGGL(kern, dim3(a.B * tpf), dim3(512), smem, st, a, tpf, skip);
  NSC_CHECK_LAUNCH("upsample_fwd");
  return NSC_OK;
}
template <int RT, int NK>
int launch_up_bwd(const UpBwdArgs& a, hipStream_t st) {
  const size_t smem = (size_t)2 * 4 * NK * ULD * sizeof(float);
  auto kern = upsample_bwd_kernel<RT, NK>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "upsample_bwd: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a.T, UTT);
  hipLaunchKernelGGL(kern, dim3(a.B * tpf), dim3(512), smem, st, a, tpf);
  NSC_CHECK_LAUNCH("upsample_bwd");
  return NSC_OK;
}
}  // namespace

extern "C" int nsc_upsample_fwd(const float* x, const float* wd, const float* wp, const float* bias, float* dwo, float* y,
                                int B, int C, int T, int K, int act, void* stream) {
  NSC_REQUIRE(x && wd && wp && y && B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_upsample_fwd: bad args");
  NSC_REQUIRE(K == UK && (C == 100 || C == 50), NSC_ERR_UNSUPPORTED, "nsc_upsample_fwd: built for K = 9, C in {100, 50} (got %d, %d)", K, C);
  NSC_REQUIRE(act == NSC_ACT_NONE || act == NSC_ACT_LRELU, NSC_ERR_BAD_ARG, "nsc_upsample_fwd: act must be none|lrelu");
  NSC_REQUIRE((long)B * C * T < (1L << 29), NSC_ERR_UNSUPPORTED, "nsc_upsample_fwd: tensor too large for 32-bit byte offsets");
  UpFwdArgs a{x, wd, wp, bias, dwo, y, B, C, T, act};
  return C == 100 ? launch_up_fwd<7, 25>(a, (hipStream_t)stream) : launch_up_fwd<4, 13>(a, (hipStream_t)stream);
}

extern "C" int nsc_upsample_bwd(const float* dz, const float* wd, const float* wp, float* dzp, float* ddw, float* dx, int B,
                                int C, int T, int K, void* stream) {
  NSC_REQUIRE(dz && wd && wp && dzp && ddw && dx && B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_upsample_bwd: bad args");
  NSC_REQUIRE(K == UK && (C == 100 || C == 50), NSC_ERR_UNSUPPORTED, "nsc_upsample_bwd: built for K = 9, C in {100, 50} (got %d, %d)", K, C);
  NSC_REQUIRE((long)B * C * T < (1L << 29), NSC_ERR_UNSUPPORTED, "nsc_upsample_bwd: tensor too large for 32-bit byte offsets");
  UpBwdArgs a{dz, wd, wp, dzp, ddw, dx, B, C, T};
  return C == 100 ? launch_up_bwd<7, 25>(a, (hipStream_t)stream) : launch_up_bwd<4, 13>(a, (hipStream_t)stream);
}
