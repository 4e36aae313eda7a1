// misc.hip - HBM-bound glue kernels of the NSC/CMRL hot path: depthwise part of SeparableConv1D, the GLU
// gate, cascade arithmetic, layout permutations, TF1 Adam, framing / overlap-add.  All fp32, [B, C, T].
#include "nsc_common.h"
#include <algorithm>

// ---------------- depthwise conv (Keras SeparableConv1D depthwise stage; nn_core_operator.py:17-21) ----------------
// One thread per 4 consecutive outputs: the 12-sample window it needs is three aligned float4 loads (first / last block
// zero at the frame edges), the taps are unrolled for K <= 9 (generic K takes the scalar kernel).  REV = 1 runs the taps
// backwards: the data gradient dx[c,t] = sum_k dy[c, t-k+padL] wd[k,c] is the same correlation with the taps flipped.
template <bool REV>
__global__ void depthwise_vec4_kernel(const float* __restrict__ x, const float* __restrict__ wd, float* __restrict__ y,
                                      int C, int T, int K, long n4) {
  const int padL = (K - 1) / 2;                      // SAME, stride 1 (K odd here: padL == padR)
  const int T4 = T >> 2;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    const int q = (int)(e % T4);
    const long row = e / T4;
    const int c = (int)(row % C);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * T);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 a = q > 0 ? xr[q - 1] : z, m = xr[q], b = q + 1 < T4 ? xr[q + 1] : z;
    const float w[12] = {a[0], a[1], a[2], a[3], m[0], m[1], m[2], m[3], b[0], b[1], b[2], b[3]};   // window: t-4 .. t+7
    f32x4 acc = z;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (k < K) {
        const float wk = wd[(REV ? K - 1 - k : k) * C + c];
        // forward: out[t+i] += wk * x[t+i + k - padL]  -> window index 4 + i + k - padL
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[4 + i + k - 4 + (4 - padL)], wk, acc[i]);
      }
    }
    reinterpret_cast<f32x4*>(y + row * T)[q] = acc;
  }
}
__global__ void depthwise_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wd, float* __restrict__ y,
                                     int C, int T, int K, long n) {
  const int padL = (K - 1) / 2;  // SAME, stride 1: pad = K-1, padL = (K-1)//2
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e % T);
    const int c = (int)((e / T) % C);
    const float* xr = x + (e - t);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int u = t + k - padL;
      if (u >= 0 && u < T) acc = fmaf(xr[u], wd[k * C + c], acc);
    }
    y[e] = acc;
  }
}
static bool depthwise_vec_ok(int T, int K) { return (T & 3) == 0 && (K & 1) && K <= 9; }
extern "C" int nsc_depthwise_fwd(const float* x, const float* wd, float* y, int B, int C, int T, int K, void* stream) {
  NSC_REQUIRE(x && wd && y && B > 0 && C > 0 && T > 0 && K > 0, NSC_ERR_BAD_ARG, "nsc_depthwise_fwd: bad args");
  const long n = (long)B * C * T;
  if (depthwise_vec_ok(T, K)) {
    hipLaunchKernelGGL(depthwise_vec4_kernel<false>, dim3(std::min<long>(4096, nsc_cdiv(n / 4, 256))), dim3(256), 0,
                       (hipStream_t)stream, x, wd, y, C, T, K, n / 4);
  } else {
    hipLaunchKernelGGL(depthwise_fwd_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0,
                       (hipStream_t)stream, x, wd, y, C, T, K, n);
  }
  NSC_CHECK_LAUNCH("depthwise_fwd");
  return NSC_OK;
}

// dx[c,t] = sum_k dy[c, t-k+padL] wd[k,c] ; dwd[k,c] += sum_{b,t} x[c,t+k-padL] dy[c,t]
__global__ void depthwise_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ wd,
                                        float* __restrict__ dx, int C, int T, int K, long n) {
  const int padL = (K - 1) / 2;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e % T);
    const int c = (int)((e / T) % C);
    const float* dr = dy + (e - t);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int u = t - k + padL;
      if (u >= 0 && u < T) acc = fmaf(dr[u], wd[k * C + c], acc);
    }
    dx[e] = acc;
  }
}
// grid (C, nsplit): block handles channel c, frames b = blockIdx.y, += gridDim.y.  The x row (with its K-1 halo) is staged
// in LDS once per frame, so a thread's K taps are LDS reads at consecutive addresses instead of K guarded global loads.
__global__ __launch_bounds__(256) void depthwise_bwd_dw_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ dy, float* __restrict__ dwd,
                                                               int B, int C, int T, int K) {
  extern __shared__ float xrow[];            // [T + 16]: xrow[j] = x[j - padL], zero outside the frame
  __shared__ float red[4][16];
  const int c = blockIdx.x, padL = (K - 1) / 2;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const float* xr = x + ((long)b * C + c) * T;
    const float* dr = dy + ((long)b * C + c) * T;
    __syncthreads();
    for (int j = threadIdx.x; j < T + 16; j += 256) {
      const int u = j - padL;
      xrow[j] = (u >= 0 && u < T) ? xr[u] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
      const float g = dr[t];
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < K) acc[k] = fmaf(xrow[t + k], g, acc[k]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    atomicAdd(dwd + k * C + c, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
  }
}
// The same with four frames in flight: all their rows are fetched before the first one is used (the kernel above pays a
// global-load round trip per frame, ~2.5 us each, with a barrier in between).  T + 16 <= 768.
#define DWB_NF 4
__global__ __launch_bounds__(256) void depthwise_bwd_dw4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ dwd, int B, int C, int T, int K) {
  extern __shared__ float xrow[];            // [DWB_NF][T + 16]
  __shared__ float red[4][16];
  const int c = blockIdx.x, padL = (K - 1) / 2, ld = T + 16;
  const int per = (B + gridDim.y - 1) / gridDim.y;
  const int b_lo = blockIdx.y * per, b_hi = min(B, b_lo + per);
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  for (int b0 = b_lo; b0 < b_hi; b0 += DWB_NF) {
    float xv[DWB_NF][3], gv[DWB_NF][3];
#pragma unroll
    for (int f = 0; f < DWB_NF; ++f) {
      const int b = min(b0 + f, b_hi - 1);                    // duplicate frames past the end are loaded, never used
      const float* xr = x + ((long)b * C + c) * T;
      const float* dr = dy + ((long)b * C + c) * T;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int j = threadIdx.x + 256 * q, u = j - padL;
        xv[f][q] = (j < ld && u >= 0 && u < T) ? xr[u] : 0.f;
        gv[f][q] = j < T ? dr[j] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < DWB_NF; ++f)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int j = threadIdx.x + 256 * q;
        if (j < ld) xrow[f * ld + j] = xv[f][q];
      }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < DWB_NF; ++f) {
      const bool live = b0 + f < b_hi;                        // block-uniform
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int t = threadIdx.x + 256 * q;
        if (live && t < T) {
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (k < K) acc[k] = fmaf(xrow[f * ld + t + k], gv[f][q], acc[k]);
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    atomicAdd(dwd + k * C + c, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
  }
}
// Round 4: the weight gradient WITHOUT the LDS row.  The two kernels above stage each x row in LDS and pay K LDS reads per FMA-row
// plus two barriers and a global round trip per group of frames (20 us for 26 MB at B = 128, C = 100, T = 256).  Here a thread owns
// FOUR consecutive time steps of one (frame, channel) row: three 16-byte loads of x (previous / own / next group: the 12-sample
// window of a K <= 9 kernel) and one of dy, 4 K FMAs from registers; a workgroup is one channel x NF4 frames, walks the batch
// grid-stride with every load of its next frames in flight, and ends in one DPP wave sum + K atomics.  T % 4 == 0, K <= 9.
template <int K>
__global__ __launch_bounds__(256) void depthwise_bwd_dw_reg_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ dwd, int B, int C, int T) {
  __shared__ float red[4][K];
  constexpr int padL = (K - 1) / 2;
  static_assert(K <= 9 && padL <= 4, "the window is the previous, own and next group of four");
  const int c = blockIdx.x;
  const int g4 = T >> 2;                                   // groups of four steps per row
  const int nitem = B * g4;                                // (frame, group) items of this channel
  float acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int it = blockIdx.y * 256 + threadIdx.x; it < nitem; it += gridDim.y * 256) {
    const int b = it / g4, g = it - b * g4;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + ((long)b * C + c) * T);
    const f32x4* dr = reinterpret_cast<const f32x4*>(dy + ((long)b * C + c) * T);
    const f32x4 xa = g > 0 ? xr[g - 1] : zero, xb = xr[g], xc = g + 1 < g4 ? xr[g + 1] : zero, d = dr[g];
    const float w[12] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3], xc[0], xc[1], xc[2], xc[3]};
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[k] = fmaf(w[4 + e + k - padL], d[e], acc[k]);     // x[4g + e + k - padL] dy[4g + e]
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float s_ = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s_;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    atomicAdd(dwd + k * C + c, (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]));
  }
}

extern "C" int nsc_depthwise_bwd(const float* x, const float* wd, const float* dy, float* dx, float* dwd, int B, int C,
                                 int T, int K, void* stream) {
  NSC_REQUIRE(x && wd && dy && B > 0 && C > 0 && T > 0 && K > 0, NSC_ERR_BAD_ARG, "nsc_depthwise_bwd: bad args");
  NSC_REQUIRE(K <= 16, NSC_ERR_UNSUPPORTED, "nsc_depthwise_bwd: K %d > 16", K);
  const long n = (long)B * C * T;
  if (dx && depthwise_vec_ok(T, K)) {
    hipLaunchKernelGGL(depthwise_vec4_kernel<true>, dim3(std::min<long>(4096, nsc_cdiv(n / 4, 256))), dim3(256), 0,
                       (hipStream_t)stream, dy, wd, dx, C, T, K, n / 4);
    NSC_CHECK_LAUNCH("depthwise_bwd_dx");
  } else if (dx) {
    hipLaunchKernelGGL(depthwise_bwd_dx_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0,
                       (hipStream_t)stream, dy, wd, dx, C, T, K, n);
    NSC_CHECK_LAUNCH("depthwise_bwd_dx");
  }
  if (dwd) {
    static const bool no_reg = NSC_PROBE_SET("NSC_DW_NO_REG");     // A/B switch for profiling
    if (K == 9 && (T & 3) == 0 && !no_reg && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0) {
      // ~2 workgroups of one channel per CU-slot: enough loads in flight, few same-address atomics (gy per element)
      const int gy = std::max(1, std::min(nsc_cdiv(B * (T / 4), 256), std::max(1, 2048 / C)));
      hipLaunchKernelGGL(depthwise_bwd_dw_reg_kernel<9>, dim3(C, gy), dim3(256), 0, (hipStream_t)stream, x, dy, dwd, B, C, T);
    } else if (T + 16 <= 768)
      hipLaunchKernelGGL(depthwise_bwd_dw4_kernel, dim3(C, nsc_cdiv(B, DWB_NF)), dim3(256),
                         DWB_NF * (T + 16) * sizeof(float), (hipStream_t)stream, x, dy, dwd, B, C, T, K);
    else
      hipLaunchKernelGGL(depthwise_bwd_dw_kernel, dim3(C, std::min(B, 16)), dim3(256), (T + 16) * sizeof(float),
                         (hipStream_t)stream, x, dy, dwd, B, C, T, K);
    NSC_CHECK_LAUNCH("depthwise_bwd_dw");
  }
  return NSC_OK;
}

// ---------------- GLU gate (nn_core_operator.py:90-100) on a fused [B, 2n, T] pre-activation ----------------
__global__ void gate_fwd_kernel(float* __restrict__ a, float* __restrict__ g, int n, int T, long total) {
  const long nT = (long)n * T;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / nT, r = e - b * nT;
    float* ab = a + b * 2 * nT;
    const float th = nsc_tanh(ab[nT + r]);
    ab[nT + r] = th;
    g[e] = ab[r] * th;
  }
}
__global__ void gate_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dg, float* __restrict__ da,
                                int n, int T, long total) {
  const long nT = (long)n * T;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / nT, r = e - b * nT;
    const float* ab = a + b * 2 * nT;
    float* db = da + b * 2 * nT;
    const float lin = ab[r], th = ab[nT + r], gg = dg[e];
    db[r] = gg * th;
    db[nT + r] = gg * lin * (1.f - th * th);
  }
}
extern "C" int nsc_gate_fwd(float* a, float* g, int B, int n, int T, void* stream) {
  NSC_REQUIRE(a && g && B > 0 && n > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gate_fwd: bad args");
  const long total = (long)B * n * T;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(std::min<long>(4096, nsc_cdiv(total, 256))), dim3(256), 0,
                     (hipStream_t)stream, a, g, n, T, total);
  NSC_CHECK_LAUNCH("gate_fwd");
  return NSC_OK;
}
extern "C" int nsc_gate_bwd(const float* a, const float* dg, float* da, int B, int n, int T, void* stream) {
  NSC_REQUIRE(a && dg && da && B > 0 && n > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gate_bwd: bad args");
  const long total = (long)B * n * T;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(std::min<long>(4096, nsc_cdiv(total, 256))), dim3(256), 0,
                     (hipStream_t)stream, a, dg, da, n, T, total);
  NSC_CHECK_LAUNCH("gate_bwd");
  return NSC_OK;
}

// ---------------- glue ----------------
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                             float a, float b, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = y ? fmaf(a, x[e], b * y[e]) : a * x[e];
}
extern "C" int nsc_axpby(const float* x, const float* y, float* out, float a, float b, long n, void* stream) {
  NSC_REQUIRE(x && out && n > 0, NSC_ERR_BAD_ARG, "nsc_axpby: bad args");
  hipLaunchKernelGGL(axpby_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, x,
                     y, out, a, b, n);
  NSC_CHECK_LAUNCH("axpby");
  return NSC_OK;
}

// One step of the cascade between two codecs (cmrl.py:49-94): decoded (+)= sc * dec and, when another codec follows, its
// input xin = rs * (x - decoded) - the same two fused multiply-adds nsc_axpby would make in two launches.
__global__ void cascade_step_kernel(const float* __restrict__ dec, float* __restrict__ decoded, int accumulate,
                                    const float* __restrict__ x, float* __restrict__ xin, float sc, float rs, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float d = accumulate ? fmaf(sc, dec[e], 1.f * decoded[e]) : sc * dec[e];
    decoded[e] = d;
    if (xin) xin[e] = fmaf(rs, x[e], -rs * d);
  }
}
extern "C" int nsc_cascade_step(const float* dec, float* decoded, int accumulate, const float* x, float* xin, float sc,
                                float rs, long n, void* stream) {
  NSC_REQUIRE(dec && decoded && n > 0 && (!xin || x), NSC_ERR_BAD_ARG, "nsc_cascade_step: bad args");
  hipLaunchKernelGGL(cascade_step_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream,
                     dec, decoded, accumulate, x, xin, sc, rs, n);
  NSC_CHECK_LAUNCH("cascade_step");
  return NSC_OK;
}

// 64 time steps per workgroup; the 4 waves split the channels (independent partial sums, then an LDS reduction): one thread
// per (b,t) walking all C channels was a 100-deep dependent chain on 128 workgroups (25 us for 13 MB)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int T,
                                                          int accumulate, int tiles_per_frame) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / tiles_per_frame, t = (blockIdx.x - b * tiles_per_frame) * 64 + lane;
  const bool live = t < T;
  const float* xb = x + (long)b * C * T + (live ? t : 0);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = wave;
  for (; c + 12 < C; c += 16) {
    s0 += xb[(long)c * T];
    s1 += xb[(long)(c + 4) * T];
    s2 += xb[(long)(c + 8) * T];
    s3 += xb[(long)(c + 12) * T];
  }
  for (; c < C; c += 4) s0 += xb[(long)c * T];
  red[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && live) {
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const long e = (long)b * T + t;
    if (accumulate) out[e] += s;
    else out[e] = s;
  }
}
extern "C" int nsc_channel_sum(const float* x, float* out, int B, int C, int T, int accumulate, void* stream) {
  NSC_REQUIRE(x && out && B > 0 && C > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_channel_sum: bad args");
  const int tpf = nsc_cdiv(T, 64);
  hipLaunchKernelGGL(channel_sum_kernel, dim3(B * tpf), dim3(256), 0, (hipStream_t)stream, x, out, C, T, accumulate, tpf);
  NSC_CHECK_LAUNCH("channel_sum");
  return NSC_OK;
}

// y[b,c,t] = ys[b, c>>1, 2t + (c&1)]   (inverse of the sub-pixel shuffle, nsc_module:158-167)
__global__ void unshuffle2_kernel(const float* __restrict__ ys, float* __restrict__ y, int C, int T, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e % T);
    const int c = (int)((e / T) % C);
    const long b = e / ((long)T * C);
    y[e] = ys[(b * (C >> 1) + (c >> 1)) * (2L * T) + 2 * t + (c & 1)];
  }
}
// ys[b, c>>1, 2t + (c&1)] = y[b,c,t]   (the sub-pixel shuffle itself, nsc_module:158-167, on [B,C,T] tensors: used by the op surface;
// the engine's up-sampling kernel shuffles in its epilogue)
__global__ void shuffle2_kernel(const float* __restrict__ y, float* __restrict__ ys, int C, int T, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long row = e / (2 * T);                 // (b, c2)
    const int u = (int)(e - row * 2 * T);         // 2t + j
    const long b = row / (C / 2);
    const int c2 = (int)(row - b * (C / 2));
    ys[e] = y[(b * C + 2 * c2 + (u & 1)) * T + (u >> 1)];
  }
}
extern "C" int nsc_shuffle2(const float* y, float* ys, int B, int C, int T, void* stream) {
  NSC_REQUIRE(ys && y && B > 0 && C > 0 && !(C & 1) && T > 0, NSC_ERR_BAD_ARG, "nsc_shuffle2: bad args");
  const long n = (long)B * C * T;
  hipLaunchKernelGGL(shuffle2_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, y, ys, C, T, n);
  NSC_CHECK_LAUNCH("shuffle2");
  return NSC_OK;
}
extern "C" int nsc_unshuffle2(const float* ys, float* y, int B, int C, int T, void* stream) {
  NSC_REQUIRE(ys && y && B > 0 && C > 0 && !(C & 1) && T > 0, NSC_ERR_BAD_ARG, "nsc_unshuffle2: bad args");
  const long n = (long)B * C * T;
  hipLaunchKernelGGL(unshuffle2_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0,
                     (hipStream_t)stream, ys, y, C, T, n);
  NSC_CHECK_LAUNCH("unshuffle2");
  return NSC_OK;
}

// y[b, c, r] = x[b, r, c] through a 32x33 LDS tile (channels_last <-> time-contiguous at the op surface)
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ x, float* __restrict__ y, int R,
                                                              int Cc) {
  __shared__ float tile[32][33];
  const long b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? x[(b * R + r) * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < Cc) y[(b * Cc + c) * R + r] = tile[tx][i];
  }
}
extern "C" int nsc_transpose_last2(const float* x, float* y, int B, int R, int Cc, void* stream) {
  NSC_REQUIRE(x && y && B > 0 && R > 0 && Cc > 0, NSC_ERR_BAD_ARG, "nsc_transpose_last2: bad args");
  hipLaunchKernelGGL(transpose_last2_kernel, dim3(nsc_cdiv(Cc, 32), nsc_cdiv(R, 32), B), dim3(256), 0,
                     (hipStream_t)stream, x, y, R, Cc);
  NSC_CHECK_LAUNCH("transpose_last2");
  return NSC_OK;
}

__global__ __launch_bounds__(256) void sum_all_kernel(const float* __restrict__ x, float* __restrict__ out, long n) {
  __shared__ float red[4];
  float s = 0.f;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) s += x[e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
// several sums in one launch: blockIdx.y = job (the bias gradients of a step's Cout = 1 convs)
struct SumBatch {
  nsc_sum_job j[NSC_SUM_MAXJ];
};
__global__ __launch_bounds__(256) void sum_all_batch_kernel(SumBatch t) {
  __shared__ float red[4];
  nsc_sum_job jb = t.j[0];
#pragma unroll
  for (int q = 1; q < NSC_SUM_MAXJ; ++q)
    if (q == (int)blockIdx.y) jb = t.j[q];
  float s = 0.f;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < jb.n; e += (long)gridDim.x * blockDim.x) s += jb.x[e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && (long)blockIdx.x * blockDim.x < jb.n) atomicAdd(jb.out, red[0] + red[1] + red[2] + red[3]);
}
extern "C" int nsc_sum_all_batch(const nsc_sum_job* jobs, int njobs, void* stream) {
  NSC_REQUIRE(jobs && njobs > 0 && njobs <= NSC_SUM_MAXJ, NSC_ERR_BAD_ARG, "nsc_sum_all_batch: 1..%d jobs", NSC_SUM_MAXJ);
  SumBatch t;
  long nmax = 0;
  for (int q = 0; q < NSC_SUM_MAXJ; ++q) {
    t.j[q] = jobs[q < njobs ? q : 0];
    NSC_REQUIRE(t.j[q].x && t.j[q].out && t.j[q].n > 0, NSC_ERR_BAD_ARG, "nsc_sum_all_batch: job %d: bad args", q);
    nmax = std::max(nmax, t.j[q].n);
  }
  hipLaunchKernelGGL(sum_all_batch_kernel, dim3(std::min<long>(256, nsc_cdiv(nmax, 1024)), njobs), dim3(256), 0,
                     (hipStream_t)stream, t);
  NSC_CHECK_LAUNCH("sum_all_batch");
  return NSC_OK;
}
extern "C" int nsc_sum_all(const float* x, float* out, long n, void* stream) {
  NSC_REQUIRE(x && out && n > 0, NSC_ERR_BAD_ARG, "nsc_sum_all: bad args");
  hipLaunchKernelGGL(sum_all_kernel, dim3(std::min<long>(256, nsc_cdiv(n, 1024))), dim3(256), 0, (hipStream_t)stream,
                     x, out, n);
  NSC_CHECK_LAUNCH("sum_all");
  return NSC_OK;
}

// ---------------- TF1 Adam on flat buffers (tf.compat.v1.train.AdamOptimizer; nsc_module:922-925) ----------------
__global__ void adam_tf1_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long n, float lr, float b1, float b2, float eps, int t,
                                const int* __restrict__ t_dev) {
  const int tt = t_dev ? t_dev[0] : t;
  const float lr_t = lr * sqrtf(1.f - powf(b2, (float)tt)) / (1.f - powf(b1, (float)tt));
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float ge = g[e];
    const float me = b1 * m[e] + (1.f - b1) * ge;
    const float ve = b2 * v[e] + (1.f - b2) * ge * ge;
    m[e] = me;
    v[e] = ve;
    p[e] -= lr_t * me / (sqrtf(ve) + eps);
  }
}
extern "C" int nsc_adam_tf1_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                                 float eps, int t, const int* t_dev, void* stream) {
  NSC_REQUIRE(p && g && m && v && n > 0, NSC_ERR_BAD_ARG, "nsc_adam_tf1_step: bad args");
  NSC_REQUIRE(t_dev || t >= 1, NSC_ERR_BAD_ARG, "nsc_adam_tf1_step: step t must be >= 1");
  hipLaunchKernelGGL(adam_tf1_kernel, dim3(std::min<long>(2048, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream,
                     p, g, m, v, n, lr, b1, b2, eps, t, t_dev);
  NSC_CHECK_LAUNCH("adam_tf1");
  return NSC_OK;
}
__global__ void increment_kernel(int* c) { if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += 1; }
extern "C" int nsc_increment(int* counter, void* stream) {
  NSC_REQUIRE(counter, NSC_ERR_BAD_ARG, "nsc_increment: null");
  hipLaunchKernelGGL(increment_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter);
  NSC_CHECK_LAUNCH("increment");
  return NSC_OK;
}
// A kernel whose duration does not depend on any cache: 256 workgroups of one wave run a dependent chain of `iters` FMAs (about 8
// cycles each).  bench.py brackets one and two launches of it with events to measure what an event bracket adds to a launch.
__global__ void spin_kernel(float* sink, int iters) {
  float x = (float)threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) x = fmaf(x, 0.999f, 0.001f);
  if (x == 123.456f) sink[0] = x;
}
extern "C" int nsc_spin(float* sink, int iters, void* stream) {
  NSC_REQUIRE(sink && iters > 0, NSC_ERR_BAD_ARG, "nsc_spin: bad args");
  hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, (hipStream_t)stream, sink, iters);
  NSC_CHECK_LAUNCH("spin");
  return NSC_OK;
}

// ---------------- framing / Hann overlap-add (utilities.py:7-39; cmrl.py:595-597) ----------------
__global__ void frame_utterance_kernel(const float* __restrict__ utt, long n, const float* __restrict__ window,
                                       float* __restrict__ frames, int nframes) {
  const long total = (long)nframes * 512;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long i = e >> 9;
    const int j = (int)(e & 511);
    const long src = i * 480 + j;
    const float v = src < n ? utt[src] : 0.f;
    frames[e] = window ? v * window[j] : v;
  }
}
extern "C" int nsc_frame_utterance(const float* utt, long n, const float* window, float* frames, int nframes,
                                   void* stream) {
  NSC_REQUIRE(utt && frames && n > 0 && nframes >= 0, NSC_ERR_BAD_ARG, "nsc_frame_utterance: bad args");
  if (nframes == 0) return NSC_OK;
  const long total = (long)nframes * 512;
  hipLaunchKernelGGL(frame_utterance_kernel, dim3(std::min<long>(4096, nsc_cdiv(total, 256))), dim3(256), 0,
                     (hipStream_t)stream, utt, n, window, frames, nframes);
  NSC_CHECK_LAUNCH("frame_utterance");
  return NSC_OK;
}

// out[s] = sum over the (at most two) frames covering sample s of frames[i, s-480 i] * win_i[s-480 i];
// win3 = the three Hann variants of utilities.py:10-15 (first, middle, last), built on the host in float64.
__device__ __forceinline__ float hann_variant(const float* __restrict__ win3, int j, int i, int nframes) {
  const int which = (i == 0) ? 0 : ((i == nframes - 1) ? 2 : 1);
  return win3[which * 512 + j];
}
__global__ void overlap_add_kernel(const float* __restrict__ frames, int nframes, const float* __restrict__ win3,
                                   float* __restrict__ out, long n) {
  for (long s = blockIdx.x * (long)blockDim.x + threadIdx.x; s < n; s += (long)gridDim.x * blockDim.x) {
    long i = s / 480;
    if (i >= nframes) i = nframes - 1;
    float acc = 0.f;
    const long j = s - i * 480;
    if (j < 512) acc += frames[i * 512 + j] * hann_variant(win3, (int)j, (int)i, nframes);
    if (i > 0) {
      const long j2 = s - (i - 1) * 480;
      if (j2 < 512) acc += frames[(i - 1) * 512 + j2] * hann_variant(win3, (int)j2, (int)(i - 1), nframes);
    }
    out[s] = acc;
  }
}
extern "C" int nsc_overlap_add(const float* frames, int nframes, const float* win3, float* out, void* stream) {
  NSC_REQUIRE(frames && out && win3 && nframes > 0, NSC_ERR_BAD_ARG, "nsc_overlap_add: bad args");
  const long n = 480L * (nframes - 1) + 512;
  hipLaunchKernelGGL(overlap_add_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0,
                     (hipStream_t)stream, frames, nframes, win3, out, n);
  NSC_CHECK_LAUNCH("overlap_add");
  return NSC_OK;
}

// dst[e] = src[idx[e]], or 0 where idx[e] < 0  (one launch rebuilds every flipped/transposed dgrad weight from the flat
// parameter buffer; negative entries are the structural zeros of the polyphase stride-2 data-gradient kernels).
// Bits 26..29 of an index = m > 0: the word is two bf16 PIECES of the split-operand images (block_split.hip: nsc_gated_block_simage_index):
// low half from src[i], high half from src[i + stride], plane (m - 1) / 5 (0 hi, 1 lo, 2 lo2), stride {20, 25, 50, 100, 1}[(m - 1) % 5].
__device__ __forceinline__ float gather_word(const float* __restrict__ src, int i) {
  if (i < 0) return 0.f;
  const int m = i >> 26;
  if (m == 0) return src[i];
  const int base = i & 0x3ffffff, plane = (m - 1) / 5, sel = (m - 1) - 5 * plane;
  const int stride = sel == 0 ? 20 : (sel == 1 ? 25 : (sel == 2 ? 50 : (sel == 3 ? 100 : 1)));
  unsigned pk[3];
  nsc_split2(src[base], src[base + stride], pk);
  return __builtin_bit_cast(float, plane == 0 ? pk[0] : (plane == 1 ? pk[1] : pk[2]));
}
// words [first, first + words) of dst, words <= 1024, by one workgroup of 256 threads: four words per thread with every level of the
// dependent chain (index -> one or two source words -> store) issued for all four at once - a plain one-word-per-thread loop is
// serial chains of three memory round trips each, and the gather is latency, not bandwidth (round 6: nsc_step_begin_chunks; the
// op surface's per-block image gathers ran 16.7 us for 0.1 M words in the plain form)
__device__ __forceinline__ void gather_chunk(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst,
                                             long first, int words) {
  int ix[4];
  float a[4], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = threadIdx.x + 256 * j;
    ix[j] = e < words ? idx[first + e] : -1;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = ix[j], m = i < 0 ? 0 : (i >> 26), base = i < 0 ? 0 : (i & 0x3ffffff);
    const int sel = m > 0 ? (m - 1) % 5 : 4;
    const int stride = m == 0 ? 0 : (sel == 0 ? 20 : (sel == 1 ? 25 : (sel == 2 ? 50 : (sel == 3 ? 100 : 1))));
    a[j] = src[base];
    b[j] = src[base + stride];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = threadIdx.x + 256 * j, i = ix[j];
    if (e < words) {
      float v = 0.f;
      if (i >= 0) {
        const int m = i >> 26;
        if (m == 0) {
          v = a[j];
        } else {
          unsigned pk[3];
          nsc_split2(a[j], b[j], pk);
          const int plane = (m - 1) / 5;
          v = __builtin_bit_cast(float, plane == 0 ? pk[0] : (plane == 1 ? pk[1] : pk[2]));
        }
      }
      dst[first + e] = v;
    }
  }
}
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                     float* __restrict__ dst, long n) {
  for (long first = blockIdx.x * 1024L; first < n; first += gridDim.x * 1024L)
    gather_chunk(src, idx, dst, first, (int)(n - first < 1024 ? n - first : 1024));
}
extern "C" int nsc_gather(const float* src, const int* idx, float* dst, long n, void* stream) {
  NSC_REQUIRE(src && idx && dst && n > 0, NSC_ERR_BAD_ARG, "nsc_gather: bad args");
  hipLaunchKernelGGL(gather_kernel, dim3(std::min<long>(8192, nsc_cdiv(n, 1024))), dim3(256), 0, (hipStream_t)stream,
                     src, idx, dst, n);
  NSC_CHECK_LAUNCH("gather");
  return NSC_OK;
}

// The three launches that open a training step in ONE (round 4): dst = gather(src, idx) as above; zero[0, zn) = 0 (the gradients and
// histograms, 16-byte aligned, zn % 4 == 0); counter[0] += 1 (the optimizer's step counter, read by the Adam launch at the END of the
// step: nothing between here and there touches it).  The zeroing used to be a runtime memset node and the increment a one-thread
// kernel - each a dispatch of ~5 us in the step's graph.
__global__ void step_begin_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, long n,
                                  float* __restrict__ zero, long zn4, int* __restrict__ counter, int gather_blocks) {
  if ((int)blockIdx.x < gather_blocks) {
#ifndef NSC_SB_VEC
#define NSC_SB_VEC 0     // A/B (make EXTRA=-DNSC_SB_VEC=1): four words per thread (16-byte index loads / stores) - measured SLOWER (45 vs 29 us)
#endif
#if NSC_SB_VEC
    const long n4 = n >> 2;
    const i32x4_t* idx4 = reinterpret_cast<const i32x4_t*>(idx);
    f32x4* dst4 = reinterpret_cast<f32x4*>(dst);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n4; e += (long)gather_blocks * blockDim.x) {
      const i32x4_t i = idx4[e];
      dst4[e] = (f32x4){gather_word(src, i[0]), gather_word(src, i[1]), gather_word(src, i[2]), gather_word(src, i[3])};
    }
    for (long e = 4 * n4 + blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gather_blocks * blockDim.x)
      dst[e] = gather_word(src, idx[e]);
#else
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gather_blocks * blockDim.x)
      dst[e] = gather_word(src, idx[e]);
#endif
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) counter[0] += 1;
  } else {
    const int zb = gridDim.x - gather_blocks;
    f32x4* z4 = reinterpret_cast<f32x4*>(zero);
    for (long e = (blockIdx.x - gather_blocks) * (long)blockDim.x + threadIdx.x; e < zn4; e += (long)zb * blockDim.x)
      z4[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}
extern "C" int nsc_step_begin(const float* src, const int* idx, float* dst, long n, float* zero, long zero_n, int* counter,
                              void* stream) {
  NSC_REQUIRE(src && idx && dst && n > 0 && zero && zero_n > 0, NSC_ERR_BAD_ARG, "nsc_step_begin: bad args");
  NSC_REQUIRE((zero_n & 3) == 0 && ((uintptr_t)zero & 15) == 0, NSC_ERR_BAD_ARG, "nsc_step_begin: the zeroed range must be 16-byte aligned and a multiple of 4 floats");
  NSC_REQUIRE((((uintptr_t)idx | (uintptr_t)dst) & 15) == 0, NSC_ERR_BAD_ARG, "nsc_step_begin: idx and dst must be 16-byte aligned");
#ifndef NSC_SB_BLOCKS
#define NSC_SB_BLOCKS 2048
#endif
  const int gb = (int)std::min<long>(NSC_SB_BLOCKS, nsc_cdiv(NSC_SB_VEC ? nsc_cdiv(n, 4) : n, 256)), zb = (int)std::min<long>(512, nsc_cdiv(zero_n / 4, 256));
  hipLaunchKernelGGL(step_begin_kernel, dim3(gb + zb), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, zero, zero_n / 4, counter, gb);
  NSC_CHECK_LAUNCH("step_begin");
  return NSC_OK;
}

// nsc_step_begin over the LIVE regions of dst only: chunks[c] = (first word, words <= 1024) of dst / idx that this step's kernels read
// (the engine records them in a step's first run: most of the index map serves the OTHER arithmetic arm and the unfused paths - 1.7 M of
// the headline step's 3.8 M words).  Everything else as nsc_step_begin.
__global__ void step_begin_chunks_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst,
                                         const int2* __restrict__ chunks, int nchunks, float* __restrict__ zero, long zn4,
                                         int* __restrict__ counter) {
  if ((int)blockIdx.x < nchunks) {
    const int2 c = chunks[blockIdx.x];
    gather_chunk(src, idx, dst, c.x, c.y);
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) counter[0] += 1;
  } else {
    const int zb = gridDim.x - nchunks;
    f32x4* z4 = reinterpret_cast<f32x4*>(zero);
    for (long e = (blockIdx.x - nchunks) * (long)blockDim.x + threadIdx.x; e < zn4; e += (long)zb * blockDim.x)
      z4[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}
extern "C" int nsc_step_begin_chunks(const float* src, const int* idx, float* dst, const int* chunks, int nchunks, float* zero, long zero_n,
                                     int* counter, void* stream) {
  NSC_REQUIRE(src && idx && dst && chunks && nchunks > 0 && zero && zero_n > 0, NSC_ERR_BAD_ARG, "nsc_step_begin_chunks: bad args");
  NSC_REQUIRE((zero_n & 3) == 0 && ((uintptr_t)zero & 15) == 0 && ((uintptr_t)chunks & 7) == 0, NSC_ERR_BAD_ARG,
              "nsc_step_begin_chunks: the zeroed range must be 16-byte aligned and a multiple of 4 floats, chunks 8-byte aligned");
  const int zb = (int)std::min<long>(512, nsc_cdiv(zero_n / 4, 256));
  hipLaunchKernelGGL(step_begin_chunks_kernel, dim3(nchunks + zb), dim3(256), 0, (hipStream_t)stream, src, idx, dst,
                     reinterpret_cast<const int2*>(chunks), nchunks, zero, zero_n / 4, counter);
  NSC_CHECK_LAUNCH("step_begin_chunks");
  return NSC_OK;
}

// elementwise product (tf.multiply of the two gate branches) and its backward given lin, th = tanh(gate)
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = a[e] * b[e];
}
extern "C" int nsc_mul(const float* a, const float* b, float* out, long n, void* stream) {
  NSC_REQUIRE(a && b && out && n > 0, NSC_ERR_BAD_ARG, "nsc_mul: bad args");
  hipLaunchKernelGGL(mul_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, a, b,
                     out, n);
  NSC_CHECK_LAUNCH("mul");
  return NSC_OK;
}
__global__ void glu_bwd_kernel(const float* __restrict__ lin, const float* __restrict__ th,
                               const float* __restrict__ dg, float* __restrict__ dlin, float* __restrict__ dgate,
                               long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float g = dg[e], t = th[e];
    dlin[e] = g * t;
    dgate[e] = g * lin[e] * (1.f - t * t);
  }
}
extern "C" int nsc_glu_bwd(const float* lin, const float* th, const float* dg, float* dlin, float* dgate, long n,
                           void* stream) {
  NSC_REQUIRE(lin && th && dg && dlin && dgate && n > 0, NSC_ERR_BAD_ARG, "nsc_glu_bwd: bad args");
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream,
                     lin, th, dg, dlin, dgate, n);
  NSC_CHECK_LAUNCH("glu_bwd");
  return NSC_OK;
}

// activation forward / backward as stand-alone ops (op surface: nn_core_operator.activation_func)
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    y[e] = nsc_apply_act(x[e], act);
}
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long n,
                               int act) {
  const int mode = act == NSC_ACT_LRELU ? 1 : (act == NSC_ACT_TANH ? 2 : 0);
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    dx[e] = dy[e] * nsc_act_grad_from_out(y[e], mode);
}
extern "C" int nsc_act_fwd(const float* x, float* y, long n, int act, void* stream) {
  NSC_REQUIRE(x && y && n > 0 && act >= 0 && act <= 2, NSC_ERR_BAD_ARG, "nsc_act_fwd: bad args");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, x,
                     y, n, act);
  NSC_CHECK_LAUNCH("act_fwd");
  return NSC_OK;
}
extern "C" int nsc_act_bwd(const float* dy, const float* y, float* dx, long n, int act, void* stream) {
  NSC_REQUIRE(dy && y && dx && n > 0 && act >= 0 && act <= 2, NSC_ERR_BAD_ARG, "nsc_act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream,
                     dy, y, dx, n, act);
  NSC_CHECK_LAUNCH("act_bwd");
  return NSC_OK;
}

// quan_loss (loss_terms_and_measures.py:257-259) on a materialised p: out[b] = mean_l sum_k sqrt(p + 1e-20);
// hist[k] += sum_{b,l} p (nullable).  One workgroup per frame.  Backward of both is elementwise (see ops.py).
__global__ __launch_bounds__(256) void p_stats_kernel(const float* __restrict__ p, int L, int nb, float* __restrict__ quan,
                                                      float* __restrict__ hist) {
  __shared__ float red[4];
  __shared__ float hs[256];
  const long b = blockIdx.x;
  const float* pb = p + b * (long)L * nb;
  // 256 % nb == 0 (the shipped 32 / 64 bins): a thread meets the SAME bin k = tid % nb in every round of the loop, so the histogram
  // rides the one pass over p (it used to be a second pass of nb threads x L serial loads: 22 of the launch's 35 us at B = 128)
  const bool one_pass = hist && (256 % nb) == 0;
  float s = 0.f, hk = 0.f;
#pragma unroll 8
  for (int e = threadIdx.x; e < L * nb; e += 256) {
    const float v = pb[e];
    s += sqrtf(v + 1e-20f);
    hk += v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  hs[threadIdx.x] = hk;
  __syncthreads();
  if (threadIdx.x == 0 && quan) quan[b] = (red[0] + red[1] + red[2] + red[3]) / (float)L;
  if (one_pass) {
    if ((int)threadIdx.x < nb) {
      float h = 0.f;
      for (int j = threadIdx.x; j < 256; j += nb) h += hs[j];
      atomicAdd(hist + threadIdx.x, h);
    }
  } else if (hist) {
    for (int k = threadIdx.x; k < nb; k += 256) {
      float h = 0.f;
      for (int l = 0; l < L; ++l) h += pb[(long)l * nb + k];
      atomicAdd(hist + k, h);
    }
  }
}
extern "C" int nsc_p_stats(const float* p, int B, int L, int nb, float* quan, float* hist, void* stream) {
  NSC_REQUIRE(p && B > 0 && L > 0 && nb > 0, NSC_ERR_BAD_ARG, "nsc_p_stats: bad args");
  hipLaunchKernelGGL(p_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p, L, nb, quan, hist);
  NSC_CHECK_LAUNCH("p_stats");
  return NSC_OK;
}
// dp[b,l,k] = gq[b]/L * 0.5/sqrt(p+1e-20) + gh[k]   (gq, gh nullable)
__global__ void p_stats_bwd_kernel(const float* __restrict__ p, const float* __restrict__ gq,
                                   const float* __restrict__ gh, float* __restrict__ dp, int L, int nb, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long b = e / ((long)L * nb);
    const int k = (int)(e % nb);
    float g = 0.f;
    if (gq) g += gq[b] / (float)L * 0.5f / sqrtf(p[e] + 1e-20f);
    if (gh) g += gh[k];
    dp[e] = g;
  }
}
extern "C" int nsc_p_stats_bwd(const float* p, const float* gq, const float* gh, float* dp, int B, int L, int nb,
                               void* stream) {
  NSC_REQUIRE(p && dp && B > 0 && L > 0 && nb > 0, NSC_ERR_BAD_ARG, "nsc_p_stats_bwd: bad args");
  const long n = (long)B * L * nb;
  hipLaunchKernelGGL(p_stats_bwd_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream,
                     p, gq, gh, dp, L, nb, n);
  NSC_CHECK_LAUNCH("p_stats_bwd");
  return NSC_OK;
}

// GLU backward writing both branch gradients into ONE [B, 2n, T] tensor (dlin | dgate) so that the two k15 data
// gradients run as a single conv with 2n input channels.
__global__ void glu_bwd_cat_kernel(const float* __restrict__ lin, const float* __restrict__ th, const float* __restrict__ dg,
                                   float* __restrict__ da, long nT, long total) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / nT, r = e - b * nT;
    const float g = dg[e], t = th[e];
    da[b * 2 * nT + r] = g * t;
    da[b * 2 * nT + nT + r] = g * lin[e] * (1.f - t * t);
  }
}
extern "C" int nsc_glu_bwd_cat(const float* lin, const float* th, const float* dg, float* da, int B, int n, int T, void* stream) {
  NSC_REQUIRE(lin && th && dg && da && B > 0 && n > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_glu_bwd_cat: bad args");
  const long total = (long)B * n * T;
  hipLaunchKernelGGL(glu_bwd_cat_kernel, dim3(std::min<long>(4096, nsc_cdiv(total, 256))), dim3(256), 0, (hipStream_t)stream,
                     lin, th, dg, da, (long)n * T, total);
  NSC_CHECK_LAUNCH("glu_bwd_cat");
  return NSC_OK;
}

// zero a device range on the stream (gradient buffers, histograms): a memset node when captured into a hipGraph
extern "C" int nsc_zero(float* p, long n, void* stream) {
  NSC_REQUIRE(p && n > 0, NSC_ERR_BAD_ARG, "nsc_zero: bad args");
  hipError_t e = hipMemsetAsync(p, 0, (size_t)n * sizeof(float), (hipStream_t)stream);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "nsc_zero: %s", hipGetErrorString(e));
  return NSC_OK;
}

// Identity of the hipGraph capture the stream is in (0: not capturing).  A host that hands out pre-zeroed or pre-built device
// buffers (nsc_amd/ops.py: the op surface's zero pool) must not reuse, inside a capture, what was prepared outside it or in an
// earlier capture: the preparing launch would be missing from the graph being recorded.
extern "C" int nsc_stream_capture_id(void* stream, unsigned long long* id) {
  NSC_REQUIRE(id, NSC_ERR_BAD_ARG, "nsc_stream_capture_id: bad args");
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  hipError_t e = hipStreamGetCaptureInfo((hipStream_t)stream, &status, &cid);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "nsc_stream_capture_id: %s", hipGetErrorString(e));
  *id = status == hipStreamCaptureStatusActive ? (cid ? cid : ~0ull) : 0ull;
  return NSC_OK;
}
