// LPC front / back end of the collaborative-quantisation path on the GPU.
// Replaces the reference's per-frame Python loops that run inside tf.py_func (lpc_utilities.py):
//   lsf2poly_after_quan        :28-33    spectrum.lsf2poly per frame, cast to float32
//   lpc_analysis_get_residual  :37-77    seven 128-sample sub-frames at hop 64, each filtered FROM REST by A(z) and
//                                        cross-faded with hanning(128) (first: flat 64 | falling half; last: rising half | flat 64)
//   lpc_synthesizer_tr         :137-156  y = res / A(z) per frame, from rest
// Call sites: nsc_module:1012-1013, 1029, 1100-1101; cmrl.py:161-162, 239, 414-415, 451.
// All three are HBM-/latency-bound byte shuffling with 16-tap filters: one frame per workgroup (or per lane for the
// recursive filter), operands in LDS / registers, double-precision accumulation like the reference's Python floats
// (the reference rounds each RESULT to float32, so the outputs here are float32 too).
#include "nsc_common.h"

#define LPC_FRAME 512
#define LPC_SUB 128
#define LPC_HALF 64
#define LPC_MAXORD 32

// ---- lsf2poly: one lane per frame.  P and Q are products of real quadratics (1 - 2 cos(w) z^-1 + z^-2): the roots
// e^{+-jw} of the even- (Q) and odd-indexed (P) frequencies; even order: P1 = P (1 - z^-1), Q1 = Q (1 + z^-1);
// a = (P1 + Q1) / 2 without its last coefficient (Kondoz, ch. 4 - what spectrum.lsf2poly implements).
__global__ __launch_bounds__(64) void lsf2poly_kernel(const float* __restrict__ lsf, float* __restrict__ poly, int B,
                                                      int order) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double P[LPC_MAXORD + 3], Q[LPC_MAXORD + 3];
  for (int i = 0; i < LPC_MAXORD + 3; ++i) P[i] = Q[i] = 0.0;
  P[0] = Q[0] = 1.0;
  int dp = 0, dq = 0;     // current degrees
  const float* w = lsf + (long)b * order;
  for (int i = 0; i < order; ++i) {
    const double c = -2.0 * cos((double)w[i]);
    double* R = (i & 1) ? P : Q;
    int& d = (i & 1) ? dp : dq;
    // R <- R * (1 + c z^-1 + z^-2), in place from the top
    for (int k = d + 2; k >= 0; --k) {
      const double r0 = k <= d ? R[k] : 0.0;
      const double r1 = (k - 1 >= 0 && k - 1 <= d) ? R[k - 1] : 0.0;
      const double r2 = (k - 2 >= 0 && k - 2 <= d) ? R[k - 2] : 0.0;
      R[k] = r0 + c * r1 + r2;
    }
    d += 2;
  }
  float* a = poly + (long)b * (order + 1);
  if (order & 1) {
    // odd order: P1 = P (1 - z^-2), Q1 = Q
    for (int k = 0; k <= order; ++k) {
      const double p1 = (k <= dp ? P[k] : 0.0) - ((k - 2 >= 0 && k - 2 <= dp) ? P[k - 2] : 0.0);
      const double q1 = k <= dq ? Q[k] : 0.0;
      a[k] = (float)(0.5 * (p1 + q1));
    }
  } else {
    for (int k = 0; k <= order; ++k) {
      const double p1 = (k <= dp ? P[k] : 0.0) - ((k - 1 >= 0 && k - 1 <= dp) ? P[k - 1] : 0.0);
      const double q1 = (k <= dq ? Q[k] : 0.0) + ((k - 1 >= 0 && k - 1 <= dq) ? Q[k - 1] : 0.0);
      a[k] = (float)(0.5 * (p1 + q1));
    }
  }
}

extern "C" int nsc_lsf2poly(const float* lsf, float* poly, int B, int order, void* stream) {
  NSC_REQUIRE(lsf && poly && B > 0, NSC_ERR_BAD_ARG, "nsc_lsf2poly: bad args");
  NSC_REQUIRE(order > 0 && order <= LPC_MAXORD, NSC_ERR_UNSUPPORTED, "nsc_lsf2poly: order %d not in 1..%d", order, LPC_MAXORD);
  hipLaunchKernelGGL(lsf2poly_kernel, dim3(nsc_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, lsf, poly, B, order);
  NSC_CHECK_LAUNCH("lsf2poly");
  return NSC_OK;
}

// ---- sub-framed FIR residual: one workgroup (256 lanes) per frame, two output samples per lane.
// res[t] = sum over the (at most two) sub-frames j that cover t of  w_j(t - 64 j) * sum_{k=0}^{min(order, t - 64 j)} a[k] x[t-k]
__device__ __forceinline__ double lpc_hann128(int n) {   // numpy.hanning(128)[n]
  return 0.5 - 0.5 * cos(6.283185307179586476925286766559 * (double)n / 127.0);
}
__global__ __launch_bounds__(256) void lpc_residual_kernel(const float* __restrict__ x, const float* __restrict__ poly,
                                                           float* __restrict__ res, int order) {
  __shared__ float xs[LPC_FRAME];
  __shared__ float as[LPC_MAXORD + 1];
  const long b = blockIdx.x;
  const int tid = threadIdx.x;
  xs[tid] = x[b * LPC_FRAME + tid];
  xs[tid + 256] = x[b * LPC_FRAME + tid + 256];
  if (tid <= order) as[tid] = poly[b * (order + 1) + tid];
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int t = tid + 256 * half;
    const int j1 = min(t / LPC_HALF, 6);            // the later sub-frame covering t (starts at 64 j1)
    double out = 0.0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = j1 - e;
      if (j < 0) continue;
      const int n = t - LPC_HALF * j;               // position inside sub-frame j
      if (n >= LPC_SUB) continue;
      double w;
      if (j == 0) w = n < LPC_HALF ? 1.0 : lpc_hann128(n);
      else if (j == 6) w = n < LPC_HALF ? lpc_hann128(n) : 1.0;
      else w = lpc_hann128(n);
      double acc = 0.0;
      const int kmax = min(order, n);
      for (int k = 0; k <= kmax; ++k) acc += (double)as[k] * (double)xs[t - k];
      out += acc * w;
    }
    res[b * LPC_FRAME + t] = (float)out;
  }
}

extern "C" int nsc_lpc_residual(const float* x, const float* poly, float* res, int B, int order, void* stream) {
  NSC_REQUIRE(x && poly && res && B > 0, NSC_ERR_BAD_ARG, "nsc_lpc_residual: bad args");
  NSC_REQUIRE(order > 0 && order <= LPC_MAXORD, NSC_ERR_UNSUPPORTED, "nsc_lpc_residual: order %d not in 1..%d", order, LPC_MAXORD);
  hipLaunchKernelGGL(lpc_residual_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, poly, res, order);
  NSC_CHECK_LAUNCH("lpc_residual");
  return NSC_OK;
}

// ---- all-pole synthesis y = res / A(z) from rest: the recursion is sequential in time, frames are independent ->
// one lane per frame, the last 16 outputs in registers (loop unrolled by 16 so that the ring index is static).
template <int ORD>
__global__ __launch_bounds__(64) void lpc_synthesis_kernel(const float* __restrict__ poly, const float* __restrict__ res,
                                                           float* __restrict__ out, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double a[ORD + 1];
#pragma unroll
  for (int k = 0; k <= ORD; ++k) a[k] = (double)poly[(long)b * (ORD + 1) + k];
  double y[ORD];            // y[i] = output at time (n - 1 - i) in ring order, rotated by the unrolled loop below
#pragma unroll
  for (int i = 0; i < ORD; ++i) y[i] = 0.0;
  const float* r = res + (long)b * LPC_FRAME;
  float* o = out + (long)b * LPC_FRAME;
  for (int n0 = 0; n0 < LPC_FRAME; n0 += ORD) {
#pragma unroll
    for (int u = 0; u < ORD; ++u) {
      // ring slot s holds the output of time (n0 + s) - ORD before this block and of time n0 + s after step s
      double acc = (double)r[n0 + u];
#pragma unroll
      for (int k = 1; k <= ORD; ++k) acc -= a[k] * y[(u - k + ORD) % ORD];   // y at time n0 + u - k
      const double v = acc / a[0];
      y[u] = v;
      o[n0 + u] = (float)v;
    }
  }
}

extern "C" int nsc_lpc_synthesis(const float* poly, const float* res, float* out, int B, int order, void* stream) {
  NSC_REQUIRE(poly && res && out && B > 0, NSC_ERR_BAD_ARG, "nsc_lpc_synthesis: bad args");
  NSC_REQUIRE(order == 16, NSC_ERR_UNSUPPORTED, "nsc_lpc_synthesis: built for LPC order 16 (got %d)", order);
  hipLaunchKernelGGL(lpc_synthesis_kernel<16>, dim3(nsc_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, poly, res, out, B);
  NSC_CHECK_LAUNCH("lpc_synthesis");
  return NSC_OK;
}
