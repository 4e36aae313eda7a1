// loss.hip - time-domain RMSE + STFT->PSD->mel->log->RMSE loss of the reference
// (loss_terms_and_measures.py:77-79 mse_loss, :178-183 tf_stft, :130-148 mfcc_transform, :151-175 mfcc_loss),
// forward AND the gradient wrt the decoded frame in one pass.  One workgroup (256 threads) per 512-sample frame.
//
// rFFT-512: hand-written radix-2 decimation-in-time complex FFT in LDS (9 stages x 256 butterflies, one per
// thread).  The decoded and target frames ride one complex transform (z = d + i*o) and are separated by
// Hermitian symmetry; the backward pass is a second complex FFT of conj(G) (G = dL/dRe + i dL/dIm, bins 0..256).
#include "nsc_common.h"
#include <algorithm>

#define NFFT 512
#define NBIN 257
#define NMEL 184

__device__ __forceinline__ int bitrev9(int v) { return (int)(__brev((unsigned)v) >> 23); }

// twiddles tw[k] = exp(-2 pi i k / 512), k = 0..255
__device__ __forceinline__ void make_twiddles(float* twr, float* twi, int tid) {
  float s, c;
  sincospif((float)tid * (1.f / 256.f), &s, &c);
  twr[tid] = c;
  twi[tid] = -s;
}

// in-place forward FFT; input must already be in bit-reversed order; result in natural order.
__device__ __forceinline__ void fft512(float* zr, float* zi, const float* twr, const float* twi, int tid) {
#pragma unroll
  for (int s = 1; s <= 9; ++s) {
    const int half = 1 << (s - 1);
    const int pos = tid & (half - 1);
    const int i0 = ((tid >> (s - 1)) << s) + pos;
    const int i1 = i0 + half;
    const int tk = pos << (9 - s);
    const float wr = twr[tk], wi = twi[tk];
    const float ur = zr[i0], ui = zi[i0];
    const float xr = zr[i1], xi = zi[i1];
    const float vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
    zr[i0] = ur + vr; zi[i0] = ui + vi;
    zr[i1] = ur - vr; zi[i1] = ui - vi;
    __syncthreads();
  }
}

// ranges (nullable): the mel matrix is banded (triangular filters), so only the non-zero band of each column / row is
// visited - same terms in the same order as the dense loops (the skipped ones are exact zeros).  Layout: [184][2]
// = [k_lo, k_hi) of each mel column, then [257][4][2] = [j_lo, j_hi) of each bin inside each of the four banks.
__global__ __launch_bounds__(256) void recon_loss_kernel(const float* __restrict__ decoded,
                                                         const float* __restrict__ target, float ct, float cf,
                                                         const float* __restrict__ gt, const float* __restrict__ gf,
                                                         const float* __restrict__ mel,
                                                         const float* __restrict__ melT, float* __restrict__ time_out,
                                                         float* __restrict__ freq_out, float* __restrict__ grad,
                                                         const int* __restrict__ ranges) {
  __shared__ float zr[NFFT], zi[NFFT], twr[256], twi[256];
  __shared__ float psd_d[NBIN], psd_o[NBIN], dre[NBIN], dim_[NBIN];
  __shared__ float sq[NMEL], gm[NMEL];
  __shared__ float red[4];
  __shared__ float bank_rms[4];
  __shared__ float s_time;
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const float* dp = decoded + b * NFFT;
  const float* op = target + b * NFFT;
  make_twiddles(twr, twi, tid);
  const float d0 = dp[tid], d1 = dp[tid + 256], o0 = op[tid], o1 = op[tid + 256];
  zr[bitrev9(tid)] = d0; zi[bitrev9(tid)] = o0;
  zr[bitrev9(tid + 256)] = d1; zi[bitrev9(tid + 256)] = o1;
  const float e0 = d0 - o0, e1 = d1 - o1;
  float ss = wave_sum(e0 * e0 + e1 * e1);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) {
    const float tl = sqrtf((red[0] + red[1] + red[2] + red[3]) * (1.f / NFFT) + 1e-7f);
    s_time = tl;
    if (time_out) time_out[b] = tl;
  }
  fft512(zr, zi, twr, twi, tid);  // ends with a barrier
  // ---- split the two real spectra, PSD = (re^2 + im^2 + 1e-7)/512 ----
  for (int k = tid; k < NBIN; k += 256) {
    const int kn = (NFFT - k) & (NFFT - 1);
    const float ar = zr[k], ai = zi[k], br = zr[kn], bi = zi[kn];
    const float Dr = 0.5f * (ar + br), Di = 0.5f * (ai - bi);
    const float Or = 0.5f * (ai + bi), Oi = -0.5f * (ar - br);
    dre[k] = Dr; dim_[k] = Di;
    psd_d[k] = (Dr * Dr + Di * Di + 1e-7f) * (1.f / NFFT);
    psd_o[k] = (Or * Or + Oi * Oi + 1e-7f) * (1.f / NFFT);
  }
  __syncthreads();
  // ---- mel banks (4 banks concatenated to 184 columns), log, squared difference ----
  float diff = 0.f, md = 0.f;
  if (tid < NMEL) {
    float mo = 0.f;
    const int klo = ranges ? ranges[2 * tid] : 0, khi = ranges ? ranges[2 * tid + 1] : NBIN;
#pragma unroll 8
    for (int k = klo; k < khi; ++k) {       // the mel loads do not depend on the sums: unrolled, eight are in flight
      const float m = mel[k * NMEL + tid];
      md = fmaf(psd_d[k], m, md);
      mo = fmaf(psd_o[k], m, mo);
    }
    diff = logf(md + 1e-7f) - logf(mo + 1e-7f);
  }
  {
    // per-bank sums of diff^2: every wave reduces its lanes bank by bank (columns 0:8 | 8:24 | 24:56 | 56:184), the (at
    // most three) waves that hold columns of a bank meet in LDS
    const float d2 = tid < NMEL ? diff * diff : 0.f;
    const int bank = tid < 8 ? 0 : (tid < 24 ? 1 : (tid < 56 ? 2 : 3));
#pragma unroll
    for (int bk = 0; bk < 4; ++bk) {
      const float sb = wave_sum(bank == bk ? d2 : 0.f);
      if ((tid & 63) == 0) sq[(tid >> 6) * 4 + bk] = sb;
    }
  }
  __syncthreads();
  if (tid < 4) {
    const float n = tid == 0 ? 8.f : (tid == 1 ? 16.f : (tid == 2 ? 32.f : 128.f));
    const float s = (sq[tid] + sq[4 + tid]) + (sq[8 + tid] + sq[12 + tid]);
    bank_rms[tid] = sqrtf(s / n + 1e-7f);
  }
  __syncthreads();
  if (tid == 0 && freq_out) freq_out[b] = 0.25f * (bank_rms[0] + bank_rms[1] + bank_rms[2] + bank_rms[3]);
  if (!grad) return;
  // ---- backward: d freq / d logmel_j = diff_j / (4 n rms); through log and the mel matmul ----
  const float wf = gf ? gf[b] : cf;
  const float wt = gt ? gt[b] : ct;
  if (tid < NMEL) {
    const int bank = tid < 8 ? 0 : (tid < 24 ? 1 : (tid < 56 ? 2 : 3));
    const float n = bank == 0 ? 8.f : (bank == 1 ? 16.f : (bank == 2 ? 32.f : 128.f));
    gm[tid] = wf * diff / (4.f * n * bank_rms[bank]) / (md + 1e-7f);
  }
  __syncthreads();
  for (int k = tid; k < NFFT; k += 256) { zr[k] = 0.f; zi[k] = 0.f; }
  __syncthreads();
  for (int k = tid; k < NBIN; k += 256) {
    float g = 0.f;
    if (ranges) {
#pragma unroll
      for (int bk = 0; bk < 4; ++bk) {
        const int jlo = ranges[2 * NMEL + 8 * k + 2 * bk], jhi = ranges[2 * NMEL + 8 * k + 2 * bk + 1];
        for (int j = jlo; j < jhi; ++j) g = fmaf(gm[j], melT[j * NBIN + k], g);
      }
    } else {
      for (int j = 0; j < NMEL; ++j) g = fmaf(gm[j], melT[j * NBIN + k], g);
    }
    g *= 2.f / NFFT;  // d psd / d re = 2 re / 512
    const int r = bitrev9(k & (NFFT - 1));
    // dx[n] = Re sum_k G_k e^{+i theta} = Re FFT(conj(G))[n]
    if (k < NFFT) { zr[r] = g * dre[k]; zi[r] = -g * dim_[k]; }
  }
  __syncthreads();
  fft512(zr, zi, twr, twi, tid);
  const float tscale = wt / (NFFT * s_time);
  grad[b * NFFT + tid] = zr[tid] + tscale * e0;
  grad[b * NFFT + tid + 256] = zr[tid + 256] + tscale * e1;
}

extern "C" int nsc_recon_loss(const float* decoded, const float* target, int B, float ct, float cf, const float* gt,
                              const float* gf, const float* mel, const float* melT, float* time_out, float* freq_out,
                              float* grad, void* stream) {
  NSC_REQUIRE(decoded && target && mel && melT && B > 0, NSC_ERR_BAD_ARG, "nsc_recon_loss: bad args");
  hipLaunchKernelGGL(recon_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, decoded, target, ct, cf, gt, gf,
                     mel, melT, time_out, freq_out, grad, (const int*)nullptr);
  NSC_CHECK_LAUNCH("recon_loss");
  return NSC_OK;
}
extern "C" int nsc_recon_loss_banded(const float* decoded, const float* target, int B, float ct, float cf, const float* gt,
                                     const float* gf, const float* mel, const float* melT, const int* ranges,
                                     float* time_out, float* freq_out, float* grad, void* stream) {
  NSC_REQUIRE(decoded && target && mel && melT && ranges && B > 0, NSC_ERR_BAD_ARG, "nsc_recon_loss_banded: bad args");
  hipLaunchKernelGGL(recon_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, decoded, target, ct, cf, gt, gf,
                     mel, melT, time_out, freq_out, grad, ranges);
  NSC_CHECK_LAUNCH("recon_loss_banded");
  return NSC_OK;
}

// Backward of (mse_loss, mfcc_loss) from what the FORWARD launch left behind (op surface, nsc_amd/ops.py: ReconLossFn): the forward
// runs nsc_recon_loss once with ct = 0, cf = 1 and keeps gfreq = d mfcc_loss[b] / d decoded[b,:] and time[b]; when the upstream
// gradients gt[b], gf[b] arrive, dL/d decoded = gf[b] gfreq + gt[b] (decoded - target) / (512 time[b]) is elementwise - instead of a
// second pass through both FFTs and the mel banks (25 us at B = 128; tf.gradients does re-traverse: loss_terms_and_measures.py:151-175).
__global__ void recon_combine_kernel(const float* __restrict__ decoded, const float* __restrict__ target,
                                     const float* __restrict__ time, const float* __restrict__ gt, const float* __restrict__ gf,
                                     const float* __restrict__ gfreq, float* __restrict__ grad, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long b = e >> 9;
    const float tscale = gt ? gt[b] / (NFFT * time[b]) : 0.f;
    grad[e] = (gf ? gf[b] * gfreq[e] : 0.f) + tscale * (decoded[e] - target[e]);
  }
}
extern "C" int nsc_recon_loss_combine(const float* decoded, const float* target, const float* time, const float* gt, const float* gf,
                                      const float* gfreq, int B, float* grad, void* stream) {
  NSC_REQUIRE(decoded && target && time && gfreq && grad && B > 0, NSC_ERR_BAD_ARG, "nsc_recon_loss_combine: bad args");
  const long n = (long)B * NFFT;
  hipLaunchKernelGGL(recon_combine_kernel, dim3(std::min<long>(2048, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, decoded,
                     target, time, gt, gf, gfreq, grad, n);
  NSC_CHECK_LAUNCH("recon_loss_combine");
  return NSC_OK;
}

// bare rFFT-512 (tf_stft): re/im [B,257], mag = sqrt(re^2 + im^2 + 1e-7)
__global__ __launch_bounds__(256) void rfft512_kernel(const float* __restrict__ sig, float* __restrict__ re,
                                                      float* __restrict__ im, float* __restrict__ mag) {
  __shared__ float zr[NFFT], zi[NFFT], twr[256], twi[256];
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  make_twiddles(twr, twi, tid);
  zr[bitrev9(tid)] = sig[b * NFFT + tid]; zi[bitrev9(tid)] = 0.f;
  zr[bitrev9(tid + 256)] = sig[b * NFFT + tid + 256]; zi[bitrev9(tid + 256)] = 0.f;
  __syncthreads();
  fft512(zr, zi, twr, twi, tid);
  for (int k = tid; k < NBIN; k += 256) {
    const float r = zr[k], i = zi[k];
    if (re) re[b * NBIN + k] = r;
    if (im) im[b * NBIN + k] = i;
    if (mag) mag[b * NBIN + k] = sqrtf(r * r + i * i + 1e-7f);
  }
}
extern "C" int nsc_rfft512(const float* sig, int B, float* re, float* im, float* mag, void* stream) {
  NSC_REQUIRE(sig && B > 0, NSC_ERR_BAD_ARG, "nsc_rfft512: bad args");
  hipLaunchKernelGGL(rfft512_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, sig, re, im, mag);
  NSC_CHECK_LAUNCH("rfft512");
  return NSC_OK;
}
