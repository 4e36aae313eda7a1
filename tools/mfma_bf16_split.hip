// tools/mfma_bf16_split.hip - OPTIONAL experiment (never the headline, not in the library): the k9 20 -> 100 conv of
// the gated block (nn_core_operator.py:101-109) on the bf16 matrix cores with fp32 operands SPLIT into bf16 pieces and fp32
// accumulation, against the exact fp32 MFMA the product uses.  Reports time per 64-step tile and the error against float64.
//   x = hi + lo (+ lo2): hi = bf16(x), lo = bf16(x - hi), lo2 = bf16(x - hi - lo)
//   x3: a.b ~ ah.bh + ah.bl + al.bh            (error ~ 2^-16 |a||b| per term)
//   x6: + ah.bl2 + al2.bh + al.bl              (error ~ 2^-24: fp32 class)
// v_mfma_f32_16x16x32_bf16 does 16x16x32 MACs in 16 cycles, v_mfma_f32_16x16x4_f32 16x16x4 in 32: 16x the rate per instruction slot,
// so x3 is 5.3x and x6 2.7x the fp32 matrix rate on paper - and the VALU is free meanwhile (fp32 MFMAs execute on the vector ALUs).
// Layout: K index = tap * 24 + c (20 channels padded to 24 = 3 groups of 8, so a lane's 8 consecutive k lie inside one tap);
// the activation tile sits TRANSPOSED in LDS as [time][24] bf16 planes: a B fragment is one 16-byte LDS read.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_split.hip -o /tmp/mfma_bf16_split && /tmp/mfma_bf16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int C = 100, NARROW = 20, K9 = 9, TT = 64, WG = 72, CP = 24, NKG = K9 * 3, NKS = (NKG + 3) / 4;   // 27 groups of 8 -> 7 k-steps of 32

__device__ __forceinline__ unsigned short f2bf(float x) {     // round to nearest even (finite inputs)
  unsigned u = __builtin_bit_cast(unsigned, x);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// ---- exact fp32 reference on the matrix pipe: one wave = one 16-row tile, 4 column tiles, K = 180 in steps of 4 ----
__global__ __launch_bounds__(512) void conv_f32_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out, int ntiles) {
  __shared__ float gs[NARROW][WG + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  float wr[K9][5];
  for (int tap = 0; tap < K9; ++tap)
    for (int u = 0; u < 5; ++u) wr[tap][u] = wave < 7 ? w[(tap * NARROW + 4 * u + kq) * C + min(wave * 16 + l15, C - 1)] : 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    for (int e = tid; e < NARROW * WG; e += 512) gs[e / WG][e % WG] = g[((long)tile * NARROW + e / WG) * WG + e % WG];
    __syncthreads();
    if (wave < 7) {
      f32x4 acc[4];
      for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < K9; ++tap)
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[tap][u], gs[4 * u + kq][c * 16 + l15 + tap], acc[c], 0, 0, 0);
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
          const int o = wave * 16 + kq * 4 + r;
          if (o < C) out[((long)tile * C + o) * TT + c * 16 + l15] = acc[c][r];
        }
    }
  }
}

// ---- bf16 split: NS = 2 (x3) or 3 (x6) pieces per operand ----
template <int NS>
__global__ __launch_bounds__(512) void conv_split_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out, int ntiles) {
  __shared__ __attribute__((aligned(16))) unsigned short gt[NS][WG][CP];      // transposed activation planes [time][24 channels]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  // weights: A fragment of k-step s: row = wave * 16 + l15, k = 32 s + 8 kq + j -> group q8 = 4 s + kq -> tap q8 / 3, channels 8 (q8 % 3) + j
  bf16x8 wa[NS][NKS];
  for (int s = 0; s < NKS; ++s) {
    const int q8 = 4 * s + kq, tap = q8 / 3, c0 = 8 * (q8 % 3);
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j, o = wave * 16 + l15;
      float x = (q8 < NKG && c < NARROW && wave < 7 && o < C) ? w[(tap * NARROW + c) * C + o] : 0.f;
      for (int p = 0; p < NS; ++p) {
        const unsigned short h = f2bf(x);
        wa[p][s][j] = (short)h;
        x -= bf2f(h);
      }
    }
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    for (int e = tid; e < CP * WG; e += 512) {          // split while staging: NS conversions + (NS - 1) subtractions per element
      const int c = e / WG, t = e % WG;
      float x = c < NARROW ? g[((long)tile * NARROW + c) * WG + t] : 0.f;
      for (int p = 0; p < NS; ++p) {
        const unsigned short h = f2bf(x);
        gt[p][t][c] = h;
        x -= bf2f(h);
      }
    }
    __syncthreads();
    if (wave < 7) {
      f32x4 acc[4];
      for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const int q8 = 4 * s + kq, tap = q8 < NKG ? q8 / 3 : 0, c0 = q8 < NKG ? 8 * (q8 % 3) : 0;   // (padded group: zero weights)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          bf16x8 b[NS];
#pragma unroll
          for (int p = 0; p < NS; ++p) b[p] = *reinterpret_cast<const bf16x8*>(&gt[p][c * 16 + l15 + tap][c0]);
          // products in order of decreasing magnitude last -> first is irrelevant in fp32 accumulation of this size; smallest first
          if (NS == 3) {
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][s], b[1], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][s], b[2], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[2][s], b[0], acc[c], 0, 0, 0);
          }
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][s], b[1], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][s], b[0], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][s], b[0], acc[c], 0, 0, 0);
        }
      }
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
          const int o = wave * 16 + kq * 4 + r;
          if (o < C) out[((long)tile * C + o) * TT + c * 16 + l15] = acc[c][r];
        }
    }
  }
}

int main() {
  const int ntiles = 8192;             // = 128 frames x 512 steps / 64 ... x 8: enough work for a stable time
  std::vector<float> hg((size_t)ntiles * NARROW * WG), hw(K9 * NARROW * C);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : hg) v = rnd() * rnd();                                   // gate products: |g| < 1, heavy near 0
  for (auto& v : hw) v = 0.082f * rnd();                                  // glorot limit of a [9, 20, 100] kernel
  float *g, *w, *o;
  hipMalloc(&g, hg.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&o, (size_t)ntiles * C * TT * 4);
  hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  const int ncheck = 32;               // tiles checked against float64
  std::vector<double> ref((size_t)ncheck * C * TT);
  double rms = 0;
  for (int tile = 0; tile < ncheck; ++tile)
    for (int oc = 0; oc < C; ++oc)
      for (int t = 0; t < TT; ++t) {
        double s = 0;
        for (int tap = 0; tap < K9; ++tap)
          for (int c = 0; c < NARROW; ++c) s += (double)hw[(tap * NARROW + c) * C + oc] * (double)hg[((size_t)tile * NARROW + c) * WG + t + tap];
        ref[((size_t)tile * C + oc) * TT + t] = s;
        rms += s * s;
      }
  rms = std::sqrt(rms / ref.size());
  std::vector<float> ho((size_t)ncheck * C * TT);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"fp32 v_mfma_f32_16x16x4_f32 (the product's arithmetic)", "bf16 x3 (hi.hi + hi.lo + lo.hi)", "bf16 x6 (three pieces, six products)"};
  printf("k9 20 -> 100 conv, %d tiles of 64 steps, 256 workgroups x 8 waves (7 row tiles); rms of the result %.4f\n", ntiles, rms);
  for (int v = 0; v < 3; ++v) {
    auto run = [&]() {
      if (v == 0) hipLaunchKernelGGL(conv_f32_kernel, dim3(256), dim3(512), 0, 0, g, w, o, ntiles);
      else if (v == 1) hipLaunchKernelGGL(conv_split_kernel<2>, dim3(256), dim3(512), 0, 0, g, w, o, ntiles);
      else hipLaunchKernelGGL(conv_split_kernel<3>, dim3(256), dim3(512), 0, 0, g, w, o, ntiles);
    };
    for (int i = 0; i < 3; ++i) run();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, e2 = 0;
    for (size_t i = 0; i < ho.size(); ++i) { const double d = std::fabs(ho[i] - ref[i]); emax = std::max(emax, d); e2 += d * d; }
    const double us = ms * 100.0, fl = 2.0 * ntiles * TT * K9 * NARROW * C;
    printf("%-58s %8.1f us  %6.1f TFLOP/s (algorithmic)   max |err| / rms %.2e   rms err / rms %.2e\n", names[v], us, fl / us / 1e6, emax / rms,
           std::sqrt(e2 / ho.size()) / rms);
  }
  return 0;
}
