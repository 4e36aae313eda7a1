// Issue rate of v_mfma_f32_16x16x32_bf16 in bare loops: NACC independent accumulators (the chain distance), 1 | 2 waves per SIMD, all 256 CUs.
// Prints cycles per MFMA per SIMD (s_memtime ticks at 100 MHz are useless for this: wall time x 2.4 GHz and the shader clock counter are shown).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_rate.hip -o /tmp/mfma_bf16_rate && /tmp/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool SAME_OPERANDS>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, unsigned long long* cyc) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) { a[i][j] = (short)(threadIdx.x * 7 + i * 3 + j); b[i][j] = (short)(threadIdx.x * 5 + i + j * 11); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 24 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[SAME_OPERANDS ? 0 : (i + rep) & 3], b[SAME_OPERANDS ? 0 : (i + 2 * rep) & 3], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, bool SAME>
void run(int waves_per_simd, float* out, unsigned long long* cyc) {
  const int iters = 2000, threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rate_kernel<NACC, SAME>), dim3(256), dim3(threads), 0, 0, out, 10, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<NACC, SAME>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 24.0 * iters * waves_per_simd;          // MFMAs per SIMD
  printf("  %d accumulators, %d wave(s)/SIMD, %s operands: %6.1f us  -> %5.1f cycles/MFMA at 2.4 GHz (wall), clock64: %5.1f per MFMA; %7.1f TFLOP/s bf16\n", NACC,
         waves_per_simd, SAME ? "same    " : "rotating", ms * 1e3, ms * 1e-3 * 2.4e9 / n, (double)c / n,
         n * 1024.0 * 2.0 * 8192 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  printf("v_mfma_f32_16x16x32_bf16, 256 workgroups (one per CU):\n");
  run<1, true>(1, out, cyc); run<2, true>(1, out, cyc); run<4, true>(1, out, cyc); run<8, true>(1, out, cyc);
  run<1, true>(2, out, cyc); run<2, true>(2, out, cyc); run<4, true>(2, out, cyc); run<8, true>(2, out, cyc); run<12, true>(2, out, cyc);
  run<4, false>(2, out, cyc); run<8, false>(2, out, cyc); run<12, false>(2, out, cyc);
  return 0;
}
