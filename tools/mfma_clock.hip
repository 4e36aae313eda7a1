// On-box: what does one fp32 MFMA cost in shader cycles, and what clock does the chip hold while issuing them?
// Each wave stamps s_memtime (shader-clock counter) and s_memrealtime (100 MHz) around a long back-to-back MFMA loop.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* stamps, int iters, float a0, float b0) {
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
  float s = 0.f;
  unsigned long long t0, t1, r0, r1;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < NACC; ++i)
      for (int j = 0; j < 16; ++j) s += acc[i][j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}
template <int SHAPE, int NACC>
void run(int wps, float* out, unsigned long long* st) {
  const int blocks = 256 * wps, iters = 40000 / NACC * (SHAPE == 16 ? 2 : 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_loop<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mfma_loop<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks * 4);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> tick, real;
  for (int w = 0; w < blocks * 4; ++w) { tick.push_back((double)h[2 * w]); real.push_back((double)h[2 * w + 1]); }
  std::sort(tick.begin(), tick.end()); std::sort(real.begin(), real.end());
  const double n = (double)iters * NACC;
  const double mt = tick[tick.size() / 2], mr = real[real.size() / 2];
  const double flops = (double)blocks * 4 * n * (SHAPE == 16 ? 2048.0 : 4096.0);
  printf("mfma_%s x%d acc, %d wave/SIMD: %6.1f TF/s | per MFMA per wave: %6.2f memtime ticks, %6.2f ns | memtime rate %.3f GHz | "
         "per SIMD: %.2f ticks/MFMA\n", SHAPE == 16 ? "16x16x4" : "32x32x2", NACC, wps, flops / ms / 1e9, mt / n, mr * 10.0 / n,
         mt / (mr * 10.0), mt / n / wps);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  unsigned long long* st; hipMalloc(&st, 8 * 2 * 4096 * 4);
  for (int wps = 1; wps <= 2; ++wps) {
    run<16, 1>(wps, out, st);
    run<16, 2>(wps, out, st);
    run<16, 4>(wps, out, st);
    run<16, 8>(wps, out, st);
    run<32, 1>(wps, out, st);
    run<32, 2>(wps, out, st);
    run<32, 4>(wps, out, st);
  }
  return 0;
}
