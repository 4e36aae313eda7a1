// On-box: sustained rate of the two fp32 MFMA shapes (16x16x4: 2048 flop / 32 cycles; 32x32x2: 4096 flop / 64 cycles - the
// same nominal rate, but the larger tile reads half the operands per flop).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak2.hip -o /tmp/mfma_peak2 && /tmp/mfma_peak2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) s += acc[i][j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape = 16; shape <= 32; shape += 16)
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
      const int blocks = 256 * waves_per_simd, iters = 20000;
      auto run = [&](int it) {
        if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f);
        else hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f);
      };
      run(1000); hipDeviceSynchronize();
      hipEventRecord(e0); run(iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 * iters * (shape == 16 ? 8 * 2048.0 : 4 * 4096.0);
      printf("mfma_f32_%s: %d wave(s)/SIMD: %.1f TFLOP/s (%.2f ms)\n", shape == 16 ? "16x16x4" : "32x32x2", waves_per_simd,
             flops / ms / 1e9, ms);
    }
  return 0;
}
