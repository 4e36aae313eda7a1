// On-box calibration of the fp32 MFMA peak (v_mfma_f32_16x16x4_f32) and an HBM copy rate.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void fill4(float4* __restrict__ b, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = make_float4(v, v, v, v);
}
__global__ void sum4(const float4* __restrict__ a, long n, float* out) {
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.f) out[0] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, 1000, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * 2048.0;
    printf("mfma_f32_16x16x4: %d wave(s)/SIMD: %.1f TFLOP/s (%.2f ms)\n", waves_per_simd, flops / ms / 1e9, ms);
  }
  const long n = 1L << 26;  // 1 GiB of float4
  float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
  hipLaunchKernelGGL(copy4, dim3(4096), dim3(256), 0, 0, a, b, n); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(copy4, dim3(4096), dim3(256), 0, 0, a, b, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("float4 copy: %.2f TB/s (read+write)\n", 5.0 * 2 * n * 16 / ms / 1e9);
  hipLaunchKernelGGL(fill4, dim3(4096), dim3(256), 0, 0, b, n, 1.f); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fill4, dim3(4096), dim3(256), 0, 0, b, n, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("float4 write-only: %.2f TB/s\n", 5.0 * n * 16 / ms / 1e9);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(sum4, dim3(4096), dim3(256), 0, 0, a, n, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("float4 read-only: %.2f TB/s\n", 5.0 * n * 16 / ms / 1e9);
  return 0;
}
