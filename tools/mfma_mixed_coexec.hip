// Do fp32 MFMAs (v_mfma_f32_16x16x4_f32: executed on the vector ALUs of gfx950) and bf16 MFMAs (v_mfma_f32_16x16x32_bf16: matrix cores) of
// the two waves that share a SIMD overlap?  512 threads per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 role B.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_mixed_coexec.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// role: 0 idle, 1 fp32 MFMA loop, 2 bf16 MFMA loop, 3 VALU fma loop
__device__ __forceinline__ float work(int role, int iters, int seed) {
  float out = 0.f;
  if (role == 1) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = seed * 0.001f, b = seed * 0.002f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
    out = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else if (role == 2) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(seed + j); b[j] = (short)(seed * 3 + j); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
    out = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else if (role == 3) {
    float x0 = seed * 0.5f, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0002f, 0.25f); x2 = fmaf(x2, 0.9999f, 0.125f); x3 = fmaf(x3, 0.9998f, 1.f); }
    out = x0 + x1 + x2 + x3;
  }
  return out;
}
__global__ __launch_bounds__(512) void k(float* out, int roleA, int roleB, int iters) {
  const int wave = threadIdx.x >> 6;
  const float v = work(wave < 4 ? roleA : roleB, iters, threadIdx.x);
  out[blockIdx.x * 512 + threadIdx.x] = v;
}
static float run(float* out, int a, int b, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, a, b, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, a, b, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int it = 4000;     // fp32: 8 MFMAs x 32 cycles = 256 cycles / iteration; bf16: 16 x 16 = 256 cycles / iteration; VALU: 64 fma
  const char* nm[] = {"idle", "fp32 MFMA", "bf16 MFMA", "VALU fma"};
  int pairs[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {2, 3}, {1, 3}};
  for (auto& p : pairs) printf("waves 0-3: %-10s | waves 4-7: %-10s : %8.1f us\n", nm[p[0]], nm[p[1]], run(out, p[0], p[1], it));
  return 0;
}
