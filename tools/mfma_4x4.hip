// On-box: v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, K = 1): lane layout (checked with exact integer data) and issue cost.
// Used for the 4 left-over channels of a 20-row operand: rows = 4 channels (A identical in every block), columns =
// 16 blocks x 4 = 64 consecutive time steps on the lanes, one instruction per k.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_4x4.hip -o /tmp/mfma_4x4 && /tmp/mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void layout(float* out) {
  const int l = threadIdx.x;
  // hypothesis: A lane l = A[block l/4][row l%4], B lane l = B[block l/4][col l%4], D reg r of lane l = D[block l/4][row r][col l%4]
  const float a = 1 + (l & 3) + 10 * (l >> 2);        // A[blk][i] = 1 + i + 10 blk
  const float b = 100 + (l & 3) + 1000 * (l >> 2);     // B[blk][j] = 100 + j + 1000 blk
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void layout_bcast(float* out) {
  const int l = threadIdx.x;
  const float a = 1 + (l & 3) + 10 * (l >> 2);
  const float b = 100 + (l & 3) + 1000 * (l >> 2);
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);   // cbsz = 4, abid = 0: every block uses block 0's A
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
template <int NACC>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* st, int iters) {
  float a = 1.f + threadIdx.x * 1e-3f, b = 2.f - threadIdx.x * 1e-3f;
  f32x4 c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// mixed stream: 4 dense 16x16x4 + 1 4x4x1 per step (what a k9 wave would issue)
__global__ __launch_bounds__(256) void mixed(float* out, unsigned long long* st, int iters) {
  float a = 1.f + threadIdx.x * 1e-3f, b = 2.f - threadIdx.x * 1e-3f;
  f32x4 c[4], d = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = d[0] + d[1] + d[2] + d[3];
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  unsigned long long* st; hipMalloc(&st, 8 * 4096);
  std::vector<float> h(256);
  for (int variant = 0; variant < 2; ++variant) {
    if (variant == 0) hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
    else hipLaunchKernelGGL(layout_bcast, dim3(1), dim3(64), 0, 0, out);
    hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int blk = l >> 2, j = l & 3;
        const float exp_ = (1 + r + 10 * (variant ? 0 : blk)) * (100.f + j + 1000 * blk);
        if (h[l * 4 + r] != exp_) { if (bad < 4) printf("  mismatch lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], exp_); ++bad; }
      }
    printf("layout check (%s): %s\n", variant ? "cbsz=4 broadcast of block 0's A" : "per-block A", bad ? "MISMATCH" : "OK: D[reg r] of lane l = A[blk][r] * B[blk][l%4]");
  }
  auto med = [&](int n) { std::vector<unsigned long long> v(n); hipMemcpy(v.data(), st, n * 8, hipMemcpyDeviceToHost); std::sort(v.begin(), v.end()); return (double)v[n / 2]; };
  const int iters = 20000;
  hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, st, iters); hipDeviceSynchronize();
  printf("4x4x1_16b, 1 accumulator : %.2f cycles per instruction\n", med(1024) / iters);
  hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, out, st, iters); hipDeviceSynchronize();
  printf("4x4x1_16b, 2 accumulators: %.2f cycles per instruction\n", med(1024) / iters / 2);
  hipLaunchKernelGGL(rate<4>, dim3(256), dim3(256), 0, 0, out, st, iters); hipDeviceSynchronize();
  printf("4x4x1_16b, 4 accumulators: %.2f cycles per instruction\n", med(1024) / iters / 4);
  hipLaunchKernelGGL(mixed, dim3(256), dim3(256), 0, 0, out, st, iters); hipDeviceSynchronize();
  printf("4 x 16x16x4 + 1 x 4x4x1  : %.2f cycles per step (4 * 32 = 128 without the small one)\n", med(1024) / iters);
  return 0;
}
