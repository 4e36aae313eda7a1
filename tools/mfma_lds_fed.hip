// On-box: cycles per v_mfma_f32_16x16x4_f32 when the B operand of every MFMA comes from LDS (what the block kernels' loops
// do), 8 waves per workgroup (2 per SIMD), one workgroup per CU.  A in registers; B fragments requested DEPTH steps ahead; a
// step = NC MFMAs on NC independent accumulators (NC column tiles).  STRIDE = row stride of the LDS tile in floats: 112 (== 16
// mod 32: the kq rows of a 32-lane group on different banks), 110 (== 14: the block data gradient's), 96 (== 0: 4-way).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_fed.hip -o /tmp/mlf && /tmp/mlf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NC, int DEPTH, int STRIDE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* st, int iters) {
  __shared__ float tile[100 * STRIDE];
  for (int i = threadIdx.x; i < 100 * STRIDE; i += 512) tile[i] = 1.f + 1e-3f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const float* yb = tile + kq * STRIDE + l15;
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = 1.f + i + lane * 1e-3f;
  f32x4 acc[NC];
  for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NSTEP = 48;                  // steps per outer iteration (compile-time LDS offsets, like the kernels)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float bb[DEPTH][NC];
    auto fetch = [&](int s) {
#pragma unroll
      for (int c = 0; c < NC; ++c) bb[s % DEPTH][c] = yb[4 * (s % 24) * STRIDE + (s / 24) * 3 + c * 16];
    };
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) fetch(i);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + DEPTH - 1 < NSTEP) fetch(s + DEPTH - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 15], bb[s % DEPTH][c], acc[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s_ = 0.f;
  for (int c = 0; c < NC; ++c) s_ += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 512 + threadIdx.x] = s_;
  if (lane == 0) st[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NC, int DEPTH, int STRIDE>
void run(float* out, unsigned long long* st) {
  const int iters = 200;
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<NC, DEPTH, STRIDE>), dim3(256), dim3(512), 0, 0, out, st, iters); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("NC %d column tiles, %d step(s) ahead, row stride %3d: %6.2f cycles per MFMA per SIMD (2 waves per SIMD)\n", NC, DEPTH - 1, STRIDE,
         (double)h[h.size() / 2] / ((double)iters * 48 * NC) / 2);
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  unsigned long long* st; hipMalloc(&st, 8 * 256 * 8);
  run<4, 2, 112>(out, st); run<4, 3, 112>(out, st); run<4, 2, 110>(out, st); run<4, 3, 110>(out, st); run<4, 2, 96>(out, st);
  run<2, 2, 112>(out, st); run<2, 3, 112>(out, st); run<2, 5, 112>(out, st); run<2, 5, 110>(out, st);
  run<1, 2, 112>(out, st); run<1, 5, 112>(out, st); run<1, 8, 112>(out, st);
  return 0;
}
