// On-box: does a v_mfma_f32_16x16x4_f32 whose accumulator lives in ARCHITECTURAL VGPRs (what hipcc allocates in the 256-register
// block kernels) issue as fast as one with an AGPR accumulator?  Inline asm pins the register class.  NACC accumulators in rotation.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_vgpr_acc.hip -o /tmp/mva && /tmp/mva
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, bool AGPR, bool DISTINCT_AB>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* st, int iters) {
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = 1.f + threadIdx.x * 1e-3f + i; b[i] = 2.f - threadIdx.x * 1e-3f - i; }
  f32x4 c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int q = DISTINCT_AB ? (i & 3) : 0;
      if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c[i]) : "v"(a[q]), "v"(b[q]));
      else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a[q]), "v"(b[q]));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NACC, bool AGPR, bool DAB>
void run(int wps, float* out, unsigned long long* st, const char* what) {
  const int blocks = 256 * wps, iters = 24000 / NACC;
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<NACC, AGPR, DAB>), dim3(blocks), dim3(256), 0, 0, out, st, iters); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-52s %d wave/SIMD: %6.2f cycles per MFMA per SIMD\n", what, wps, (double)h[h.size() / 2] / ((double)iters * NACC) / wps);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  unsigned long long* st; hipMalloc(&st, 8 * 4096 * 4);
  for (int wps = 1; wps <= 2; ++wps) {
    run<2, true, false>(wps, out, st, "AGPR acc x2, same A/B");
    run<2, false, false>(wps, out, st, "VGPR acc x2, same A/B");
    run<4, true, false>(wps, out, st, "AGPR acc x4, same A/B");
    run<4, false, false>(wps, out, st, "VGPR acc x4, same A/B");
    run<4, true, true>(wps, out, st, "AGPR acc x4, distinct A/B");
    run<4, false, true>(wps, out, st, "VGPR acc x4, distinct A/B");
    run<8, true, true>(wps, out, st, "AGPR acc x8, distinct A/B");
    run<8, false, true>(wps, out, st, "VGPR acc x8, distinct A/B");
    run<1, true, false>(wps, out, st, "AGPR acc x1 (dependent chain)");
    run<1, false, false>(wps, out, st, "VGPR acc x1 (dependent chain)");
  }
  return 0;
}
