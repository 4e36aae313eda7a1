// On-box: do fp32 MFMAs (v_mfma_f32_16x16x4_f32 / 32x32x2) co-execute with ordinary VALU work of the OTHER wave on the
// same SIMD?  One 512-thread workgroup per CU: waves 0-3 run a back-to-back MFMA loop, waves 4-7 a v_fma_f32 loop (eight
// independent chains) or an LDS-read loop.  Cycles per wave, each role alone and both together.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE, int OTHER>   // OTHER: 0 v_fma, 1 ds_read_b32, 2 v_exp_f32
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int n_mfma, int n_other, float a0) {
  __shared__ float lds[4096];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = a0 + i;
  __syncthreads();
  float s = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  if (wave < 4) {
    float a = a0 + threadIdx.x * 1e-3f, b = a0 - threadIdx.x * 1e-3f;
    if (SHAPE == 16) {
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < n_mfma; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      }
      t1 = __builtin_amdgcn_s_memtime();
      s = c0[0] + c1[1] + c0[2] + c1[3];
    } else {
      f32x16 c0, c1;
      for (int j = 0; j < 16; ++j) { c0[j] = 0.f; c1[j] = 0.f; }
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < n_mfma; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      }
      t1 = __builtin_amdgcn_s_memtime();
      for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
    }
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 + i + threadIdx.x;
    const float m = 1.0001f, ad = 0.5f;
    t0 = __builtin_amdgcn_s_memtime();
    if (OTHER == 0) {
      for (int it = 0; it < n_other; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(ad));
      }
    } else if (OTHER == 1) {
      int idx = (threadIdx.x & 63);
      for (int it = 0; it < n_other; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += lds[(idx + 64 * i + it) & 4095];
      }
    } else {
      for (int it = 0; it < n_other; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int SHAPE, int OTHER>
void run(float* out, unsigned long long* st, int n_mfma, int n_other, const char* what) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<SHAPE, OTHER>), dim3(256), dim3(512), 0, 0, out, st, n_mfma, n_other, 1.f);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> m, o;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : o).push_back((double)h[b * 8 + w]);
  std::sort(m.begin(), m.end()); std::sort(o.begin(), o.end());
  printf("%-44s mfma wave: %8.0f cycles (%6.2f / MFMA)   other wave: %8.0f cycles (%6.2f / instr)\n", what, m[m.size() / 2],
         n_mfma ? m[m.size() / 2] / (2.0 * n_mfma) : 0.0, o[o.size() / 2], n_other ? o[o.size() / 2] / (8.0 * n_other) : 0.0);
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  unsigned long long* st; hipMalloc(&st, 8 * 256 * 8);
  const int NM = 4000, NO = 4000;
  run<16, 0>(out, st, NM, 0, "16x16x4 alone");
  run<16, 0>(out, st, 0, NO, "v_fma alone");
  run<16, 0>(out, st, NM, NO, "16x16x4 + v_fma (32000 each... ratio 1:4)");
  run<16, 0>(out, st, NM, NO / 4, "16x16x4 + v_fma (1 fma per MFMA)");
  run<16, 0>(out, st, NM, NO * 2, "16x16x4 + v_fma (8 fma per MFMA)");
  run<32, 0>(out, st, NM / 2, 0, "32x32x2 alone");
  run<32, 0>(out, st, NM / 2, NO, "32x32x2 + v_fma (8 fma per MFMA)");
  run<16, 1>(out, st, 0, NO, "ds_read alone");
  run<16, 1>(out, st, NM, NO, "16x16x4 + ds_read (4 reads per MFMA)");
  run<16, 2>(out, st, 0, NO, "v_exp alone");
  run<16, 2>(out, st, NM, NO, "16x16x4 + v_exp (4 exp per MFMA)");
  return 0;
}
