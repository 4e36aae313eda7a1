// LDS-DMA check: global_load_lds_dwordx4 issued from inline asm writes m0 + lane * 16 (what the block kernels rely on for the k15 tables).
// hipcc --offload-arch=gfx950 -O3 tools/glds_check.hip -o /tmp/glds_check && /tmp/glds_check
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, int n) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // hidden LDS-DMA: 16 bytes per lane, 1 KiB per wave instruction, destination = m0 + lane * 16
  const unsigned lds0 = (unsigned)(unsigned long long)sm;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float* g = src + ((i * 8 + wave) * 64 + lane) * 4;
    const unsigned m0v = lds0 + (i * 8 + wave) * 1024;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(m0v) : "memory", "m0");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int e = tid; e < n; e += blockDim.x) dst[e] = sm[e] * 2.f;
}
int main() {
  const int n = 4 * 8 * 256;
  float *s, *d; hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
  float* h = new float[n]; for (int i = 0; i < n; ++i) h[i] = i;
  hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(512), n * 4, 0, s, d, n);
  hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < n; ++i) bad += h[i] != 2.f * i;
  printf("bad %d of %d\n", bad, n);
  return bad != 0;
}
