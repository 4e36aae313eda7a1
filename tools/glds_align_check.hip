// Does the LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane to m0 + lane * 16) accept an LDS base that is only 8- or 4-byte aligned,
// and a partial wave (exec mask)?  One wave loads 28 lanes' worth (a 110-float row of the data-gradient tiles = 27.5 chunks) to
// base + SHIFT floats; the LDS array is read back whole.  Prints, per SHIFT, whether every float landed at base + SHIFT + lane * 4 + e.
// hipcc --offload-arch=gfx950 -O3 tools/glds_align_check.hip -o /tmp/glds_align_check && /tmp/glds_align_check
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, int shift) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63;
  for (int e = lane; e < 512; e += 64) sm[e] = -1.f;
  __builtin_amdgcn_s_barrier();
  const unsigned m0v = (unsigned)(unsigned long long)sm + 4u * (unsigned)shift;
  if (lane < 28) {
    const float* g = src + lane * 4;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(m0v) : "memory", "m0");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int e = lane; e < 512; e += 64) dst[e] = sm[e];
}
int main() {
  float *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 4096);
  float h[512]; for (int i = 0; i < 512; ++i) h[i] = 100.f + i;
  hipMemcpy(s, h, 2048, hipMemcpyHostToDevice);
  for (int shift : {0, 2, 1, 3, 6}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, s, d, shift);
    float o[512]; hipMemcpy(o, d, 2048, hipMemcpyDeviceToHost);
    int ok = 0, stray = 0;
    for (int i = 0; i < 512; ++i) {
      const int j = i - shift;
      if (j >= 0 && j < 112) ok += o[i] == 100.f + j;
      else stray += o[i] != -1.f;
    }
    printf("base + %d floats (%2d-byte aligned): %3d of 112 floats in place, %d stray writes; first words:", shift, (shift * 4) % 16 == 0 ? 16 : ((shift * 4) % 8 == 0 ? 8 : 4), ok, stray);
    for (int i = 0; i < 8; ++i) printf(" %.0f", o[i]);
    printf("\n");
  }
  return 0;
}
