// The BUFFER form of the LDS-DMA: buffer_load_dwordx4 ... lds (16 bytes per lane to m0 + lane * 16) - does it exist on gfx950, and do lanes whose
// offset is out of range write ZEROS (what a row-wise DMA staging of frame tiles needs for the columns outside the frame)?
// hipcc --offload-arch=gfx950 -O3 tools/blds_check.hip -o /tmp/blds_check && /tmp/blds_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, int nvalid_bytes) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63;
  for (int e = lane; e < 512; e += 64) sm[e] = -1.f;
  __builtin_amdgcn_s_barrier();
  const unsigned long a = (unsigned long)src;
  const i32x4 rsrc = {(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFF), nvalid_bytes, 0x00020000};
  const unsigned m0v = (unsigned)(unsigned long long)sm + 8u;          // an 8-byte aligned LDS base
  const int voff = lane < 40 ? lane * 16 : 0x7ffffff0;                  // lanes 40.. point far out of range
  asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(m0v) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int e = lane; e < 512; e += 64) dst[e] = sm[e];
}
int main() {
  float *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 4096);
  float h[512]; for (int i = 0; i < 512; ++i) h[i] = 100.f + i;
  hipMemcpy(s, h, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, s, d, 32 * 16);     // 32 lanes' worth of bytes are in range
  hipError_t e = hipDeviceSynchronize();
  float o[512]; hipMemcpy(o, d, 2048, hipMemcpyDeviceToHost);
  int ok = 0, zeros = 0, untouched = 0;
  for (int l = 0; l < 64; ++l)
    for (int q = 0; q < 4; ++q) {
      const float v = o[2 + l * 4 + q];
      if (l < 32) ok += v == 100.f + l * 4 + q;
      else { zeros += v == 0.f; untouched += v == -1.f; }
    }
  printf("%s; in-range lanes 0..31: %d of 128 floats in place; out-of-range lanes 32..63 (offset past num_records | far away): %d zeros, %d untouched of 128\n",
         hipGetErrorString(e), ok, zeros, untouched);
  return 0;
}
