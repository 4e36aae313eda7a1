// Write-only HBM stream rate on this box for several store forms (bounds the quantizer forward, which writes 32 of every
// 34 bytes it moves).  hipcc --offload-arch=gfx950 -O3 tools/write_peak.hip -o /tmp/write_peak && /tmp/write_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void fill_plain(f32x4* __restrict__ b, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = (f32x4){v, v, v, v};
}
__global__ void fill_nt(f32x4* __restrict__ b, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    __builtin_nontemporal_store((f32x4){v, v, v, v}, b + i);
}
// each workgroup owns a contiguous chunk (like a quantizer workgroup writing one frame's p)
__global__ void fill_chunk(f32x4* __restrict__ b, long n, float v, long chunk) {
  for (long c = blockIdx.x; c * chunk < n; c += gridDim.x) {
    f32x4* p = b + c * chunk;
    for (long i = threadIdx.x; i < chunk && c * chunk + i < n; i += blockDim.x) p[i] = (f32x4){v, v, v, v};
  }
}
__global__ void fill_chunk_nt(f32x4* __restrict__ b, long n, float v, long chunk) {
  for (long c = blockIdx.x; c * chunk < n; c += gridDim.x) {
    f32x4* p = b + c * chunk;
    for (long i = threadIdx.x; i < chunk && c * chunk + i < n; i += blockDim.x) __builtin_nontemporal_store((f32x4){v, v, v, v}, p + i);
  }
}
int main() {
  const long n = 1L << 26;  // 1 GiB
  f32x4* b; hipMalloc(&b, n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
#define RUN(name, launch)                                                              \
  launch; hipDeviceSynchronize(); hipEventRecord(e0);                                  \
  for (int r = 0; r < 5; ++r) { launch; }                                              \
  hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);       \
  printf("%-42s %.2f TB/s\n", name, 5.0 * n * 16 / ms / 1e9);
  for (int g : {1024, 4096, 16384}) {
    char nm[64];
    snprintf(nm, 64, "plain float4, grid %d", g); RUN(nm, hipLaunchKernelGGL(fill_plain, dim3(g), dim3(256), 0, 0, b, n, 1.f));
    snprintf(nm, 64, "nontemporal float4, grid %d", g); RUN(nm, hipLaunchKernelGGL(fill_nt, dim3(g), dim3(256), 0, 0, b, n, 1.f));
  }
  for (long ch : {2048L, 8192L}) {   // 32 KB (one frame of p) and 128 KB chunks
    char nm[64];
    snprintf(nm, 64, "chunked %ld KB/WG plain, grid 2048", ch * 16 / 1024); RUN(nm, hipLaunchKernelGGL(fill_chunk, dim3(2048), dim3(256), 0, 0, b, n, 1.f, ch));
    snprintf(nm, 64, "chunked %ld KB/WG nt, grid 2048", ch * 16 / 1024); RUN(nm, hipLaunchKernelGGL(fill_chunk_nt, dim3(2048), dim3(256), 0, 0, b, n, 1.f, ch));
  }
  return 0;
}
