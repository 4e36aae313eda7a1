// tools/drain_probe.hip - what the END of a kernel that wrote a lot costs the next dispatch, and whether the store flavour changes it.
// A graph of 40 x { writer (256 workgroups x 512 lanes, MB megabytes of float4 stores), tiny (one workgroup, one add) } is replayed and
// compared with 40 x writer and 40 x tiny alone; writers: plain stores | sc0 sc1 (write-through) | nt.  If the dirty lines a writer
// leaves in the eight L2s are flushed at its end-of-kernel release, write-through stores should make { writer, tiny } cheaper.
//   hipcc --offload-arch=gfx950 -O3 tools/drain_probe.hip -o /tmp/drain_probe && /tmp/drain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));


template <int AUX>
__global__ __launch_bounds__(512) void writer_kernel(float* out, long n4, float v) {
  const long per = (n4 + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(n4, lo + per);
  const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(out + 4 * lo, 0, (unsigned)((hi - lo) * 16), 0x00020000);
  const f32x4 x = {v, v + 1.f, v + 2.f, v + 3.f};
  for (long i = threadIdx.x; i < hi - lo; i += 512) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, x), d, (int)(i * 16), 0, AUX);
}
__global__ void tiny_kernel(int* c) { if (threadIdx.x == 0) c[0] += 1; }
__global__ __launch_bounds__(512) void reader_kernel(const float* in, long n4, float* sink) {
  const long per = (n4 + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(n4, lo + per);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (long i = lo + threadIdx.x; i < hi; i += 512) s += reinterpret_cast<const f32x4*>(in)[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[0] = 1.f;
}

static void launch_writer(int mode, float* o, long n4, hipStream_t st) {
  if (mode == 0) hipLaunchKernelGGL(writer_kernel<0>, dim3(256), dim3(512), 0, st, o, n4, 1.f);
  else if (mode == 1) hipLaunchKernelGGL(writer_kernel<0x11>, dim3(256), dim3(512), 0, st, o, n4, 1.f);   // sc0 | sc1
  else hipLaunchKernelGGL(writer_kernel<2>, dim3(256), dim3(512), 0, st, o, n4, 1.f);                      // nt (slc bit of the aux field)
}

int main() {
  const int N = 40;
  hipStream_t st; hipStreamCreate(&st);
  int* cnt; hipMalloc(&cnt, 4); hipMemset(cnt, 0, 4);
  float* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* mname[3] = {"plain", "sc0 sc1", "nt"};
  for (int mb : {1, 8, 32, 128}) {
    const long n4 = (long)mb * 1024 * 1024 / 16;
    float* buf; hipMalloc(&buf, n4 * 16);
    auto time_graph = [&](auto body) {
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int i = 0; i < N; ++i) body();
      hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, st); hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
      return 1e3 * ms / (5 * N);
    };
    const double t_tiny = time_graph([&] { hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st, cnt); });
    printf("%3d MB per writer launch; tiny alone %.2f us per launch\n", mb, t_tiny);
    for (int mode = 0; mode < 3; ++mode) {
      const double tw = time_graph([&] { launch_writer(mode, buf, n4, st); });
      const double twt = time_graph([&] { launch_writer(mode, buf, n4, st); hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st, cnt); });
      const double twr = time_graph([&] { launch_writer(mode, buf, n4, st); hipLaunchKernelGGL(reader_kernel, dim3(256), dim3(512), 0, st, buf, n4, sink); });
      printf("  %-8s writer %7.2f us (%5.0f GB/s)   writer + tiny %7.2f (tiny costs %5.2f)   writer + reader of the same bytes %7.2f\n", mname[mode], tw,
             mb * 1.048576e6 / tw / 1e3, twt, twt - tw, twr);
    }
    hipFree(buf);
  }
  return 0;
}
