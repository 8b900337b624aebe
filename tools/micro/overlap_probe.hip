// Does a latency-bound chain of small kernels on one stream overlap a chip-filling kernel on another?  (round 3)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/overlap_probe.hip -o tools/micro/overlap_probe.bin
// big  : grid of NB work-groups x 256 threads, each spinning on fp32 FMAs for ~T_big
// small: grid of NS work-groups x 256 threads doing ONE dependent global-memory round trip chain (latency-bound),
//        launched CH times back to back on its stream (each launch depends on the previous one through the stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void big(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; a = a * b - 0.5f; a = a * b + 0.25f; a = a * b - 0.25f; }
  if (a == 12345.f) out[0] = a;
}
__global__ void small(const int* __restrict__ chase, int* out, int hops) {
  int p = (blockIdx.x * 256 + threadIdx.x) & 0xffff;
  for (int i = 0; i < hops; ++i) p = chase[p];
  if (p == -1) out[0] = p;
}

int main() {
  float* fo; int *chase, *io;
  CK(hipMalloc(&fo, 4)); CK(hipMalloc(&io, 4)); CK(hipMalloc(&chase, 65536 * 4));
  std::vector<int> h(65536);
  for (int i = 0; i < 65536; ++i) h[i] = (i * 7919 + 13) & 0xffff;
  CK(hipMemcpy(chase, h.data(), 65536 * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, ea, eb, ef;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb)); CK(hipEventCreate(&ef));
  auto time = [&](auto fn, const char* what) {
    for (int w = 0; w < 2; ++w) fn();
    hipDeviceSynchronize();
    hipEventRecord(e0, sa);
    for (int r = 0; r < 5; ++r) fn();
    hipEventRecord(e1, sa);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.1f us\n", what, ms * 1e3f / 5);
    return 0;
  };
  for (int nb : {256 * 2, 256 * 8, 256 * 32}) {
    const int iters = 2000000 / (nb / 256) ;      // total work constant-ish: ~1-2 ms
    for (int ns : {64, 512}) {
      const int CH = 20, hops = 40;
      char buf[128];
      auto A = [&]() { hipLaunchKernelGGL(big, dim3(nb), dim3(256), 0, sa, fo, iters / 8); };
      auto B = [&]() { for (int c = 0; c < CH; ++c) hipLaunchKernelGGL(small, dim3(ns), dim3(256), 0, sa, chase, io, hops); };
      snprintf(buf, sizeof buf, "big %5d WGs alone", nb); time(A, buf);
      snprintf(buf, sizeof buf, "chain of %d x small(%d WGs) alone", CH, ns); time(B, buf);
      auto S = [&]() { A(); B(); };
      time(S, "  serial on one stream");
      auto P = [&]() {          // fork from sa, chain on sb, join
        hipEventRecord(ef, sa); hipStreamWaitEvent(sb, ef, 0);
        hipLaunchKernelGGL(big, dim3(nb), dim3(256), 0, sa, fo, iters / 8);
        for (int c = 0; c < CH; ++c) hipLaunchKernelGGL(small, dim3(ns), dim3(256), 0, sb, chase, io, hops);
        hipEventRecord(eb, sb); hipStreamWaitEvent(sa, eb, 0);
      };
      time(P, "  two streams (eager)");
      // the same captured into a graph
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal);
      P();
      hipStreamEndCapture(sa, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      auto G = [&]() { hipGraphLaunch(ge, sa); };
      time(G, "  two streams (hipGraph)");
    }
  }
  return 0;
}
