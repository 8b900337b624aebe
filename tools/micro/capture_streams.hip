// Which multi-stream capture patterns does this ROCm's hipStreamEndCapture / hipGraphLaunch survive?
// hipcc --offload-arch=gfx950 -O2 tools/micro/capture_streams.hip -o tools/micro/capture_streams.bin
// usage: capture_streams.bin <nside> <pattern>   pattern bits: 1 = side->side waits, 2 = origin->side waits mid-capture,
//        4 = same event waited by several streams, 8 = side stream with no kernel, 16 = many rounds (50)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(err_)); exit(2); } } while (0)
__global__ void k(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main(int argc, char** argv) {
  const int nside = argc > 1 ? atoi(argv[1]) : 2, pat = argc > 2 ? atoi(argv[2]) : 0;
  const int rounds = (pat & 16) ? 50 : 4;
  float* buf; CK(hipMalloc(&buf, (nside + 1) * 4096 * sizeof(float)));
  hipStream_t s0; CK(hipStreamCreate(&s0));
  std::vector<hipStream_t> side(nside);
  for (auto& s : side) CK(hipStreamCreate(&s));
  auto ev = []() { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); return e; };
  hipEvent_t fork = ev();
  std::vector<hipEvent_t> joins;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, s0));
  for (auto& s : side) CK(hipStreamWaitEvent(s, fork, 0));
  for (int r = 0; r < rounds; ++r) {
    k<<<16, 256, 0, s0>>>(buf, 4096);
    for (int i = 0; i < nside; ++i)
      if (!((pat & 8) && i == nside - 1)) k<<<16, 256, 0, side[i]>>>(buf + (i + 1) * 4096, 4096);
    if ((pat & 1) && nside >= 2) {            // side[1] waits for side[0]
      hipEvent_t e = ev(); CK(hipEventRecord(e, side[0])); CK(hipStreamWaitEvent(side[1], e, 0));
      if (pat & 4) CK(hipStreamWaitEvent(s0, e, 0));
    }
    if (pat & 2) {                            // sides wait for the origin mid-capture
      hipEvent_t e = ev(); CK(hipEventRecord(e, s0));
      for (auto& s : side) CK(hipStreamWaitEvent(s, e, 0));
    }
  }
  for (auto& s : side) { hipEvent_t e = ev(); CK(hipEventRecord(e, s)); CK(hipStreamWaitEvent(s0, e, 0)); }
  hipGraph_t g; CK(hipStreamEndCapture(s0, &g));
  hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s0));
  CK(hipStreamSynchronize(s0));
  printf("OK nside=%d pattern=%d\n", nside, pat);
  return 0;
}
