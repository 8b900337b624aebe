// Replays the executor's exact launch / event pattern (tools/micro/pattern_s<N>.txt: per step "stream record nwait waits..")
// with dummy kernels, to tell a runtime problem with the PATTERN from one with the real kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(err_)); exit(2); } } while (0)
__global__ void k(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "r"); int ns, n; if (!f || fscanf(f, "%d %d", &ns, &n) != 2) return 3;
  const int flags = argc > 2 ? atoi(argv[2]) : 2;
  const int relay = argc > 3 ? atoi(argv[3]) : 0;   // event flags: 0 default(timing), 2 disable timing, 1 blocking sync
  std::vector<int> st(n), rec(n); std::vector<std::vector<int>> waits(n);
  for (int i = 0; i < n; ++i) { int nw; fscanf(f, "%d %d %d", &st[i], &rec[i], &nw); waits[i].resize(nw); for (auto& w : waits[i]) fscanf(f, "%d", &w); }
  float* buf; CK(hipMalloc(&buf, 4096 * sizeof(float) * 8));
  std::vector<hipStream_t> s(ns); for (auto& x : s) CK(hipStreamCreate(&x));
  auto ev = [&]() { hipEvent_t e; CK(hipEventCreateWithFlags(&e, flags)); return e; };
  std::vector<hipEvent_t> evs(n, nullptr); for (int i = 0; i < n; ++i) if (rec[i]) evs[i] = ev();
  hipEvent_t fork = ev(); std::vector<hipEvent_t> join(ns); for (auto& e : join) e = ev();
  CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, s[0]));
  for (int j = 1; j < ns; ++j) CK(hipStreamWaitEvent(s[j], fork, 0));
  for (int i = 0; i < n; ++i) {
    for (int w : waits[i]) {
      // relay: a lower-numbered side stream never waits directly on a higher-numbered side stream's event
      if (relay && st[i] != 0 && st[w] != 0 && st[w] > st[i]) {
        CK(hipStreamWaitEvent(s[0], evs[w], 0));
        hipEvent_t r = ev(); CK(hipEventRecord(r, s[0])); CK(hipStreamWaitEvent(s[st[i]], r, 0));
      } else {
        CK(hipStreamWaitEvent(s[st[i]], evs[w], 0));
      }
    }
    k<<<16, 256, 0, s[st[i]]>>>(buf + st[i] * 4096, 4096);
    if (rec[i]) CK(hipEventRecord(evs[i], s[st[i]]));
  }
  for (int j = 1; j < ns; ++j) { CK(hipEventRecord(join[j], s[j])); CK(hipStreamWaitEvent(s[0], join[j], 0)); }
  hipGraph_t g; CK(hipStreamEndCapture(s[0], &g));
  hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s[0]));
  CK(hipStreamSynchronize(s[0]));
  printf("OK %s streams=%d steps=%d flags=%d relay=%d\n", argv[1], ns, n, flags, relay);
  return 0;
}
