// What does ONE dependent kernel launch cost on this box, by how it is issued?  (round 5: the reference's speed protocol,
// exp/pennaction/eval_speed2d.py, runs 16 frames per call through ~620 launches -- half of its step time is this floor.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
// Variants, N dependent launches each (every launch reads what the previous one wrote):
//   graph    : stream capture -> hipGraphInstantiate -> hipGraphLaunch
//   eager    : hipLaunchKernelGGL in a C loop on one stream (host keeps the queue full)
// for a trivial kernel at several grid / work-group shapes.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void bump(float* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += 1.0f;
}
__global__ void bump_lds(float* x, int n) {
  extern __shared__ float lds[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  lds[threadIdx.x] = i < n ? x[i] : 0.f;
  __syncthreads();
  if (i < n) x[i] = lds[threadIdx.x] + 1.0f;
}

// distinct code objects: the same trivial work behind different (and differently long) instruction streams
template <int V>
__global__ void bump_v(float* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? x[i] : 0.f;
#pragma unroll
  for (int k = 0; k < 64 * (V + 1); ++k) v = v * 1.0000001f + (float)(k + V) * 1e-9f;      // (V + 1) * 64 dependent FMAs, unrolled
  if (i < n) x[i] = v + 1.0f;
}
struct Fat { float* x; int n; int pad[62]; };      // a 264-byte by-value argument like dh_conv_args
__global__ void bump_fat(const Fat a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) a.x[i] += 1.0f + (float)a.pad[5];
}
__global__ void sweep(float* x, size_t n) {        // writes n floats: dirty lines the next launch boundary has to deal with
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] += 1.0f;
}

__global__ void bump2(float* x, float* far, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = x[i] + far[i]; x[i] = v + 1.0f; far[i] = v; }
}
__global__ void bump5(float* x, const float* a, const float* b, const float* c, float* d, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = x[i] + a[i] + b[i] + c[i]; x[i] = v + 1.0f; d[i] = v; }
}

template <typename F>
static int time_graph(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int N, int REP, const char* what, F issue) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) issue(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  // host cost of hipGraphLaunch itself: the GPU idle before the call, wall clock around the call alone
  double host = 0;
  for (int r = 0; r < 5; ++r) {
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s));
    host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  CK(hipStreamSynchronize(s));
  printf("%-72s graph %6.2f us/node   host: hipGraphLaunch %6.2f us/node\n", what, ms * 1e3 / (REP * N), host / (5 * N));
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  return 0;
}

int main() {
  float* x;
  CK(hipMalloc(&x, 1 << 24));
  CK(hipMemset(x, 0, 1 << 24));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)bump_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  const int N = 600, REP = 20;
  struct Shape { int grid, block, lds; const char* what; };
  const Shape shapes[] = {{1, 64, 0, "1 x 64"}, {20, 256, 0, "20 x 256"}, {20, 1024, 0, "20 x 1024"},
                          {20, 1024, 68 * 1024, "20 x 1024, 68 KB LDS"}, {256, 256, 0, "256 x 256"},
                          {2048, 256, 0, "2048 x 256"}};
  for (const Shape& sh : shapes) {
    const int n = sh.grid * sh.block;
    auto launch = [&]() {
      if (sh.lds) hipLaunchKernelGGL(bump_lds, dim3(sh.grid), dim3(sh.block), sh.lds, s, x, n);
      else hipLaunchKernelGGL(bump, dim3(sh.grid), dim3(sh.block), 0, s, x, n);
    };
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms_g; CK(hipEventElapsedTime(&ms_g, e0, e1));
    // eager
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < REP; ++r)
      for (int i = 0; i < N; ++i) launch();
    auto t1 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms_e; CK(hipEventElapsedTime(&ms_e, e0, e1));
    const double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / (REP * N);
    printf("%-24s graph %6.2f us/node   eager %6.2f us/launch (host issue %5.2f us/launch)\n", sh.what,
           ms_g * 1e3 / (REP * N), ms_e * 1e3 / (REP * N), host_us);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  // ---- what raises the floor inside a real forward? ------------------------------------------------------------
  float* big;
  CK(hipMalloc(&big, (size_t)64 << 20));
  CK(hipMemset(big, 0, (size_t)64 << 20));
  const int n20 = 20 * 256;
  time_graph(s, e0, e1, N, REP, "same kernel, 20 x 256", [&](int) { hipLaunchKernelGGL(bump, dim3(20), dim3(256), 0, s, x, n20); });
  time_graph(s, e0, e1, N, REP, "8 distinct kernels round-robin (64..512 unrolled FMAs), 20 x 256", [&](int i) {
    switch (i & 7) {
      case 0: hipLaunchKernelGGL(bump_v<0>, dim3(20), dim3(256), 0, s, x, n20); break;
      case 1: hipLaunchKernelGGL(bump_v<1>, dim3(20), dim3(256), 0, s, x, n20); break;
      case 2: hipLaunchKernelGGL(bump_v<2>, dim3(20), dim3(256), 0, s, x, n20); break;
      case 3: hipLaunchKernelGGL(bump_v<3>, dim3(20), dim3(256), 0, s, x, n20); break;
      case 4: hipLaunchKernelGGL(bump_v<4>, dim3(20), dim3(256), 0, s, x, n20); break;
      case 5: hipLaunchKernelGGL(bump_v<5>, dim3(20), dim3(256), 0, s, x, n20); break;
      case 6: hipLaunchKernelGGL(bump_v<6>, dim3(20), dim3(256), 0, s, x, n20); break;
      default: hipLaunchKernelGGL(bump_v<7>, dim3(20), dim3(256), 0, s, x, n20); break;
    }
  });
  time_graph(s, e0, e1, N, REP, "one long kernel (512 unrolled FMAs) repeated, 20 x 256", [&](int) { hipLaunchKernelGGL(bump_v<7>, dim3(20), dim3(256), 0, s, x, n20); });
  Fat fat{}; fat.x = x; fat.n = n20;
  time_graph(s, e0, e1, N, REP, "264-byte by-value argument, 20 x 256", [&](int) { hipLaunchKernelGGL(bump_fat, dim3(20), dim3(256), 0, s, fat); });
  for (size_t mb : {1, 4, 16}) {
    char what[96];
    snprintf(what, sizeof what, "tiny kernel alternating with a %zu MB sweep (per PAIR of nodes)", mb);
    const size_t nf = (mb << 20) / 4;
    time_graph(s, e0, e1, N / 2, REP, what, [&](int) {
      hipLaunchKernelGGL(sweep, dim3(512), dim3(256), 0, s, big, nf);
      hipLaunchKernelGGL(bump, dim3(20), dim3(256), 0, s, x, n20);
    });
    snprintf(what, sizeof what, "the %zu MB sweep alone", mb);
    time_graph(s, e0, e1, N / 2, REP, what, [&](int) { hipLaunchKernelGGL(sweep, dim3(512), dim3(256), 0, s, big, nf); });
  }
  // address-translation reach: node i works on 20 KB at offset (i * stride) of a 2 GB allocation (each node still
  // depends on its predecessor through x)
  {
    float* huge;
    const size_t hb = (size_t)2 << 30;
    CK(hipMalloc(&huge, hb));
    CK(hipMemset(huge, 0, hb));
    for (size_t stride_mb : {0, 2, 3, 64}) {
      char what[96];
      snprintf(what, sizeof what, "tiny kernel, node i touches 20 KB at offset i x %zu MB of a 2 GB buffer (+ x)", stride_mb);
      time_graph(s, e0, e1, N, REP, what, [&](int i) {
        float* at = huge + (((size_t)i * stride_mb << 20) % (hb - (1 << 20))) / 4;
        hipLaunchKernelGGL(bump2, dim3(20), dim3(256), 0, s, x, at, n20);
      });
    }
    // the same with FOUR far-apart operands per node (input, weights, residual, output of a conv)
    time_graph(s, e0, e1, N, REP, "tiny kernel, four operands 37 / 91 / 153 / 211 MB strides apart per node", [&](int i) {
      auto at = [&](size_t mb) { return huge + (((size_t)i * mb << 20) % (hb - (1 << 20))) / 4; };
      hipLaunchKernelGGL(bump5, dim3(20), dim3(256), 0, s, x, at(37), at(91), at(153), at(211), n20);
    });
    (void)hipFree(huge);
  }
  // two chains in ONE graph (fork / join through events, as a multi-stream plan is captured): per node of BOTH chains
  {
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    float* x2 = x + (1 << 20);
    bool first = true;
    time_graph(s, e0, e1, N, REP, "two independent chains of N/2 nodes on two captured streams, 20 x 256", [&](int i) {
      if (first) { (void)hipEventRecord(fork, s); (void)hipStreamWaitEvent(s2, fork, 0); first = false; }
      if (i & 1) hipLaunchKernelGGL(bump, dim3(20), dim3(256), 0, s2, x2, n20);
      else hipLaunchKernelGGL(bump, dim3(20), dim3(256), 0, s, x, n20);
      if (i == N - 1) { (void)hipEventRecord(join, s2); (void)hipStreamWaitEvent(s, join, 0); first = true; }
    });
  }
  // a tiny kernel that READS 256 KB the previous (tiny) node did not write: cold after the boundary's invalidate?
  time_graph(s, e0, e1, N, REP, "tiny kernel reading 1 MB written long ago (sweep grid 512 x 256, read-modify-write)", [&](int) {
    hipLaunchKernelGGL(sweep, dim3(512), dim3(256), 0, s, big, (size_t)(1 << 20) / 4); });
  return 0;
}
