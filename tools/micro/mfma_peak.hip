// Micro-benchmark: what fraction of the nominal fp32 MFMA peak can back-to-back v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32 issue reach on gfx950 with W waves per SIMD and A independent accumulators per wave?
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[ACC];
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[ACC];
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < ACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int acc, double flop_per_mfma, int wgs_per_cu) {
  float* out; hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
  const int iters = 4000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 256>>>(out, 100, 1.f, 2.f); hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 8 * acc * flop_per_mfma;
  printf("%-10s acc=%d waves/SIMD=%d : %.2f ms  %.1f TFLOP/s\n", name, acc, wgs_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 4; ++w) {
    run("32x32x2", k32<1>, 1, 4096, w);
    run("32x32x2", k32<2>, 2, 4096, w);
    run("32x32x2", k32<4>, 4, 4096, w);
    run("16x16x4", k16<2>, 2, 2048, w);
    run("16x16x4", k16<4>, 4, 2048, w);
  }
  return 0;
}
