// Micro-benchmark: what bounds the K loop of the split-bf16 GEMM (gemm1x1s.hip) when memory is taken away?
// One "K-step" = two chunks of [NR/2 ds_read_b128 -> 6 MFMA -> s_waitcnt lgkmcnt(0) -> 12 MFMA with NV/2 VALU spread
// behind them], optionally closed by s_barrier; v_mfma_f32_32x32x16_bf16 on three accumulators in rotation, exactly
// the kernel's stream.  WPS = waves per SIMD (work-groups of 256 threads, WPS of them per CU through the LDS size).
// Reported: matrix-pipe busy fraction = 32 cycles x MFMAs per SIMD / elapsed shader cycles (s_memtime), and the clock.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/split_loop_model.hip -o /tmp/slm && /tmp/slm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NR, int NV, bool BAR, int NT = 256>
__global__ __launch_bounds__(NT) void k(float* out, long long* clk, int iters) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < 2048; i += NT) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  f32x16 acc[3];
  for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[3], b[3];
  for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(float)(threadIdx.x + i); b[i][e] = (__bf16)(float)(e + i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
  const unsigned la = (unsigned)((threadIdx.x & 63) * 16);
  float4 q[NR > 0 ? NR / 2 : 1];
  float qs = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int r = 0; r < NR / 2; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[r]) : "v"(la), "n"(r * 1024));
#pragma unroll
      for (int m = 0; m < 6; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 3], b[(m / 3) % 3], acc[m % 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (NR > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        qs += q[0].x + q[NR / 2 - 1].y;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m + 1) % 3], b[(m / 3) % 3], acc[m % 3], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < (NV / 2 + 11 - m) / 12; ++f)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m + f) & 7]) : "v"(v[(m + f + 3) & 7]), "v"(qs));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = qs;
  for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NR, int NV, bool BAR>
void run(int wps) {
  float* out; long long* clk;
  const int grid = 256 * wps;
  hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&clk, grid * sizeof(long long));
  const int iters = 4000;
  const size_t lds = wps == 1 ? 100 * 1024 : wps == 2 ? 72 * 1024 : 36 * 1024;
  auto kern = k<NR, NV, BAR>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 256, lds>>>(out, clk, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<grid, 256, lds>>>(out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c0; hipMemcpy(&c0, clk, sizeof(c0), hipMemcpyDeviceToHost);
  const double mf = 36.0 * iters * wps;                     // MFMAs per SIMD
  printf("waves/SIMD %d  reads %2d  valu %3d  barrier %d : %.3f ms  s_memtime ticks %lld (%.0f MHz)  pipe busy %.1f %% of wall @2.4 GHz\n", wps, NR, NV, (int)BAR, ms, c0, c0 / (ms * 1e3), 100.0 * mf * 32 / (ms * 1e-3 * 2.4e9));
  hipFree(out); hipFree(clk);
}

int main() {
  for (int wps = 1; wps <= 2; ++wps) {
    run<0, 0, false>(wps); run<0, 0, true>(wps);
    run<22, 0, false>(wps); run<22, 0, true>(wps);
    run<0, 104, false>(wps); run<0, 152, false>(wps); run<0, 152, true>(wps);
    run<22, 104, true>(wps); run<22, 152, true>(wps); run<22, 56, true>(wps);
  }
  return 0;
}
