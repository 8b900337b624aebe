// Micro-benchmark: do fp32 MFMA (v_mfma_f32_32x32x2_f32) and fp32 VALU work issued by ANOTHER wave of the same SIMD
// overlap on gfx950, or do they share the SIMD's fp32 throughput?  (Both peak at 64 FLOP/clk/SIMD = 157.3 TFLOP/s.)
// (prio: s_setprio 3 on the MFMA waves (1) or on the other waves (2).)
// Work-group = 8 waves: waves 0-3 (one per SIMD) run `mf` MFMA batches, waves 4-7 run `va` VALU batches of the chosen
// kind.  Reported: time of MFMA alone, VALU alone, both together.  "both ~ max" = separate pipes, "both ~ sum" = shared.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_share.hip -o /tmp/mvs && /tmp/mvs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: v_fma_f32   1: v_max_i32 (integer VALU)   2: v_pk_fma_f32   3: ds_read_b128 (LDS only, no VALU)
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int mf, int va, float a, float b, int prio) {
  __shared__ float4 lds[1024];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lds[threadIdx.x] = make_float4(a, b, a, b);
  lds[threadIdx.x + 512] = make_float4(b, a, b, a);
  __syncthreads();
  float s = 0.f;
  if (wave < 4) {
    if (prio == 1) __builtin_amdgcn_s_setprio(3);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < mf; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    if (prio == 2) __builtin_amdgcn_s_setprio(3);
    float v[16];
    int iv[16];
    for (int i = 0; i < 16; ++i) { v[i] = a * i + threadIdx.x; iv[i] = (int)threadIdx.x * (i + 1); }
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < va; ++it) {
      if constexpr (KIND == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], a, b);
      } else if constexpr (KIND == 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_max_i32 %0, %1, %0" : "+v"(iv[i]) : "v"(it));
      } else if constexpr (KIND == 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 16; i += 2)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&v[i]) : "v"(*(double*)&v[(i + 2) & 15]), "v"(*(double*)&v[(i + 4) & 15]));
      } else {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          float4 t = lds[(threadIdx.x + u * 16 + it) & 1023];
          q.x += t.x;
        }
      }
    }
    for (int i = 0; i < 16; ++i) s += v[i] + iv[i];
    s += q.x;
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
float timeit(K kern, float* out, int mf, int va, int prio = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<256, 512>>>(out, mf > 0 ? 10 : 0, va > 0 ? 10 : 0, 1.f, 2.f, prio); hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<256, 512>>>(out, mf, va, 1.0001f, 0.5f, prio);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int KIND>
void run(const char* name, int va) {
  float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
  const int mf = 4000;                       // 4000 x 32 MFMAs x 64 cycles = 8.2 M cycles per wave
  const float tm = timeit(k<KIND>, out, mf, 0), tv = timeit(k<KIND>, out, 0, va), tb = timeit(k<KIND>, out, mf, va);
  const float tb1 = timeit(k<KIND>, out, mf, va, 1), tb2 = timeit(k<KIND>, out, mf, va, 2);
  printf("%-12s both with MFMA waves at prio 3: %.3f ms | with the OTHER waves at prio 3: %.3f ms\n", name, tb1, tb2);
  printf("%-12s mfma alone %.3f ms | other alone %.3f ms | both %.3f ms | max %.3f sum %.3f  -> overlap %.0f %%\n", name, tm,
         tv, tb, tm > tv ? tm : tv, tm + tv, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
  hipFree(out);
}

int main() {
  run<0>("v_fma_f32", 16000);      // 16000 x 128 fma
  run<0>("v_fma_f32", 4000);
  run<1>("v_max_i32", 16000);
  run<2>("v_pk_fma_f32", 16000);
  run<3>("ds_read_b128", 8000);
  return 0;
}
