// Micro-benchmark: how many other instructions of the SAME wave hide under one v_mfma_f32_32x32x2_f32 (64 cycles in
// the matrix pipe) -- and under one v_mfma_f32_32x32x16_bf16 (32 cycles) -- on gfx950?  One wave per SIMD (256-thread work-groups, one per CU), 4 independent accumulators,
// NF filler instructions issued after every MFMA.  Reported: cycles per MFMA (wall / #MFMA at the measured clock
// assumption of 2.4 GHz) -- 64 = fillers are free, 64 + NF * c = they are not.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_inwave_fill.hip -o /tmp/mif && /tmp/mif
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_fma_f32  1: v_max_i32  2: ds_read_b128 (results consumed at the end of the iteration)
template <int KIND, int NF, bool BF = false>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  __shared__ float4 lds[1024];
  lds[threadIdx.x] = make_float4(a, b, a, b);
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x = a + threadIdx.x, y = b;
  float v[16];
  int iv[16];
  for (int i = 0; i < 16; ++i) { v[i] = a * i + threadIdx.x; iv[i] = (int)threadIdx.x * (i + 1); }
  const unsigned la = (unsigned)(threadIdx.x * 16);
  float4 q[NF > 0 ? NF : 1];
  float qs = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (BF) {
        bf16x8 xa, yb;
        for (int e = 0; e < 8; ++e) { xa[e] = (__bf16)x; yb[e] = (__bf16)y; }
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, yb, acc[u & 3], 0, 0, 0);
      } else {
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[u & 3], 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(u * NF + f) & 15]) : "v"(a), "v"(b));
        else if constexpr (KIND == 1) asm volatile("v_max_i32 %0, %1, %0" : "+v"(iv[(u * NF + f) & 15]) : "v"(it));
        else asm volatile("ds_read_b128 %0, %1" : "=v"(q[f]) : "v"(la));
      }
      if constexpr (KIND == 2 && NF > 0) {
        if (u == 7) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          qs += q[0].x;
        }
      }
    }
  }
  float s = qs;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i] + iv[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NF, bool BF = false>
void run(const char* name) {
  float* out; hipMalloc(&out, 256 * 256 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, NF, BF><<<256, 256>>>(out, 10, 1.f, 2.f); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND, NF, BF><<<256, 256>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-5s %-12s fillers/MFMA %2d : %.3f ms  -> %.1f cycles per MFMA @2.4 GHz\n", BF ? "bf16" : "f32", name, NF, ms, ms * 1e-3 * 2.4e9 / (iters * 8.0));
  hipFree(out);
}

int main() {
  run<0, 0>("none");
  run<0, 2>("v_fma_f32"); run<0, 4>("v_fma_f32"); run<0, 8>("v_fma_f32"); run<0, 12>("v_fma_f32"); run<0, 16>("v_fma_f32");
  run<1, 4>("v_max_i32"); run<1, 8>("v_max_i32"); run<1, 16>("v_max_i32");
  run<2, 1>("ds_read_b128"); run<2, 2>("ds_read_b128"); run<2, 4>("ds_read_b128");
  // the same under v_mfma_f32_32x32x16_bf16 (32 cycles in the matrix pipe)
  run<0, 0, true>("none");
  run<0, 2, true>("v_fma_f32"); run<0, 4, true>("v_fma_f32"); run<0, 6, true>("v_fma_f32"); run<0, 8, true>("v_fma_f32"); run<0, 12, true>("v_fma_f32");
  run<1, 4, true>("v_max_i32"); run<1, 8, true>("v_max_i32");
  run<2, 1, true>("ds_read_b128"); run<2, 2, true>("ds_read_b128"); run<2, 4, true>("ds_read_b128");
  return 0;
}
