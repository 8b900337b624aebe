// Micro-benchmark, second part: the split-bf16 GEMM's K loop WITH its LDS-DMA stream (buffer_load ... lds).
// Wave stream per K-step (32 k-values): H halves-of-chunks x 2 chunks, each [reads -> 6 MFMA -> wait -> 12 MFMA (+VALU)],
// one s_barrier per K-step followed by the wave's P DMA pieces (1 KiB each), either back to back (SPREAD = 0) or one
// piece every few MFMAs (SPREAD = 1).  TN = 3: H = 1 (36 MFMA per K-step, the 32 x 96 wave tile); TN = 6: H = 2 (72 MFMA,
// 32 x 192).  SRC: 0 = every work-group re-reads a 256 KiB window (L2 hits), 1 = streams its own region (HBM).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/split_loop_model2.hip -o /tmp/slm2 && /tmp/slm2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename R>
__device__ __forceinline__ void dma16(R rs, float* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, voff, soff, 0, 0);
#endif
}

template <int H, int NRH, int NVC, int P, bool SPREAD, int SRC, int NT, bool BPC = false>
__global__ __launch_bounds__(NT) void k(float* out, const float* src, int iters, unsigned region) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4* lds = (float4*)smem;
  for (int i = threadIdx.x; i < 2048; i += NT) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  constexpr int NA = 3 * H;
  f32x16 acc[NA];
  for (int i = 0; i < NA; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[3], b[3];
  for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(float)(threadIdx.x + i); b[i][e] = (__bf16)(float)(e + i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
  const unsigned la = (unsigned)((threadIdx.x & 63) * 16);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float4 q[NRH > 0 ? NRH : 1];
  float qs = 0.f;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)0x7ffffff0, 0x00020000);
  const unsigned base = SRC == 0 ? (blockIdx.x & 7) * 262144u : blockIdx.x * region;   // region: 4 MiB, walked cyclically
  const unsigned voff = base + (threadIdx.x & 63) * 16 + wave * 1024 * P;
  float* dst = smem + 8192 + wave * 256 * P;                       // 1 KiB per piece, behind the read area
  int piece = 0;
  for (int it = 0; it < iters; ++it) {
    const int soff = SRC == 0 ? (it & 3) * (NT / 64) * 1024 * P : (it % 56) * (NT / 64) * 1024 * P;   // (BPC: both chunks re-read the same window)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int hh = c * H + h;                                   // 0 .. 2H-1
        if (hh == 0 || (BPC && h == 0)) {                          // BPC: 16-k K-steps, one barrier + P pieces per chunk
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (!SPREAD) {
#pragma unroll
            for (int pc = 0; pc < P; ++pc) dma16(rs, dst + pc * 256, voff + pc * 1024, soff);
          }
        }
#pragma unroll
        for (int r = 0; r < NRH; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[r]) : "v"(la), "n"(r * 1024));
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          acc[h * 3 + m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 3], b[(m / 3) % 3], acc[h * 3 + m % 3], 0, 0, 0);
          if (SPREAD && (m % 3) == 2) {                              // one piece every third MFMA until the wave's share is out
            constexpr int SLOTS = 2 * H * 6;                         // 6 slots per half
            const int slot = (BPC ? h : hh) * 6 + (m / 3);
            if (slot < P) dma16(rs, dst + slot * 256, voff + slot * 1024, soff);
            (void)SLOTS;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NRH > 0) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          qs += q[0].x + q[NRH - 1].y;
        }
        __builtin_amdgcn_sched_barrier(0);
        const int nv = (h == H - 1) ? NVC : 0;                      // the A split sits in the chunk's last half
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          acc[h * 3 + m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m + 1) % 3], b[(m / 3) % 3], acc[h * 3 + m % 3], 0, 0, 0);
          if (SPREAD && (m % 3) == 2) {
            const int slot = (BPC ? h : hh) * 6 + 2 + (m / 3);
            if (slot < P) dma16(rs, dst + slot * 256, voff + slot * 1024, soff);
          }
#pragma unroll
          for (int f = 0; f < (nv + 11 - m) / 12; ++f)
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m + f) & 7]) : "v"(v[(m + f + 3) & 7]), "v"(qs));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  (void)piece;
  float s = qs + smem[8192 + threadIdx.x];
  for (int i = 0; i < NA; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * NT + threadIdx.x] = s;
}

static float* g_src = nullptr;
template <int H, int NRH, int NVC, int P, bool SPREAD, int SRC, int NT, bool BPC = false>
void run(int wg_per_cu, const char* what) {
  float* out;
  const int grid = 256 * wg_per_cu;
  hipMalloc(&out, grid * NT * sizeof(float));
  const int iters = 1500;
  const unsigned region = 4u << 20;                                            // bytes a work-group walks when streaming
  static_assert(56 * (NT / 64) * 1024 * P + 64 * 1024 <= (4 << 20), "region");
  const size_t lds = wg_per_cu == 1 ? 120 * 1024 : 72 * 1024;
  auto kern = k<H, NRH, NVC, P, SPREAD, SRC, NT, BPC>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, NT, lds>>>(out, g_src, 600, region); hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<grid, NT, lds>>>(out, g_src, iters, region);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = wg_per_cu * (NT / 256.0);
  const double mf = 36.0 * H * iters * waves_per_simd;
  const double gbs = (double)grid * (NT / 64) * P * (BPC ? 2 : 1) * 1024.0 * iters / (ms * 1e-3) / 1e9;
  printf("%-34s waves/SIMD %.0f MFMA/K-step %d reads %2d valu %3d pieces %2d %s %s : %.3f ms  pipe busy %.1f %%  DMA %.1f TB/s\n", what, waves_per_simd,
         36 * H, NRH * 2 * H, NVC * 2, P, SPREAD ? "spread" : "burst ", SRC ? "hbm" : "l2 ", ms, 100.0 * mf * 32 / (ms * 1e-3 * 2.4e9), gbs / 1e3);
  hipFree(out);
}

int main() {
  hipMalloc(&g_src, (size_t)3 << 30);
  hipMemset(g_src, 0, (size_t)3 << 30);
  // today's kernel: 128 x 96 tile, 4 waves, two work-groups per CU, 36 MFMA / 22 reads / ~150 VALU / 9 pieces per wave and K-step
  run<1, 11, 76, 0, false, 0, 256>(2, "32x96 no dma");
  run<1, 11, 76, 9, false, 0, 256>(2, "32x96 dma burst l2");
  run<1, 11, 76, 9, true, 0, 256>(2, "32x96 dma spread l2");
  run<1, 11, 76, 9, false, 1, 256>(2, "32x96 dma burst hbm");
  run<1, 11, 76, 9, true, 1, 256>(2, "32x96 dma spread hbm");
  run<1, 11, 52, 9, true, 0, 256>(2, "32x96 spread, 104 valu");
  // 256 x 192 tile, 8 waves of 32 x 192 (one work-group per CU): 72 MFMA / 40 reads / ~150 VALU / 9 pieces
  run<2, 10, 76, 0, false, 0, 512>(1, "32x192 no dma");
  run<2, 10, 76, 9, false, 0, 512>(1, "32x192 dma burst l2");
  run<2, 10, 76, 9, true, 0, 512>(1, "32x192 dma spread l2");
  run<2, 10, 76, 9, true, 1, 512>(1, "32x192 dma spread hbm");
  run<2, 10, 60, 9, true, 0, 512>(1, "32x192 spread, 120 valu");
  // 128 x 192 tile, 4 waves of 32 x 192, one work-group per CU (one wave per SIMD): 72 MFMA / 40 reads / 13 pieces
  run<2, 10, 76, 13, true, 0, 256>(1, "32x192 4 waves spread");
  // 128 x 192 tile, 4 waves of 32 x 192, 16-k K-steps (52 KB of LDS: two work-groups per CU): 7 pieces per chunk
  run<2, 10, 76, 7, false, 0, 256, true>(2, "32x192 4w x2 bk16 burst");
  run<2, 10, 76, 7, true, 0, 256, true>(2, "32x192 4w x2 bk16 spread");
  run<2, 10, 76, 0, true, 0, 256, true>(2, "32x192 4w x2 bk16 no dma");
  run<2, 10, 76, 7, true, 1, 256, true>(2, "32x192 4w x2 bk16 spread hbm");
  // [r06] VERDICT r05 item 6 priced: the activations arrive PRE-SPLIT (three bf16 planes written by the producer's epilogue:
  // 6 bytes per element instead of 4) -- no split VALU in the K loop, one more LDS-DMA piece per wave and 16-k K-step
  // (A stage 3 KiB instead of 2 KiB per wave), one more ds_read_b128 per chunk (11 instead of 10 per half in this model)
  run<2, 11, 0, 8, true, 0, 256, true>(2, "pre-split A: bk16 spread l2");
  run<2, 11, 0, 8, true, 1, 256, true>(2, "pre-split A: bk16 spread hbm");
  run<2, 10, 0, 7, true, 0, 256, true>(2, "(no split VALU, nothing else) l2");
  run<2, 10, 0, 7, true, 1, 256, true>(2, "(no split VALU, nothing else) hbm");
  return 0;
}
