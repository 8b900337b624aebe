// Micro-benchmark: does the bf16 MFMA rate depend on the operand DATA (switching power -> clock)?
// 2 waves per SIMD, nothing but v_mfma_f32_32x32x16_bf16 on three accumulators; operands are 12 register sets loaded
// once from a buffer that holds (0) one repeated small value, (1) random bf16 values of magnitude ~1, (2) random bits.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_data_power.hip -o /tmp/mdp && /tmp/mdp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float* out, const f32x4* src, int iters) {
  f32x16 acc[3];
  for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[6], b[6];
  for (int i = 0; i < 6; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(i * 2 + 0) * 256 + threadIdx.x]);
    b[i] = __builtin_bit_cast(bf16x8, src[(i * 2 + 1) * 256 + threadIdx.x]);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 36; ++m)
      acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 6], b[(m / 2) % 6], acc[m % 3], 0, 0, 0);
    if ((it & 63) == 63)
      for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;       // keep the sums finite
  }
  float s = 0.f;
  for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void kf(float* out, const float* src, int iters) {
  f32x16 acc[3];
  for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[12], b[12];
  for (int i = 0; i < 12; ++i) { a[i] = src[(i * 2) * 256 + threadIdx.x]; b[i] = src[(i * 2 + 1) * 256 + threadIdx.x]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 36; ++m)
      acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m % 12], b[(m / 2) % 12], acc[m % 3], 0, 0, 0);
    if ((it & 63) == 63)
      for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
  }
  float s = 0.f;
  for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out; f32x4* src;
  hipMalloc(&out, 512 * 256 * sizeof(float));
  hipMalloc(&src, 12 * 256 * 16);
  const char* names[3] = {"constant", "random values ~1", "random bits"};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<unsigned short> h(12 * 256 * 8);
    srand(1);
    for (auto& v : h) {
      if (mode == 0) v = 0x3f80;                                            // 1.0
      else if (mode == 1) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));   // +-[0.5, 1)
      else v = (unsigned short)(rand() & 0xffff) & 0xbfff;                   // any bits, exponent kept below inf/nan
    }
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<512, 256>>>(out, src, 2000); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = 20000;
      hipEventRecord(e0);
      k<<<512, 256>>>(out, src, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-18s run %d: %.2f ms  -> %.0f MHz effective matrix-pipe clock (%.2f PFLOP/s)\n", names[mode], rep, ms,
             36.0 * iters * 2 * 32 / (ms * 1e-3) / 1e6, 512.0 * 4 * 36 * iters * 32768 / (ms * 1e-3) / 1e15);
    }
  }
  // the same question for v_mfma_f32_32x32x2_f32 (64 cycles per instruction)
  float* fsrc; hipMalloc(&fsrc, 24 * 256 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<float> h(24 * 256);
    srand(2);
    for (auto& v : h) v = mode == 0 ? 1.0f : (float)((rand() & 0xffff) - 32768) / 32768.0f + (float)(rand() & 0xff) * 1e-6f;
    hipMemcpy(fsrc, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kf<<<512, 256>>>(out, fsrc, 1000); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = 10000;
      hipEventRecord(e0);
      kf<<<512, 256>>>(out, fsrc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("f32 %-14s run %d: %.2f ms  -> %.0f MHz effective matrix-pipe clock (%.1f TFLOP/s)\n", mode ? "random values" : "constant", rep, ms,
             36.0 * iters * 2 * 64 / (ms * 1e-3) / 1e6, 512.0 * 4 * 36 * iters * 4096 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
