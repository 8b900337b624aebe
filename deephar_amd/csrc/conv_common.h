// Shared pieces of the two MFMA convolution kernels (conv_igemm.hip: general implicit GEMM with register
// staging; gemm1x1.hip: pointwise GEMM with LDS-DMA double buffering).
#pragma once
#include "dh_kernels.h"

namespace dh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// XCD-aware bijective remap: block b runs on XCD b%8; give each XCD a contiguous run of tiles so the
// workgroups that share one activation tile (consecutive N-tiles) hit the same L2.
__device__ __forceinline__ int xcd_tile(int b, int nwg) {
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Epilogue.  C/D layout of a 32x32 MFMA tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Each wave stages one 32 x (TN*32) row block through its private LDS slab and reads it back row-wise, so
// the BN affine / residual loads / stores are 16-byte wide and whole output rows are contiguous.
//   out = relu?( acc*post_scale + post_shift + res1[m] + res2[mo] ), optionally written 2x up-sampled.
template <int WM, int WN, int TM, int TN, bool UP2>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][TN], float* smem, int m0,
                                              int n0, int M, int epi_vec) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int LDC = TN * 32 + 4;
  constexpr int ROW4 = TN * 8;                 // float4 per staged row
  float* sC = smem + wave * 32 * LDC;
  const int ohw = p.OH * p.OW;
  const bool vec = epi_vec != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sC[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + j * 32 + li] = acc[i][j][r];
    __syncthreads();
#pragma unroll 2
    for (int f = lane; f < 32 * ROW4; f += 64) {
      const int row = f / ROW4, c4 = f - row * ROW4;
      const int m = m0 + (wm * TM + i) * 32 + row;
      const int n = n0 + wn * TN * 32 + c4 * 4;
      if (m >= M || n >= p.Cout) continue;
      float4 v = *reinterpret_cast<const float4*>(&sC[row * LDC + c4 * 4]);
      size_t mo[4];
      int nout = 1;
      if constexpr (UP2) {
        const int fr = m / ohw;
        const int rem = m - fr * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
        for (int d = 0; d < 4; ++d)
          mo[d] = ((size_t)fr * 2 * p.OH + 2 * oh + (d >> 1)) * (2 * p.OW) + 2 * ow + (d & 1);
        nout = 4;
      } else {
        mo[0] = (size_t)m;
      }
      if (vec) {
        if (p.post_scale != nullptr) {
          const float4 sc = *reinterpret_cast<const float4*>(p.post_scale + n);
          const float4 sh = *reinterpret_cast<const float4*>(p.post_shift + n);
          v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        }
        if (p.res1 != nullptr) {
          const float4 r = *reinterpret_cast<const float4*>(p.res1 + (size_t)m * p.ldr1 + n);
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
#pragma unroll
        for (int d = 0; d < (UP2 ? 4 : 1); ++d) {
          float4 o = v;
          if (p.res2 != nullptr) {
            const float4 r = *reinterpret_cast<const float4*>(p.res2 + mo[d] * p.ldr2 + n);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
          }
          if (p.post_relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
          }
          *reinterpret_cast<float4*>(p.y + mo[d] * p.ldy + n) = o;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= p.Cout) break;
          float t = (&v.x)[e];
          if (p.post_scale != nullptr) t = t * p.post_scale[n + e] + p.post_shift[n + e];
          if (p.res1 != nullptr) t += p.res1[(size_t)m * p.ldr1 + n + e];
          for (int d = 0; d < nout; ++d) {
            float o = t;
            if (p.res2 != nullptr) o += p.res2[mo[d] * p.ldr2 + n + e];
            if (p.post_relu) o = fmaxf(o, 0.f);
            p.y[mo[d] * p.ldy + n + e] = o;
          }
        }
      }
    }
  }
}

}  // namespace dh
