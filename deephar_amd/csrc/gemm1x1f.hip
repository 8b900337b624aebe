// Pointwise convolution on FINE tiles for gfx950: the tiling of the fp32 GEMM family for launches with fewer 32 x 32 tiles
// than the chip has SIMDs -- the pose stream of SPNet at a couple of clips per call (exp/pennaction/eval_speed2d.py; the
// pointwise halves of common.residual_unit / sepconv2d on the 16 x 16 ... 4 x 4 levels, spnet.py:251-314), where a
// 1 024 x 480 x 480 GEMM ran 15.7 us on the LDS-DMA kernel for 3 us of matrix work.
//
// Why it is slow there (DESIGN.md 3.1 [r05]): v_mfma_f32_32x32x2_f32 issues once per 64 cycles, so one wave walking
// K = 480 for its 32 x 32 tile is a serial chain of 240 MFMAs = 6.4 us whatever the memory system does, and with fewer
// tiles than SIMDs nothing runs beside it.  v_mfma_f32_16x16x4_f32 has the same MAC rate but a quarter of the tile:
// four times the waves, each with a chain of K / 4 MFMAs at 40 cycles = 2.0 us for K = 480.
//
// Bit-identical to the rest of the family, so the autotuner may pick it by TIME (batch size included) without moving a
// bit.  The 32 x 32 x 2 kernels accumulate an output element over k in the order 0, 4, 1, 5, 2, 6, 3, 7 of every group of
// eight (MFMA e of sub-step S multiplies k = 8 S + e in lane half 0 and k = 8 S + 4 + e in lane half 1).  A 16 x 16 x 4 MFMA
// takes its four k from the four lane groups lg = lane / 16, in that order; here lane groups 0, 2 hold the float4 of
// k = 8 p .. 8 p + 3 and lane groups 1, 3 that of k = 8 p + 4 .. 8 p + 7, and the two MFMAs of a pair p feed
//   MFMA 1: (x, x, y, y) of groups (0, 1, 2, 3) = k (8p, 8p + 4, 8p + 1, 8p + 5)
//   MFMA 2: (z, z, w, w)                         = k (8p + 2, 8p + 6, 8p + 3, 8p + 7)
// -- the same products added in the same order, one accumulator chain.  (One v_cndmask per operand and MFMA: paid in
// full beside an fp32 MFMA, which is why this tiling loses wherever the chip is full, and why it does not matter here.)
//
// Work-group = four waves = the four 16 x 16 quadrants of a 32 x 32 region (the two waves of a row pair read the same A
// rows: L1 hits); no LDS staging and no barrier in the K loop: a lane loads its operands straight from global memory
// (A: 16 bytes of its row, B: the packed weights' 16-byte unit of its column), a chunk of eight pairs ahead of the one
// being multiplied.  ReLU / BatchNormalization prologue and BN / residual / ReLU epilogue as in gemm1x1.hip (same
// arithmetic: fmaf for the prologue, t * scale + shift for the epilogue); the half-resolution residual, the fused
// up-sampling and the pooled second output are not built here (DH_EUNSUPPORTED: the autotuner skips the tiling).
#include "conv_common.h"

namespace dh {
namespace {

constexpr int FG_DEPTH = 8;          // pairs (8 k each) per chunk; two chunks of registers

template <bool RELU, bool PRE>
__global__ __launch_bounds__(256) void gemm1x1_fine_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float fg_tab[];      // PRE: [2][Kp] scale, shift (zero beyond K)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int hi = lg & 1;               // which k-group of the pair this lane holds
  const bool sel = (lg >> 1) != 0;     // which elements of it feed the MFMAs: (x, z) or (y, w)
  const int M = p.N * p.OH * p.OW;
  const int tiles_n = (p.Cout + 31) / 32;
  const int m0 = (blockIdx.x / tiles_n) * 32 + (wave >> 1) * 16;
  const int n0 = (blockIdx.x % tiles_n) * 32 + (wave & 1) * 16;

  int m = m0 + li;
  m = m < M ? m : M - 1;
  const float* xrow = p.x + (size_t)m * p.ldx + 4 * hi;
  const int col = n0 + li < p.Np ? n0 + li : p.Np - 1;
  const float* wcol = p.w + ((size_t)hi * p.Np + col) * 4;      // packed [Kp / 4][Np][4]: k-group 2 p + hi, column col
  const size_t wstep = (size_t)2 * p.Np * 4;                    // floats per pair
  const int pairs = p.Kp / 8;

  if constexpr (PRE) {
    for (int i = tid; i < p.Kp; i += 256) {
      fg_tab[i] = i < p.K ? p.pre_scale[i] : 0.f;
      fg_tab[p.Kp + i] = i < p.K ? p.pre_shift[i] : 0.f;
    }
  }

  // this lane's four outputs: rows m0 + 4 lg + r, column n0 + li; their BN / residual values fly during the K loop
  const int ncol = n0 + li;
  const bool cok = ncol < p.Cout;
  float psc = 1.f, psh = 0.f, r1v[4] = {0.f, 0.f, 0.f, 0.f}, r2v[4] = {0.f, 0.f, 0.f, 0.f};
  if (cok && p.post_scale != nullptr) { psc = p.post_scale[ncol]; psh = p.post_shift[ncol]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mm = m0 + 4 * lg + r;
    if (cok && mm < M) {
      if (p.res1 != nullptr) r1v[r] = p.res1[(size_t)mm * p.ldr1 + ncol];
      if (p.res2 != nullptr) r2v[r] = p.res2[(size_t)mm * p.ldr2 + ncol];
    }
  }

  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  float4 fa0[FG_DEPTH], fb0[FG_DEPTH], fa1[FG_DEPTH], fb1[FG_DEPTH];

  auto fetch = [&](int p0, float4 (&fa)[FG_DEPTH], float4 (&fb)[FG_DEPTH]) {
#pragma unroll
    for (int d = 0; d < FG_DEPTH; ++d) {
      const int pp = p0 + d < pairs ? p0 + d : pairs - 1;            // (past the end: a valid address, never multiplied)
      // k >= K only occurs inside the last pair(s); there the weights are zero and the tail of this row / the next
      // row's head is finite data (dh_conv2d_f32: inputs must be finite) -- the same products the family forms
      const int k0 = 8 * pp + 4 * hi;
      fa[d] = k0 < p.K ? *reinterpret_cast<const float4*>(xrow + 8 * pp) : make_float4(0.f, 0.f, 0.f, 0.f);
      fb[d] = *reinterpret_cast<const float4*>(wcol + (size_t)pp * wstep);
    }
  };
  auto multiply = [&](int p0, const float4 (&fa)[FG_DEPTH], const float4 (&fb)[FG_DEPTH]) {
#pragma unroll
    for (int d = 0; d < FG_DEPTH; ++d) {
      if (p0 + d < pairs) {                                            // (wave-uniform)
        float4 a = fa[d];
        if constexpr (PRE) {
          const float4 sc = *reinterpret_cast<const float4*>(fg_tab + 8 * (p0 + d) + 4 * hi);
          const float4 sh = *reinterpret_cast<const float4*>(fg_tab + p.Kp + 8 * (p0 + d) + 4 * hi);
          a = make_float4(fmaf(a.x, sc.x, sh.x), fmaf(a.y, sc.y, sh.y), fmaf(a.z, sc.z, sh.z), fmaf(a.w, sc.w, sh.w));
        }
        if constexpr (RELU) {
          a = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
        }
        const float4 b = fb[d];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel ? a.y : a.x, sel ? b.y : b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel ? a.w : a.z, sel ? b.w : b.z, acc, 0, 0, 0);
      }
    }
  };

  fetch(0, fa0, fb0);
  if constexpr (PRE) __syncthreads();                                  // the table is staged
  for (int p0 = 0; p0 < pairs; p0 += 2 * FG_DEPTH) {
    if (p0 + FG_DEPTH < pairs) fetch(p0 + FG_DEPTH, fa1, fb1);
    multiply(p0, fa0, fb0);
    if (p0 + 2 * FG_DEPTH < pairs) fetch(p0 + 2 * FG_DEPTH, fa0, fb0);
    multiply(p0 + FG_DEPTH, fa1, fb1);
  }

  // C layout of a 16 x 16 tile: register r of lane (li, lg) = row 4 lg + r, column li
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mm = m0 + 4 * lg + r;
    if (!cok || mm >= M) continue;
    float t = acc[r];
    if (p.post_scale != nullptr) t = t * psc + psh;
    if (p.res1 != nullptr) t += r1v[r];
    if (p.res2 != nullptr) t += r2v[r];
    if (p.post_relu) t = fmaxf(t, 0.f);
    p.y[(size_t)mm * p.ldy + ncol] = t;
  }
}

template <bool RELU, bool PRE>
int launch_fine(const ConvArgs& a, unsigned tiles, hipStream_t s) {
  const size_t lds = PRE ? (size_t)2 * a.Kp * sizeof(float) : 0;
  hipLaunchKernelGGL((gemm1x1_fine_kernel<RELU, PRE>), dim3(tiles), dim3(256), lds, s, a);
  return check_launch();
}

}  // namespace

bool gemm1x1_eligible(const ConvArgs& a);

// Tiling 9 of the LDS-DMA family's index space (tile_cfg 18 of dh_conv2d_f32): plain pointwise convolutions only.
int launch_gemm1x1_fine(const ConvArgs& a, hipStream_t s) {
  if (!gemm1x1_eligible(a)) return DH_EUNSUPPORTED;
  const bool pointwise = a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0;
  if (!pointwise || a.up2 || a.res2_down || a.y_pool != nullptr || a.w_split || a.x_u8 || a.Kp % 8 != 0 ||
      (a.pre_scale != nullptr && a.Kp > 4096))
    return DH_EUNSUPPORTED;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = ((M + 31) / 32) * ((a.Cout + 31) / 32);
  if (tiles <= 0 || tiles > 0x7fffffffLL) return DH_EINVAL;
  const unsigned t = (unsigned)tiles;
  if (a.pre_scale != nullptr) return a.pre_relu ? launch_fine<true, true>(a, t, s) : launch_fine<false, true>(a, t, s);
  return a.pre_relu ? launch_fine<true, false>(a, t, s) : launch_fine<false, false>(a, t, s);
}

}  // namespace dh
