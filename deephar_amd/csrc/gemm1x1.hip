// Pointwise (1x1, stride 1, no padding) convolution = GEMM [M = N*H*W, K = Cin] x [K, Cout] for gfx950.
// 80 % of ReceptionNet's MACs: the pointwise half of every SeparableConv2D plus the
// 1x1 projections (reference deephar/layers.py:74-80, 258-301; models/reception.py:43-59,145-164).
//
// Same MFMA core and fragment mapping as conv_igemm.hip (v_mfma_f32_32x32x2_f32, exact fp32), but the
// staging is LDS-DMA: global_load_lds_dwordx4 writes both operand tiles straight into LDS, double buffered,
// so a K-step costs no staging VGPRs, no ds_write and ONE barrier; the DMA of tile k+1 is in flight
// during the whole MFMA block of tile k.
//   A (activations, rows = pixels): LDS image [BM][32] floats, 128-byte rows, 16-byte slots XOR-swizzled
//      with (row & 7).  The DMA destination is lane-linear (wave base + lane*16), so the swizzle lives on the
//      per-lane SOURCE address and on the fragment read (both-sides-or-neither).
//   B (weights): host-packed [K/4][N][4] -> the LDS image is the global image, copied linearly.
//   ReLU-on-load (act_conv_bn) is applied to the A fragments after the ds_read.
//   Rows >= M and k >= K are clamped to valid addresses: padded weight rows are zero, tail rows never stored.
#include "conv_common.h"

namespace dh {
namespace {

constexpr int BK = 32;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int WM, int WN, int TM, int TN, bool UP2>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm1x1_kernel(const ConvArgs p, const int epi_vec) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int APASS = BM * 8 / NT;
  constexpr int BPASS = 8 * BN / NT;
  constexpr int STAGE = BM * BK + BK * BN;  // floats per stage
  static_assert(BM * 8 % NT == 0 && (8 * BN) % NT == 0, "tile/thread mismatch");

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int M = p.N * p.OH * p.OW;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  // ---- per-thread DMA sources (fixed over K except for the k offset)
  const float* a_src[APASS];
  int a_slot[APASS];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int r = (tid >> 3) + ps * (NT / 8);       // row inside the tile; lane-linear LDS slot' = tid&7
    int m = m0 + r;
    m = m < M ? m : M - 1;
    a_src[ps] = p.x + (size_t)m * p.ldx;
    a_slot[ps] = ((tid & 7) ^ (r & 7)) * 4;          // swizzle on the source side
  }
  const float* b_src[BPASS];
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int idx = tid + q * NT;
    const int kq = idx / BN, j = idx - kq * BN;
    const int col = n0 + j < p.Np ? n0 + j : 0;
    b_src[q] = p.w + ((size_t)kq * p.Np + col) * 4;
  }
  const size_t b_step = (size_t)8 * p.Np * 4;       // floats per K-step in the packed weight

  auto issue = [&](int kt, int stage) {
    float* sA = smem + stage * STAGE;
    float* sB = sA + BM * BK;
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      int k = kt * BK + a_slot[ps];
      k = k < p.K ? k : 0;
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[ps] + k), (lptr_t)(sA + (ps * NT + wave * 64) * 4), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[q] + kt * b_step), (lptr_t)(sB + (q * NT + wave * 64) * 4),
                                       16, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.Kp / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // fragment read offsets (floats) inside a stage, per A tile row block
  int a_off[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = ((wm * TM + i) * 32 + li) * BK;
  const int sw = li & 7;   // (row & 7) of this lane's rows: tile row blocks start at multiples of 32

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    const float* sA = smem + cur * STAGE;
    const float* sB = sA + BM * BK;
    // fragments of sub-step s+1 are fetched from LDS while the MFMAs of sub-step s run (register double buffer)
    float4 fa[2][TM], fb[2][TN];
    auto fetch = [&](int s, float4 (&a)[TM], float4 (&b)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        a[i] = *reinterpret_cast<const float4*>(&sA[a_off[i] + (((s * 2 + lh) ^ sw) << 2)]);
        if (p.pre_relu) {
          a[i].x = fmaxf(a[i].x, 0.f); a[i].y = fmaxf(a[i].y, 0.f);
          a[i].z = fmaxf(a[i].z, 0.f); a[i].w = fmaxf(a[i].w, 0.f);
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const float4*>(&sB[((s * 2 + lh) * BN + (wn * TN + j) * 32 + li) * 4]);
    };
    fetch(0, fa[0], fb[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) fetch(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][i].x, fb[s & 1][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][i].y, fb[s & 1][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][i].z, fb[s & 1][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][i].w, fb[s & 1][j].w, acc[i][j], 0, 0, 0);
        }
    }
    // tile kt+1 has landed (this wave's DMA) and every wave is done reading tile kt
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  conv_epilogue<WM, WN, TM, TN, UP2>(p, acc, smem, m0, n0, M, epi_vec);
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(const ConvArgs& a, int epi, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = ((M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  if (tiles <= 0 || tiles > 0x7fffffffLL) return DH_EINVAL;
  constexpr int kStage = 2 * (BM * BK + BK * BN), kEpi = WM * WN * 32 * (TN * 32 + 4);
  constexpr size_t lds = (size_t)(kStage > kEpi ? kStage : kEpi) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (a.up2) {
    if constexpr (TM * TN >= 6) {
      return DH_EUNSUPPORTED;
    } else {
      auto kern = gemm1x1_kernel<WM, WN, TM, TN, true>;
      if (lds > 64 * 1024) {
        static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)lds), true);
        (void)once;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), lds, s, a, epi);
    }
  } else {
    auto kern = gemm1x1_kernel<WM, WN, TM, TN, false>;
    if (lds > 64 * 1024) {
      static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)lds), true);
      (void)once;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), lds, s, a, epi);
  }
  return check_launch();
}

}  // namespace

bool gemm1x1_eligible(const ConvArgs& a) {
  return a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0 && a.pre_scale == nullptr &&
         a.H == a.OH && a.W == a.OW && a.Cin % 4 == 0 && a.ldx % 4 == 0 &&
         (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;
}

int launch_gemm1x1(const ConvArgs& a, int cfg, int epi, hipStream_t s) {
  if (!gemm1x1_eligible(a)) return DH_EUNSUPPORTED;
  switch (cfg) {
    case 0: return launch_cfg<2, 2, 2, 3>(a, epi, s);
    case 1: return launch_cfg<2, 2, 2, 2>(a, epi, s);
    case 2: return launch_cfg<4, 1, 1, 3>(a, epi, s);
    case 3: return launch_cfg<4, 1, 1, 2>(a, epi, s);
    case 4: return launch_cfg<4, 1, 1, 1>(a, epi, s);
    case 5: return launch_cfg<2, 1, 1, 3>(a, epi, s);
    case 6: return launch_cfg<2, 1, 1, 2>(a, epi, s);
    case 7: return launch_cfg<2, 1, 1, 1>(a, epi, s);
    case 8: return launch_cfg<1, 1, 1, 1>(a, epi, s);
  }
  return DH_EINVAL;
}

}  // namespace dh
