// Split-bf16 ("bf16x3") variant of the LDS-DMA GEMM of gemm1x1.hip: the same pointwise / K x K implicit GEMM, the same
// fp32 inputs, outputs, epilogue and K order, but every fp32 operand is split EXACTLY into three bf16 parts
//   x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (round to nearest even)
// and a product is evaluated on the bf16 matrix cores as the six largest of the nine partial products
//   a b ~ a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1          (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|)
// accumulated in fp32 inside v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs cover 16 k-values in 6 x 32 = 192 cycles per
// SIMD; the fp32 MFMA (v_mfma_f32_32x32x2_f32, vector rate) needs 8 x 64 = 512 -- and, unlike the fp32 MFMA, the bf16
// MFMA leaves the SIMD's other issue ports free, so the splitting VALU work and the LDS reads run underneath it
// (profiles/r02_sepconv_fusion_study.md section 3 for the fp32 side of that statement).
// The per-product error (2^-23 relative, typically 2^-25) is below what the fp32 accumulation itself contributes
// (sqrt(K) roundings of the running sum), so results stay within the same 1e-3 px of the fp64 oracle
// (profiles/parity_r02_bf16x3.json); they are NOT bit-identical to the fp32-MFMA path, which remains the default.
// Weights are split once on the host (dh_conv2d_pack_weights_split_host / engine/packing.py):
//   [K/8][3 parts][Np][8 bf16]  -- one 16-byte unit per (k-group, part, column) = one lane's B operand.
// Reference layers replaced: as gemm1x1.hip (deephar/layers.py:74-80, 258-301; models/reception.py:43-98).
#include "conv_common.h"

namespace dh {
namespace {

constexpr int BK = 32;

__device__ __attribute__((aligned(128))) float g_zero_page_s[BK] = {};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ float4 lds_rd(unsigned addr) {
  float4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void lgkm_wait() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float relu1(float v) {
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}

// (x0, x1) -> packed bf16 pair (v_cvt_pk_bf16_f32, RNE) and the exact residuals
__device__ __forceinline__ unsigned split_pair(float& x0, float& x1) {
  const f32x2 v = {x0, x1};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const unsigned p = __builtin_bit_cast(unsigned, h);
  x0 -= __uint_as_float(p << 16);
  x1 -= __uint_as_float(p & 0xffff0000u);
  return p;
}
__device__ __forceinline__ unsigned pack_pair(float x0, float x1) {
  const f32x2 v = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

struct Frag3 { bf16x8 p[3]; };

// 8 consecutive k of one row (two float4) -> three bf16x8 operands
template <bool RELU>
__device__ __forceinline__ Frag3 split8(float4 lo4, float4 hi4) {
  float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  if constexpr (RELU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = relu1(x[i]);
  }
  unsigned a[4], b[4], c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = split_pair(x[2 * i], x[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = split_pair(x[2 * i], x[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = pack_pair(x[2 * i], x[2 * i + 1]);
  Frag3 f;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f.p[0] = __builtin_bit_cast(bf16x8, (u32x4){a[0], a[1], a[2], a[3]});
  f.p[1] = __builtin_bit_cast(bf16x8, (u32x4){b[0], b[1], b[2], b[3]});
  f.p[2] = __builtin_bit_cast(bf16x8, (u32x4){c[0], c[1], c[2], c[3]});
  return f;
}

__device__ __forceinline__ bf16x8 as_bf(float4 v) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(bf16x8, (f32x4){v.x, v.y, v.z, v.w});
}

// six partial products, smallest first
__device__ __forceinline__ void mfma6(const Frag3& a, const bf16x8 (&b)[3], f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b[0], c, 0, 0, 0);
}

template <int WM, int WN, int TM, int TN, bool UP2, bool RELU, bool KXK = false>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm1x1s_kernel(const ConvArgs p, const int epi_vec) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int APASS = BM * 8 / NT;
  constexpr int BPASS = (12 * BN + NT - 1) / NT;         // 16-byte units per thread: 4 k-groups x 3 parts x BN
  constexpr int BROWS = BPASS * NT;                      // padded unit count of the B stage
  constexpr int STAGE = BM * BK + BROWS * 4;             // floats per stage
  static_assert(BM * 8 % NT == 0, "tile/thread mismatch");

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

  // ---- per-thread DMA sources (as gemm1x1_kernel)
  const float* a_src[APASS];
  int a_slot[APASS];
  int a_pix[KXK ? APASS : 1], a_ih0[KXK ? APASS : 1], a_iw0[KXK ? APASS : 1];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int r = (tid >> 3) + ps * (NT / 8);
    int m = m0 + r;
    m = m < M ? m : M - 1;
    a_slot[ps] = ((tid & 7) ^ (r & 7)) * 4;
    if constexpr (KXK) {
      const int n = m / (p.OH * p.OW);
      const int rem = m - n * (p.OH * p.OW);
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      a_pix[ps] = n * p.H * p.W;
      a_ih0[ps] = oh * p.SH - p.PT;
      a_iw0[ps] = ow * p.SW - p.PL;
      a_src[ps] = p.x;
    } else {
      a_src[ps] = p.x + (size_t)m * p.ldx;
    }
  }
  const int chunks_per_tap = KXK ? p.Cin / BK : 1;
  // packed split weight: 16-byte unit (kg, part, n) at ((kg * 3 + part) * Np + n) * 4 floats
  const float* b_src[BPASS];
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int idx = tid + q * NT;
    int r = idx / BN;
    const int j = idx - r * BN;
    r = r < 12 ? r : 11;
    const int col = n0 + j < p.Np ? n0 + j : 0;
    b_src[q] = p.w + ((size_t)r * p.Np + col) * 4;
  }
  const size_t b_step = (size_t)12 * p.Np * 4;           // floats per K-step in the packed weight

  auto issue = [&](int kt, int stage) {
    float* sA = smem + stage * STAGE;
    float* sB = sA + BM * BK;
    int kh = 0, kw = 0, c0 = 0;
    if constexpr (KXK) {
      const int tap = kt / chunks_per_tap;
      c0 = (kt - tap * chunks_per_tap) * BK;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      const float* src;
      if constexpr (KXK) {
        const int ih = a_ih0[ps] + kh, iw = a_iw0[ps] + kw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && kt * BK < p.K;
        src = ok ? p.x + (size_t)(a_pix[ps] + ih * p.W + iw) * p.ldx + c0 + a_slot[ps] : g_zero_page_s + a_slot[ps];
      } else {
        int k = kt * BK + a_slot[ps];
        k = k < p.K ? k : 0;
        src = a_src[ps] + k;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + (ps * NT + wave * 64) * 4), 16, 0, 0);
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

  // ---- fragment read addresses (LDS byte offsets), stage 0.  A: row li of the wave's 32-row block, 8 consecutive k
  // of k-group (2c + lh) = slots 2(2c+lh), 2(2c+lh)+1 of the 128-byte row, XOR-swizzled with (row & 7) = (li & 7).
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  unsigned a_base[TM][2][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        a_base[i][c][h] = lds0 + (unsigned)(((wm * TM + i) * 32 + li) * BK * 4 +
                                            (((2 * (2 * c + lh) + h) ^ (li & 7)) << 4));
  // B: unit ((2c + lh) * 3 + part) * BN + column
  const unsigned b_base = lds0 + (unsigned)(BM * BK * 4) + (unsigned)((lh * 3 * BN + wn * TN * 32 + li) * 16);

  EpiPrefetch<TM, TN> pre;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt == nk - 1) pre.template issue<WM, WN>(p, m0, n0, M, epi_vec);
    const unsigned so = (unsigned)(cur * STAGE * 4);
    const unsigned bo = b_base + so;

    float4 ra[2][TM][2];
    float4 rb[2][TN][3];
    // chunk 0 operands
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ra[0][i][0] = lds_rd<0>(a_base[i][0][0] + so);
      ra[0][i][1] = lds_rd<0>(a_base[i][0][1] + so);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      rb[0][j][0] = lds_rd<0>(bo + (unsigned)(j * 512));
      rb[0][j][1] = lds_rd<0>(bo + (unsigned)(BN * 16 + j * 512));
      rb[0][j][2] = lds_rd<0>(bo + (unsigned)(2 * BN * 16 + j * 512));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);          // DMA of the next tile flies during this K-step
    lgkm_wait();
    // chunk 1 operands in flight while chunk 0 is split and multiplied
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ra[1][i][0] = lds_rd<0>(a_base[i][1][0] + so);
      ra[1][i][1] = lds_rd<0>(a_base[i][1][1] + so);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      rb[1][j][0] = lds_rd<0>(bo + (unsigned)(6 * BN * 16 + j * 512));
      rb[1][j][1] = lds_rd<0>(bo + (unsigned)(7 * BN * 16 + j * 512));
      rb[1][j][2] = lds_rd<0>(bo + (unsigned)(8 * BN * 16 + j * 512));
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      Frag3 fa[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = split8<RELU>(ra[0][i][0], ra[0][i][1]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const bf16x8 fb[3] = {as_bf(rb[0][j][0]), as_bf(rb[0][j][1]), as_bf(rb[0][j][2])};
#pragma unroll
        for (int i = 0; i < TM; ++i) mfma6(fa[i], fb, acc[i][j]);
      }
    }
    lgkm_wait();
    {
      Frag3 fa[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = split8<RELU>(ra[1][i][0], ra[1][i][1]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const bf16x8 fb[3] = {as_bf(rb[1][j][0]), as_bf(rb[1][j][1]), as_bf(rb[1][j][2])};
#pragma unroll
        for (int i = 0; i < TM; ++i) mfma6(fa[i], fb, acc[i][j]);
      }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  conv_epilogue<WM, WN, TM, TN, UP2, true>(p, acc, smem, m0, n0, M, epi_vec, pre);
}

template <int WM, int WN, int TM, int TN, bool UP2, bool RELU, bool KXK = false>
int launch_variant(const ConvArgs& a, int epi, unsigned tiles, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr int BPASS = (12 * BN + NT - 1) / NT;
  constexpr int kStage = 2 * (BM * BK + BPASS * NT * 4), kEpi = WM * WN * 32 * (TN * 32 + 4);
  constexpr size_t lds = (size_t)(kStage > kEpi ? kStage : kEpi) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = gemm1x1s_kernel<WM, WN, TM, TN, UP2, RELU, KXK>;
  if (lds > 64 * 1024) {
    static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds), true);
    (void)once;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(NT), lds, s, a, epi);
  return check_launch();
}

template <int WM, int WN, int TM, int TN>
int launch_cfg(const ConvArgs& a, int epi, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = ((M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  if (tiles <= 0 || tiles > 0x7fffffffLL) return DH_EINVAL;
  const unsigned t = (unsigned)tiles;
  if (a.up2) {
    if constexpr (TM * TN >= 6) {
      return DH_EUNSUPPORTED;
    } else {
      return a.pre_relu ? launch_variant<WM, WN, TM, TN, true, true>(a, epi, t, s)
                        : launch_variant<WM, WN, TM, TN, true, false>(a, epi, t, s);
    }
  }
  if (!(a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0))
    return a.pre_relu ? launch_variant<WM, WN, TM, TN, false, true, true>(a, epi, t, s)
                      : launch_variant<WM, WN, TM, TN, false, false, true>(a, epi, t, s);
  return a.pre_relu ? launch_variant<WM, WN, TM, TN, false, true>(a, epi, t, s)
                    : launch_variant<WM, WN, TM, TN, false, false>(a, epi, t, s);
}

}  // namespace

bool gemm1x1_eligible(const ConvArgs& a);

int launch_gemm1x1_split(const ConvArgs& a, int cfg, int epi, hipStream_t s) {
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
