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

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// buffer_load_dwordx4 ... offen lds: 16 bytes per lane from (descriptor base + voff + soff) to LDS (wave-uniform `dst`
// + lane * 16).  A 32-bit per-lane offset fixed over K plus a SCALAR K-step offset: no per-step 64-bit address VALU
// (5.7 VALU per MFMA were measured in the first version of this kernel; about 4 hide under a bf16 MFMA).  An
// out-of-range offset loads zeros.  The builtin only exists for the gfx950 pass (see sepconv_fused.hip).
template <typename RSRC>
__device__ __forceinline__ void dma16(RSRC rs, float* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, voff, soff, 0, 0);
#endif
}
constexpr unsigned OOB = 0xfffffff0u;

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

// Six partial products, smallest first: (a part, b part) = (2,0) (0,2) (1,1) (1,0) (0,1) (0,0).  They are issued
// product-major over all TM x TN tiles of the wave: consecutive MFMAs then write DIFFERENT accumulators.  Issued
// tile-major (six dependent MFMAs on one accumulator back to back) every MFMA waits out its predecessor's result
// latency, which is longer than the 32-cycle issue interval of v_mfma_f32_32x32x16_bf16 (measured: matrix pipe 42 % busy,
// 56 % of the wave cycles in issue stalls).  The order per accumulator -- and with it every result bit -- is unchanged.
template <int TM, int TN>
__device__ __forceinline__ void mfma6_tiles(const Frag3 (&a)[TM], const bf16x8 (&b)[TN][3], f32x16 (&acc)[TM][TN]) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].p[PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
}

// NS = number of LDS stages: 2 (next K-step in flight) or 3 (two K-steps in flight: the split kernel's K-step is short
// enough that one DMA round trip no longer fits under it).
template <int WM, int WN, int TM, int TN, bool UP2, bool RELU, bool KXK = false, int NS = 2>
__global__ __launch_bounds__(WM* WN * 64, WM * WN >= 8 ? 2 : 2) void gemm1x1s_kernel(const ConvArgs p, const int epi_vec) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int APASS = BM * 8 / NT;
  constexpr int BPASS = (12 * BN + NT - 1) / NT;         // 16-byte units per thread: 4 k-groups x 3 parts x BN
  constexpr int BROWS = 12 * BN;                         // 16-byte units of the B stage (the last pass is partial)
  constexpr int STAGE = BM * BK + BROWS * 4;             // floats per stage
  static_assert(BM * 8 % NT == 0 && (12 * BN) % 64 == 0, "tile/thread mismatch");

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

  // ---- per-thread DMA sources: byte offsets into two buffer descriptors (activations, packed split weight)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)(((unsigned)(p.N * p.H * p.W - 1) * p.ldx + (unsigned)p.Cin) * 4u), 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((unsigned)p.Kp * p.Np * 6u),
                                                      0x00020000);
  unsigned a_off[APASS];                                  // pointwise: (pixel * ldx + slot) * 4, fixed over K
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
      a_off[ps] = 0;
    } else {
      a_off[ps] = ((unsigned)m * p.ldx + a_slot[ps]) * 4u;
    }
  }
  const int chunks_per_tap = KXK ? p.Cin / BK : 1;
  // packed split weight: 16-byte unit (kg, part, n) at ((kg * 3 + part) * Np + n) * 16 bytes
  unsigned b_off[BPASS];
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int idx = tid + q * NT;
    const int r = idx / BN;
    const int j = idx - r * BN;
    b_off[q] = r < 12 && n0 + j < p.Np ? ((unsigned)r * p.Np + n0 + j) * 16u : OOB;
  }
  const int b_step = 12 * p.Np * 16;                      // bytes per K-step in the packed weight

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
      if constexpr (KXK) {                                // padding taps: out-of-range offset -> zeros
        const int ih = a_ih0[ps] + kh, iw = a_iw0[ps] + kw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && kt * BK < p.K;
        const unsigned off = ok ? ((unsigned)(a_pix[ps] + ih * p.W + iw) * p.ldx + c0 + a_slot[ps]) * 4u : OOB;
        dma16(rs_x, sA + (ps * NT + wave_u * 64) * 4, off, 0);
      } else {                                            // k >= K reads the next pixel (finite) or zeros: weights there are 0
        dma16(rs_x, sA + (ps * NT + wave_u * 64) * 4, a_off[ps], kt * BK * 4);
      }
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q)
      if ((q + 1) * NT <= BROWS || q * NT + wave_u * 64 < BROWS)     // whole waves: 12 * BN is a multiple of 64
        dma16(rs_w, sB + (q * NT + wave_u * 64) * 4, b_off[q], kt * b_step);
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
  if constexpr (NS == 3) {
    if (nk > 1) issue(1, 1);
  }
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
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt == nk - 1) pre.template issue<WM, WN>(p, m0, n0, M, epi_vec);
    const unsigned so = (unsigned)(cur * STAGE * 4);
    const unsigned bo = b_base + so;

    int nxt = cur + NS - 1;                           // stage that was read NS-1 ... 1 K-steps ago: free
    nxt = nxt >= NS ? nxt - NS : nxt;
    const bool more = kt + NS - 1 < nk;
    if constexpr (TM * TN >= 6) {
      // big per-wave tile (64 x 96): LDS fragment traffic per MFMA is what bounds this kernel (each wave re-reads its
      // B columns: 0.61 KB per MFMA at 32 x 96, 0.36 KB at 64 x 96), and 96 accumulators leave room for ONE chunk of
      // operands at a time -- no operand double buffering, the partner wave of the SIMD covers the read latency
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float4 ra1[TM][2], rb1[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ra1[i][0] = lds_rd<0>(a_base[i][c][0] + so);
          ra1[i][1] = lds_rd<0>(a_base[i][c][1] + so);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          rb1[j][0] = lds_rd<0>(bo + (unsigned)((6 * c) * BN * 16 + j * 512));
          rb1[j][1] = lds_rd<0>(bo + (unsigned)((6 * c + 1) * BN * 16 + j * 512));
          rb1[j][2] = lds_rd<0>(bo + (unsigned)((6 * c + 2) * BN * 16 + j * 512));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c == 0 && more) issue(kt + NS - 1, nxt);
        lgkm_wait();
        Frag3 fa[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = split8<RELU>(ra1[i][0], ra1[i][1]);
        bf16x8 fb[TN][3];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q) fb[j][q] = as_bf(rb1[j][q]);
        mfma6_tiles<TM, TN>(fa, fb, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
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
    if (more) issue(kt + NS - 1, nxt);                // DMA of K-step kt+NS-1 flies during this one (and the next)
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
    // chunk 0 is split with nothing to hide behind; chunk 1 is split UNDER chunk 0's MFMAs: the scheduler is told to
    // place VPM VALU instructions after every MFMA (about four hide under a v_mfma_f32_32x32x16_bf16, measured).
    Frag3 fa0[TM], fa1[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa0[i] = split8<RELU>(ra[0][i][0], ra[0][i][1]);
    lgkm_wait();                                      // chunk 1 operands have landed
    {
      bf16x8 fb[TN][3];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) fb[j][q] = as_bf(rb[0][j][q]);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa1[i] = split8<RELU>(ra[1][i][0], ra[1][i][1]);
      mfma6_tiles<TM, TN>(fa0, fb, acc);
      constexpr int VPM = (52 + 6 * TN - 1) / (6 * TN);
#pragma unroll
      for (int u = 0; u < 6 * TM * TN; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);    // VPM VALU
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      bf16x8 fb[TN][3];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) fb[j][q] = as_bf(rb[1][j][q]);
      mfma6_tiles<TM, TN>(fa1, fb, acc);
    }

    }

    // K-step kt+1 must have landed: with three stages the loads just issued (K-step kt+2) may stay in flight
    // (in-order counter: "at most the loads of K-step kt+2 outstanding"; a wave that sat out the partial last B pass
    // issued one load less)
    if (NS == 3 && more) {
      if (BPASS * NT <= BROWS || (BPASS - 1) * NT + wave_u * 64 < BROWS)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APASS + BPASS) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APASS + BPASS - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    cur = cur + 1 == NS ? 0 : cur + 1;
  }

  conv_epilogue<WM, WN, TM, TN, UP2, true>(p, acc, smem, m0, n0, M, epi_vec, pre);
}

template <int WM, int WN, int TM, int TN, bool UP2, bool RELU, bool KXK = false, int NS = 2>
int launch_variant(const ConvArgs& a, int epi, unsigned tiles, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr int BPASS = (12 * BN + NT - 1) / NT;
  constexpr int kStage = NS * (BM * BK + 12 * BN * 4), kEpi = WM * WN * 32 * (TN * 32 + 4);
  constexpr size_t lds = (size_t)(kStage > kEpi ? kStage : kEpi) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = gemm1x1s_kernel<WM, WN, TM, TN, UP2, RELU, KXK, NS>;
  if (lds > 64 * 1024) {
    static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds), true);
    (void)once;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(NT), lds, s, a, epi);
  return check_launch();
}

template <int WM, int WN, int TM, int TN, int NS = 2>
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
      return a.pre_relu ? launch_variant<WM, WN, TM, TN, true, true, false, NS>(a, epi, t, s)
                        : launch_variant<WM, WN, TM, TN, true, false, false, NS>(a, epi, t, s);
    }
  }
  if (!(a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0))
    return a.pre_relu ? launch_variant<WM, WN, TM, TN, false, true, true, NS>(a, epi, t, s)
                      : launch_variant<WM, WN, TM, TN, false, false, true, NS>(a, epi, t, s);
  return a.pre_relu ? launch_variant<WM, WN, TM, TN, false, true, false, NS>(a, epi, t, s)
                    : launch_variant<WM, WN, TM, TN, false, false, false, NS>(a, epi, t, s);
}

}  // namespace

bool gemm1x1_eligible(const ConvArgs& a);

int launch_gemm1x1_split(const ConvArgs& a, int cfg, int epi, hipStream_t s) {
  if (!gemm1x1_eligible(a)) return DH_EUNSUPPORTED;
  // 32-bit byte offsets into the buffer descriptors
  if ((long long)a.N * a.H * a.W * a.ldx * 4 > 0xf0000000LL || (long long)a.Kp * a.Np * 6 > 0xf0000000LL)
    return DH_EUNSUPPORTED;
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
    // one work-group per CU, three LDS stages: two K-steps of DMA in flight, bigger tiles = less L2 -> LDS traffic per MAC
    case 9: return launch_cfg<8, 1, 1, 3, 3>(a, epi, s);      // 256 x 96, 8 waves
    case 10: return launch_cfg<4, 1, 1, 3, 3>(a, epi, s);     // 128 x 96, 4 waves
    case 11: return launch_cfg<8, 1, 1, 2, 3>(a, epi, s);     // 256 x 64
    case 12: return launch_cfg<4, 2, 1, 3, 2>(a, epi, s);     // 128 x 192, 8 waves, two stages
    case 13: return launch_cfg<4, 2, 2, 3, 2>(a, epi, s);     // 256 x 192, 8 waves of 64 x 96
  }
  return DH_EINVAL;
}

int gemm1x1_split_num_cfgs() { return 14; }

}  // namespace dh
