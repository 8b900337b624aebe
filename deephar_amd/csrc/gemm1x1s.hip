// Split-bf16 ("bf16x3") variant of the LDS-DMA GEMM of gemm1x1.hip: the same pointwise / K x K implicit GEMM, the same
// fp32 inputs, outputs, epilogue and K order, but every fp32 operand is split EXACTLY into three bf16 parts
//   x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (round to nearest even)
// and a product is evaluated on the bf16 matrix cores as the six largest of the nine partial products
//   a b ~ a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1          (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|)
// accumulated in fp32 inside v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs cover 16 k-values in 6 x 32 = 192 cycles per
// SIMD; the fp32 MFMA (v_mfma_f32_32x32x2_f32, vector rate) needs 8 x 64 = 512 -- and, unlike the fp32 MFMA, the bf16
// MFMA lets about four VALU instructions per MFMA issue underneath it (LDS reads and DMA pieces do not hide): the K loops
// below are arranged around that budget (profiles/r02_sepconv_fusion_study.md section 3, profiles/r02_split_loop_model.txt).
// The nominal 2.5 PFLOP/s is only sustained on constant operands; on real data the bf16 cores run at ~1.9 PFLOP/s
// (profiles/r02_mfma_data_power.txt).
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
#ifndef DH_WIDE_CHROWS
#define DH_WIDE_CHROWS 6
#endif
constexpr int WIDE_CHROWS = DH_WIDE_CHROWS;   // epilogue rows per chunk of the wide tiling (see conv_epilogue)

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// buffer_load_dwordx4 ... offen lds: 16 bytes per lane from (descriptor base + voff + soff) to LDS (wave-uniform `dst`
// + lane * 16).  A 32-bit per-lane offset fixed over K plus a SCALAR K-step offset: no per-step 64-bit address VALU
// (5.7 VALU per MFMA were measured in the first version of this kernel; about 4 hide under a bf16 MFMA).  An
// out-of-range offset loads zeros.  The builtin only exists for the gfx950 pass (see gemm1x1.hip dma16).
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

// products [T0, T1) of the same sequence
template <int TM, int TN, int T0, int T1>
__device__ __forceinline__ void mfma_products(const Frag3 (&a)[TM], const bf16x8 (&b)[TN][3], f32x16 (&acc)[TM][TN]) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int t = T0; t < T1; ++t)
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
  constexpr bool PIPELINED = TM == 1;                    // software-pipelined K loop (below); else one chunk at a time
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
  // wait until at most `groups` of this wave's K-step DMA groups are outstanding (in-order counter; a wave that sits
  // out the partial last B pass issues one load less per group)
  const bool full_group = BPASS * NT <= BROWS || (BPASS - 1) * NT + wave_u * 64 < BROWS;
  auto wait_groups = [&](int groups) {
    constexpr int G = APASS + BPASS;
    if (groups == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (groups == 1) {
      if (full_group) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G - 1) : "memory");
    } else {
      if (full_group) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G - 2) : "memory");
    }
  };
  issue(0, 0);
  if constexpr (!PIPELINED) {
    if constexpr (NS == 3) {
      if (nk > 1) issue(1, 1);
    }
    wait_groups(0);
  } else {                                            // pipelined loop: K-steps 1 .. NS-1 in flight from the start
    if (nk > 1) issue(1, 1);
    if constexpr (NS == 3) {
      if (nk > 2) issue(2, 2);
    }
    wait_groups(nk - 1 < NS - 1 ? nk - 1 : NS - 1);
  }
  __syncthreads();

  // ---- fragment read addresses (LDS byte offsets), stage 0.  A: row li of the wave's 32-row block, 8 consecutive k
  // of k-group (2c + lh) = slots 2(2c+lh), 2(2c+lh)+1 of the 128-byte row, XOR-swizzled with (row & 7) = (li & 7).
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  unsigned a_base[2][TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        a_base[c][i][h] = lds0 + (unsigned)(((wm * TM + i) * 32 + li) * BK * 4 +
                                            (((2 * (2 * c + lh) + h) ^ (li & 7)) << 4));
  // B: unit ((2c + lh) * 3 + part) * BN + column
  const unsigned b_base = lds0 + (unsigned)(BM * BK * 4) + (unsigned)((lh * 3 * BN + wn * TN * 32 + li) * 16);

  EpiPrefetch<TM, TN> pre;
  if constexpr (!PIPELINED) {
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt == nk - 1) pre.template issue<WM, WN>(p, m0, n0, M, epi_vec);
    const unsigned so = (unsigned)(cur * STAGE * 4);
    const unsigned bo = b_base + so;

    int nxt = cur + NS - 1;                           // stage that was read NS-1 ... 1 K-steps ago: free
    nxt = nxt >= NS ? nxt - NS : nxt;
    const bool more = kt + NS - 1 < nk;
    {
      // big per-wave tile (64 x 96): LDS fragment traffic per MFMA is what bounds this kernel (each wave re-reads its
      // B columns: 0.61 KB per MFMA at 32 x 96, 0.36 KB at 64 x 96), and 96 accumulators leave room for ONE chunk of
      // operands at a time -- no operand double buffering, the partner wave of the SIMD covers the read latency
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float4 ra1[TM][2], rb1[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ra1[i][0] = lds_rd<0>(a_base[c][i][0] + so);
          ra1[i][1] = lds_rd<0>(a_base[c][i][1] + so);
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
  } else {
  // ---- software-pipelined loop (per-wave tiles of up to 5 MFMA tiles): the operands of chunk c+1 (16 k-values) are
  // read from LDS and split WHILE chunk c is multiplied, across K-steps too, so that between two MFMAs the wave only
  // ever issues what hides under them.  Measured before this structure (SQ counters, 64 x 32 x 32 x 576 -> 576): matrix
  // pipe 56 % busy; per wave and K-step 1152 cycles of MFMA and about as much again of exposed LDS round trips, split
  // arithmetic and barrier, which two uncoordinated work-groups per CU do not hide from each other (both waves of a
  // SIMD fall into step: the pipe is shared while both multiply and idle while both fetch).
  // One barrier per K-step, at the top of its SECOND chunk: by then every wave has finished reading stage kt-1 ... so
  // the DMA of K-step kt+NS-1 may overwrite it, and K-step kt+1 (issued NS-1 K-steps earlier) has landed.
  struct Ops { Frag3 a[TM]; bf16x8 b[TN][3]; };
  Ops o0, o1;
  float4 ra[TM][2];
  auto read_chunk = [&](Ops& o, const unsigned (&ab)[TM][2], unsigned so, unsigned boff) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ra[i][0] = lds_rd<0>(ab[i][0] + so);
      ra[i][1] = lds_rd<0>(ab[i][1] + so);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) o.b[j][q] = as_bf(lds_rd<0>(b_base + so + boff + (unsigned)(q * BN * 16 + j * 512)));
  };
  constexpr int NPRE = 2;                                  // products issued before the wait for the next operands
  constexpr int VPM = (52 * TM + (6 - NPRE) * TM * TN - 1) / ((6 - NPRE) * TM * TN);
  // multiply `cur`; when `fetch`, the reads for `nxt` are already in flight: wait for them after NPRE products and
  // split the A rows under the remaining MFMAs (VPM VALU instructions behind each; about four hide, measured)
  auto multiply = [&](const Ops& cur, Ops& nxt, bool fetch) {
    __builtin_amdgcn_sched_barrier(0);
    mfma_products<TM, TN, 0, NPRE>(cur.a, cur.b, acc);
    if (fetch) {
      lgkm_wait();
#pragma unroll
      for (int i = 0; i < TM; ++i) nxt.a[i] = split8<RELU>(ra[i][0], ra[i][1]);
      mfma_products<TM, TN, NPRE, 6>(cur.a, cur.b, acc);
#pragma unroll
      for (int i = 0; i < TM; ++i)                              // keep the split HERE (it would be sunk to its first use)
#pragma unroll
        for (int q = 0; q < 3; ++q) asm volatile("" : "+v"(nxt.a[i].p[q]));
#pragma unroll
      for (int u = 0; u < (6 - NPRE) * TM * TN; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);    // VPM VALU
      }
    } else {
      mfma_products<TM, TN, NPRE, 6>(cur.a, cur.b, acc);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // chunk (0, 0)
  read_chunk(o0, a_base[0], 0u, 0u);
  lgkm_wait();
#pragma unroll
  for (int i = 0; i < TM; ++i) o0.a[i] = split8<RELU>(ra[i][0], ra[i][1]);

  int cur = 0;
  for (int kt = 0; kt < nk - 1; ++kt) {               // the last K-step is peeled: no branch around the MFMA streams
    const unsigned so = (unsigned)(cur * STAGE * 4);
    // first chunk of the K-step: fetch its second one
    read_chunk(o1, a_base[1], so, (unsigned)(6 * BN * 16));
    multiply(o0, o1, true);
    // second chunk: open K-step kt+1
    const int nst = cur + 1 == NS ? 0 : cur + 1;
    wait_groups(NS == 3 && kt + 2 < nk ? 1 : 0);
    __syncthreads();
    if (kt + NS < nk) issue(kt + NS, cur);            // stage of K-step kt: every wave has read both of its chunks
    read_chunk(o0, a_base[0], (unsigned)(nst * STAGE * 4), 0u);
    multiply(o1, o0, true);
    cur = nst;
  }
  pre.template issue<WM, WN>(p, m0, n0, M, epi_vec);
  read_chunk(o1, a_base[1], (unsigned)(cur * STAGE * 4), (unsigned)(6 * BN * 16));
  multiply(o0, o1, true);
  multiply(o1, o0, false);
  }

  conv_epilogue<WM, WN, TM, TN, UP2, true>(p, acc, smem, m0, n0, M, epi_vec, pre);
}

// ---------------------------------------------------------------------------------------------------------------
// Wide per-wave tile: 32 rows x 192 columns per wave, WM waves stacked over M (work-group tile 32 WM x 192), 16-k
// K-steps, two LDS stages (52 KB at WM = 4: two work-groups per CU, so one tile's epilogue runs beside another's K loop).
// What bounds the 32 x 96 tiling above is not the matrix pipe but everything the wave issues between its MFMAs
// (tools/micro/split_loop_model*.hip: that instruction mix reaches 58-62 % pipe occupancy with memory taken away, the
// mix of this tiling 80-84 %): the activation split costs 52 VALU per 32 rows x 8 k however many columns use it, and an
// LDS-DMA piece costs its wave ~60 issue cycles.  Twice the columns per wave halve the VALU per MFMA, 128 x 192 instead
// of 128 x 96 per work-group takes the DMA pieces per MFMA from 0.25 to 0.18.
// The 96 accumulator registers leave room for ONE set of B operands per half tile (3 column tiles x 3 parts): the K-step
// is processed as two halves (column tiles 0-2, then 3-5), each half's B operands are read from LDS while the other
// half is multiplied, the next K-step's activation rows are read at the start of the second half and split under its
// MFMAs.  One barrier per K-step, at the start of its second half: every wave has read the whole stage by then (it is
// refilled with K-step kt+2), and K-step kt+1 -- issued one K-step earlier -- has landed.
// Per accumulator the products arrive in the same order as in every other tiling: results are bit-identical.
template <int WM, bool UP2, bool RELU, bool KXK>
__global__ __launch_bounds__(WM * 64, 2) void gemm1x1s_wide_kernel(const ConvArgs p, const int epi_vec) {
  constexpr int NT = WM * 64;
  constexpr int BKW = 16;
  constexpr int BM = WM * 32, BN = 192;
  constexpr int APASS = BM * 4 / NT;                      // = 2: 16-byte units of the A stage per thread
  constexpr int BROWS = 6 * BN;                          // 16-byte units of the B stage: 2 k-groups x 3 parts x BN
  constexpr int BPASS = (BROWS + NT - 1) / NT;
  constexpr int A_BYTES = BM * BKW * 4;
  constexpr int STAGE_BYTES = A_BYTES + BROWS * 16;
  static_assert(STAGE_BYTES + 2 * BN * 16 + 5 * 512 < 65536, "ds_read immediate offsets");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int M = p.N * p.OH * p.OW;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)(((unsigned)(p.N * p.H * p.W - 1) * p.ldx + (unsigned)p.Cin) * 4u), 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((unsigned)p.Kp * p.Np * 6u),
                                                      0x00020000);
  // A stage: 64-byte rows of four 16-byte slots, slot XOR ((row >> 2) & 3): the 16 lanes ds_read_b128 serves per cycle
  // (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a wave's block) then fall on 16 different bank groups
  unsigned a_off[APASS];
  int a_slot[APASS];
  int a_pix[KXK ? APASS : 1], a_ih0[KXK ? APASS : 1], a_iw0[KXK ? APASS : 1];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int r = (tid >> 2) + ps * (NT / 4);
    int m = m0 + r;
    m = m < M ? m : M - 1;
    a_slot[ps] = ((tid & 3) ^ ((r >> 2) & 3)) * 4;
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
  const int chunks_per_tap = KXK ? p.Cin / BKW : 1;
  unsigned b_off[BPASS];
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int idx = tid + q * NT;
    const int r = idx / BN;
    const int j = idx - r * BN;
    b_off[q] = r < 6 && n0 + j < p.Np ? ((unsigned)r * p.Np + n0 + j) * 16u : OOB;
  }
  const int b_step = 6 * p.Np * 16;                       // bytes per 16-k K-step in the packed weight
  // the last B pass is partial: the waves beyond it send their piece (out-of-range offset: zeros, no fetch) to a dump
  // slot behind the stages, so that every wave issues the same branch-free sequence
  constexpr int DUMP_BYTES = 2 * STAGE_BYTES;

  auto issue = [&](int kt, int stage) {
    float* sA = smem + stage * (STAGE_BYTES / 4);
    float* sB = sA + A_BYTES / 4;
    int kh = 0, kw = 0, c0 = 0;
    if constexpr (KXK) {
      const int tap = kt / chunks_per_tap;
      c0 = (kt - tap * chunks_per_tap) * BKW;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      if constexpr (KXK) {
        const int ih = a_ih0[ps] + kh, iw = a_iw0[ps] + kw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && kt * BKW < p.K;
        const unsigned off = ok ? ((unsigned)(a_pix[ps] + ih * p.W + iw) * p.ldx + c0 + a_slot[ps]) * 4u : OOB;
        dma16(rs_x, sA + (ps * NT + wave_u * 64) * 4, off, 0);
      } else {
        dma16(rs_x, sA + (ps * NT + wave_u * 64) * 4, a_off[ps], kt * BKW * 4);
      }
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      if ((q + 1) * NT <= BROWS) {
        dma16(rs_w, sB + (q * NT + wave_u * 64) * 4, b_off[q], kt * b_step);
      } else {
        const bool mine = q * NT + wave_u * 64 < BROWS;   // wave-uniform
        dma16(rs_w, mine ? sB + (q * NT + wave_u * 64) * 4 : smem + DUMP_BYTES / 4 + wave_u * 256, b_off[q], kt * b_step);
      }
    }
  };

  f32x16 acc[2][1][3];                                    // [half][1][column tile of the half]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][0][j][r] = 0.f;

  const int nk = p.Kp / BKW;                              // even, >= 2 (Kp is a multiple of 32)
  issue(0, 0);
  issue(1, 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APASS + BPASS) : "memory");
  __syncthreads();

  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned sw = (unsigned)((li >> 2) & 3);
  const unsigned a_rd0 = lds0 + (unsigned)((wave * 32 + li) * 64) + (((2u * lh) ^ sw) << 4);
  const unsigned a_rd1 = lds0 + (unsigned)((wave * 32 + li) * 64) + (((2u * lh + 1u) ^ sw) << 4);
  const unsigned b_rd = lds0 + (unsigned)A_BYTES + (unsigned)((lh * 3 * BN + li) * 16);

  Frag3 fa0[1], fa1[1];
  bf16x8 bq0[3][3], bq1[3][3];                            // B operands of column tiles 0-2 / 3-5
  float4 ra[2];
  // stage and half are compile-time: every ds_read is one base register plus an immediate
#define DH_RD_B1(DST, ST, HALF, J)                                                                      \
  DST[J][0] = as_bf(lds_rd<(ST) * STAGE_BYTES + ((HALF) * 3 + (J)) * 512>(b_rd));                       \
  DST[J][1] = as_bf(lds_rd<(ST) * STAGE_BYTES + BN * 16 + ((HALF) * 3 + (J)) * 512>(b_rd));             \
  DST[J][2] = as_bf(lds_rd<(ST) * STAGE_BYTES + 2 * BN * 16 + ((HALF) * 3 + (J)) * 512>(b_rd));
#define DH_RD_B(DST, ST, HALF) DH_RD_B1(DST, ST, HALF, 0) DH_RD_B1(DST, ST, HALF, 1) DH_RD_B1(DST, ST, HALF, 2)
#define DH_RD_A(ST)                                        \
  ra[0] = lds_rd<(ST) * STAGE_BYTES>(a_rd0);               \
  ra[1] = lds_rd<(ST) * STAGE_BYTES>(a_rd1);

  constexpr int VPM = (52 + 11) / 12;
  // one K-step: multiply (CUR, stage ST).  MODE 0: open K-step kt+1 (stage 1 - ST), split its rows into NXT and refill
  // stage ST with K-step kt+2; MODE 1: the same without the refill (last but one); MODE 2: last K-step.
  // The refill's DMA pieces cost the wave ~60 issue cycles each: they go out one behind each of the next MFMAs instead
  // of in one burst behind the barrier (tools/micro/split_loop_model2.hip: 80 -> 84 % pipe occupancy).
#define DH_KSTEP(CUR, NXT, ST, MODE)                                                                             \
  {                                                                                                              \
    DH_RD_B(bq1, ST, 1)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    mfma_products<1, 3, 0, 6>(CUR, bq0, acc[0]);                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    lgkm_wait();                                                                                                 \
    if constexpr ((MODE) != 2) {                                                                                 \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
      __syncthreads();                                                                                           \
      DH_RD_B(bq0, 1 - (ST), 0)                                                                                  \
      DH_RD_A(1 - (ST))                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
      if constexpr ((MODE) == 0) issue(kt + 2, ST);                                                              \
      mfma_products<1, 3, 0, 2>(CUR, bq1, acc[1]);                                                               \
      if constexpr ((MODE) == 0) {                                                                               \
        _Pragma("unroll") for (int u = 0; u < 5; ++u) {                                                          \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                     \
        }                                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        __builtin_amdgcn_sched_group_barrier(0x020, APASS + BPASS - 5, 0);                                       \
      }                                                                                                          \
      lgkm_wait();                                                                                               \
      NXT[0] = split8<RELU>(ra[0], ra[1]);                                                                       \
      mfma_products<1, 3, 2, 6>(CUR, bq1, acc[1]);                                                               \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) asm volatile("" : "+v"(NXT[0].p[q]));                        \
      _Pragma("unroll") for (int u = 0; u < 12; ++u) {                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                                                     \
      }                                                                                                          \
    } else {                                                                                                     \
      pre0.template issue<WM, 1>(p, m0, n0, M, epi_vec);                                                         \
      mfma_products<1, 3, 0, 6>(CUR, bq1, acc[1]);                                                               \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  }

  EpiPrefetch<1, 3> pre0;
  // K-step 0: rows and the first half's B operands
  DH_RD_B(bq0, 0, 0)
  DH_RD_A(0)
  lgkm_wait();
  fa0[0] = split8<RELU>(ra[0], ra[1]);

  int kt = 0;
  for (; kt < nk - 2; kt += 2) {
    DH_KSTEP(fa0, fa1, 0, 0)
    ++kt;
    DH_KSTEP(fa1, fa0, 1, 0)
    --kt;
  }
  DH_KSTEP(fa0, fa1, 0, 1)
  ++kt;
  DH_KSTEP(fa1, fa0, 1, 2)
#undef DH_KSTEP
#undef DH_RD_A
#undef DH_RD_B
#undef DH_RD_B1

  // columns 0-95 (residual rows requested during the last K-step), then 96-191 through the same slab.  Holding the
  // second slice's residual rows in registers across the first slice's row loop spills (measured: 80 dwords); its
  // round trip is covered by the other work-group of the CU instead.
  conv_epilogue<WM, 1, 1, 3, UP2, true, EpiNoHook, true, WIDE_CHROWS>(p, acc[0], smem, m0, n0, M, epi_vec, pre0);
  conv_epilogue<WM, 1, 1, 3, UP2, false, EpiNoHook, true, WIDE_CHROWS>(p, acc[1], smem, m0, n0 + 96, M, epi_vec, pre0);
}

template <int WM, bool UP2, bool RELU, bool KXK>
int launch_wide_variant(const ConvArgs& a, int epi, unsigned tiles, hipStream_t s) {
  constexpr int kStage = 2 * (WM * 32 * 16 * 4 + 6 * 192 * 16) + WM * 1024, kEpi = WM * 32 * (3 * 32 + 4) * 4;
  constexpr size_t lds = kStage > kEpi ? kStage : kEpi;
  auto kern = gemm1x1s_wide_kernel<WM, UP2, RELU, KXK>;
  if (lds > 64 * 1024) {
    static LdsLimit lim;
    lim.raise((const void*)kern, (int)lds);
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * 64), lds, s, a, epi);
  return check_launch();
}

template <int WM>
int launch_wide(const ConvArgs& a, int epi, hipStream_t s) {
  constexpr int BM = WM * 32, BN = 192;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = ((M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  if (tiles <= 0 || tiles > 0x7fffffffLL) return DH_EINVAL;
  const unsigned t = (unsigned)tiles;
  if (a.up2)
    return a.pre_relu ? launch_wide_variant<WM, true, true, false>(a, epi, t, s)
                      : launch_wide_variant<WM, true, false, false>(a, epi, t, s);
  if (!(a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0))
    return a.pre_relu ? launch_wide_variant<WM, false, true, true>(a, epi, t, s)
                      : launch_wide_variant<WM, false, false, true>(a, epi, t, s);
  return a.pre_relu ? launch_wide_variant<WM, false, true, false>(a, epi, t, s)
                    : launch_wide_variant<WM, false, false, false>(a, epi, t, s);
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
    static LdsLimit lim;
    lim.raise((const void*)kern, (int)lds);
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
  if (a.y_pool != nullptr && !conv_epilogue_pools_for<WM, TM, false>(a)) return DH_EUNSUPPORTED;
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

// What dh_conv2d_f32 accepts with w_split = 1 (the weight pointer itself is not looked at: a binding asks before it packs).
bool gemm1x1_split_eligible(const ConvArgs& a0) {
  ConvArgs a = a0;
  a.w = reinterpret_cast<const float*>(uintptr_t(16));
  a.w_split = 0;                       // (conv_is_skinny answers for the fp32 packing)
  if (a.x_u8 || a.pre_scale != nullptr || conv_is_skinny(a) || !gemm1x1_eligible(a)) return false;   // (no BN prologue on the split path)
  // 32-bit byte offsets into the buffer descriptors
  return (long long)a.N * a.H * a.W * a.ldx * 4 <= 0xf0000000LL && (long long)a.Kp * a.Np * 6 <= 0xf0000000LL;
}

int launch_gemm1x1_split(const ConvArgs& a, int cfg, int epi, hipStream_t s) {
  if (!gemm1x1_eligible(a) || !gemm1x1_split_eligible(a)) return DH_EUNSUPPORTED;
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
    case 14: return launch_wide<4>(a, epi, s);                // 128 x 192, 4 waves of 32 x 192, 16-k K-steps
    case 15: return launch_wide<2>(a, epi, s);                // 64 x 192
  }
  return DH_EINVAL;
}

int gemm1x1_split_num_cfgs() { return 16; }

}  // namespace dh
