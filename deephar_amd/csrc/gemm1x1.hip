// Pointwise (1x1, stride 1, no padding) convolution = GEMM [M = N*H*W, K = Cin] x [K, Cout] for gfx950.
// 80 % of ReceptionNet's MACs: the pointwise half of every SeparableConv2D plus the
// 1x1 projections (reference deephar/layers.py:74-80, 258-301; models/reception.py:43-59,145-164).
//
// Same MFMA core and fragment mapping as conv_igemm.hip (v_mfma_f32_32x32x2_f32, exact fp32), but the
// staging is LDS-DMA: buffer_load_dwordx4 ... lds writes both operand tiles straight into LDS, double buffered,
// so a K-step costs no staging VGPRs, no ds_write and ONE barrier; the DMA of tile k+1 is in flight
// during the whole MFMA block of tile k.
//   A (activations, rows = pixels): LDS image [BM][32] floats, 128-byte rows, 16-byte slots XOR-swizzled
//      with (row & 7).  The DMA destination is lane-linear (wave base + lane*16), so the swizzle lives on the
//      per-lane SOURCE address and on the fragment read (both-sides-or-neither).
//   B (weights): host-packed [K/4][N][4] -> the LDS image is the global image, copied linearly.
//   ReLU-on-load (act_conv_bn) is applied to the A fragments after the ds_read; so is a BatchNormalization prologue
//   (PRE: x * scale[k] + shift[k] before the ReLU -- the pre-activation 1x1 convs of SPNet's residual units,
//   common.py:25-67, whose input has other readers and therefore cannot take the BN in its producer's epilogue): the
//   per-channel scale / shift sit in LDS behind the two operand stages and are read with the fragments.
//   Rows >= M and k >= K are clamped to valid addresses: padded weight rows are zero, tail rows never stored.
#include <algorithm>

#include "conv_common.h"
#include "dw_lds.h"

namespace dh {
namespace {

constexpr int BK = 32;

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned OOB = 0xfffffff0u;   // a buffer offset beyond every descriptor used here: the load returns zeros

// buffer_load_dwordx4 ... offen lds: 16 bytes per lane from (descriptor base + voff + soff) to LDS (wave-uniform `dst`
// + lane * 16).  The builtin only exists for the gfx950 pass: hipcc's host pass silently drops a kernel TEMPLATE whose
// body names it (no host stub is emitted), hence the guard.
template <typename RSRC>
__device__ __forceinline__ void dma16(RSRC rs, float* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, voff, soff, 0, 0);
#endif
}

// Fragment reads go through inline asm on purpose: with an LDS-DMA in flight hipcc cannot prove that the DMA's
// LDS destination (the other stage) does not alias an ordinary ds_read and puts `s_waitcnt vmcnt(0)` in front
// of the first fragment read of every K-step, which serialises the DMA with the MFMA block it is meant to
// overlap.  An asm ds_read is invisible to that pass; its completion is awaited by hand (lgkm_wait) and the
// consumers are fenced behind the wait with sched_barrier (hipcc would otherwise hoist the MFMAs).
template <int OFF>
__device__ __forceinline__ float4 lds_rd(unsigned addr) {
  float4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void lgkm_wait() {
  __builtin_amdgcn_sched_barrier(0);   // MFMAs issued before the wait stay before it: they are what hides the read
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// ReLU as ONE integer instruction per element: non-negative floats order like non-negative ints and everything with
// the sign bit set is a negative int, so max_i32(bits, 0) is relu(v) (-0 -> +0).  fmaxf costs two v_max_f32 (a
// canonicalising one first), and on gfx950 nothing issues beside an fp32 MFMA (profiles/r02_sepconv_fusion_study.md):
// every VALU instruction of the K loop is paid in full -- 32 instead of 16 per K-step was 4-8 % of the ReLU-on-load
// GEMMs.  (An asm v_max_f32 is not an option: hipcc pads no VALU->MFMA hazard wait states after inline asm.)
__device__ __forceinline__ float relu1(float v) {
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}
__device__ __forceinline__ float4 relu4(float4 v) { return make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)); }

// PRE: scale / shift of the four k values this lane's A fragment of sub-step S covers (k = kt*32 + (2S + lh)*4 + e)
template <bool PRE, int S>
__device__ __forceinline__ void fetch_affine(unsigned t_sc, unsigned t_sh, float4& sc, float4& sh) {
  if constexpr (PRE) {
    sc = lds_rd<S * 32>(t_sc);
    sh = lds_rd<S * 32>(t_sh);
  }
}

template <int TM, int TN, int BN, int BOFF, int S>
__device__ __forceinline__ void fetch_frags(const unsigned (&a_addr)[TM][4], unsigned b_addr, float4 (&fa)[TM],
                                            float4 (&fb)[TN]) {
  fa[0] = lds_rd<0>(a_addr[0][S]);
  if constexpr (TM > 1) fa[1] = lds_rd<0>(a_addr[1][S]);
  constexpr int BO = BOFF + S * 2 * BN * 16;
  fb[0] = lds_rd<BO>(b_addr);
  if constexpr (TN > 1) fb[1] = lds_rd<BO + 512>(b_addr);
  if constexpr (TN > 2) fb[2] = lds_rd<BO + 1024>(b_addr);
}

// KXK: the same kernel as an implicit GEMM over a K x K (strided, zero-padded) convolution whose Cin is a multiple
// of 32: K-step kt covers 32 channels of ONE filter tap, so each staged row is still one contiguous 128-byte run -- of
// the tap's input pixel, or of a page of zeros when the tap falls into the padding (ReLU-on-load keeps zeros zero).
// The kernel body as a device function: `block` of `nblocks` is the work-group's index among the GEMM work-groups (the
// kernel's blockIdx.x / gridDim.x, or its share of a grouped launch: conv_dw_group_kernel below), `smem` the work-group's
// dynamic LDS; threads 0 .. WM * WN * 64 - 1 run it.
template <int WM, int WN, int TM, int TN, bool UP2, bool RELU, bool KXK = false, bool PRE = false>
__device__ __forceinline__ void gemm1x1_body(const ConvArgs& p, const int epi_vec, const int block, const int nblocks,
                                             float* const smem) {
  static_assert(!(PRE && (KXK || UP2)), "the BN prologue is built for the plain pointwise form");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int APASS = BM * 8 / NT;
  constexpr int BPASS = 8 * BN / NT;
  constexpr int STAGE = BM * BK + BK * BN;  // floats per stage
  static_assert(BM * 8 % NT == 0 && (8 * BN) % NT == 0, "tile/thread mismatch");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int M = p.N * p.OH * p.OW;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile = xcd_tile(block, nblocks);
  const int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  // ---- per-thread DMA sources: byte offsets into two buffer descriptors (activations, packed weight), fixed over K;
  // the K-step offset is SCALAR (soffset of buffer_load ... lds), so a K-step costs no 64-bit address VALU -- on gfx950
  // every VALU instruction beside an fp32 MFMA is paid in full (profiles/r02_sepconv_fusion_study.md section 3)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)(((unsigned)(p.N * p.H * p.W - 1) * p.ldx + (unsigned)p.Cin) * 4u), 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((unsigned)p.Kp * p.Np * 4u),
                                                      0x00020000);
  unsigned a_off[APASS];                            // pointwise: (pixel * ldx + slot) * 4
  int a_slot[APASS];
  int a_pix[KXK ? APASS : 1], a_ih0[KXK ? APASS : 1], a_iw0[KXK ? APASS : 1];   // KXK: frame base pixel, top-left tap
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int r = (tid >> 3) + ps * (NT / 8);       // row inside the tile; lane-linear LDS slot' = tid&7
    int m = m0 + r;
    m = m < M ? m : M - 1;
    a_slot[ps] = ((tid & 7) ^ (r & 7)) * 4;          // swizzle on the source side
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
  unsigned b_off[BPASS];
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int idx = tid + q * NT;
    const int kq = idx / BN, j = idx - kq * BN;
    const int col = n0 + j < p.Np ? n0 + j : 0;
    b_off[q] = ((unsigned)kq * p.Np + col) * 16u;
  }
  const int b_step = 8 * p.Np * 16;                 // bytes per K-step in the packed weight

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
      if constexpr (KXK) {                          // padding taps: out-of-range offset -> the DMA writes zeros
        const int ih = a_ih0[ps] + kh, iw = a_iw0[ps] + kw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && kt * BK < p.K;
        const unsigned off = ok ? ((unsigned)(a_pix[ps] + ih * p.W + iw) * p.ldx + c0 + a_slot[ps]) * 4u : OOB;
        dma16(rs_x, sA + (ps * NT + wave_u * 64) * 4, off, 0);
      } else {                                      // k >= K: the next pixel's (finite) data or zeros, times zero weights
        dma16(rs_x, sA + (ps * NT + wave_u * 64) * 4, a_off[ps], kt * BK * 4);
      }
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) dma16(rs_w, sB + (q * NT + wave_u * 64) * 4, b_off[q], kt * b_step);
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
  if constexpr (PRE) {              // scale | shift tables behind the stages, zero beyond K (k >= K meets zero weights)
    float* tab = smem + 2 * STAGE;
    for (int i = tid; i < p.Kp; i += NT) {
      tab[i] = i < p.K ? p.pre_scale[i] : 0.f;
      tab[p.Kp + i] = i < p.K ? p.pre_shift[i] : 0.f;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- fragment read addresses (LDS byte offsets), stage 0; tile row blocks start at multiples of 32 so
  // (row & 7) == (lane & 7) and the XOR swizzle only depends on the lane
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  unsigned a_base[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a_base[i][q] = lds0 + (unsigned)(((wm * TM + i) * 32 + li) * BK * 4 + (((q * 2 + lh) ^ (li & 7)) << 4));
  const unsigned b_base = lds0 + (unsigned)((lh * BN + wn * TN * 32 + li) * 16);
  constexpr int BOFF = BM * BK * 4;   // B tile follows the A tile inside a stage

  auto mfma_block = [&](const float4 (&a)[TM], const float4 (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
      }
  };
  auto relu_frags = [&](float4 (&a)[TM], const float4& sc, const float4& sh) {
    if constexpr (PRE) {            // same fused multiply-add as the general kernel's prologue (conv_igemm.hip): same bits
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = make_float4(fmaf(a[i].x, sc.x, sh.x), fmaf(a[i].y, sc.y, sh.y), fmaf(a[i].z, sc.z, sh.z), fmaf(a[i].w, sc.w, sh.w));
    }
    if constexpr (RELU) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = relu4(a[i]);
    }
  };
  const unsigned t_sc0 = lds0 + (unsigned)(2 * STAGE * 4 + lh * 16);

  // NB: an asm ds_read result must be awaited inside the basic block that issued it -- register copies the
  // allocator inserts at block boundaries (loop back-edge, if/else joins) would otherwise copy registers the
  // data has not reached yet.  Hence every fetch and its wait live in the straight-line body of one K-step.
  // (Fetching tile kt+1's first fragments before the loop back-edge was tried: correct with a tail wait, but
  // the loop-carried fragment registers cost more than the hidden LDS latency buys: 111 vs 118 TFLOP/s.)
  EpiPrefetch<TM, TN> pre;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt == nk - 1) pre.template issue<WM, WN>(p, m0, n0, M, epi_vec);   // lands during the last MFMA block
    const unsigned so = (unsigned)(cur * STAGE * 4);
    unsigned a_addr[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) a_addr[i][q] = a_base[i][q] + so;
    const unsigned b_addr = b_base + so;

    // sub-step s+1's fragments are in flight (asm ds_read) while sub-step s's MFMAs run; the first read of the
    // K-step has no MFMAs to hide behind, so the DMA issue of the next tile (address VALU + 6 loads) goes there
    float4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    float4 sc0, sh0, sc1, sh1;                        // PRE only
    const unsigned t_sc = t_sc0 + (unsigned)(kt * BK * 4), t_sh = t_sc + (unsigned)(p.Kp * 4);
    fetch_frags<TM, TN, BN, BOFF, 0>(a_addr, b_addr, fa0, fb0);
    fetch_affine<PRE, 0>(t_sc, t_sh, sc0, sh0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);          // DMA of the next tile flies during this MFMA block
    lgkm_wait();
    fetch_frags<TM, TN, BN, BOFF, 1>(a_addr, b_addr, fa1, fb1);
    fetch_affine<PRE, 1>(t_sc, t_sh, sc1, sh1);
    __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs they overlap with
    relu_frags(fa0, sc0, sh0);
    mfma_block(fa0, fb0);
    lgkm_wait();
    fetch_frags<TM, TN, BN, BOFF, 2>(a_addr, b_addr, fa0, fb0);
    fetch_affine<PRE, 2>(t_sc, t_sh, sc0, sh0);
    __builtin_amdgcn_sched_barrier(0);
    relu_frags(fa1, sc1, sh1);
    mfma_block(fa1, fb1);
    lgkm_wait();
    fetch_frags<TM, TN, BN, BOFF, 3>(a_addr, b_addr, fa1, fb1);
    fetch_affine<PRE, 3>(t_sc, t_sh, sc1, sh1);
    __builtin_amdgcn_sched_barrier(0);
    relu_frags(fa0, sc0, sh0);
    mfma_block(fa0, fb0);
    lgkm_wait();
    relu_frags(fa1, sc1, sh1);
    mfma_block(fa1, fb1);

    // tile kt+1 has landed (this wave's DMA) and every wave is done reading tile kt
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  conv_epilogue<WM, WN, TM, TN, UP2, true>(p, acc, smem, m0, n0, M, epi_vec, pre);
}

template <int WM, int WN, int TM, int TN, bool UP2, bool RELU, bool KXK = false, bool PRE = false>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm1x1_kernel(const ConvArgs p, const int epi_vec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gemm1x1_body<WM, WN, TM, TN, UP2, RELU, KXK, PRE>(p, epi_vec, (int)blockIdx.x, (int)gridDim.x, smem);
}

// [r06] Grouped launch for the LATENCY regime: the 1x1 shortcut convolution of a residual unit and the depthwise convolution
// of its main path (deephar/models/common.py:25-67: `shortcut = conv2d(relu(BN(x)))` beside `sepconv2d(relu(BN(x)))`) read
// the same tensor and do not depend on each other, but as two nodes of a one-stream graph they run one after the other, and
// a node costs ~5 us plus its work however small (profiles/r06_speed2d_timeline.md); putting one of them on another stream
// costs more than it saves (profiles/r06_helper_stream_experiment.txt).  Here ONE launch runs both: work-groups
// [0, nb_conv) are the GEMM's, the rest the depthwise kernel's (dw_lds.h), each running exactly the code of its stand-alone
// kernel -- same bits.  Work-groups are 256 threads; the waves a body does not need end at once (s_barrier counts only the
// waves still alive).
template <int WM, bool PRE, bool RELU, int KS, int DNT, int MAXT, int MAXN, bool AFF, bool DRELU>
__global__ __launch_bounds__(256) void conv_dw_group_kernel(const ConvArgs pc, const int epi_vec, const int nb_conv,
                                                            const DwArgs pd, const int tw, const int rows,
                                                            const int twh_magic) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < nb_conv) {
    if (threadIdx.x < WM * 64)
      gemm1x1_body<WM, 1, 1, 1, false, RELU, false, PRE>(pc, epi_vec, (int)blockIdx.x, nb_conv, smem);
  } else {
    if (threadIdx.x < DNT)
      dwl::dwconv_lds_body<KS, DNT, MAXT, MAXN, AFF, DRELU>(pd, tw, rows, twh_magic, (int)blockIdx.x - nb_conv,
                                                            reinterpret_cast<float4*>(smem));
  }
}

constexpr int kMaxPreKp = 4096;    // BN prologue: scale + shift tables of at most 2 x 16 KB in LDS

template <int WM, int WN, int TM, int TN, bool UP2, bool RELU, bool KXK = false, bool PRE = false>
int launch_variant(const ConvArgs& a, int epi, unsigned tiles, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr int kStage = 2 * (BM * BK + BK * BN), kEpi = WM * WN * 32 * (TN * 32 + 4);
  constexpr size_t kLds = (size_t)(kStage > kEpi ? kStage : kEpi) * sizeof(float);
  static_assert(kLds + (PRE ? 2 * kMaxPreKp * sizeof(float) : 0) <= 160 * 1024, "LDS budget");
  // PRE: the tables sit behind the two stages (the epilogue slab, when larger, only starts after the K loop)
  const size_t lds = PRE ? std::max(kLds, (size_t)(kStage + 2 * a.Kp) * sizeof(float)) : kLds;
  auto kern = gemm1x1_kernel<WM, WN, TM, TN, UP2, RELU, KXK, PRE>;
  if (kLds + (PRE ? 2 * kMaxPreKp * sizeof(float) : 0) > 64 * 1024) {
    static LdsLimit lim;
    lim.raise((const void*)kern, (int)(kLds + (PRE ? 2 * kMaxPreKp * sizeof(float) : 0)));
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
  if (a.y_pool != nullptr && !conv_epilogue_pools_for<WM, TM, false>(a)) return DH_EUNSUPPORTED;
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
  if (a.pre_scale != nullptr)       // BatchNormalization (+ ReLU) prologue on the A fragments
    return a.pre_relu ? launch_variant<WM, WN, TM, TN, false, true, false, true>(a, epi, t, s)
                      : launch_variant<WM, WN, TM, TN, false, false, false, true>(a, epi, t, s);
  return a.pre_relu ? launch_variant<WM, WN, TM, TN, false, true>(a, epi, t, s)
                    : launch_variant<WM, WN, TM, TN, false, false>(a, epi, t, s);
}

}  // namespace

bool gemm1x1_eligible(const ConvArgs& a) {
  const bool aligned = a.ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;
  const bool pointwise = a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0 &&
                         a.H == a.OH && a.W == a.OW && a.Cin % 4 == 0;
  // K x K: a K-step must not straddle filter taps, the up-sampling epilogue is only built for the pointwise form
  const bool kxk = a.Cin % BK == 0 && !a.up2 && a.KH >= 1 && a.KW >= 1 && a.SH >= 1 && a.SW >= 1 && a.PT >= 0 &&
                   a.PL >= 0;
  // a BatchNormalization prologue: the plain pointwise form only (no zero padding to apply it around, no fused
  // up-sampling), fp32 weights, tables that fit LDS
  const bool pre_ok = a.pre_scale == nullptr || (pointwise && !a.up2 && a.w_split == 0 && a.Kp <= kMaxPreKp);
  const bool fits32 = (long long)a.N * a.H * a.W * a.ldx * 4 <= 0xf0000000LL;      // 32-bit buffer offsets
  return aligned && pre_ok && fits32 && (pointwise || kxk);
}

bool conv_is_skinny(const ConvArgs& a);
bool conv_stem_eligible(const ConvArgs& a);

namespace {

template <int WM, int KS, int DNT, int MAXT, int MAXN>
int launch_group(const ConvArgs& a, int epi, const DwArgs& d, const dwl::DwLdsGeom& g, hipStream_t s) {
  constexpr int BM = WM * 32, BN = 32;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = ((M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  if (tiles <= 0 || tiles + (long long)g.blocks > 0x7fffffffLL) return DH_EINVAL;
  constexpr int kStage = 2 * (BM * BK + BK * BN), kEpi = WM * 32 * (32 + 4);
  const size_t lds_conv = (size_t)std::max(kStage > kEpi ? kStage : kEpi, kStage + 2 * a.Kp) * sizeof(float);
  const size_t lds = std::max(lds_conv, g.lds);
  auto kern = conv_dw_group_kernel<WM, true, true, KS, DNT, MAXT, MAXN, true, true>;
  static LdsLimit lim;
  lim.raise((const void*)kern, 160 * 1024);
  if (lds > 160 * 1024) return DH_EUNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles + g.blocks), dim3(256), lds, s, a, epi, (int)tiles, d, g.tw, g.rows,
                     65536 / (g.tw + KS - 1) + 1);
  return check_launch();
}

}  // namespace

// The 1x1 shortcut convolution `a` (BatchNormalization + ReLU prologue, fp32 tap-major weights, LDS-DMA GEMM family) and the
// 5x5 depthwise convolution `d` (BatchNormalization + ReLU prologue, LDS-tiled kernel) in one launch; DH_EUNSUPPORTED for
// anything else -- the caller then launches the two on their own, with the same result bits.
int launch_conv_dw_group(const ConvArgs& a, const DwArgs& d, hipStream_t s) {
  if (a.N <= 0 || a.Cin <= 0 || a.Cout <= 0 || a.OH <= 0 || a.OW <= 0 || d.N <= 0 || d.C <= 0) return DH_EINVAL;
  if (a.Kp % BK != 0 || a.Np % 32 != 0 || a.Kp < a.K || a.Np < a.Cout || a.K != a.KH * a.KW * a.Cin) return DH_EINVAL;
  const bool pointwise = a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0 && a.H == a.OH && a.W == a.OW;
  if (!pointwise || a.up2 || a.y_pool != nullptr || a.x_u8 || a.w_split || a.pre_scale == nullptr || !a.pre_relu ||
      a.Kp > kMaxPreKp || !gemm1x1_eligible(a) || conv_is_skinny(a) || conv_stem_eligible(a))
    return DH_EUNSUPPORTED;
  if (a.res2_down && (a.res2 == nullptr || (a.OH & 1) || (a.OW & 1))) return DH_EINVAL;
  dwl::DwLdsGeom g;
  if (d.KW != 5 || d.pre_scale == nullptr || !d.pre_relu || !dwl::dw_lds_geometry(d, g)) return DH_EUNSUPPORTED;
  if (d.up_in && ((d.H & 1) || (d.W & 1))) return DH_EINVAL;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int epi0 = (a.Cout % 4 == 0) && (a.ldy % 4 == 0) && al16(a.y) && (a.res1 == nullptr || (a.ldr1 % 4 == 0 && al16(a.res1))) &&
                   (a.res2 == nullptr || (a.ldr2 % 4 == 0 && al16(a.res2))) &&
                   (a.post_scale == nullptr || (al16(a.post_scale) && al16(a.post_shift)));
  const int epi = epi_with_direct(a, epi0);
  const long long M = (long long)a.N * a.OH * a.OW;
  if (M > 0x7fffffffLL) return DH_EINVAL;
  // (all tilings of the family give the same bits: the group picks its own -- 128-row tiles once they fill the chip)
  const bool wide = ((M + 127) / 128) * ((a.Cout + 31) / 32) >= 256;
  if (g.nt == 256) return wide ? launch_group<4, 5, 256, 5, 10>(a, epi, d, g, s) : launch_group<2, 5, 256, 5, 10>(a, epi, d, g, s);
  return wide ? launch_group<4, 5, 64, 6, 12>(a, epi, d, g, s) : launch_group<2, 5, 64, 6, 12>(a, epi, d, g, s);
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

int gemm1x1_num_cfgs() { return 9; }

}  // namespace dh
