// Fused SeparableConv2D for gfx950: depthwise K x K (stride 1, TF-SAME) computed on the fly as the A operand of
// the pointwise MFMA GEMM -- the depthwise result never exists in HBM.
//
// Replaces keras SeparableConv2D as the reference uses it: layers.sepconv2d (deephar/layers.py:74-80),
// separable_act_conv_bn (:288-301: ReLU -> depthwise -> pointwise -> BN), reception._sepconv_residual
// (models/reception.py:43-59: + residual add), the up-sampling merge of the hourglass (reception.py:122-127) and
// common.residual_unit's depthwise branch (models/common.py:47-48).  The unfused pair (spatial.hip
// dwconv_*_kernel + gemm1x1.hip) writes and re-reads the [M, Cin] depthwise tensor: 151 MB each way per
// separable conv at 32x32x576, batch 64.
//
// Work-group = WM waves, output tile = (BM = 32*WM consecutive pixels = R whole image rows) x (BN = 32*TN output
// channels); wave w owns rows [32w, 32w+32) x all BN columns (TN accumulator tiles of 32x32).  K (= Cin) is walked in
// steps of 16 channels:
//   1. LDS-DMA (buffer_load_dwordx4 ... lds) brings the halo'd input tile [(R+KS-1) x (W+KS-1)] x 16 channels, the
//      KS*KS x 16 depthwise taps and the [16 x BN] slice of the packed pointwise weight; padding columns / rows
//      outside the batch are out-of-range buffer offsets, which load zeros.
//   2. depthwise stage (VALU): thread = (channel quad, column, pair of rows); 6 x 5 LDS reads feed 2 x 25 fmaf in
//      the SAME (kh, kw) order as dwconv_lds_kernel / dwconv_kernel, ReLU applied on the way in; rows that belong
//      to the neighbouring frame are redirected to an LDS block of zeros.  Result -> LDS A tile [BM][16], 16-byte
//      slots XOR-swizzled with (row >> 2) & 3 so the MFMA fragment reads are bank-conflict free.
//   3. MFMA stage: v_mfma_f32_32x32x2_f32 over the A tile and the double-buffered B slice; the DMA of the next
//      K-step flies during it.  K is summed in exactly the order of gemm1x1_kernel (k-quad pairs (0,1),(2,3),...),
//      so the fused kernel is BIT-IDENTICAL to the unfused pair -- asserted by tests/test_gpu_ops.py.
// Two work-groups share a CU (<= 80 KB LDS, <= 256 VGPR+AGPR): one's depthwise stage / barriers / epilogue run
// under the other's MFMA stage.  With Cout > BN the depthwise work is repeated per N tile (2x at 576 -> 576): it is
// 25/BN of the MFMA work and runs on the VALU pipe, which the matrix pipe does not use.
#include "conv_common.h"

namespace dh {
namespace {

constexpr int SK = 16;                 // channels per K-step
constexpr int ZERO_FLOATS = 128;       // LDS block of zeros (512 B): source of out-of-frame rows

typedef __attribute__((address_space(3))) void* lptr_t;

// buffer_load_dwordx4 ... offen lds: 16 bytes per lane from (descriptor base + voff + soff) to LDS (wave-uniform `dst`
// + lane * 16); an out-of-range voff loads zeros.  The builtin only exists for the gfx950 pass: hipcc's host pass
// silently drops a kernel TEMPLATE whose body names it (no host stub is emitted), hence the guard.
template <typename RSRC>
__device__ __forceinline__ void dma16(RSRC rs, float* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, voff, soff, 0, 0);
#endif
}

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
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
// ReLU as ONE integer instruction per element: non-negative floats order like non-negative ints, anything with the sign
// bit set is a negative int, so max_i32(bits, 0) == bits of relu(v) (-0 -> +0).  `lo` = 0 for ReLU, INT_MIN for the
// identity: branch-free on a kernel argument, and no canonicalising v_max_f32 pair as fmaxf would emit.
__device__ __forceinline__ float relu1(float v, int lo) {
  const int b = __float_as_int(v);
  return __int_as_float(b > lo ? b : lo);
}
__device__ __forceinline__ float4 relu4(float4 v, int lo) {
  return make_float4(relu1(v.x, lo), relu1(v.y, lo), relu1(v.z, lo), relu1(v.w, lo));
}

template <int N, int OFF0, int STRIDE>
struct BRead {
  template <int J = 0>
  static __device__ __forceinline__ void run(unsigned addr, float4 (&f)[N]) {
    if constexpr (J < N) {
      f[J] = lds_rd<OFF0 + J * STRIDE>(addr);
      run<J + 1>(addr, f);
    }
  }
};

__device__ __forceinline__ void mfma4(const float4& a, const float4& b, f32x16& c16) {
  c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c16, 0, 0, 0);
  c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c16, 0, 0, 0);
  c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c16, 0, 0, 0);
  c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c16, 0, 0, 0);
}

// Step I of the 2*NG (sub-step, group) sequence: wait for this step's fragments, put the next step's reads in flight,
// run this step's 12 MFMAs.  Straight-line by construction (an asm ds_read must be awaited in the block that issued it).
template <int TN, int BN, int I, int GT>
__device__ __forceinline__ void mfma_step(unsigned a_addr1, unsigned bo, float4 (&fa)[2], float4 (&fb)[2][GT],
                                          f32x16 (&acc)[TN]) {
  constexpr int NG = TN / GT;
  if constexpr (I < 2 * NG) {
    constexpr int S = I / NG, G = I % NG;
    lgkm_wait();
    if constexpr (I + 1 < 2 * NG) {
      constexpr int S1 = (I + 1) / NG, G1 = (I + 1) % NG;
      if constexpr (S1 != S) fa[1] = lds_rd<0>(a_addr1);
      BRead<GT, S1 * 2 * BN * 16 + G1 * GT * 512, 512>::run(bo, fb[(I + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j) mfma4(fa[S], fb[I & 1][j], acc[G * GT + j]);
    mfma_step<TN, BN, I + 1, GT>(a_addr1, bo, fa, fb, acc);
  }
}

// p: the POINTWISE convolution's arguments (KH = KW = 1; x / H / W / Cin / ldx describe the input of the depthwise
// stage, which has the same spatial size); dw: [KS*KS][Cin] depthwise taps.
template <int WM, int TN, int KS, bool UP2>
__global__ __launch_bounds__(WM * 64, 2) void sepconv_fused_kernel(const ConvArgs p, const float* __restrict__ dw,
                                                                  const int pad, const int epi_vec) {
  constexpr int NT = WM * 64;
  constexpr int BM = WM * 32;
  constexpr int BN = TN * 32;
  constexpr int SLOTS_B = (4 * BN + NT - 1) / NT;
  constexpr int MAXH = WM >= 4 ? 5 : 7;                       // halo transfers per thread (W <= 32: see launch_sep)
  static_assert(KS * KS * 4 <= NT, "one tap transfer per thread");
  constexpr int GT = 3, NG = TN / GT;                         // B fragments are fetched in groups of 3 tiles
  static_assert(TN % GT == 0, "TN must be a multiple of 3");
  constexpr int BSTAGE = SLOTS_B * NT * 4;                    // floats per B stage

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  const int W = p.W, H = p.H;
  const int WP = W + KS - 1;
  const int R = BM / W;                                       // image rows per tile
  const int rows_total = p.N * H;
  const int M = rows_total * W;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n;
  const int m0 = tm * BM;
  const int n0 = (tile % tiles_n) * BN;
  const int row0 = tm * R;

  float* sZero = smem;
  float* sHalo = smem + ZERO_FLOATS;                          // [MAXH * NT] float4, pixel-major, quad-minor
  float* sDww = sHalo + MAXH * NT * 4;                        // [KS*KS][4 quads] float4 (NT transfers)
  float* sA = sDww + NT * 4;                                  // [BM][16]
  float* sB = sA + BM * SK;                                   // 2 x [4 k-quads][BN] float4

  if (tid < ZERO_FLOATS / 4) reinterpret_cast<float4*>(sZero)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- DMA sources: buffer_load_dwordx4 ... lds through three buffer descriptors (wave-uniform base + size), per-lane
  // 32-bit byte offsets fixed over K, the channel / K-step offset in the scalar offset.  An offset beyond the buffer
  // (OOB) makes the load return zeros: that is the zero padding of the halo and the filler of unused transfers.
  constexpr unsigned OOB = 0xfffffff0u;
  const int HP4 = (R + KS - 1) * WP * 4;
  unsigned h_off[MAXH];
#pragma unroll
  for (int s = 0; s < MAXH; ++s) {
    const int t = s * NT + tid;
    const int px = t >> 2, q = t & 3;
    const int rr = px / WP, cc = px - rr * WP;
    const int g = row0 - pad + rr, iw = cc - pad;
    const bool ok = t < HP4 && (unsigned)g < (unsigned)rows_total && (unsigned)iw < (unsigned)W;
    h_off[s] = ok ? (unsigned)((g * W + iw) * p.ldx + q * 4) * 4u : OOB;
  }
  unsigned b_off[SLOTS_B];
#pragma unroll
  for (int s = 0; s < SLOTS_B; ++s) {
    const int t = s * NT + tid;
    const int kq = t / BN;
    const int j = t - kq * BN;
    b_off[s] = kq < 4 && n0 + j < p.Np ? (unsigned)((kq * p.Np + n0 + j) * 4) * 4u : OOB;
  }
  const unsigned w_off = tid < KS * KS * 4 ? (unsigned)((tid >> 2) * p.Cin + (tid & 3) * 4) * 4u : OOB;
  const int b_step = 4 * p.Np * 16;                          // bytes per K-step in the packed pointwise weight
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  const int x_bytes = (int)((unsigned)rows_total * W * p.ldx * 4u), w_bytes = (int)((unsigned)p.Kp * p.Np * 4u);
  auto issue = [&](int kt, int stage) {
    // (descriptors are built here, from kernel arguments only: four SGPRs each, hoisted out of the K loop)
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, x_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, w_bytes, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dw), 0, KS * KS * p.Cin * 4, 0x00020000);
    const int cb = kt * SK * 4;                               // byte offset of this K-step's channels
#pragma unroll
    for (int s = 0; s < MAXH; ++s)
      dma16(rs_x, sHalo + (s * NT + wave_u * 64) * 4, h_off[s], cb);
    dma16(rs_d, sDww + wave_u * 64 * 4, w_off, cb);
    float* dstB = sB + stage * BSTAGE;
#pragma unroll
    for (int s = 0; s < SLOTS_B; ++s)
      dma16(rs_w, dstB + (s * NT + wave_u * 64) * 4, b_off[s], kt * b_step);
  };

  // ---- depthwise stage bookkeeping: thread = (quad q, column c, row pair rp)
  const int q = tid & 3, pp = tid >> 2;
  const int rp = pp / W, c = pp - rp * W;
  const int h0 = (row0 + 2 * rp) % H;                          // image row of the upper output of the pair
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const float4* rowp[KS + 1];
#pragma unroll
  for (int i = 0; i <= KS; ++i) {
    const bool ok = (unsigned)(h0 + i - pad) < (unsigned)H;
    rowp[i] = ok ? reinterpret_cast<const float4*>(sHalo) + ((2 * rp + i) * WP + c) * 4 + q
                 : reinterpret_cast<const float4*>(sZero) + q;
  }
  const int ma = 2 * rp * W + c, mb = ma + W;                   // tile rows of the two outputs
  float4* a_dst0 = reinterpret_cast<float4*>(sA) + ma * 4 + (q ^ ((ma >> 2) & 3));
  float4* a_dst1 = reinterpret_cast<float4*>(sA) + mb * 4 + (q ^ ((mb >> 2) & 3));
  const float4* wq = reinterpret_cast<const float4*>(sDww) + q;
  const int relu_lo = p.pre_relu != 0 ? 0 : (int)0x80000000;

  // ---- MFMA stage bookkeeping
  const int arow = wave * 32 + li;
  const unsigned a_sw = (unsigned)((arow >> 2) & 3);
  const unsigned a_base = lds0 + (unsigned)((ZERO_FLOATS + MAXH * NT * 4 + NT * 4) * 4) + (unsigned)(arow * 64);
  const unsigned a_addr0 = a_base + (((0u + lh) ^ a_sw) << 4);
  const unsigned a_addr1 = a_base + (((2u + lh) ^ a_sw) << 4);
  const unsigned b_base = a_base - (unsigned)(arow * 64) + (unsigned)(BM * SK * 4) + (unsigned)((lh * BN + li) * 16);

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nk = (p.Cin + SK - 1) / SK;
  issue(0, 0);

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // halo / taps / B slice of this K-step have landed (this wave's DMA), every wave is past MFMA(kt-1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- depthwise stage: 2 outputs (rows 2rp, 2rp+1 of the tile, column c) x 4 channels.  Iteration kh reads ONE new
    // input row and ONE row of taps: the upper output takes input row kh, the lower one input row kh+1, both with
    // taps w[kh][*] -- per output the sum still runs over (kh, kw) in lexicographic order.  The scheduling fences keep
    // the compiler from hoisting all 30 + 25 LDS reads to the top (the accumulators leave ~100 registers).
    {
      float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
      float4 vc[KS], vn[KS], wr[KS];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) vc[kw] = rowp[0][kw * 4];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) vc[kw] = relu4(vc[kw], relu_lo);
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          vn[kw] = rowp[kh + 1][kw * 4];
          wr[kw] = wq[(kh * KS + kw) * 4];
        }
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) vn[kw] = relu4(vn[kw], relu_lo);
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          o0 = fma4(vc[kw], wr[kw], o0);
          o1 = fma4(vn[kw], wr[kw], o1);
        }
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) vc[kw] = vn[kw];
        __builtin_amdgcn_sched_barrier(0);
      }
      *a_dst0 = o0;
      *a_dst1 = o1;
    }
    __syncthreads();                       // A tile complete; halo and taps may be overwritten

    // ---- MFMA stage: 2 sub-steps (k-quad pairs (0,1), (2,3)) x NG groups of 3 accumulator tiles; the fragments of
    // group g+1 are in flight (asm ds_read, hand-placed waits) while the 12 MFMAs of group g run.  The very first
    // reads have no MFMAs to hide behind: the next K-step's DMA issue (address VALU + ~11 loads) goes there.
    const unsigned bo = b_base + (unsigned)(cur * BSTAGE * 4);
    float4 fa[2], fb[2][GT];
    fa[0] = lds_rd<0>(a_addr0);
    BRead<GT, 0, 512>::run(bo, fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    mfma_step<TN, BN, 0, GT>(a_addr1, bo, fa, fb, acc);
  }

  // ---- epilogue: the shared conv epilogue in chunks of 3 accumulator tiles (its LDS slab is 32 x 100 floats per wave)
  constexpr int CH = TN % 3 == 0 ? 3 : (TN % 2 == 0 ? 2 : 1);
#pragma unroll
  for (int c0 = 0; c0 < TN; c0 += CH) {
    f32x16 part[1][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) part[0][j] = acc[c0 + j];
    EpiPrefetch<1, CH> none;
    conv_epilogue<WM, 1, 1, CH, UP2, false>(p, part, smem, m0, n0 + c0 * 32, M, epi_vec, none);
  }
}

struct SepCfg { int wm, tn; };
constexpr SepCfg kSepCfgs[] = {{4, 9}, {4, 3}, {2, 9}, {2, 3}};
constexpr int kNumSepCfgs = sizeof(kSepCfgs) / sizeof(kSepCfgs[0]);

template <int WM, int TN, int KS, bool UP2>
int launch_sep(const ConvArgs& a, const float* dw, int epi, hipStream_t s) {
  constexpr int NT = WM * 64, BM = WM * 32, BN = TN * 32;
  constexpr int SLOTS_B = (4 * BN + NT - 1) / NT;
  constexpr int MAXH = WM >= 4 ? 5 : 7;
  constexpr int CH = TN % 3 == 0 ? 3 : (TN % 2 == 0 ? 2 : 1);
  if (BM % (2 * a.W) != 0 || (a.H & 1)) return DH_EUNSUPPORTED;        // whole row pairs per tile, pairs inside a frame
  const int R = BM / a.W;
  const int hp4 = (R + KS - 1) * (a.W + KS - 1) * 4;
  const int nslot_h = (hp4 + NT - 1) / NT;
  if (nslot_h > MAXH) return DH_EUNSUPPORTED;
  const size_t main_f = (size_t)ZERO_FLOATS + (size_t)MAXH * NT * 4 + NT * 4 + (size_t)BM * SK +
                        2 * (size_t)SLOTS_B * NT * 4;
  const size_t epi_f = (size_t)WM * 32 * (CH * 32 + 4);
  const size_t lds = (main_f > epi_f ? main_f : epi_f) * sizeof(float);
  if (lds > 160 * 1024) return DH_EUNSUPPORTED;
  const long long rows = (long long)a.N * a.H;
  const long long tiles = ((rows + R - 1) / R) * ((a.Cout + BN - 1) / BN);
  if (tiles <= 0 || tiles > 0x7fffffffLL || rows * a.W * (long long)a.ldx * 4 > 0xf0000000LL) return DH_EINVAL;
  auto kern = sepconv_fused_kernel<WM, TN, KS, UP2>;
  if (lds > 64 * 1024) {
    static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024), true);
    (void)once;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), lds, s, a, dw, (KS - 1) / 2, epi);
  return check_launch();
}

template <int WM, int TN>
int launch_sep_cfg(const ConvArgs& a, const float* dw, int ks, int epi, hipStream_t s) {
  if (a.up2) {
    if (ks == 5) return launch_sep<WM, TN, 5, true>(a, dw, epi, s);
    return launch_sep<WM, TN, 3, true>(a, dw, epi, s);
  }
  if (ks == 5) return launch_sep<WM, TN, 5, false>(a, dw, epi, s);
  return launch_sep<WM, TN, 3, false>(a, dw, epi, s);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int sepconv_num_cfgs() { return kNumSepCfgs; }

// Heuristic when the caller does not autotune: widest N tile that still gives every CU two work-groups.
int sepconv_pick_cfg(int M, int Cout) {
  for (int i = 0; i < kNumSepCfgs; ++i) {
    const long long tiles = ((long long)(M + kSepCfgs[i].wm * 32 - 1) / (kSepCfgs[i].wm * 32)) *
                            ((Cout + kSepCfgs[i].tn * 32 - 1) / (kSepCfgs[i].tn * 32));
    if (tiles >= 512) return i;
  }
  return kNumSepCfgs - 1;
}

int launch_sepconv_fused(const ConvArgs& a, const float* dw, int dkh, int dkw, int dpt, int dpl, int cfg,
                         hipStream_t s) {
  if (dw == nullptr || a.N <= 0 || a.Cin <= 0 || a.Cout <= 0) return DH_EINVAL;
  if (a.y_pool != nullptr) return DH_EUNSUPPORTED;     // (the pooled second output lives in the shared LDS-slab epilogue)
  const int ks = dkh;
  const bool shape = dkh == dkw && (ks == 5 || ks == 3) && dpt == (ks - 1) / 2 && dpl == (ks - 1) / 2 &&
                     a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0 && a.H == a.OH &&
                     a.W == a.OW && a.Cin % SK == 0 && a.K == a.Cin && a.pre_scale == nullptr && !a.x_u8;
  const bool aligned = a.ldx % 4 == 0 && al16(a.x) && al16(a.w) && al16(dw);
  if (!shape || !aligned) return DH_EUNSUPPORTED;
  const int M = a.N * a.OH * a.OW;
  if (cfg < 0) cfg = sepconv_pick_cfg(M, a.Cout);
  // 16-byte epilogue (as in conv_igemm.hip): every pointer / ld the vector path touches is a multiple of 4 floats
  const bool vec = a.Cout % 4 == 0 && a.ldy % 4 == 0 && al16(a.y) &&
                   (a.post_scale == nullptr || (al16(a.post_scale) && al16(a.post_shift))) &&
                   (a.res1 == nullptr || (a.ldr1 % 4 == 0 && al16(a.res1))) &&
                   (a.res2 == nullptr || (a.ldr2 % 4 == 0 && al16(a.res2)));
  const int epi = vec ? 1 : 0;
  switch (cfg) {
    case 0: return launch_sep_cfg<4, 9>(a, dw, ks, epi, s);
    case 1: return launch_sep_cfg<4, 3>(a, dw, ks, epi, s);
    case 2: return launch_sep_cfg<2, 9>(a, dw, ks, epi, s);
    case 3: return launch_sep_cfg<2, 3>(a, dw, ks, epi, s);
  }
  return DH_EINVAL;
}

}  // namespace dh
