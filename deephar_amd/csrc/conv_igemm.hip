// Implicit-GEMM convolution for gfx950 on the fp32 matrix cores.
//
// Replaces what Keras Conv2D / the pointwise half of SeparableConv2D lower to in the reference
// (deephar/layers.py:66-80, 202-325), with the surrounding BatchNormalization / ReLU / add layers folded
// into the prologue and epilogue.  fp32 in, fp32 accumulate (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf
// chain, so numerics are plain IEEE fp32) -- the 1e-3 px parity bar rules out bf16/fp8.
//
// Tiling (wave = 64 lanes, 32x32 MFMA tiles):
//   workgroup = WM x WN waves, each wave owns TM x TN tiles of 32x32 -> BM = 32*WM*TM, BN = 32*WN*TN
//   K is walked in steps of BK = 32.  A (activations) is gathered global -> registers -> LDS with row
//   stride 36 floats so that the ds_read_b128 fragment reads are bank-conflict free; B (weights) is
//   host-packed as [K/4][N][4] so both its global->LDS copy and its fragment reads are 16-byte wide.
//   One LDS stage + register prefetch of the next K-step (loads are issued before the MFMA block and
//   written to LDS after it), two workgroups per CU so one block's MFMAs cover the other's barriers.
//   A b128 fragment read hands each lane 4 consecutive k; MFMA t consumes element t of both operands, so
//   the lane halves (k, k+4) pair up -- a permutation of the K order, which the sum does not care about.
//   blockIdx is remapped so that the N-tiles sharing one A tile run back to back on the same XCD (L2).
#include "conv_common.h"

namespace dh {

namespace {

constexpr int BK = 32;
constexpr int LDA = BK + 4;  // floats; 144 B rows keep ds_read_b128 conflict-free (odd multiple of 16 B)

template <int WM, int WN, int TM, int TN, bool VEC4, bool UP2>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv_igemm_kernel(const ConvArgs p, const int epi_vec) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int APASS = BM * 8 / NT;       // float4 A loads per thread per K-step
  constexpr int AROWS = NT / 8;            // rows covered per pass
  constexpr int BPASS = 8 * BN / NT;       // float4 B loads per thread per K-step
  static_assert(BM * 8 % NT == 0 && (8 * BN) % NT == 0, "tile/thread mismatch");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;             // [BM][LDA]
  float* sB = smem + BM * LDA;  // [8][BN][4]

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

  // ---- per-thread A-row bookkeeping (fixed over the K loop)
  const int a_col = (tid & 7) * 4;
  int a_pix[APASS], a_ih0[APASS], a_iw0[APASS];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int m = m0 + (tid >> 3) + ps * AROWS;
    if (m < M) {
      const int n = m / (p.OH * p.OW);
      const int rem = m - n * (p.OH * p.OW);
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      a_pix[ps] = n * p.H * p.W;
      a_ih0[ps] = oh * p.SH - p.PT;
      a_iw0[ps] = ow * p.SW - p.PL;
    } else {
      a_pix[ps] = 0;
      a_ih0[ps] = -(1 << 28);
      a_iw0[ps] = 0;
    }
  }

  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(p.w);
  const int nk = p.Kp / BK;
  const bool has_pre = p.pre_scale != nullptr;

  // B columns handled by this thread are fixed over the K loop
  int b_off[BPASS];
  unsigned bmask = 0;
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int idx = tid + q * NT;
    const int kq = idx / BN, j = idx - kq * BN;
    const bool ok = n0 + j < p.Np;
    b_off[q] = kq * p.Np + (ok ? n0 + j : 0);
    bmask |= (ok ? 1u : 0u) << q;
  }

  // Raw register staging: loads are issued unconditionally (clamped addresses) so that nothing waits on
  // them until store_tile(), i.e. until after the MFMA block of the current K-step.
  float4 ra[APASS], rb[BPASS];
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned amask = 0;   // VEC4: bit ps ; scalar path: bit (ps*4 + e)
  int lut_c[4] = {0, 0, 0, 0};   // scalar path with uint8 input: LUT row (channel * 256) of the staged elements

  // running decode of this thread's k index (k = kt*32 + a_col) into (kh, kw, c); VEC4 only
  int t_c = a_col % p.Cin, t_kw, t_kh;
  {
    const int tap = a_col / p.Cin;
    t_kh = tap / p.KW;
    t_kw = tap - t_kh * p.KW;
  }

  auto load_tile = [&](int kt) {
    amask = 0;
    if constexpr (VEC4) {
      const bool kvalid = kt * BK + a_col < p.K;
      if (has_pre) {
        psc = *reinterpret_cast<const float4*>(p.pre_scale + t_c);
        psh = *reinterpret_cast<const float4*>(p.pre_shift + t_c);
      }
#pragma unroll
      for (int ps = 0; ps < APASS; ++ps) {
        const int ih = a_ih0[ps] + t_kh, iw = a_iw0[ps] + t_kw;
        const bool ok = kvalid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const size_t off = ok ? (size_t)(a_pix[ps] + ih * p.W + iw) * p.ldx + t_c : 0;
        ra[ps] = *reinterpret_cast<const float4*>(p.x + off);
        amask |= (ok ? 1u : 0u) << ps;
      }
      // advance to the next K-step
      t_c += BK;
      while (t_c >= p.Cin) {
        t_c -= p.Cin;
        if (++t_kw == p.KW) { t_kw = 0; ++t_kh; }
      }
    } else {
      const int k0 = kt * BK + a_col;
      float sc[4], sh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + e;
        const int tap = k / p.Cin;
        const int cc = k - tap * p.Cin;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const bool kv = k < p.K;
        sc[e] = has_pre ? p.pre_scale[cc] : 1.f;
        sh[e] = has_pre ? p.pre_shift[cc] : 0.f;
        lut_c[e] = (kv ? cc : 0) * 256;
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps) {
          const int ih = a_ih0[ps] + kh, iw = a_iw0[ps] + kw;
          const bool ok = kv && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          const size_t off = ok ? (size_t)(a_pix[ps] + ih * p.W + iw) * p.ldx + cc : 0;
          // uint8 frames: stage the raw byte (exact in fp32); the LUT is applied in store_tile()
          (&ra[ps].x)[e] = p.x_u8 ? (float)reinterpret_cast<const unsigned char*>(p.x)[off] : p.x[off];
          amask |= (ok ? 1u : 0u) << (ps * 4 + e);
        }
      }
      psc = make_float4(sc[0], sc[1], sc[2], sc[3]);
      psh = make_float4(sh[0], sh[1], sh[2], sh[3]);
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) rb[q] = w4[(size_t)kt * 8 * p.Np + b_off[q]];
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      float4 v = ra[ps];
      if constexpr (!VEC4) {
        if (p.x_u8) {
          v.x = p.in_lut[lut_c[0] + (int)v.x]; v.y = p.in_lut[lut_c[1] + (int)v.y];
          v.z = p.in_lut[lut_c[2] + (int)v.z]; v.w = p.in_lut[lut_c[3] + (int)v.w];
        }
      }
      if (has_pre) {
        v.x = v.x * psc.x + psh.x; v.y = v.y * psc.y + psh.y;
        v.z = v.z * psc.z + psh.z; v.w = v.w * psc.w + psh.w;
      }
      if (p.pre_relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if constexpr (VEC4) {
        if (!((amask >> ps) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const unsigned mk = amask >> (ps * 4);
        if (!(mk & 1u)) v.x = 0.f;
        if (!(mk & 2u)) v.y = 0.f;
        if (!(mk & 4u)) v.z = 0.f;
        if (!(mk & 8u)) v.w = 0.f;
      }
      *reinterpret_cast<float4*>(&sA[((tid >> 3) + ps * AROWS) * LDA + a_col]) = v;
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      float4 v = rb[q];
      if (!((bmask >> q) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&sB[(tid + q * NT) * 4]) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_tile(0);
  store_tile();
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) load_tile(kt + 1);

#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const float4*>(&sA[((wm * TM + i) * 32 + li) * LDA + s * 8 + lh * 4]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const float4*>(&sB[((s * 2 + lh) * BN + (wn * TN + j) * 32 + li) * 4]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }

  EpiPrefetch<TM, TN> pre;   // (register-staged kernel: no spare VGPRs to hold a residual tile across the K loop)
  conv_epilogue<WM, WN, TM, TN, UP2, false>(p, acc, smem, m0, n0, M, epi_vec, pre);
}

struct Cfg { int wm, wn, tm, tn; };
constexpr Cfg kCfgs[] = {
    {2, 2, 2, 3},  // 0: 128 x 192
    {2, 2, 2, 2},  // 1: 128 x 128
    {4, 1, 1, 3},  // 2: 128 x  96
    {4, 1, 1, 2},  // 3: 128 x  64
    {4, 1, 1, 1},  // 4: 128 x  32
    {2, 1, 1, 3},  // 5:  64 x  96
    {2, 1, 1, 2},  // 6:  64 x  64
    {2, 1, 1, 1},  // 7:  64 x  32
    {1, 1, 1, 1},  // 8:  32 x  32
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

template <int WM, int WN, int TM, int TN>
int launch_cfg(const ConvArgs& a, bool vec4, int epi, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = ((M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  if (tiles <= 0 || tiles > 0x7fffffffLL) return DH_EINVAL;
  constexpr int kStage = BM * LDA + BK * BN, kEpi = WM * WN * 32 * (TN * 32 + 4);
  const size_t lds = (size_t)(kStage > kEpi ? kStage : kEpi) * sizeof(float);
  if (a.y_pool != nullptr && !conv_epilogue_pools_for<WM, TM, false>(a)) return DH_EUNSUPPORTED;
  if (a.up2) {
    if (!vec4) return DH_EUNSUPPORTED;
    if constexpr (TM * TN >= 6) return DH_EUNSUPPORTED;  // would spill; the dispatcher never asks for it
    else
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN, true, true>), dim3((unsigned)tiles), dim3(NT), lds, s, a, epi);
  } else if (vec4) {
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN, true, false>), dim3((unsigned)tiles), dim3(NT), lds, s, a, epi);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN, false, false>), dim3((unsigned)tiles), dim3(NT), lds, s, a, epi);
  }
  return check_launch();
}

}  // namespace

bool gemm1x1_eligible(const ConvArgs& a);
int launch_gemm1x1(const ConvArgs& a, int cfg, int epi, hipStream_t s);
int launch_gemm1x1_split(const ConvArgs& a, int cfg, int epi, hipStream_t s);
int gemm1x1_split_num_cfgs();
int gemm1x1_num_cfgs();
bool conv_is_skinny(const ConvArgs& a);
int launch_conv_splitk(const ConvArgs& a, hipStream_t s);
int launch_conv_halo(const ConvArgs& a, int cfg, int epi, hipStream_t s);
bool conv_stem_eligible(const ConvArgs& a);
int launch_conv_stem(const ConvArgs& a, hipStream_t s);

// cfg 0..8: general implicit-GEMM kernel; cfg 9..17: the same tile shapes on the LDS-DMA kernel
int conv_igemm_num_cfgs() { return kNumCfgs + gemm1x1_num_cfgs(); }

// Default tiling when the caller does not autotune (measured on MI355X, tools/bench_ops.py): four waves
// side by side along M with narrow per-wave tiles (more resident waves per SIMD) beat the 2x2-wave layouts;
// memory/epilogue-bound shapes (tiny K or tiny M) want the narrowest tile.
int conv_igemm_pick_cfg(int M, int Cout) {
  const int np = (Cout + 31) / 32 * 32;
  auto blocks = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((np + bn - 1) / bn); };
  int bn = 32;
  if (np % 96 == 0) bn = 96;
  else if (np % 64 == 0) bn = 64;
  if (bn == 96) return blocks(128, 96) >= 512 ? 2 : (blocks(128, 32) >= 512 ? 4 : 7);
  if (bn == 64) return blocks(128, 64) >= 512 ? 3 : (blocks(128, 32) >= 512 ? 4 : 7);
  if (blocks(128, 32) >= 512) return 4;
  if (blocks(64, 32) >= 256) return 7;
  return 8;
}

int launch_conv_igemm(const ConvArgs& a, int cfg, hipStream_t s) {
  if (a.N <= 0 || a.Cin <= 0 || a.Cout <= 0 || a.OH <= 0 || a.OW <= 0) return DH_EINVAL;
  if (a.Kp % BK != 0 || a.Np % 32 != 0 || a.Kp < a.K || a.Np < a.Cout || a.K != a.KH * a.KW * a.Cin)
    return DH_EINVAL;
  if ((long long)a.N * a.H * a.W > 0x7fffffffLL || (long long)a.N * a.OH * a.OW * (a.up2 ? 4 : 1) > 0x7fffffffLL)
    return DH_EINVAL;
  if (a.x_u8 && a.in_lut == nullptr) return DH_EINVAL;
  // tiny output, long reduction: the in-work-group split-K kernel, whatever tiling was asked for (shape rule: the
  // result bits of a layer must not depend on a timing-based choice)
  if (a.y_pool != nullptr) {             // pooled second output: one image row per wave (pairs of waves pool), or [r06] two /
                                         // four whole image rows per wave (OW = 16 / 8: the wave pools alone)
    const bool ok = (a.OW == 32 || ((a.OW == 16 || a.OW == 8) && (a.OH * a.OW) % 32 == 0)) && a.OH % 2 == 0 && !a.up2 && !a.x_u8 && a.ldyp % 4 == 0 && a.Cout % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(a.y_pool) & 15) == 0 && a.ldy % 4 == 0 && a.w_split != 2 &&
                    (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 && !conv_is_skinny(a);
    if (!ok) return DH_EUNSUPPORTED;
  }
  if (a.x_resample && !conv_is_skinny(a)) return DH_EUNSUPPORTED;   // (resampling on load: the skinny-conv kernel only)
  if (conv_is_skinny(a)) {
    if (a.res2_down) return DH_EUNSUPPORTED;          // (the split-K kernel has its own, simpler epilogue)
    return cfg < kNumCfgs + gemm1x1_num_cfgs() ? launch_conv_splitk(a, s) : DH_EINVAL;
  }
  // first layer (3 input channels, stride 2): its own kernel, by a rule on the layer's geometry like the split-K layers --
  // it pairs the k values of an MFMA differently from the tap-major kernels, so its last bits differ from theirs and a
  // layer must never move between the two on a timing-based choice
  if (conv_stem_eligible(a)) return cfg < kNumCfgs + gemm1x1_num_cfgs() ? launch_conv_stem(a, s) : DH_EINVAL;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int epi = (a.Cout % 4 == 0) && (a.ldy % 4 == 0) && al16(a.y) &&
                  (a.res1 == nullptr || (a.ldr1 % 4 == 0 && al16(a.res1))) &&
                  (a.res2 == nullptr || (a.ldr2 % 4 == 0 && al16(a.res2))) &&
                  (a.post_scale == nullptr || (al16(a.post_scale) && al16(a.post_shift)));
  if (a.y_pool != nullptr && !epi) return DH_EUNSUPPORTED;
  if (a.w_split == 2) return launch_conv_halo(a, cfg, epi_with_direct(a, epi, false), s);   // chunk-major fp32 packing: the halo-resident kernel only
  if (cfg < 0 && a.y_pool != nullptr) {
    cfg = conv_igemm_pick_cfg(a.N * a.OH * a.OW, a.Cout);
    if (cfg == 8 && a.OW == 32) cfg = 7;  // 32 x 32 has no wave pair
    if (cfg == 0 || cfg == 1) cfg = 2;    // 64-row waves do not pool
    if (!a.w_split && gemm1x1_eligible(a)) cfg += kNumCfgs;
  }
  if (cfg < 0)
    cfg = conv_igemm_pick_cfg(a.N * a.OH * a.OW, a.Cout) + (!a.w_split && !a.x_u8 && gemm1x1_eligible(a) ? kNumCfgs : 0);
  if (!a.w_split && cfg >= kNumCfgs + gemm1x1_num_cfgs()) return DH_EINVAL;
  if (a.x_u8 && cfg >= kNumCfgs) return DH_EUNSUPPORTED;
  if (a.w_split) {                       // split-bf16 weights: only the LDS-DMA GEMM family reads that packing
    if (a.x_u8 || !gemm1x1_eligible(a)) return DH_EUNSUPPORTED;
    if (cfg >= gemm1x1_split_num_cfgs()) return DH_EINVAL;
    if (a.up2 && cfg == 0) cfg = 2;
    return launch_gemm1x1_split(a, cfg, epi, s);
  }
  if (cfg >= kNumCfgs) {
    cfg -= kNumCfgs;
    if (a.up2 && cfg == 0) cfg = 2;
    return launch_gemm1x1(a, cfg, epi_with_direct(a, epi), s);   // interior tiles: epilogue straight from the accumulators
  }
  if (a.up2 && cfg == 0) cfg = 2;  // 128x192 + fused up-sampling epilogue exceeds the register budget
  const bool vec4 = !a.x_u8 && (a.Cin % 4 == 0) && (a.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) &&
                    (a.pre_scale == nullptr || (((reinterpret_cast<uintptr_t>(a.pre_scale) |
                                                  reinterpret_cast<uintptr_t>(a.pre_shift)) & 15) == 0));
  switch (cfg) {
    case 0: return launch_cfg<2, 2, 2, 3>(a, vec4, epi, s);
    case 1: return launch_cfg<2, 2, 2, 2>(a, vec4, epi, s);
    case 2: return launch_cfg<4, 1, 1, 3>(a, vec4, epi, s);
    case 3: return launch_cfg<4, 1, 1, 2>(a, vec4, epi, s);
    case 4: return launch_cfg<4, 1, 1, 1>(a, vec4, epi, s);
    case 5: return launch_cfg<2, 1, 1, 3>(a, vec4, epi, s);
    case 6: return launch_cfg<2, 1, 1, 2>(a, vec4, epi, s);
    case 7: return launch_cfg<2, 1, 1, 1>(a, vec4, epi, s);
    case 8: return launch_cfg<1, 1, 1, 1>(a, vec4, epi, s);
  }
  return DH_EINVAL;
}

}  // namespace dh
