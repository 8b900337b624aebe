// SeparableConv2D (+ ReLU prologue, BN / residual epilogue) in ONE launch on 8 x 8 maps -- the three `_sepconv_residual`
// units of the lowest hourglass level (reference deephar/models/reception.py:43-59,112-118: relu -> depthwise 5x5 ->
// pointwise 288 -> 288 -> BN -> + x).  As two launches (dwconv_lds_kernel + gemm1x1_kernel) such a unit takes 12.5 + 15.5 us
// at batch 64 for 0.68 GFLOP of matrix work (4.3 us at the fp32 MFMA rate): both launches are latency chains -- launch,
// first round trip, nine barrier-separated K-steps, epilogue -- not bandwidth or issue bound (profiles/r03_midsize_gemm_
// ablation.md).  A frame of this level is 64 pixels x 288 channels = 74 KB: it FITS IN LDS, and so does the depthwise
// result.  Work-group = (frame, 96 output channels), six waves:
//   1. the frame's input, through the ReLU, goes to LDS in one round of loads (every load in flight at once);
//   2. the depthwise stage runs entirely out of LDS -- thread = (channel quad, two output rows): same tap order and fmaf
//      chain as dwconv_lds_kernel, rows outside the map contribute through zeroed weights, columns outside it are skipped
//      at compile time (adding 0 * w changes no bit) -- and writes the GEMM's A operand [64][Cin] back to LDS;
//   3. the pointwise GEMM has its whole A operand resident: no barrier inside the K loop; every lane fetches its B
//      fragments (host-packed [K/4][N][4]: a float4 IS four consecutive k of one column) straight from global memory /
//      L2, one K-step ahead; same fragment mapping and k order as gemm1x1_kernel / conv_igemm_kernel: same bits;
//   4. the shared epilogue (conv_common.h), direct path: BN, residual (the unit's input, read from global), store.
// The three column slices of a frame repeat stages 1-2 (0.06 GFLOP of FMAs per launch in total): cheaper than a second
// launch.  dh_conv2d_f32 takes this path when dh_conv_args.dw_w is set (include/deephar_hip.h); the planner sets it by a
// rule on the layer's geometry (engine/planner.py: sepconv8_rule = sepconv8_eligible below).
#include "conv_common.h"

namespace dh {
namespace {

constexpr int kPix = 64;            // 8 x 8
constexpr int kSide = 8;
constexpr int kNT = 384;            // six waves: 2 (row blocks of 32 pixels) x 3 (column tiles of 32)
constexpr int kMaxC = 288;          // 64 x C (input) + 64 x (C + 4) (depthwise result) floats of LDS: 148.5 KB at C = 288

__device__ __forceinline__ float4 sc_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 sc_fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float sc_relu1(float v) {     // one integer max per element (see gemm1x1.hip: relu1)
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}

template <int KS, bool RELU>
__global__ __launch_bounds__(kNT, 1) void sepconv8_kernel(const ConvArgs p, const int epi_vec) {
  constexpr int PD = (KS - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = p.Cin, c4 = C >> 2;
  const int PA = C + 4;                                  // A pitch: 16-byte aligned rows, bank-staggered like LDA = BK + 4
  float4* xs4 = reinterpret_cast<float4*>(smem);         // [64][c4]  relu(x) of the frame
  float* as = smem + kPix * C;                           // [64][PA]  depthwise result = the GEMM's A operand

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nsl = p.Cout / 96;
  const int frame = blockIdx.x / nsl, slice = blockIdx.x - frame * nsl;
  const int n0 = slice * 96;
  const int M = p.N * kPix, m0 = frame * kPix;
  const unsigned magic = (1u << 20) / (unsigned)c4 + 1u;   // i / c4 for i < 2^20 / c4 (here i < 64 * 72)

  // ---- 1. frame -> LDS (all loads of a thread in flight before the first store)
  {
    const float* xf = p.x + (size_t)frame * kPix * p.ldx;
    constexpr int NI = (kPix * (kMaxC / 4) + kNT - 1) / kNT;    // 12
    float4 v[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * kNT;
      const int px = (int)(((unsigned)i * magic) >> 20), q = i - px * c4;
      v[it] = i < kPix * c4 ? sc_ld4(xf + (size_t)px * p.ldx + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * kNT;
      float4 t = v[it];
      if constexpr (RELU) t = make_float4(sc_relu1(t.x), sc_relu1(t.y), sc_relu1(t.z), sc_relu1(t.w));
      if (i < kPix * c4) xs4[i] = t;
    }
  }
  __syncthreads();

  // ---- 2. depthwise KS x KS, stride 1, zero padding PD: thread = (channel quad q, output rows 2 sg, 2 sg + 1)
  if (tid < 4 * c4) {
    const int sg = (int)(((unsigned)tid * magic) >> 20), q = tid - sg * c4;
    const int r0 = sg * 2;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[2][2][4];                                 // [output row][column half][pixel]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[a][b][o] = zero;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
      float4 wv[KS];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) wv[kw] = sc_ld4(p.dw_w + (size_t)(kh * KS + kw) * C + q * 4);
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int ir = r0 + rr + kh - PD;
        const bool rok = (unsigned)ir < (unsigned)kSide;
        const float4* src = xs4 + (size_t)((rok ? ir : 0) * kSide) * c4 + q;
        float4 we[KS];                                   // a row outside the map: zero weights (0 * finite adds nothing)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) we[kw] = rok ? wv[kw] : zero;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int j = 0; j < 4 + KS - 1; ++j) {
            const int iw = half * 4 - PD + j;            // compile-time: columns outside the map are skipped
            if (iw < 0 || iw >= kSide) continue;
            const float4 v = src[iw * c4];
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
              const int o = j - kw;
              if (o >= 0 && o < 4) acc[rr][half][o] = sc_fma4(v, we[kw], acc[rr][half][o]);
            }
          }
      }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int o = 0; o < 4; ++o)
          *reinterpret_cast<float4*>(&as[((r0 + rr) * kSide + half * 4 + o) * PA + q * 4]) = acc[rr][half][o];
  }
  // ---- 3. pointwise GEMM: wave (wm, wn) owns rows wm * 32 .. + 31, columns n0 + wn * 32 .. + 31.  The B fragments of
  // THREE K-steps are in flight (a K-step is 16 MFMAs = 0.4 us per wave, an L2 round trip is > 1 us); the first three
  // are issued before the barrier that publishes the depthwise result.
  const int wm = wave / 3, wn = wave - wm * 3;
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(p.w);
  const int col = n0 + wn * 32 + li;
  const int nk = p.Kp / 32;
  auto load_b = [&](float4 (&b)[4], int kt) {
    const int kc = kt < nk ? kt : nk - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = w4[(size_t)(kc * 8 + s * 2 + lh) * p.Np + col];
  };
  float4 b0[4], b1[4], b2[4];
  load_b(b0, 0);
  load_b(b1, 1);
  load_b(b2, 2);
  __syncthreads();

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  const float* arow = as + (wm * 32 + li) * PA + lh * 4;
  auto kstep = [&](const float4 (&b)[4], int kt) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(arow + kt * 32 + s * 8);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[s].x, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[s].y, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[s].z, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[s].w, acc[0][0], 0, 0, 0);
    }
  };
  EpiPrefetch<1, 1> pre;
  for (int kt = 0; kt < nk; kt += 3) {
    if (kt + 3 >= nk) pre.template issue<2, 3>(p, m0, n0, M, epi_vec);   // residual / BN values land during the last steps
    kstep(b0, kt);
    if (kt + 3 < nk) load_b(b0, kt + 3);
    if (kt + 1 < nk) {
      kstep(b1, kt + 1);
      if (kt + 4 < nk) load_b(b1, kt + 4);
    }
    if (kt + 2 < nk) {
      kstep(b2, kt + 2);
      if (kt + 5 < nk) load_b(b2, kt + 5);
    }
  }
  conv_epilogue<2, 3, 1, 1, false, true>(p, acc, smem, m0, n0, M, epi_vec, pre);
}

template <int KS>
int launch_sepconv8_ks(const ConvArgs& a, int epi, hipStream_t s) {
  const size_t lds = (size_t)(kPix * a.Cin + kPix * (a.Cin + 4)) * sizeof(float);
  const size_t slab = (size_t)6 * 32 * (32 + 4) * sizeof(float);            // staged epilogue (non-direct tiles)
  const long long blocks = (long long)a.N * (a.Cout / 96);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DH_EINVAL;
  auto kern = a.pre_relu ? sepconv8_kernel<KS, true> : sepconv8_kernel<KS, false>;
  static LdsLimit lim_t, lim_f;
  (a.pre_relu ? lim_t : lim_f).raise((const void*)kern, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kNT), lds > slab ? lds : slab, s, a, epi);
  return check_launch();
}

}  // namespace

// The layers the one-launch SeparableConv2D takes (mirrored by engine/planner.py: sepconv8_rule): 8 x 8 maps, a
// pointwise convolution behind a depthwise 3x3 / 5x5 'same' convolution, 32 <= Cin <= 288 in multiples of 32, Cout in
// multiples of 96, fp32 weights, ReLU (no BatchNormalization) prologue, 16-byte aligned views.
bool sepconv8_eligible(const ConvArgs& a) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool geom = a.dw_w != nullptr && a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PT == 0 && a.PL == 0 &&
                    a.H == kSide && a.W == kSide && a.OH == kSide && a.OW == kSide && a.Cin % 32 == 0 && a.Cin >= 32 &&
                    a.Cin <= kMaxC && a.Cout % 96 == 0 && a.K == a.Cin && a.Kp == a.Cin &&
                    a.dw_kh == a.dw_kw && (a.dw_kh == 3 || a.dw_kh == 5) && a.dw_pt == (a.dw_kh - 1) / 2 &&
                    a.dw_pl == (a.dw_kw - 1) / 2;
  const bool plain = a.pre_scale == nullptr && !a.up2 && !a.x_u8 && a.w_split == 0 && a.y_pool == nullptr;
  const bool aligned = a.ldx % 4 == 0 && al16(a.x) && al16(a.w) && al16(a.dw_w);
  return geom && plain && aligned;
}

int launch_sepconv8(const ConvArgs& a, int epi, hipStream_t s) {
  if (!sepconv8_eligible(a)) return DH_EUNSUPPORTED;
  return a.dw_kh == 5 ? launch_sepconv8_ks<5>(a, epi, s) : launch_sepconv8_ks<3>(a, epi, s);
}

}  // namespace dh
