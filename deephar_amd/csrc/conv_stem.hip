// First-layer convolution for gfx950: 3 input channels, stride 2, 128 output columns -- ReceptionNet's stem conv 3x3 3->32
// (reference deephar/models/reception.py:61-66, conv_bn_act) and SPNet's entry conv 7x7 3->64 (deephar/models/spnet.py:
// 317-322) on 256 x 256 frames.  The general implicit-GEMM kernel gathers such a layer element by element (Cin = 3 is not
// a float4): 0.14 / 0.37 of the fp32 MFMA rate (round-3 profile).  Here:
//   * work-group = TWO full output rows of one frame (256 consecutive output pixels x all Cout), walked over
//     `pairs_per_wg` consecutive row pairs; the input rows those outputs touch (KH + 2 rows x (254 + KW) pixels x 3
//     channels) are copied into LDS as they lie in memory -- NHWC with C = 3 makes an image row one contiguous run of floats,
//     zero padding = out-of-image positions written as 0 -- so the copy is a bounds-checked linear copy, and for uint8
//     frames (dh_conv_args.x_u8) a byte copy through the normalisation table;
//   * im2col happens in the fragment ADDRESS: for a fixed kernel row kh the taps (kw, c) of an output pixel are KW * 3
//     CONSECUTIVE floats of the LDS row starting at pixel 2 * ow, so lane (li, lh) reads A[row li][k = 2j + lh] of MFMA j
//     with one ds_read_b32 at an immediate offset; KW * 3 is padded to even with a zero weight (the extra float read is
//     a neighbouring input value, replaced by 0 in the pad lane);
//   * the weights [KH][KW*3 (+pad)][Cout] are staged into LDS once per work-group from the standard packed layout
//     ([Kp/4][Np][4], K = (kh, kw, c)): no second weight packing;
//   * K runs (kh, kw, c) ascending, two consecutive k per MFMA; the tap-major kernels pair k with k + 4 inside blocks of
//     eight, so the last bits differ from theirs: dh_conv2d_f32 picks this kernel by a RULE on the layer's geometry
//     (conv_stem_eligible / dh_conv2d_uses_first_layer_kernel), whatever tiling is asked for -- never by timing;
//   * epilogue (BN, ReLU; the layer has no residual) straight from the accumulators, dword stores with scalar row offsets.
#include "conv_common.h"

namespace dh {
namespace {

constexpr int kStemOW = 128;

template <int KH, int KW, int TN, bool U8>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(const ConvArgs p, const int pairs_per_wg) {
  constexpr int C = 3, KL = KW * C, KLP = (KL + 1) & ~1, NJ = KLP / 2;
  constexpr int OW = kStemOW, IWH = (OW - 1) * 2 + KW;   // halo pixels per input row
  constexpr int RF = IWH * C;                            // floats per halo row
  constexpr int RP = RF + 3;                             // row pitch: room for the one float the K padding reads past the row
  constexpr int HR = KH + 2;                             // input rows under two output rows
  constexpr int BN = TN * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                                      // [KH][KLP][BN]
  float* halo = smem + KH * KLP * BN;                    // [HR][RP]
  float* lut = halo + HR * RP;                           // U8: [3][256]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  // ---- weights: standard packing, element (k, n) at ((k >> 2) * Np + n) * 4 + (k & 3), k = kh * KL + kl
  // [r06] every weight load of the thread is in flight at once and lands in LDS behind the first halo request below: the
  // straightforward loop (load, wait, ds_write, next) compiled to exactly that -- up to 39 DEPENDENT memory round trips at the
  // start of every work-group (7 x 7: 9 856 weights / 256 threads).
  constexpr int NWL = (KH * KLP * BN + 255) / 256;
  float wst[NWL];
#pragma unroll
  for (int q = 0; q < NWL; ++q) {
    const int idx = tid + q * 256;
    const int n = idx % BN, kk = idx / BN;
    const int kl = kk % KLP, kh = kk / KLP;
    const int k = kh * KL + kl;
    const bool ok = idx < KH * KLP * BN && kl < KL && n < p.Cout;
    const float v = p.w[ok ? ((size_t)(k >> 2) * p.Np + n) * 4 + (k & 3) : 0];
    wst[q] = ok ? v : 0.f;
  }
  float lst[U8 ? 3 : 1];
  if constexpr (U8) {
#pragma unroll
    for (int i = 0; i < 3; ++i) lst[i] = p.in_lut[tid + i * 256];
  }

  const int pairs = p.OH >> 1;
  const int groups = pairs / pairs_per_wg;               // row-pair groups per frame (launcher: divides)
  const int wg = xcd_tile(blockIdx.x, gridDim.x);        // neighbouring row groups share halo rows: same XCD, same L2
  const int n = wg / groups;
  const int pair0 = (wg - n * groups) * pairs_per_wg;
  const size_t frame = (size_t)n * p.H * p.W * C;        // element offset of the frame (floats or bytes)
  const int row_elems = p.W * C;

  // ---- halo: HR rows x RP floats, a linear copy of the image rows with zeros outside the image.  [r05] The copy of row
  // pair pr + 1 is issued into registers BEFORE the MFMA block of pair pr and lands in LDS after it: the global round trip
  // runs under the matrix work (round 4 waited for it between two barriers; the SPNet first layer sat at 0.55).
  constexpr int NE = (HR * RP + 255) / 256;              // fully unrolled: every load of the copy is in flight at once
  float stg[U8 ? 1 : NE];                                // float frames: the value (0 outside the image)
  int sti[U8 ? NE : 1];                                  // uint8 frames: index into the normalisation table, -1 outside
  auto issue = [&](int pr) {
    const int ih0 = (pair0 + pr) * 4 - p.PT;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = tid + q * 256;
      if (e >= HR * RP) break;
      const int r = e / RP, f = e - r * RP;
      const int ih = ih0 + r, fo = f - C * p.PL;
      const bool ok = f < RF && (unsigned)ih < (unsigned)p.H && (unsigned)fo < (unsigned)row_elems;
      const size_t off = frame + (size_t)(ok ? ih : 0) * row_elems + (ok ? fo : 0);
      if constexpr (U8) {
        const int ch = (fo + 3 * 64) % 3;                // (fo >= -3 * PL > -192)
        const int b = reinterpret_cast<const unsigned char*>(p.x)[off];
        sti[q] = ok ? ch * 256 + b : -1;
      } else {
        const float v = p.x[off];
        stg[q] = ok ? v : 0.f;
      }
    }
  };
  auto land = [&]() {
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = tid + q * 256;
      if (e >= HR * RP) break;
      if constexpr (U8) halo[e] = sti[q] >= 0 ? lut[sti[q]] : 0.f;
      else halo[e] = stg[q];
    }
  };
  issue(0);
#pragma unroll
  for (int q = 0; q < NWL; ++q) {
    const int idx = tid + q * 256;
    if (idx < KH * KLP * BN) Bs[idx] = wst[q];
  }
  if constexpr (U8) {
#pragma unroll
    for (int i = 0; i < 3; ++i) lut[tid + i * 256] = lst[i];
  }
  __syncthreads();                                       // weights and table staged (the table is read by land())

  // ---- fragment addresses (float indices), fixed over the row pairs
  int a_base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    a_base[i] = (wave >> 1) * 2 * RP + ((wave & 1) * 64 + i * 32 + li) * 2 * C + lh;
  const int b_base = lh * BN + li;

  float dsc[TN], dsh[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int col = t * 32 + li;
    dsc[t] = p.post_scale != nullptr && col < p.Cout ? p.post_scale[col] : 1.f;
    dsh[t] = p.post_scale != nullptr && col < p.Cout ? p.post_shift[col] : 0.f;
  }
  const long long M = (long long)p.N * p.OH * OW;
  const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(((unsigned)(M - 1) * p.ldy + (unsigned)p.Cout) * 4u), 0x00020000);

  for (int pr = 0; pr < pairs_per_wg; ++pr) {
    const int oh0 = (pair0 + pr) * 2;
    if (pr > 0) __syncthreads();                         // every wave is done reading the previous halo
    land();
    __syncthreads();
    if (pr + 1 < pairs_per_wg) issue(pr + 1);

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float a[2], b[TN];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = halo[a_base[i] + kh * RP + 2 * j];
          if ((KL & 1) && j == NJ - 1) a[i] = lh ? 0.f : a[i];       // the K pad slot holds a neighbouring input value: an Inf / NaN
        }                                                            // there must not reach this output through 0 * x
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = Bs[b_base + (kh * KLP + 2 * j) * BN + t * 32];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[t], acc[i][t], 0, 0, 0);
      }

    // ---- epilogue: register r of a lane is row (r & 3) + 8 * (r >> 2) + 4 * lh, column li of the 32 x 32 tile
    const int m_wave = (n * p.OH + oh0 + (wave >> 1)) * OW + (wave & 1) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vo = ((m_wave + i * 32 + 4 * lh) * p.ldy + li) * 4;
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        if (t * 32 + li < p.Cout) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][t][r];
            if (p.post_scale != nullptr) v = v * dsc[t] + dsh[t];
            if (p.post_relu) v = fmaxf(v, 0.f);
            buf_st1_stream(rs_y, vo, (((r & 3) + 8 * (r >> 2)) * p.ldy + t * 32) * 4, v);
          }
        }
      }
    }
  }
}

template <int KH, int KW, int TN, bool U8>
int launch_stem_variant(const ConvArgs& a, hipStream_t s) {
  constexpr int C = 3, KL = KW * C, KLP = (KL + 1) & ~1;
  constexpr int RP = ((kStemOW - 1) * 2 + KW) * C + 3, HR = KH + 2;
  constexpr size_t lds = (size_t)(KH * KLP * TN * 32 + HR * RP + (U8 ? 3 * 256 : 0)) * sizeof(float);
  static_assert(2 * lds <= 160 * 1024, "two work-groups per CU");
  const int pairs = a.OH / 2;
  const int ppw = pairs % 4 == 0 ? 4 : (pairs % 2 == 0 ? 2 : 1);
  const long long blocks = (long long)a.N * (pairs / ppw);
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DH_EINVAL;
  auto kern = conv_stem_kernel<KH, KW, TN, U8>;
  if (lds > 64 * 1024) {
    static LdsLimit lim;
    lim.raise((const void*)kern, (int)lds);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a, ppw);
  return check_launch();
}

template <int KH, int KW, int TN>
int launch_stem_tn(const ConvArgs& a, hipStream_t s) {
  return a.x_u8 ? launch_stem_variant<KH, KW, TN, true>(a, s) : launch_stem_variant<KH, KW, TN, false>(a, s);
}

}  // namespace

// The layers this kernel takes (a rule on the layer's geometry; launch_conv_igemm asks it for every launch):
// 3 dense input channels, stride 2, 3x3 or 7x7, 128 output columns, an even number of output rows, 32 or 64 output
// channels, fp32 tap-major weights, BN / ReLU epilogue only.
bool conv_stem_eligible(const ConvArgs& a) {
  const bool geom = a.Cin == 3 && a.ldx == 3 && a.SH == 2 && a.SW == 2 && a.KH == a.KW && (a.KH == 3 || a.KH == 7) &&
                    a.OW == kStemOW && (a.OH & 1) == 0 && (a.Cout == 32 || a.Cout == 64) && a.PT >= 0 && a.PL >= 0 &&
                    a.PL <= a.KW && a.K == a.KH * a.KW * 3;
  const bool plain = a.pre_scale == nullptr && !a.pre_relu && a.res1 == nullptr && a.res2 == nullptr && !a.up2 &&
                     a.y_pool == nullptr && a.w_split == 0 && (!a.x_u8 || a.in_lut != nullptr);
  const bool fits = (long long)a.N * a.OH * a.OW * a.ldy * 4 < 0x7fffffffLL && (reinterpret_cast<uintptr_t>(a.y) & 3) == 0;
  return geom && plain && fits;
}

int launch_conv_stem(const ConvArgs& a, hipStream_t s) {
  if (!conv_stem_eligible(a)) return DH_EUNSUPPORTED;
  if (a.KH == 3) return a.Cout == 32 ? launch_stem_tn<3, 3, 1>(a, s) : launch_stem_tn<3, 3, 2>(a, s);
  return a.Cout == 32 ? launch_stem_tn<7, 7, 1>(a, s) : launch_stem_tn<7, 7, 2>(a, s);
}

}  // namespace dh
