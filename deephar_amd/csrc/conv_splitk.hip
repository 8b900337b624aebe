// Convolutions with a tiny output map (the action heads: 3x1 .. 3x5 and 1x1 convolutions over a [frames x joints] "image"
// of 16-256 positions per clip, reference deephar/models/spnet.py:51-148, action.py:20-42; the heat-map heads of the
// coarse SPNet levels, spnet.py:24-48): the MFMA kernels of conv_igemm.hip / gemm1x1.hip give one 32 x 32 output tile
// to one wave, which then walks all of K alone.  v_mfma_f32_32x32x2_f32 issues once per 64 cycles, so a K = 480
// reduction is a serial chain of 240 MFMAs = 6.4 us on ONE of the chip's 1024 SIMDs, whatever the memory system does;
// at two 8-frame clips per call (exp/pennaction/eval_speed2d.py) 400 of the 621 launches of the last prediction
// block's model are of this kind and ran 15-33 us each (profiles/r05_speed2d_first.json).
//
// Here a work-group of 4 / 8 / 16 waves owns a 16 x 16 output tile (v_mfma_f32_16x16x4_f32: the same MAC rate per SIMD as
// the 32 x 32 form, four times as many work-groups -- a [2 clips x 8 x 8] x 160 map is 80 tiles on 80 CUs instead of 20 on
// 20, and the A rows of a quad are 64 contiguous bytes) and splits K into contiguous runs of k-group quads (16 k each:
// lane group lg = lane / 16 takes k-group 4 q + lg).  Per chunk of eight quads a lane issues ALL its loads back to back
// -- A: the four consecutive channels of its pixel's tap (one 16-byte load; four dwords when Cin % 4 != 0), B: the
// packed weights' 16-byte unit of its column -- and only then touches them: one memory round trip per chunk, not one per
// group [r05: round 2's kernel applied the BatchNormalization prologue inside the load helper, i.e. waited for every
// pair before issuing the next -- 27 us for K = 1440].  The prologue's per-channel
// scale / shift sit in LDS (staged while the first chunk is in flight); (tap, channel) of a k-group comes from one
// multiply-high, no integer divide.  The partial tiles are summed through LDS in wave order, then BN / residuals / ReLU.
//
// The K order differs from the other conv kernels (NWV partial sums over contiguous K runs), so this kernel is picked
// by a SHAPE rule (conv_is_skinny: per-frame geometry only), never by timing, batch size or alignment, and the number
// of waves follows from K alone: a layer always computes the same bits.
#include <cstdlib>

#include "conv_common.h"

namespace dh {
namespace {

constexpr int SK_MAX_CIN = 4096; // prologue table in LDS: 2 x Cin floats

// RS = dh_conv_args.x_resample [r06]: 0 = x as it is; 1 = x is stored at HALF resolution and read as UpSampling2D((2, 2))(x)
// (input pixel (h, w) = x(h / 2, w / 2)); 2 / 3 = x is stored at DOUBLE resolution and read through MaxPooling2D((2, 2)) /
// layers.max_min_pooling((2, 2)) (max, or max + min, of the four pixels) -- the prologue and the zero padding act on the
// resampled pixels, exactly as if the up-sampling / pooling launch had written them out.
// `block`: the tile index (blockIdx.x, or the index inside one convolution's range of a paired launch).  A work-group that
// was launched with more than NWV waves (the pair's other convolution needs them) retires the extra ones here: a wave that
// has ended no longer counts at s_barrier.
// SEG [r06] (dh_conv2d_seg_f32): the input is a concatenation that was never written -- channels [0, c_split) are the 2 x 2
// maximum (window stride (pool_sh, 2), 'same' padding) of x, stored at [N, H * pool_sh, 2 W, c_split]; channels [c_split, Cin)
// are x2 as stored, [N, H, W, Cin - c_split].  Every element is read as the maximum of four loads (the same pixel four times
// for the second segment): one code path, all loads of a chunk in flight together.
template <int NWV, bool VEC, int RS, bool SEG = false>
__device__ __forceinline__ void conv_skinny_body(const ConvArgs& p, const unsigned magic_cin, const unsigned magic_kw, float* sk_lds,
                                                 const int block, const ConvSeg seg = ConvSeg{}) {
  float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(sk_lds);            // [NWV][4][64] partial tiles
  float* pre_tab = sk_lds + NWV * 4 * 64;                                        // [2][Cin4]: scale, shift (BN prologue only)
  const int tid = threadIdx.x;
  if (tid >= NWV * 64) return;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;               // v_mfma_f32_16x16x4_f32: A[row li][k lg], B[k lg][col li]
  const int M = p.N * p.OH * p.OW;
  const int tiles_n = (p.Cout + 15) / 16;
  const int m0 = (block / tiles_n) * 16;
  const int n0 = (block % tiles_n) * 16;
  const bool aff = p.pre_scale != nullptr;
  const int cin4 = (p.Cin + 3) & ~3;
  constexpr int SK_DEPTH = (RS >= 2 || SEG) ? (VEC ? 4 : 2) : (VEC ? 8 : 4);   // k-group quads in flight per wave (dword form: 4 loads per quad and
                                                          // lane; pooled input: four pixels per element)
  // physical extent of x, and the pixel pitch between the four pixels of a pooled window
  const int PH = SEG ? p.H * seg.pool_sh : (RS == 1 ? p.H >> 1 : (RS >= 2 ? p.H << 1 : p.H));
  const int PW = SEG ? p.W << 1 : (RS == 1 ? p.W >> 1 : (RS >= 2 ? p.W << 1 : p.W));

  // this lane's output pixel (A operand row) and its top-left input position
  int m = m0 + li;
  m = m < M ? m : M - 1;
  const int fr = m / (p.OH * p.OW);
  const int rem = m - fr * (p.OH * p.OW);
  const int oh = rem / p.OW, ow = rem - oh * p.OW;
  const int ih0 = oh * p.SH - p.PT, iw0 = ow * p.SW - p.PL;
  const float* xf = p.x + (size_t)fr * PH * PW * p.ldx;
  const float* xf2 = SEG && seg.x2 != nullptr ? seg.x2 + (size_t)fr * p.H * p.W * seg.ldx2 : xf;
  const int ncol = n0 + li < p.Np ? n0 + li : p.Np - 1;   // (Np is a multiple of 32: always in range; belt and braces)
  const float* wcol = p.w + (size_t)ncol * 4;             // packed [Kp / 4][Np][4]: unit (k-group, column)
  const int quads = p.Kp / 16;                            // 16 k each: lane group lg takes k-group 4 q + lg
  const int per_wave = (quads + NWV - 1) / NWV;
  const int q_begin = wave * per_wave;
  const int q_end = q_begin + per_wave < quads ? q_begin + per_wave : quads;

  f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};     // two chains: a dependent 16x16x4 MFMA stalls 8 of 40 cycles

  // element k -> its address relative to xf, or -1 outside the image / past K (zero padding applies AFTER the prologue)
  auto locate = [&](int k, int& c) -> int {
    const int tap = (int)__umulhi((unsigned)k, magic_cin);            // k / Cin
    c = k - tap * p.Cin;
    const int kh = (int)(((unsigned)tap * magic_kw) >> 16);            // tap / KW
    const int kw = tap - kh * p.KW;
    const int ih = ih0 + kh, iw = iw0 + kw;
    const bool ok = k < p.K && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    if constexpr (RS == 1) return ok ? ((ih >> 1) * PW + (iw >> 1)) * p.ldx + c : -1;
    if constexpr (RS >= 2) return ok ? (2 * ih * PW + 2 * iw) * p.ldx + c : -1;          // top-left pixel of the 2 x 2 window
    return ok ? (ih * p.W + iw) * p.ldx + c : -1;
  };
  // SEG: element k -> (base, offset of the window's first pixel, step to the next column, step to the next row); -1 as above
  auto locate_seg = [&](int k, int& c, const float*& base, int& dcol, int& drow) -> int {
    const int tap = (int)__umulhi((unsigned)k, magic_cin);
    c = k - tap * p.Cin;
    const int kh = (int)(((unsigned)tap * magic_kw) >> 16);
    const int kw = tap - kh * p.KW;
    const int ih = ih0 + kh, iw = iw0 + kw;
    const bool ok = k < p.K && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    const bool first = c < seg.c_split;
    base = first ? xf : xf2;
    const int r0 = ih * seg.pool_sh;
    dcol = first ? p.ldx : 0;
    drow = first && r0 + 1 < PH ? PW * p.ldx : 0;                       // ('same' padding: the last row's window is one row high)
    if (!ok) return -1;
    return first ? (r0 * PW + 2 * iw) * p.ldx + c : (ih * p.W + iw) * seg.ldx2 + (c - seg.c_split);
  };
  // one input element (or float4 of them) through the pooling window
  auto pool4 = [&](float a, float b, float c_, float d) -> float {
    const float mx = fmaxf(fmaxf(a, b), fmaxf(c_, d));
    if constexpr (RS == 3) return mx + fminf(fminf(a, b), fminf(c_, d));
    return mx;
  };

  bool tab_ready = !aff;
  for (int q0 = q_begin; q0 < q_end || !tab_ready; q0 += SK_DEPTH) {
    float4 fa[SK_DEPTH], fb[SK_DEPTH];
    int ch[SK_DEPTH];              // VEC: channel of the group's first element (table index); scalar: k of the first element
    unsigned okm = 0;              // VEC: bit d = group d inside the image; scalar: 4 bits per group
#pragma unroll
    for (int d = 0; d < SK_DEPTH; ++d) {
      const int q = q0 + d;
      const bool live = q < q_end;
      const int kg = 4 * (live ? q : 0) + lg;
      fb[d] = *reinterpret_cast<const float4*>(wcol + (size_t)kg * p.Np * 4);
      const int k0 = 4 * kg;
      if constexpr (VEC && SEG) {
        int c, dcol, drow;
        const float* base;
        const int off = locate_seg(k0, c, base, dcol, drow);
        const bool ok = live && off >= 0;
        const float* q0 = base + (ok ? off : 0);
        const float4 v00 = *reinterpret_cast<const float4*>(q0), v01 = *reinterpret_cast<const float4*>(q0 + dcol);
        const float4 v10 = *reinterpret_cast<const float4*>(q0 + drow), v11 = *reinterpret_cast<const float4*>(q0 + drow + dcol);
        fa[d] = make_float4(pool4(v00.x, v01.x, v10.x, v11.x), pool4(v00.y, v01.y, v10.y, v11.y),
                            pool4(v00.z, v01.z, v10.z, v11.z), pool4(v00.w, v01.w, v10.w, v11.w));
        ch[d] = c;
        okm |= (ok ? 1u : 0u) << d;
      } else if constexpr (VEC) {
        int c;
        const int off = locate(k0, c);
        const bool ok = live && off >= 0;
        if constexpr (RS >= 2) {
          const float* q0 = xf + (ok ? off : 0);
          const float4 v00 = *reinterpret_cast<const float4*>(q0), v01 = *reinterpret_cast<const float4*>(q0 + p.ldx);
          const float4 v10 = *reinterpret_cast<const float4*>(q0 + (size_t)PW * p.ldx);
          const float4 v11 = *reinterpret_cast<const float4*>(q0 + (size_t)(PW + 1) * p.ldx);
          fa[d] = make_float4(pool4(v00.x, v01.x, v10.x, v11.x), pool4(v00.y, v01.y, v10.y, v11.y),
                              pool4(v00.z, v01.z, v10.z, v11.z), pool4(v00.w, v01.w, v10.w, v11.w));
        } else {
          fa[d] = *reinterpret_cast<const float4*>(xf + (ok ? off : 0));
        }
        ch[d] = c;
        okm |= (ok ? 1u : 0u) << d;
      } else {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int c;
          if constexpr (SEG) {
            int dcol, drow;
            const float* base;
            const int off = locate_seg(k0 + j, c, base, dcol, drow);
            const bool ok = live && off >= 0;
            const float* q0 = base + (ok ? off : 0);
            e[j] = pool4(q0[0], q0[dcol], q0[drow], q0[drow + dcol]);
            okm |= (ok ? 1u : 0u) << (4 * d + j);
            continue;
          }
          const int off = locate(k0 + j, c);
          const bool ok = live && off >= 0;
          if constexpr (RS >= 2) {
            const float* q0 = xf + (ok ? off : 0);
            e[j] = pool4(q0[0], q0[p.ldx], q0[(size_t)PW * p.ldx], q0[(size_t)(PW + 1) * p.ldx]);
          } else {
            e[j] = xf[ok ? off : 0];
          }
          okm |= (ok ? 1u : 0u) << (4 * d + j);
        }
        fa[d] = make_float4(e[0], e[1], e[2], e[3]);
        ch[d] = k0;
      }
    }
    if (!tab_ready) {                                       // first pass: the table lands while the chunk is in flight
      for (int i = tid; i < cin4; i += NWV * 64) {
        pre_tab[i] = i < p.Cin ? p.pre_scale[i] : 0.f;
        pre_tab[cin4 + i] = i < p.Cin ? p.pre_shift[i] : 0.f;
      }
      __syncthreads();
      tab_ready = true;
    }
#pragma unroll
    for (int d = 0; d < SK_DEPTH; ++d) {
      float a[4] = {fa[d].x, fa[d].y, fa[d].z, fa[d].w};
      if constexpr (VEC) {
        if (aff) {
          const float4 sc = *reinterpret_cast<const float4*>(pre_tab + ch[d]);
          const float4 sh = *reinterpret_cast<const float4*>(pre_tab + cin4 + ch[d]);
          a[0] = fmaf(a[0], sc.x, sh.x); a[1] = fmaf(a[1], sc.y, sh.y); a[2] = fmaf(a[2], sc.z, sh.z); a[3] = fmaf(a[3], sc.w, sh.w);
        }
        const bool ok = (okm >> d) & 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (p.pre_relu) a[j] = fmaxf(a[j], 0.f);
          a[j] = ok ? a[j] : 0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (aff) {
            const int k = ch[d] + j;
            const int c = k - (int)__umulhi((unsigned)k, magic_cin) * p.Cin;
            a[j] = fmaf(a[j], pre_tab[c], pre_tab[cin4 + c]);
          }
          if (p.pre_relu) a[j] = fmaxf(a[j], 0.f);
          a[j] = ((okm >> (4 * d + j)) & 1u) ? a[j] : 0.f;
        }
      }
      // MFMA j multiplies element j of every lane group's k-group: k = 4 (4 q + lg) + j (a dead group is zeros on the A side)
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], fb[d].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], fb[d].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], fb[d].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], fb[d].w, acc1, 0, 0, 0);
    }
  }

  // C layout of the 16x16 tile: register r of lane (li, lg) = row 4 lg + r, column li.  Thread t < 256 sums and writes
  // output t; its BN / residual values are fetched BEFORE the partial tiles go through LDS, so the epilogue's global round
  // trip runs under the reduction instead of behind it.
  const int ohw = p.OH * p.OW;
  const int er = tid >> 6, el = tid & 63;                                  // (tid < 256: waves 0 .. 3)
  const int mm = m0 + 4 * (el >> 4) + er, n = n0 + (el & 15);
  const bool mine = tid < 256 && mm < M && n < p.Cout;
  float psc = 1.f, psh = 0.f, r1v = 0.f, r2v = 0.f;
  if (mine) {
    if (p.post_scale != nullptr) { psc = p.post_scale[n]; psh = p.post_shift[n]; }
    if (p.res1 != nullptr) r1v = p.res1[(size_t)mm * p.ldr1 + n];
    if (p.res2 != nullptr && !p.up2) r2v = p.res2[(size_t)mm * p.ldr2 + n];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  __syncthreads();
  if (mine) {
    float t = red[0][er][el];
#pragma unroll
    for (int w = 1; w < NWV; ++w) t += red[w][er][el];
    if (p.post_scale != nullptr) t = t * psc + psh;
    if (p.res1 != nullptr) t += r1v;
    if (p.up2) {
      const int f2 = mm / ohw;
      const int rm = mm - f2 * ohw;
      const int o_h = rm / p.OW, o_w = rm - o_h * p.OW;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const size_t mo = ((size_t)f2 * 2 * p.OH + 2 * o_h + (d >> 1)) * (2 * p.OW) + 2 * o_w + (d & 1);
        float o = t;
        if (p.res2 != nullptr) o += p.res2[mo * p.ldr2 + n];
        if (p.post_relu) o = fmaxf(o, 0.f);
        p.y[mo * p.ldy + n] = o;
      }
    } else {
      if (p.res2 != nullptr) t += r2v;
      if (p.post_relu) t = fmaxf(t, 0.f);
      p.y[(size_t)mm * p.ldy + n] = t;
    }
  }
}

template <int NWV, bool VEC, int RS>
__global__ __launch_bounds__(NWV * 64) void conv_skinny_kernel(const ConvArgs p, const unsigned magic_cin, const unsigned magic_kw) {
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];
  conv_skinny_body<NWV, VEC, RS>(p, magic_cin, magic_kw, sk_lds, (int)blockIdx.x);
}

template <int NWV, bool VEC>
__global__ __launch_bounds__(NWV * 64) void conv_skinny_seg_kernel(const ConvArgs p, const ConvSeg seg, const unsigned magic_cin,
                                                                    const unsigned magic_kw) {
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];
  conv_skinny_body<NWV, VEC, 2, true>(p, magic_cin, magic_kw, sk_lds, (int)blockIdx.x, seg);
}

// [r06] Two INDEPENDENT skinny convolutions in one launch (dh_conv2d_pair_f32): work-groups [0, tiles_a) run the first,
// the rest the second, each with the code of its own instantiation (wave count from its K, vector / scalar loads) -- the
// bits of either are those of its own launch.  `code` = 2 * log2(NWV / 4) + VEC.  Plain inputs only (x_resample = 0).
template <int NMAX>
__device__ __forceinline__ void conv_skinny_dispatch(const ConvArgs& p, const int code, const unsigned magic_cin, const unsigned magic_kw,
                                                     float* lds, const int block) {
  switch (code) {
    case 0: conv_skinny_body<4, false, 0>(p, magic_cin, magic_kw, lds, block); break;
    case 1: conv_skinny_body<4, true, 0>(p, magic_cin, magic_kw, lds, block); break;
    case 2: if constexpr (NMAX >= 8) conv_skinny_body<8, false, 0>(p, magic_cin, magic_kw, lds, block); break;
    case 3: if constexpr (NMAX >= 8) conv_skinny_body<8, true, 0>(p, magic_cin, magic_kw, lds, block); break;
    case 4: if constexpr (NMAX >= 16) conv_skinny_body<16, false, 0>(p, magic_cin, magic_kw, lds, block); break;
    case 5: if constexpr (NMAX >= 16) conv_skinny_body<16, true, 0>(p, magic_cin, magic_kw, lds, block); break;
  }
}

template <int NMAX>
__global__ __launch_bounds__(NMAX * 64) void conv_skinny_pair_kernel(const ConvArgs pa, const ConvArgs pb, const int tiles_a, const int code_a,
                                                                      const int code_b, const unsigned magic_cin_a, const unsigned magic_kw_a,
                                                                      const unsigned magic_cin_b, const unsigned magic_kw_b) {
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];
  if ((int)blockIdx.x < tiles_a)
    conv_skinny_dispatch<NMAX>(pa, code_a, magic_cin_a, magic_kw_a, sk_lds, (int)blockIdx.x);
  else
    conv_skinny_dispatch<NMAX>(pb, code_b, magic_cin_b, magic_kw_b, sk_lds, (int)blockIdx.x - tiles_a);
}

template <int NWV, int RS>
int launch_skinny_rs(const ConvArgs& a, unsigned tiles, bool vec, hipStream_t s) {
  const unsigned magic_cin = (unsigned)((1ull << 32) / (unsigned)a.Cin + 1ull);       // k / Cin = umulhi(k, magic), k * Cin < 2^32
  const unsigned magic_kw = (1u << 16) / (unsigned)a.KW + 1u;                          // tap / KW for tap, KW < 2^8 (conv_is_skinny)
  const size_t lds = ((size_t)NWV * 4 * 64 + (a.pre_scale != nullptr ? (size_t)2 * ((a.Cin + 3) & ~3) : 0)) * sizeof(float);
  constexpr int lds_max = (NWV * 4 * 64 + 2 * SK_MAX_CIN) * (int)sizeof(float);      // <= 48 KB
  if (vec) {
    static LdsLimit lim;
    if (lds_max > 64 * 1024) lim.raise((const void*)conv_skinny_kernel<NWV, true, RS>, lds_max);
    hipLaunchKernelGGL((conv_skinny_kernel<NWV, true, RS>), dim3(tiles), dim3(NWV * 64), lds, s, a, magic_cin, magic_kw);
  } else {
    static LdsLimit lim;
    if (lds_max > 64 * 1024) lim.raise((const void*)conv_skinny_kernel<NWV, false, RS>, lds_max);
    hipLaunchKernelGGL((conv_skinny_kernel<NWV, false, RS>), dim3(tiles), dim3(NWV * 64), lds, s, a, magic_cin, magic_kw);
  }
  return check_launch();
}

template <int NWV>
int launch_skinny(const ConvArgs& a, unsigned tiles, bool vec, hipStream_t s) {
  switch (a.x_resample) {
    case 0: return launch_skinny_rs<NWV, 0>(a, tiles, vec, s);
    case 1: return launch_skinny_rs<NWV, 1>(a, tiles, vec, s);
    case 2: return launch_skinny_rs<NWV, 2>(a, tiles, vec, s);
    case 3: return launch_skinny_rs<NWV, 3>(a, tiles, vec, s);
  }
  return DH_EINVAL;
}

}  // namespace

// Shape rule -- deterministic, independent of timing, of the batch size and of buffer alignment (the bits of a layer
// must not depend on any of them): at most 256 output positions per frame / clip, at most 256 output channels, a
// reduction of at least 64 (shorter ones are one or two K-steps of the general kernel anyway).  [r05: was K >= 768 and
// Cin % 4 == 0 -- the action heads' 1x1 / 3x3 convolutions with K = 70 .. 720 ran 15-33 us on one wave per tile.]
bool conv_is_skinny(const ConvArgs& a) {
  if (a.x_u8 || a.w_split) return false;
  // kernel extent: `locate` decodes tap -> (kh, kw) as (tap * magic_kw) >> 16, exact while tap * KW < 2^16 -- KH * KW < 256
  // keeps tap < 256 and KW < 256 (KW < 256 alone admitted KW = 255, KH >= 2, where tap >= 258 decodes wrong)
  // DEEPHAR_SKINNY_MIN_K: A/B aid for the rule's K threshold (read once; planner.split_k_rule reads the same variable)
  static const int min_k = [] { const char* e = getenv("DEEPHAR_SKINNY_MIN_K"); return e != nullptr ? atoi(e) : 64; }();
  return a.OH * a.OW <= 256 && a.K >= min_k && a.Cout <= 256 && a.Cin >= 2 && a.Cin <= SK_MAX_CIN && a.KH >= 1 && a.KW >= 1 &&
         (long long)a.KH * a.KW < 256 && (long long)a.K * a.Cin < (1ll << 31);
}

// waves per tile: from K alone (16 k per quad, eight quads per wave and chunk)
int conv_skinny_waves(int Kp) {
  const int quads = Kp / 16;
  return quads > 32 ? 16 : (quads > 8 ? 8 : 4);
}

namespace {
struct SkinnyLaunch { long long tiles; int nwv; bool vec; unsigned magic_cin, magic_kw; size_t lds; };
int skinny_launch_shape(const ConvArgs& a, SkinnyLaunch* o) {
  const long long M = (long long)a.N * a.OH * a.OW;
  o->tiles = ((M + 15) / 16) * ((a.Cout + 15) / 16);
  if (o->tiles <= 0 || o->tiles > 0x3fffffffLL || (long long)a.H * a.W * a.ldx > 0x7fffffffLL || a.Kp % 16 != 0) return DH_EINVAL;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  o->vec = a.Cin % 4 == 0 && a.ldx % 4 == 0 && al16(a.x);
  o->nwv = conv_skinny_waves(a.Kp);
  o->magic_cin = (unsigned)((1ull << 32) / (unsigned)a.Cin + 1ull);
  o->magic_kw = (1u << 16) / (unsigned)a.KW + 1u;
  o->lds = ((size_t)o->nwv * 4 * 64 + (a.pre_scale != nullptr ? (size_t)2 * ((a.Cin + 3) & ~3) : 0)) * sizeof(float);
  return DH_OK;
}
template <int NMAX>
int launch_pair(const ConvArgs& a, const ConvArgs& b, const SkinnyLaunch& la, const SkinnyLaunch& lb, hipStream_t s) {
  auto code = [](const SkinnyLaunch& l) { return 2 * (l.nwv == 16 ? 2 : (l.nwv == 8 ? 1 : 0)) + (l.vec ? 1 : 0); };
  const size_t lds = la.lds > lb.lds ? la.lds : lb.lds;
  hipLaunchKernelGGL((conv_skinny_pair_kernel<NMAX>), dim3((unsigned)(la.tiles + lb.tiles)), dim3(NMAX * 64), lds, s, a, b, (int)la.tiles,
                     code(la), code(lb), la.magic_cin, la.magic_kw, lb.magic_cin, lb.magic_kw);
  return check_launch();
}
}  // namespace

// dh_conv2d_pair_f32: both convolutions on the skinny-conv kernel (conv_is_skinny), inputs as stored (x_resample = 0), no
// second residual at half resolution -- anything else is for the two entry points.
int launch_conv_skinny_pair(const ConvArgs& a, const ConvArgs& b, hipStream_t s) {
  if (!conv_is_skinny(a) || !conv_is_skinny(b) || a.x_resample || b.x_resample || a.res2_down || b.res2_down || a.y_pool != nullptr ||
      b.y_pool != nullptr)
    return DH_EUNSUPPORTED;
  SkinnyLaunch la, lb;
  if (skinny_launch_shape(a, &la) != DH_OK || skinny_launch_shape(b, &lb) != DH_OK) return DH_EINVAL;
  const int nmax = la.nwv > lb.nwv ? la.nwv : lb.nwv;
  switch (nmax) {
    case 16: return launch_pair<16>(a, b, la, lb, s);
    case 8: return launch_pair<8>(a, b, la, lb, s);
    default: return launch_pair<4>(a, b, la, lb, s);
  }
}

namespace {
template <int NWV>
int launch_seg(const ConvArgs& a, const ConvSeg& seg, const SkinnyLaunch& l, hipStream_t s) {
  if (l.vec)
    hipLaunchKernelGGL((conv_skinny_seg_kernel<NWV, true>), dim3((unsigned)l.tiles), dim3(NWV * 64), l.lds, s, a, seg, l.magic_cin, l.magic_kw);
  else
    hipLaunchKernelGGL((conv_skinny_seg_kernel<NWV, false>), dim3((unsigned)l.tiles), dim3(NWV * 64), l.lds, s, a, seg, l.magic_cin, l.magic_kw);
  return check_launch();
}
}  // namespace

// dh_conv2d_seg_f32: a skinny-conv layer whose input is concatenate([MaxPooling2D((2, 2), strides=(pool_sh, 2))(x), x2]) read in
// place.  a.H, a.W: the extent the convolution sees (the pooled one); a.Cin: both segments; a.x_resample must be 0.
int launch_conv_splitk_seg(const ConvArgs& a, const ConvSeg& seg, hipStream_t s) {
  if (!conv_is_skinny(a) || a.x_resample || a.res2_down || a.y_pool != nullptr) return DH_EUNSUPPORTED;
  if (seg.c_split <= 0 || seg.c_split > a.Cin || (seg.pool_sh != 1 && seg.pool_sh != 2) ||
      (seg.c_split < a.Cin && (seg.x2 == nullptr || seg.ldx2 < a.Cin - seg.c_split)) || a.ldx < seg.c_split)
    return DH_EINVAL;
  SkinnyLaunch l;
  if (skinny_launch_shape(a, &l) != DH_OK) return DH_EINVAL;
  if ((long long)a.H * seg.pool_sh * a.W * 2 * a.ldx > 0x7fffffffLL || (long long)a.H * a.W * seg.ldx2 > 0x7fffffffLL) return DH_EINVAL;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  l.vec = l.vec && seg.c_split % 4 == 0 && (seg.x2 == nullptr || (seg.ldx2 % 4 == 0 && al16(seg.x2)));
  switch (l.nwv) {
    case 16: return launch_seg<16>(a, seg, l, s);
    case 8: return launch_seg<8>(a, seg, l, s);
    default: return launch_seg<4>(a, seg, l, s);
  }
}

int launch_conv_splitk(const ConvArgs& a, hipStream_t s) {
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = ((M + 15) / 16) * ((a.Cout + 15) / 16);
  if (tiles <= 0 || tiles > 0x7fffffffLL || (long long)a.H * a.W * a.ldx * (a.x_resample >= 2 ? 4 : 1) > 0x7fffffffLL ||
      a.Kp % 16 != 0)
    return DH_EINVAL;
  if (a.x_resample < 0 || a.x_resample > 3 || (a.x_resample == 1 && ((a.H | a.W) & 1))) return DH_EINVAL;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // (the vector form is a property of the layer wherever the engine lays tensors out 16-byte aligned; both forms load
  //  the same values and run the same arithmetic: same bits)
  const bool vec = a.Cin % 4 == 0 && a.ldx % 4 == 0 && al16(a.x);
  switch (conv_skinny_waves(a.Kp)) {
    case 16: return launch_skinny<16>(a, (unsigned)tiles, vec, s);
    case 8: return launch_skinny<8>(a, (unsigned)tiles, vec, s);
    default: return launch_skinny<4>(a, (unsigned)tiles, vec, s);
  }
}

}  // namespace dh
