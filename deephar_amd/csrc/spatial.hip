// HBM-bound spatial kernels for gfx950: depthwise KxK conv, max / max+min pooling, nearest x2
// up-sampling (+add), element-wise glue, channel-slab copies, zero padding.
// All NHWC fp32; lanes run over channels (float4 = 16 B per lane, coalesced 128-B+ segments per pixel),
// rows of the grid over pixels.  Replaces Keras SeparableConv2D's depthwise half, MaxPooling2D,
// UpSampling2D, add, concatenate (reference deephar/layers.py:74-104, reception.py:74,86,108-127).
#include "dh_kernels.h"
#include "dw_lds.h"

namespace dh {
namespace {

using namespace dwl;

// ------------------------------------------------------------------------------------------------
// Depthwise conv.  Thread = (channel quad, output row, strip of TW output columns).  The strip walks a
// sliding window along W so each input float4 is loaded once per (row, kh) instead of KW times.
// ------------------------------------------------------------------------------------------------
template <int KS, int TW>
__global__ __launch_bounds__(256) void dwconv_kernel(const DwArgs p) {
  const int c4n = p.C >> 2;
  const int strips = (p.W + TW - 1) / TW;
  const long long total = (long long)p.N * p.H * strips * c4n;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % c4n) * 4;
    long long t = idx / c4n;
    const int st = (int)(t % strips); t /= strips;
    const int oh = (int)(t % p.H);
    const int n = (int)(t / p.H);
    const int ow0 = st * TW;

    const bool aff = p.pre_scale != nullptr;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero;
    if (aff) { sc = ld4(p.pre_scale + c); sh = ld4(p.pre_shift + c); }

    float4 acc[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = zero;

#pragma unroll 1
    for (int kh = 0; kh < KS; ++kh) {   // one row of taps at a time keeps ~110 VGPRs (4 waves/SIMD)
      const int ih = oh - p.PT + kh;
      const bool rok = (unsigned)ih < (unsigned)p.H;
      // up_in: x is stored at half resolution and read as UpSampling2D((2, 2))(x): input pixel (ih, iw) = x(ih / 2, iw / 2)
      const int ush = p.up_in ? 1 : 0;
      const float* row = p.x + ((size_t)(n * (p.H >> ush) + ((rok ? ih : 0) >> ush)) * (p.W >> ush)) * p.ldx + c;
      // all loads of the row are issued unconditionally (clamped), masked afterwards: no wait between them
      float4 in[TW + KS - 1];
#pragma unroll
      for (int j = 0; j < TW + KS - 1; ++j) {
        const int iw = ow0 - p.PL + j;
        const bool ok = (unsigned)iw < (unsigned)p.W;
        in[j] = ld4(row + (size_t)((ok ? iw : 0) >> ush) * p.ldx);
      }
      float4 wv[KS];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) wv[kw] = ld4(p.w + (size_t)(kh * KS + kw) * p.C + c);
#pragma unroll
      for (int j = 0; j < TW + KS - 1; ++j) {
        const int iw = ow0 - p.PL + j;
        float4 v = in[j];
        if (aff) v = fma4(v, sc, sh);
        if (p.pre_relu) v = max4(v, zero);
        if (!(rok && (unsigned)iw < (unsigned)p.W)) v = zero;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          const int o = j - kw;
          if (o >= 0 && o < TW) acc[o] = fma4(v, wv[kw], acc[o]);
        }
      }
    }
    float* out = p.y + ((size_t)(n * p.H + oh) * p.W + ow0) * p.ldy + c;
#pragma unroll
    for (int i = 0; i < TW; ++i)
      if (ow0 + i < p.W) st4(out + (size_t)i * p.ldy, acc[i]);
  }
}

// LDS-tiled depthwise conv: the body lives in dw_lds.h (shared with the grouped launch of gemm1x1.hip)
template <int KS, int NT, int MAXT, int MAXN, bool AFF, bool RELU>
__global__ __launch_bounds__(NT) void dwconv_lds_kernel(const DwArgs p, const int tw, const int rows, const int twh_magic) {
  extern __shared__ __attribute__((aligned(16))) float4 dsm[];
  dwconv_lds_body<KS, NT, MAXT, MAXN, AFF, RELU>(p, tw, rows, twh_magic, (int)blockIdx.x, dsm);
}

// Generic fallback (any KW, C not a multiple of 4): one thread per output element.
__global__ __launch_bounds__(256) void dwconv_generic_kernel(const DwArgs p) {
  const long long total = (long long)p.N * p.H * p.W * p.C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % p.C);
    long long t = idx / p.C;
    const int ow = (int)(t % p.W); t /= p.W;
    const int oh = (int)(t % p.H);
    const int n = (int)(t / p.H);
    float acc = 0.f;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int ih = oh - p.PT + kh;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int iw = ow - p.PL + kw;
        if ((unsigned)iw >= (unsigned)p.W) continue;
        const int ush = p.up_in ? 1 : 0;
        float v = p.x[((size_t)(n * (p.H >> ush) + (ih >> ush)) * (p.W >> ush) + (iw >> ush)) * p.ldx + c];
        if (p.pre_scale != nullptr) v = fmaf(v, p.pre_scale[c], p.pre_shift[c]);
        if (p.pre_relu) v = fmaxf(v, 0.f);
        acc = fmaf(v, p.w[(size_t)(kh * p.KW + kw) * p.C + c], acc);
      }
    }
    p.y[((size_t)(n * p.H + oh) * p.W + ow) * p.ldy + c] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void pool_kernel(const PoolArgs p) {
  constexpr int V = VEC ? 4 : 1;
  const int cn = p.C / V;
  const long long total = (long long)p.N * p.OH * p.OW * cn;
  // (round 6 tried the XCD-contiguous block order here -- xcd_order, dh_kernels.h: overlapping windows of the 3 x 3 / stride-2
  //  pooling then share an L2 and the fetched bytes drop from 1.54x to 1.17x the input, but the launch gets SLOWER: 430 -> 455 us
  //  on SPNet-NTU's 128 x 128 x 96 pooling, 34 -> 42 us at 16 frames; profiles/r06_pmc_all_launches.md.  Round-robin order kept.)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cn) * V;
    long long t = idx / cn;
    const int ow = (int)(t % p.OW); t /= p.OW;
    const int oh = (int)(t % p.OH);
    const int n = (int)(t / p.OH);
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    for (int kh = 0; kh < p.KH; ++kh) {
      const int ih = oh * p.SH - p.PT + kh;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int iw = ow * p.SW - p.PL + kw;
        if ((unsigned)iw >= (unsigned)p.W) continue;
        const float* src = p.x + ((size_t)(n * p.H + ih) * p.W + iw) * p.ldx + c;
        float4 v;
        if constexpr (VEC) v = ld4(src);
        else v = make_float4(*src, 0.f, 0.f, 0.f);
        mx = max4(mx, v);
        mn = min4(mn, v);
      }
    }
    float4 r = mx;
    if (p.mode == 1) r = make_float4(mx.x + mn.x, mx.y + mn.y, mx.z + mn.z, mx.w + mn.w);
    float* dst = p.y + ((size_t)(n * p.OH + oh) * p.OW + ow) * p.ldy + c;
    if constexpr (VEC) st4(dst, r);
    else *dst = r.x;
  }
}

// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void upsample2x_add_kernel(const float* __restrict__ a, int lda,
                                                             const float* __restrict__ b, int ldb,
                                                             float* __restrict__ y, int ldy, int N, int H,
                                                             int W, int C) {
  constexpr int V = VEC ? 4 : 1;
  const int cn = C / V;
  const long long total = (long long)N * H * W * cn;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cn) * V;
    long long t = idx / cn;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const size_t po = (size_t)(n * H + h) * W + w;
    const size_t pb = (size_t)(n * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1);
    if constexpr (VEC) {
      float4 v = ld4(b + pb * ldb + c);
      if (a != nullptr) {
        const float4 u = ld4(a + po * lda + c);
        v = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
      }
      st4(y + po * ldy + c, v);
    } else {
      float v = b[pb * ldb + c];
      if (a != nullptr) v = a[po * lda + c] + v;
      y[po * ldy + c] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eltwise_kernel(const EltArgs p) {
  const long long total = p.npix * p.C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % p.C);
    const long long px = idx / p.C;
    float v = p.a[px * p.lda + c];
    if (p.scale != nullptr) v = v * p.scale[c] + p.shift[c];
    if (p.op == 0) {
      if (p.b != nullptr) v += p.b[px * p.ldb + (p.bcast_b ? 0 : c)];
      if (p.c != nullptr) v += p.c[px * p.ldc + c];
    } else if (p.op == 1) {
      v *= p.b[px * p.ldb + (p.bcast_b ? 0 : c)];
    } else if (p.op == 2) {
      if (p.b != nullptr) v += p.b[px * p.ldb + (p.bcast_b ? 0 : c)];
      v = 1.f / (1.f + expf(-v));
    }
    if (p.relu) v = fmaxf(v, 0.f);
    p.y[px * p.ldy + c] = v;
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ x, int ldx,
                                                            float* __restrict__ y, int ldy, long long npix,
                                                            int C) {
  constexpr int V = VEC ? 4 : 1;
  const int cn = C / V;
  const long long total = npix * cn;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cn) * V;
    const long long px = idx / cn;
    if constexpr (VEC) st4(y + px * ldy + c, ld4(x + px * ldx + c));
    else y[px * ldy + c] = x[px * ldx + c];
  }
}

__global__ __launch_bounds__(256) void zeropad_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int B, int H, int W, int C, int OH, int OW, int PT,
                                                      int PL) {
  const long long total = (long long)B * OH * OW * C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int w = (int)(t % OW); t /= OW;
    const int h = (int)(t % OH);
    const int b = (int)(t / OH);
    const int ih = h - PT, iw = w - PL;
    y[idx] = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                 ? x[((size_t)(b * H + ih) * W + iw) * C + c] : 0.f;
  }
}

inline unsigned grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  const long long cap = 256LL * 16;  // 256 CUs x 16 workgroups, grid-stride beyond that
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int KS, int NT, int MAXT, int MAXN, bool AFF, bool RELU>
int launch_dw_lds_variant(const DwArgs& a, int tw, int rows, unsigned blocks, size_t lds, hipStream_t s) {
  auto kern = dwconv_lds_kernel<KS, NT, MAXT, MAXN, AFF, RELU>;
  if (lds > 64 * 1024) {
    static LdsLimit lim;
    lim.raise((const void*)kern, (int)lds);
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, s, a, tw, rows, 65536 / (tw + KS - 1) + 1);
  return check_launch();
}

template <int KS, int NT, int MAXT, int MAXN>
int launch_dw_lds_ks(const DwArgs& a, int tw, int rows, unsigned blocks, size_t lds, hipStream_t s) {
  const int twh = tw + KS - 1;
  if ((KS - 1) * twh > MAXT * (NT / DW_CQ) || rows * twh > MAXN * (NT / DW_CQ)) return DH_EUNSUPPORTED;
  const bool aff = a.pre_scale != nullptr;
  if (aff) return a.pre_relu ? launch_dw_lds_variant<KS, NT, MAXT, MAXN, true, true>(a, tw, rows, blocks, lds, s)
                             : launch_dw_lds_variant<KS, NT, MAXT, MAXN, true, false>(a, tw, rows, blocks, lds, s);
  return a.pre_relu ? launch_dw_lds_variant<KS, NT, MAXT, MAXN, false, true>(a, tw, rows, blocks, lds, s)
                    : launch_dw_lds_variant<KS, NT, MAXT, MAXN, false, false>(a, tw, rows, blocks, lds, s);
}

int launch_dw_lds(const DwArgs& a, int tw, int rows, int nt, unsigned blocks, size_t lds, hipStream_t s) {
  // slots per thread: 256 threads = 32 tile pixels per slot (5 x 36-pixel rows above, 8 x 36 or 16 x 20 new pixels per
  // band), 64 threads = 8 pixels per slot (4 x 12 above, 8 x 12 new)
  if (a.KW == 5) return nt == 256 ? launch_dw_lds_ks<5, 256, 5, 10>(a, tw, rows, blocks, lds, s)
                                  : launch_dw_lds_ks<5, 64, 6, 12>(a, tw, rows, blocks, lds, s);
  return nt == 256 ? launch_dw_lds_ks<3, 256, 3, 10>(a, tw, rows, blocks, lds, s)
                   : launch_dw_lds_ks<3, 64, 3, 10>(a, tw, rows, blocks, lds, s);
}


}  // namespace

int launch_dwconv(const DwArgs& a, hipStream_t s) {
  if (a.N <= 0 || a.C <= 0 || a.KH <= 0 || a.KW <= 0) return DH_EINVAL;
  if (a.up_in && ((a.H & 1) || (a.W & 1))) return DH_EINVAL;        // an up-sampled input has even extents
  const bool vec = (a.C % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) && al16(a.x) && al16(a.y) &&
                   al16(a.w) && (a.pre_scale == nullptr || (al16(a.pre_scale) && al16(a.pre_shift)));
  if (vec && a.KH == a.KW && (a.KW == 5 || a.KW == 3) && a.C % 32 == 0 && a.W >= DW_LDS_MIN_W && a.W % 8 == 0) {
    // LDS-tiled path: column tile = min(W, 32), 8-column strips; the work-group is sized to the map --
    //   W >= 32: 256 threads = 8 channel quads x 4 strips x 8 rows, bands of 8 rows walked inside the work-group
    //   W == 16: 256 threads x 16 rows, one band (128 threads x 8 rows, two bands, 38 KB of LDS: 19.4 vs 18.0 us, not used)
    //   W ==  8:  64 threads x  8 rows: the whole 8 x 8 map of a 32-channel chunk in one band (10.7 us; the register
    //             kernel these maps ran on before took 13.0)
    const int tw = a.W >= 32 ? 32 : a.W;
    const int nt = tw == 8 ? 64 : 256;
    int rows = nt / (8 * (tw / 8));
    if (rows > a.H) rows = a.H;                                      // (8 or 16, or the even H of an up-sampled input: even)
    const long long blocks = (long long)a.N * ((a.W + tw - 1) / tw) * (a.C / 32);    // bands are walked inside
    const size_t lds = ((size_t)(rows + a.KW - 1) * (tw + a.KW - 1) * DW_PITCH + (size_t)a.KW * a.KW * DW_CQ + 1) * 16;
    const bool fits31 = (long long)a.H * a.W * a.ldx * 4 < 0x7fffffffLL && (long long)a.H * a.W * a.ldy * 4 < 0x7fffffffLL;
    if (blocks <= 0x7fffffffLL && fits31) {
      const int rc = launch_dw_lds(a, tw, rows, nt, (unsigned)blocks, lds, s);
      if (rc != DH_EUNSUPPORTED) return rc;
    }
  }
  if (vec && a.KH == a.KW && (a.KW == 5 || a.KW == 3 || a.KW == 1)) {
    const int TW = a.W >= 16 ? 8 : 4;
    const long long total = (long long)a.N * a.H * ((a.W + TW - 1) / TW) * (a.C / 4);
    const dim3 g(grid_for(total)), b(256);
    if (a.KW == 5 && TW == 8) hipLaunchKernelGGL((dwconv_kernel<5, 8>), g, b, 0, s, a);
    else if (a.KW == 5) hipLaunchKernelGGL((dwconv_kernel<5, 4>), g, b, 0, s, a);
    else if (a.KW == 3 && TW == 8) hipLaunchKernelGGL((dwconv_kernel<3, 8>), g, b, 0, s, a);
    else if (a.KW == 3) hipLaunchKernelGGL((dwconv_kernel<3, 4>), g, b, 0, s, a);
    else if (TW == 8) hipLaunchKernelGGL((dwconv_kernel<1, 8>), g, b, 0, s, a);
    else hipLaunchKernelGGL((dwconv_kernel<1, 4>), g, b, 0, s, a);
  } else {
    const long long total = (long long)a.N * a.H * a.W * a.C;
    hipLaunchKernelGGL(dwconv_generic_kernel, dim3(grid_for(total)), dim3(256), 0, s, a);
  }
  return check_launch();
}

int launch_pool(const PoolArgs& a, hipStream_t s) {
  if (a.N <= 0 || a.C <= 0 || a.OH <= 0 || a.OW <= 0) return DH_EINVAL;
  const bool vec = (a.C % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) && al16(a.x) && al16(a.y);
  const long long total = (long long)a.N * a.OH * a.OW * (vec ? a.C / 4 : a.C);
  if (vec) hipLaunchKernelGGL(pool_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(pool_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, a);
  return check_launch();
}

int launch_upsample2x_add(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int N, int H,
                          int W, int C, hipStream_t s) {
  if (N <= 0 || C <= 0 || (H & 1) || (W & 1)) return DH_EINVAL;
  const bool vec = (C % 4 == 0) && (ldb % 4 == 0) && (ldy % 4 == 0) && al16(b) && al16(y) &&
                   (a == nullptr || ((lda % 4 == 0) && al16(a)));
  const long long total = (long long)N * H * W * (vec ? C / 4 : C);
  if (vec)
    hipLaunchKernelGGL(upsample2x_add_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, a, lda, b, ldb, y,
                       ldy, N, H, W, C);
  else
    hipLaunchKernelGGL(upsample2x_add_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, a, lda, b, ldb, y,
                       ldy, N, H, W, C);
  return check_launch();
}

int launch_eltwise(const EltArgs& a, hipStream_t s) {
  if (a.npix <= 0 || a.C <= 0 || a.a == nullptr || a.y == nullptr) return DH_EINVAL;
  if (a.op == 1 && a.b == nullptr) return DH_EINVAL;
  hipLaunchKernelGGL(eltwise_kernel, dim3(grid_for(a.npix * a.C)), dim3(256), 0, s, a);
  return check_launch();
}

int launch_copy_channels(const float* x, int ldx, float* y, int ldy, long long npix, int C, hipStream_t s) {
  if (npix <= 0 || C <= 0) return DH_EINVAL;
  const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && al16(x) && al16(y);
  const long long total = npix * (vec ? C / 4 : C);
  if (vec) hipLaunchKernelGGL(copy_channels_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, x, ldx, y, ldy, npix, C);
  else hipLaunchKernelGGL(copy_channels_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, x, ldx, y, ldy, npix, C);
  return check_launch();
}

int launch_zeropad(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, int PT, int PL,
                   hipStream_t s) {
  if (B <= 0 || C <= 0 || PT < 0 || PL < 0 || OH < H + PT || OW < W + PL) return DH_EINVAL;
  hipLaunchKernelGGL(zeropad_kernel, dim3(grid_for((long long)B * OH * OW * C)), dim3(256), 0, s, x, y, B, H, W, C,
                     OH, OW, PT, PL);
  return check_launch();
}

// ---- uint8 frames -> normalised fp32 (utils/transform.normalize_channels, transform.py:212-231) -------------
namespace {
__global__ __launch_bounds__(256) void normalize_u8_kernel(const unsigned char* __restrict__ x,
                                                           const float* __restrict__ lut, float* __restrict__ y,
                                                           long long total, int C) {
  __shared__ float s_lut[4 * 256];
  const int rows = C < 4 ? C : 4;
  for (int i = threadIdx.x; i < rows * 256; i += blockDim.x) s_lut[i] = lut[i];
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    y[i] = c < 4 ? s_lut[c * 256 + x[i]] : lut[c * 256 + x[i]];
  }
}
}  // namespace

int launch_normalize_u8(const unsigned char* x, const float* lut, float* y, long long n_pixels, int C, hipStream_t s) {
  if (n_pixels <= 0 || C <= 0) return DH_EINVAL;
  const long long total = n_pixels * C;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
  normalize_u8_kernel<<<blocks, 256, 0, s>>>(x, lut, y, total, C);
  return hipGetLastError() == hipSuccess ? DH_OK : DH_ELAUNCH;
}

}  // namespace dh
