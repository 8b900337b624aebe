// The LDS-tiled depthwise convolution (spatial.hip: dwconv_lds_kernel) as a device function, shared with the grouped launch
// of gemm1x1.hip (a residual unit's 1x1 shortcut convolution and the depthwise convolution of its main path read the same
// tensor and do not depend on each other: one launch runs both, conv_dw_group_kernel).
#pragma once
#include "dh_kernels.h"

namespace dh {
namespace dwl {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 min4(float4 a, float4 b) {
  return make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w));
}


// ------------------------------------------------------------------------------------------------
// LDS-tiled depthwise conv (W >= 8, C % 32 == 0): workgroup = (image, 32-channel chunk, full-width column tile),
// walking down bands of rows.  The halo'd input tile is fetched from global ONCE (prologue affine / ReLU / zero
// padding applied on the way in), every one of the KSxKS re-reads comes from LDS, the filter taps of the chunk sit in
// LDS too.  Thread = (channel quad, 8-column strip, row).
//
// The kernel is bound by VALU issue, not by HBM (round-3 instruction count: 400 packed FMAs per thread and band
// against 530 other VALU instructions), so everything around the FMAs is kept off the vector ALU:
//   * loads are buffer loads through a descriptor over ONE frame: per-slot byte offsets are computed once, a band
//     costs one v_add per load, and rows above / below the frame and halo columns left / right of it are out-of-range
//     offsets -- the load returns zeros, which is the padding (after ReLU; the affine variant masks explicitly);
//   * LDS tile addresses are computed once per thread (slots beyond the tile write one dump cell, unpredicated);
//   * stores are buffer stores with the column step in the scalar offset.
// ------------------------------------------------------------------------------------------------
constexpr int DW_CQ = 8;        // channel quads per workgroup (32 channels)
#ifndef DW_LDS_MIN_W
#define DW_LDS_MIN_W 8
#endif
constexpr int DW_PITCH = 9;     // float4 per tile pixel (8 + 1 pad: spreads strips over LDS banks)
constexpr int DW_OOB = (int)0x80000000u;   // + any band offset (< 2 GB, checked at launch) stays out of range

typedef unsigned int dw_u4 __attribute__((ext_vector_type(4)));
template <typename RSRC>
__device__ __forceinline__ float4 buf_ld4(RSRC rs, int voff) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#if defined(__HIP_DEVICE_COMPILE__)
  const dw_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
  r = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
#endif
  return r;
}
template <typename RSRC>
__device__ __forceinline__ void buf_st4(RSRC rs, int voff, int soff, float4 r) {
#if defined(__HIP_DEVICE_COMPILE__)
  dw_u4 v;
  v.x = __float_as_uint(r.x); v.y = __float_as_uint(r.y); v.z = __float_as_uint(r.z); v.w = __float_as_uint(r.w);
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
#endif
}
// relu as one integer max per element (see gemm1x1.hip: relu1)
__device__ __forceinline__ float dw_relu1(float v) {
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}
__device__ __forceinline__ float4 dw_relu4(float4 v) {
  return make_float4(dw_relu1(v.x), dw_relu1(v.y), dw_relu1(v.z), dw_relu1(v.w));
}

// MAXT / MAXN: load slots per thread for the KS - 1 rows above the first band / for the `rows` new rows of a band
// `block`: the work-group's index among the depthwise work-groups (the kernel's blockIdx.x, or its offset inside a grouped
// launch, gemm1x1.hip: conv_dw_group_kernel); `dsm`: the work-group's dynamic LDS; the NT threads 0 .. NT-1 of the
// work-group run it (all of them: it contains work-group barriers).
template <int KS, int NT, int MAXT, int MAXN, bool AFF, bool RELU>
__device__ __forceinline__ void dwconv_lds_body(const DwArgs& p, const int tw, const int rows, const int twh_magic,
                                                const int block, float4* dsm) {
  constexpr int SP = NT / DW_CQ;                    // tile pixels per load slot
  const int tid = threadIdx.x;
  const int chunks = p.C / 32;
  const int bands = (p.H + rows - 1) / rows;
  const int ctiles = (p.W + tw - 1) / tw;
  int b = block;
  const int cchunk = b % chunks; b /= chunks;
  const int ct = b % ctiles;
  const int n = b / ctiles;
  const int c0 = cchunk * 32, w0 = ct * tw;
  const int th = rows + KS - 1, twh = tw + KS - 1;
  float4* wts = dsm;                                // [KS*KS][DW_CQ]
  float4* tile = dsm + KS * KS * DW_CQ;             // ring of th rows: [th][twh][DW_PITCH], + one dump cell
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

  // [r06] Every load the work-group needs before its first FMA -- filter taps, prologue affine, the KS - 1 rows above the first
  // band, the first band's rows -- is ISSUED before the first one is consumed: one memory round trip in front of the arithmetic
  // instead of three (taps -> LDS, top rows -> LDS, band 0 -> LDS).  At a couple of clips per call a depthwise launch is a single
  // band on a handful of CUs and those round trips were a third of its 9-13 us; at a throughput batch the other work-group of
  // the CU covered them.  (Band k + 1 is still NOT requested under band k's arithmetic: see below.)
  constexpr int WPT = (KS * KS * DW_CQ + NT - 1) / NT;
  float4 wreg[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int i = tid + k * NT;
    const int ic = i < KS * KS * DW_CQ ? i : 0;
    wreg[k] = ld4(p.w + (size_t)(ic / DW_CQ) * p.C + c0 + (ic % DW_CQ) * 4);
  }
  const int q = tid & (DW_CQ - 1);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = zero;
  if constexpr (AFF) { sc = ld4(p.pre_scale + c0 + q * 4); sh = ld4(p.pre_shift + c0 + q * 4); }

  // up_in: x is stored at HALF resolution, [N, H / 2, W / 2, C], and read as UpSampling2D((2, 2))(x) -- window pixel (r, c)
  // is x(r >> 1, c >> 1); rows / columns outside the full-resolution frame stay out-of-range offsets (an arithmetic shift
  // keeps a negative row negative, a row >= H lands behind the half-resolution frame)
  const int ush = p.up_in ? 1 : 0;
  const int hw_in = (p.H >> ush) * (p.W >> ush), w_in = p.W >> ush;
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x) + (size_t)n * hw_in * p.ldx, 0,
      (int)(((unsigned)(hw_in - 1) * p.ldx + (unsigned)p.C) * 4u), 0x00020000);
  const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(
      p.y + (size_t)n * p.H * p.W * p.ldy, 0, (int)(((unsigned)(p.H * p.W - 1) * p.ldy + (unsigned)p.C) * 4u), 0x00020000);
  const int row_bytes = twh * DW_PITCH * 16, ring_bytes = th * row_bytes;
  const int dump = ring_bytes;                      // byte offset (from `tile`) of the cell that swallows unused slots
  // px / twh for px < 1024 as multiply + shift (twh_magic = 65536 / twh + 1): an integer divide is ~40 VALU instructions
  auto row_of = [&](int px) { return (px * twh_magic) >> 16; };
  auto px_offset = [&](int px, int first_row) {     // frame byte offset of window pixel px, window starting at first_row
    const int tr = row_of(px), tc = px - tr * twh;
    const int iw = w0 - p.PL + tc;
    return (unsigned)iw < (unsigned)p.W ? ((((first_row + tr) >> ush) * w_in + (iw >> ush)) * p.ldx + c0 + q * 4) * 4 : DW_OOB;
  };

  // ---- the KS - 1 rows above the first band (input rows -PT .. KS - 2 - PT) go to ring rows 0 .. KS - 2, once: requested here,
  // written to the ring below (behind the first band's requests)
  float4 top[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int px = (tid >> 3) + j * SP;
    top[j] = buf_ld4(rs_x, px < (KS - 1) * twh ? px_offset(px, -p.PT) : DW_OOB);
  }
  // ---- every band then brings in only its `rows` NEW rows (window rows KS - 1 .. th - 1 = input rows r0 + KS - 1 - PT
  // ...): the KS - 1 rows it shares with the band above stay where they are.  Input row i lives in ring row (i + PT) % th.
  // Slot j of this thread = pixel (tid >> 3) + j * SP of the new-row window, quad q; its frame offset for band 0 and its
  // ring offset for band 0 are computed once, a band adds one scalar to each (and wraps the ring offset).
  int voff[MAXN], loff[MAXN];
  int trow[AFF ? MAXN : 1];                           // affine variant: the slot's window row, -1 = never valid
#pragma unroll
  for (int j = 0; j < MAXN; ++j) {
    const int px = (tid >> 3) + j * SP;
    const bool in = px < rows * twh;
    voff[j] = in ? px_offset(px, KS - 1 - p.PT) : DW_OOB;
    loff[j] = in ? (KS - 1) * row_bytes + (px * DW_PITCH + q) * 16 : -1;
    if constexpr (AFF) trow[j] = (in && voff[j] != DW_OOB) ? row_of(px) : -1;
  }
  // The loads of a band are issued back to back (one round trip per band) and are NOT overlapped with the previous
  // band's arithmetic: with the next band in flight during the FMAs the kernel is 6 % faster in isolation (62 vs 66 us on
  // 64 x 32 x 32 x 576), but on most boxes of the pool the firmware then drops the engine clock of the WHOLE forward by
  // 5 % (2213 vs 2338 MHz; the GEMMs lose 6 %, the step 4 %) -- profiles/r03_dvfs_study.md.  The other work-group of the
  // CU covers the wait.
  float4 stage[MAXN];
  auto fetch = [&](int band) {
    const int boff = band * (rows >> ush) * w_in * p.ldx * 4;          // (up_in: bands of an even number of rows, see launch)
#pragma unroll
    for (int j = 0; j < MAXN; ++j) stage[j] = buf_ld4(rs_x, voff[j] + boff);
  };
  fetch(0);
  // ---- now the LDS side of the prologue: filter taps, then the top rows (prologue affine / ReLU / zero padding on the way in)
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int i = tid + k * NT;
    if (i < KS * KS * DW_CQ) wts[i] = wreg[k];
  }
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int px = (tid >> 3) + j * SP;
    float4 v = top[j];
    if constexpr (AFF) {
      const int tr = row_of(px), iw = w0 - p.PL + px - tr * twh;
      v = fma4(v, sc, sh);
      if (!((unsigned)(tr - p.PT) < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)) v = zero;
    }
    if constexpr (RELU) v = dw_relu4(v);
    if (px < (KS - 1) * twh) tile[px * DW_PITCH + q] = v;
  }
  const int strips = tw >> 3;
  const int strip = (tid >> 3) % strips;
  const int row = (tid >> 3) / strips;
  const int y_off = ((row * p.W + w0 + strip * 8) * p.ldy + c0 + q * 4) * 4;
  const int y_col = p.ldy * 4;
  char* const tile_b = reinterpret_cast<char*>(tile);

  int ring0 = 0;                                    // ring row of this band's window row 0 (= r0 % th)
  for (int band = 0; band < bands; ++band) {
    const int r0 = band * rows;
    const int shift = ring0 * row_bytes;
    if (band > 0) fetch(band);
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      float4 v = stage[j];
      if constexpr (AFF) {
        v = fma4(v, sc, sh);
        if (!(trow[j] >= 0 && (unsigned)(r0 + KS - 1 - p.PT + trow[j]) < (unsigned)p.H)) v = zero;   // pad AFTER the affine
      }
      if constexpr (RELU) v = dw_relu4(v);
      int o = loff[j] + shift;
      o = o >= ring_bytes ? o - ring_bytes : o;
      o = loff[j] < 0 ? dump : o;
      *reinterpret_cast<float4*>(tile_b + o) = v;
    }
    __syncthreads();

    if (row < rows && r0 + row < p.H && w0 + strip * 8 < p.W) {   // (W % 8 == 0: a strip is inside the frame or outside)
      float4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = zero;
      int rr = ring0 + row;                          // ring row of window row (row + kh)
      rr = rr >= th ? rr - th : rr;
#pragma unroll 1                                   // one kernel row of LDS reads in flight: keeps 2+ waves per SIMD
      for (int kh = 0; kh < KS; ++kh) {
        float4 wv[KS];
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) wv[kw] = wts[(kh * KS + kw) * DW_CQ + q];
        const float4* src = tile + (rr * twh + strip * 8) * DW_PITCH + q;
#pragma unroll
        for (int j = 0; j < 8 + KS - 1; ++j) {
          const float4 v = src[j * DW_PITCH];
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const int o = j - kw;
            if (o >= 0 && o < 8) acc[o] = fma4(v, wv[kw], acc[o]);
          }
        }
        rr = rr + 1 == th ? 0 : rr + 1;
      }
      const int yo = y_off + r0 * p.W * p.ldy * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) buf_st4(rs_y, yo, i * y_col, acc[i]);
    }
    __syncthreads();                                // every wave is done reading the rows the next band overwrites
    ring0 += rows;
    ring0 = ring0 >= th ? ring0 - th : ring0;
  }
}

// Launch geometry of the LDS-tiled kernel for a layer, or false when the layer runs on another depthwise kernel:
// column tile = min(W, 32), 8-column strips; the work-group is sized to the map --
//   W >= 32: 256 threads = 8 channel quads x 4 strips x 8 rows, bands of 8 rows walked inside the work-group
//   W == 16: 256 threads x 16 rows, one band (128 threads x 8 rows, two bands, 38 KB of LDS: 19.4 vs 18.0 us, not used)
//   W ==  8:  64 threads x  8 rows: the whole 8 x 8 map of a 32-channel chunk in one band (10.7 us; the register
//             kernel these maps ran on before took 13.0)
struct DwLdsGeom {
  int tw, rows, nt;
  unsigned blocks;
  size_t lds;
};
inline bool dw_lds_geometry(const DwArgs& a, DwLdsGeom& g) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool vec = (a.C % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) && al16(a.x) && al16(a.y) && al16(a.w) &&
                   (a.pre_scale == nullptr || (al16(a.pre_scale) && al16(a.pre_shift)));
  if (!(vec && a.KH == a.KW && (a.KW == 5 || a.KW == 3) && a.C % 32 == 0 && a.W >= DW_LDS_MIN_W && a.W % 8 == 0)) return false;
  g.tw = a.W >= 32 ? 32 : a.W;
  g.nt = g.tw == 8 ? 64 : 256;
  g.rows = g.nt / (8 * (g.tw / 8));
  if (g.rows > a.H) g.rows = a.H;                                    // (8 or 16, or the even H of an up-sampled input: even)
  const long long blocks = (long long)a.N * ((a.W + g.tw - 1) / g.tw) * (a.C / 32);    // bands are walked inside
  g.lds = ((size_t)(g.rows + a.KW - 1) * (g.tw + a.KW - 1) * DW_PITCH + (size_t)a.KW * a.KW * DW_CQ + 1) * 16;
  const bool fits31 = (long long)a.H * a.W * a.ldx * 4 < 0x7fffffffLL && (long long)a.H * a.W * a.ldy * 4 < 0x7fffffffLL;
  if (blocks > 0x7fffffffLL || !fits31) return false;
  g.blocks = (unsigned)blocks;
  // slots per thread: 256 threads = 32 tile pixels per slot (5 x 36-pixel rows above, 8 x 36 or 16 x 20 new pixels per
  // band), 64 threads = 8 pixels per slot (4 x 12 above, 8 x 12 new)
  const int twh = g.tw + a.KW - 1, maxt = a.KW == 5 ? (g.nt == 256 ? 5 : 6) : 3, maxn = a.KW == 5 ? (g.nt == 256 ? 10 : 12) : 10;
  return (a.KW - 1) * twh <= maxt * (g.nt / DW_CQ) && g.rows * twh <= maxn * (g.nt / DW_CQ);
}

}  // namespace dwl
}  // namespace dh
