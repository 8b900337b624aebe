// Shared pieces of the two MFMA convolution kernels (conv_igemm.hip: general implicit GEMM with register
// staging; gemm1x1.hip: pointwise GEMM with LDS-DMA double buffering).
#pragma once
#include <type_traits>

#include "dh_kernels.h"

namespace dh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Epilogue traffic is streamed once (residual tiles in, output tile out), so it goes through non-temporal
// accesses: it does not allocate in the caches and the operand tiles that other work-groups are about to re-read
// stay resident.  Measured on the 65536 x 576 x 576 GEMM: 4-7 % faster (382 -> 356..368 us), thin GEMMs up to 11 %.
__device__ __forceinline__ void st4_stream(float* ptr, float4 v) {
  f32x4v t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4v*>(ptr));
}
__device__ __forceinline__ float4 ld4_stream(const float* ptr) {
  const f32x4v t = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(ptr));
  return make_float4(t.x, t.y, t.z, t.w);
}

// Streaming dword accesses through a buffer descriptor: per-lane byte offset in a VGPR, the wave-uniform part of the
// address in the SCALAR offset -- an access costs no vector-ALU address arithmetic.  aux = 2: non-temporal.
// (The builtins only exist in the device pass.)
template <typename RSRC>
__device__ __forceinline__ float buf_ld1_stream(RSRC rs, int voff, int soff) {
  float r = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
  r = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 2));
#endif
  return r;
}
template <typename RSRC>
__device__ __forceinline__ void buf_st1_stream(RSRC rs, int voff, int soff, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 2);
#endif
}
__device__ __forceinline__ float& f4c(float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// Launch-side half of the direct epilogue: `epi` (1 = 16-byte epilogue possible) gains bit 1 when interior tiles may
// store straight from the accumulators -- no up-sampling, no pooled second output, a second residual only at half
// resolution, and every byte offset of y / res1 / res2 inside 31 bits (buffer offsets are 32-bit).
inline int epi_with_direct(const ConvArgs& a, int epi, bool half_res_residual = true) {
  const long long M = (long long)a.N * a.OH * a.OW;
  const int ohw = a.OH * a.OW;
  // a second residual only in its half-resolution form, on maps where a wave's 32 pixels sit in one frame, start at an
  // even image row and split into shifts: 2^a x 2^b >= 32 pixels, at least 8 wide
  const bool r2ok = a.res2 == nullptr || (half_res_residual && a.res2_down && (a.OW & (a.OW - 1)) == 0 && (ohw & (ohw - 1)) == 0 && a.OW >= 8 &&
                                          ohw >= 32 && (M / 4) * a.ldr2 * 4 < 0x7fffffffLL);
  const bool direct = epi && !a.up2 && a.y_pool == nullptr && r2ok && M * a.ldy * 4 < 0x7fffffffLL &&
                      (a.res1 == nullptr || M * a.ldr1 * 4 < 0x7fffffffLL);
  return epi | (direct ? 2 : 0);
}

// XCD-aware bijective remap: block b runs on XCD b%8; give each XCD a contiguous run of tiles so the
// workgroups that share one activation tile (consecutive N-tiles) hit the same L2.
__device__ __forceinline__ int xcd_tile(int b, int nwg) {
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// Epilogue.  C/D layout of a 32x32 MFMA tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Each wave stages one 32 x (TN*32) row block through its private LDS slab and reads it back row-wise, so
// the BN affine / residual loads / stores are 16-byte wide and whole output rows are contiguous.
//   out = relu?( acc*post_scale + post_shift + res1[m] + res2[mo] ), optionally written 2x up-sampled, or with
//   res2 read at half resolution (res2_down: the UpSampling2D sits on the residual instead of on the result).
//   Optionally also writes the 2x2 max-pooled output (y_pool).
//
// The epilogue is latency-, not bandwidth-bound: a work-group that loads its residual tile only after the last
// MFMA sits through one HBM round trip per dependent batch while the other work-groups of the CU -- started
// together, same amount of work -- sit in the same phase, so nothing fills the matrix pipe.  Hence:
//   * EpiPrefetch::issue() puts ALL residual (res1) loads of the tile in flight (registers) and is called by the
//     K loop one step before its end, so the round trip overlaps the last MFMA block;
//   * the row loop is fully unrolled, branch-free (clamped addresses, predicated stores): every remaining load
//     of a row block is issued before the first use;
//   * only the first LDS staging needs a work-group barrier (the stage buffers are being re-used); the slab is
//     wave-private after that.
template <int TM, int TN>
struct EpiPrefetch {
  static constexpr int ROW4 = TN * 8;      // float4 per staged row
  static constexpr int IT = 4 * TN;        // row-loop iterations per 32-row block (32*ROW4 / 64 lanes)
  // holding a whole residual tile costs 4*IT*TM VGPRs across the last K-step: only the one-row-block tilings
  // (128x32 .. 128x96 per work-group) have them to spare
  static constexpr bool kEnabled = (TM == 1);
  float4 r1[kEnabled ? TM : 1][kEnabled ? IT : 1];
  float dsc[kEnabled ? TN : 1], dsh[kEnabled ? TN : 1];   // direct epilogue: this lane's BN scale / shift per column tile

  // epi_vec bit 1 (set by the launcher: no up-sampling, no pooled output, one residual at most, 31-bit offsets) + a tile
  // that lies fully inside the output: the epilogue runs straight from the accumulators (conv_epilogue, direct path)
  template <int WM, int WN>
  static __device__ __forceinline__ bool direct_tile(const ConvArgs& p, int m0, int n0, int M, int epi_vec) {
    return (epi_vec & 2) != 0 && m0 + WM * TM * 32 <= M && n0 + WN * TN * 32 <= p.Cout;
  }

  template <int WM, int WN>
  __device__ __forceinline__ void issue(const ConvArgs& p, int m0, int n0, int M, int epi_vec) {
    if constexpr (kEnabled) {
      if (direct_tile<WM, WN>(p, m0, n0, M, epi_vec)) {
        // accumulator layout: register r of lane (li, lh) is row (r & 3) + 8 * (r >> 2) + 4 * lh, column li of the tile
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int li = lane & 31, lh = lane >> 5;
        const int ncol = n0 + wn * TN * 32 + li;
        if (p.post_scale != nullptr) {
#pragma unroll
          for (int j = 0; j < TN; ++j) { dsc[j] = p.post_scale[ncol + j * 32]; dsh[j] = p.post_shift[ncol + j * 32]; }
        }
        if (p.res1 == nullptr) return;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res1), 0,
                                                          (int)(((unsigned)(M - 1) * p.ldr1 + (unsigned)p.Cout) * 4u), 0x00020000);
        const int vo = ((m0 + wm * TM * 32 + 4 * lh) * p.ldr1 + ncol) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              f4c(r1[i][j * 4 + (r >> 2)], r & 3) = buf_ld1_stream(rs, vo, ((i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldr1 + j * 32) * 4);
        return;
      }
      if (epi_vec == 0 || p.res1 == nullptr) return;
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int wm = wave / WN, wn = wave % WN;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int f = lane + 64 * it;
          const int row = f / ROW4, c4 = f - row * ROW4;
          int m = m0 + (wm * TM + i) * 32 + row;
          int n = n0 + wn * TN * 32 + c4 * 4;
          m = m < M ? m : M - 1;
          n = n < p.Cout ? n : p.Cout - 4;
          r1[i][it] = ld4_stream(p.res1 + (size_t)m * p.ldr1 + n);
        }
    }
  }
};

// Tilings whose epilogue can also write the 2x2 max-pooled output (dh_conv_args.y_pool): every wave owns ONE 32-row
// block = one image row at OW == 32, and the waves come in (even, odd) pairs over M.
template <int WM, int TM, bool UP2>
constexpr bool conv_epilogue_pools() { return TM == 1 && WM % 2 == 0 && !UP2; }
// [r06] OW == 16 / OW == 8: a wave's 32-row block is two / four WHOLE image rows (OH * OW a multiple of 32, so a block
// never straddles frames and starts at an even row) -- the wave pools its own slab, no partner needed: any TM == 1 tiling.
template <int TM, bool UP2>
constexpr bool conv_epilogue_pools_in_wave() { return TM == 1 && !UP2; }
// launch-side: may this tiling serve a.y_pool?
template <int WM, int TM, bool UP2>
inline bool conv_epilogue_pools_for(const ConvArgs& a) {
  return a.OW == 32 ? conv_epilogue_pools<WM, TM, UP2>() : conv_epilogue_pools_in_wave<TM, UP2>();
}

struct EpiNoHook {
  __device__ __forceinline__ void operator()() const {}
};

// PRE: the caller issued EpiPrefetch::issue() (and the tiling holds a prefetched tile)
// `staged` runs once the last row block's accumulators sit in the LDS slab (their registers are free from there on):
// a kernel that finishes its tile in several column slices puts the next slice's residual prefetch there.
// CHROWS > 0: the row loop of the LDS-slab path runs in chunks of CHROWS iterations, each chunk loading ITS residual rows
// first (instead of every residual row of the block before the first store): a kernel whose accumulators leave few
// registers (the split-bf16 wide tiling holds 96 of them for its other column slice) does not spill.
template <int WM, int WN, int TM, int TN, bool UP2, bool PRE, typename HOOK = EpiNoHook, bool DIRECT_R2 = true, int CHROWS = 0>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][TN], float* smem, int m0,
                                              int n0, int M, int epi_vec, const EpiPrefetch<TM, TN>& pre,
                                              HOOK staged = HOOK()) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int LDC = TN * 32 + 4;
  constexpr int ROW4 = TN * 8;
  constexpr int IT = 4 * TN;
  constexpr int NSC = (64 % ROW4 == 0) ? 1 : 3;   // distinct column groups a lane meets over the row loop
  constexpr bool kPre = PRE && EpiPrefetch<TM, TN>::kEnabled;
  constexpr bool kPool = conv_epilogue_pools<WM, TM, UP2>();
  constexpr bool kPoolWave = conv_epilogue_pools_in_wave<TM, UP2>();
  if constexpr (!UP2 && kPre && std::is_same<HOOK, EpiNoHook>::value) {
    if (EpiPrefetch<TM, TN>::template direct_tile<WM, WN>(p, m0, n0, M, epi_vec)) {
      // Direct path (interior tiles of plain launches): no LDS staging, no work-group barrier, no wait between the last
      // MFMA and the first store.  Register r of a lane is one row of one column: a store instruction writes two
      // 128-byte row segments, BN scale / shift are one value per lane and column tile, the row step is a scalar offset.
      const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(((unsigned)(M - 1) * p.ldy + (unsigned)p.Cout) * 4u),
                                                          0x00020000);
      const int ncol = n0 + wn * TN * 32 + li;
      const int vo = ((m0 + wm * TM * 32 + 4 * lh) * p.ldy + ncol) * 4;
      float sc[TN], sh[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        sc[j] = pre.dsc[j];
        sh[j] = pre.dsh[j];
      }
      // Second residual at HALF resolution (res2_down; the launcher only sets the flag on maps of 2^a x 2^b >= 32 pixels
      // with OW >= 8): the wave's 32 pixels start at an even image row, so with t = (r & 3) + 8 * (r >> 2) + 4 * lh the
      // source pixel splits into a wave-uniform part per register r and 2 * lh * ldr2 per lane -- again no vector ALU.
      // Registers r and r ^ 1 of a lane read the same source pixel: 8 loads per column tile.
      float r2[DIRECT_R2 ? TN : 1][8];
      if (DIRECT_R2 && p.res2 != nullptr) {                                   // (uniform) direct tiles carry res2 only as res2_down
        const auto rs_2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.res2), 0, (int)(((unsigned)((M >> 2) - 1) * p.ldr2 + (unsigned)p.Cout) * 4u), 0x00020000);
        const int ohw = p.OH * p.OW, ow_sh = __ffs(p.OW) - 1;
        const int mb = __builtin_amdgcn_readfirstlane(m0 + wm * 32);
        const int fr = mb >> (__ffs(ohw) - 1), pos = mb & (ohw - 1);
        const int oh0 = pos >> ow_sh, ow0 = pos & (p.OW - 1);
        const int v2 = (2 * lh * p.ldr2 + ncol) * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            const int tr = 2 * (h & 1) + 8 * (h >> 1);              // rows (2h, 2h + 1) -> registers r = 2h, 2h + 1
            const int oh = oh0 + (tr >> ow_sh), ow = ow0 + (tr & (p.OW - 1));
            const int src = (fr * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1);
            r2[j][h] = buf_ld1_stream(rs_2, v2, (src * p.ldr2 + j * 32) * 4);
          }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float t = acc[i][j][r];
            if (p.post_scale != nullptr) t = t * sc[j] + sh[j];
            if (p.res1 != nullptr) t += f4c(pre.r1[i][j * 4 + (r >> 2)], r & 3);
            if (DIRECT_R2 && p.res2 != nullptr) t += r2[DIRECT_R2 ? j : 0][r >> 1];
            if (p.post_relu) t = fmaxf(t, 0.f);
            buf_st1_stream(rs_y, vo, ((i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldy + j * 32) * 4, t);
          }
      return;
    }
  }
  float* sC = smem + wave * 32 * LDC;
  const int ohw = p.OH * p.OW;
  const bool vec = epi_vec != 0;
  const bool pow2 = (p.OW & (p.OW - 1)) == 0 && (ohw & (ohw - 1)) == 0;      // (uniform) maps of 2^a x 2^b pixels: shifts
  const int ow_sh = __ffs(p.OW) - 1, ohw_sh = __ffs(ohw) - 1;
  __syncthreads();                            // every wave is done with the operand stages
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slab reads of the previous block
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sC[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + j * 32 + li] = acc[i][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // wave-private slab: no barrier needed
    if (i == TM - 1) staged();
    if (vec) {
      // pass 1: every load of this row block that is not in flight already
      float4 rr[kPre ? 1 : IT], sc[NSC], sh[NSC];
      int ncol[NSC];
#pragma unroll
      for (int q = 0; q < NSC; ++q) {
        const int c4 = (lane + 64 * q) % ROW4;
        const int n = n0 + wn * TN * 32 + c4 * 4;
        ncol[q] = n < p.Cout ? n : p.Cout - 4;
        if (p.post_scale != nullptr) {
          sc[q] = *reinterpret_cast<const float4*>(p.post_scale + ncol[q]);
          sh[q] = *reinterpret_cast<const float4*>(p.post_shift + ncol[q]);
        }
      }
      if constexpr (!kPre && CHROWS == 0) {
        if (p.res1 != nullptr) {
#pragma unroll
          for (int it = 0; it < IT; ++it) {
            const int row = (lane + 64 * it) / ROW4;
            const int m = m0 + (wm * TM + i) * 32 + row;
            rr[it] = ld4_stream(p.res1 + (size_t)(m < M ? m : M - 1) * p.ldr1 + ncol[it % NSC]);
          }
        }
      }
      // second residual (read at OUTPUT resolution): also loaded before the first store -- the compiler must
      // assume y may alias res2 and would otherwise keep every load behind the previous iteration's store, one
      // memory round trip per row.  Up-sampling writes (and reads) 4 positions per row: chunks of CH rows.
      constexpr int CH = UP2 ? (IT < 4 ? IT : 4) : (CHROWS > 0 && CHROWS < IT ? CHROWS : IT);     // rows per chunk: <= 16 float4 of res2 in flight
      static_assert(IT % CH == 0, "chunks of equal size");
      constexpr int ND = UP2 ? 4 : 1;
#pragma unroll
      for (int c0 = 0; c0 < IT; c0 += CH) {
        float4 r2[CH][ND];
        size_t mo[CH][ND];
        if constexpr (!kPre && CHROWS > 0) {             // this chunk's first-residual rows
          if (p.res1 != nullptr) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
              const int row = (lane + 64 * (c0 + u)) / ROW4;
              const int m = m0 + (wm * TM + i) * 32 + row;
              rr[c0 + u] = ld4_stream(p.res1 + (size_t)(m < M ? m : M - 1) * p.ldr1 + ncol[(c0 + u) % NSC]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int it = c0 + u;
          const int row = (lane + 64 * it) / ROW4;
          const int m = m0 + (wm * TM + i) * 32 + row;
          const int mc = m < M ? m : M - 1;
          if constexpr (UP2) {
            int fr, oh, ow;
            if (pow2) {
              fr = mc >> ohw_sh;
              const int rem = mc & (ohw - 1);
              oh = rem >> ow_sh;
              ow = rem & (p.OW - 1);
            } else {
              fr = mc / ohw;
              const int rem = mc - fr * ohw;
              oh = rem / p.OW;
              ow = rem - oh * p.OW;
            }
#pragma unroll
            for (int d = 0; d < 4; ++d)
              mo[u][d] = ((size_t)fr * 2 * p.OH + 2 * oh + (d >> 1)) * (2 * p.OW) + 2 * ow + (d & 1);
          } else {
            mo[u][0] = (size_t)mc;
          }
          if (p.res2 != nullptr) {
            if (!UP2 && p.res2_down) {
              // res2 lives at HALF resolution: out(oh, ow) += res2(oh / 2, ow / 2) = add([., UpSampling2D(res2)]); the four
              // pixels that share a source row hit it in L2 (rows of a tile are consecutive pixels)
              int fr, oh, ow;
              if (pow2) {                                    // no integer divide on this hardware: ~40 VALU each, and the
                fr = mc >> ohw_sh;                           // epilogue's VALU is paid in full beside the other waves' MFMAs
                const int rem = mc & (ohw - 1);
                oh = rem >> ow_sh;
                ow = rem & (p.OW - 1);
              } else {
                fr = mc / ohw;
                const int rem = mc - fr * ohw;
                oh = rem / p.OW;
                ow = rem - oh * p.OW;
              }
              const size_t mr = ((size_t)fr * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1);
              r2[u][0] = *reinterpret_cast<const float4*>(p.res2 + mr * p.ldr2 + ncol[it % NSC]);
            } else {
#pragma unroll
              for (int d = 0; d < ND; ++d)
                r2[u][d] = ld4_stream(p.res2 + mo[u][d] * p.ldr2 + ncol[it % NSC]);
            }
          }
        }
        // combine + store
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int it = c0 + u;
          const int f = lane + 64 * it;
          const int row = f / ROW4, c4 = f - row * ROW4;
          const int m = m0 + (wm * TM + i) * 32 + row;
          const int nc = ncol[it % NSC];
          const bool ok = m < M && n0 + wn * TN * 32 + c4 * 4 < p.Cout;
          float4 t = *reinterpret_cast<const float4*>(&sC[row * LDC + c4 * 4]);
          if (p.post_scale != nullptr) {
            const float4 a = sc[it % NSC], b = sh[it % NSC];
            t.x = t.x * a.x + b.x; t.y = t.y * a.y + b.y; t.z = t.z * a.z + b.z; t.w = t.w * a.w + b.w;
          }
          if (p.res1 != nullptr) {
            float4 r;
            if constexpr (kPre) r = pre.r1[i][it]; else r = rr[it];
            t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
          }
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            float4 o = t;
            if (p.res2 != nullptr) { o.x += r2[u][d].x; o.y += r2[u][d].y; o.z += r2[u][d].z; o.w += r2[u][d].w; }
            if (p.post_relu) {
              o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            if (ok) st4_stream(p.y + mo[u][d] * p.ldy + nc, o);
            if constexpr (kPool || kPoolWave) {                    // keep the final values: they are pooled below
              if (p.y_pool != nullptr) *reinterpret_cast<float4*>(&sC[row * LDC + c4 * 4]) = o;
            }
          }
        }
      }
    } else {
      for (int f = lane; f < 32 * ROW4; f += 64) {
        const int row = f / ROW4, c4 = f - row * ROW4;
        const int m = m0 + (wm * TM + i) * 32 + row;
        const int n = n0 + wn * TN * 32 + c4 * 4;
        if (m >= M || n >= p.Cout) continue;
        const float4 v = *reinterpret_cast<const float4*>(&sC[row * LDC + c4 * 4]);
        size_t mo[4];
        int nout = 1;
        if constexpr (UP2) {
          const int fr = m / ohw;
          const int rem = m - fr * ohw;
          const int oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
          for (int d = 0; d < 4; ++d)
            mo[d] = ((size_t)fr * 2 * p.OH + 2 * oh + (d >> 1)) * (2 * p.OW) + 2 * ow + (d & 1);
          nout = 4;
        } else {
          mo[0] = (size_t)m;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= p.Cout) break;
          float t = (&v.x)[e];
          if (p.post_scale != nullptr) t = t * p.post_scale[n + e] + p.post_shift[n + e];
          if (p.res1 != nullptr) t += p.res1[(size_t)m * p.ldr1 + n + e];
          for (int d = 0; d < nout; ++d) {
            float o = t;
            if (p.res2 != nullptr) {
              size_t mr = mo[d];
              if (!UP2 && p.res2_down) {
                const int fr = m / ohw;
                const int rem = m - fr * ohw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                mr = ((size_t)fr * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1);
              }
              o += p.res2[mr * p.ldr2 + n + e];
            }
            if (p.post_relu) o = fmaxf(o, 0.f);
            p.y[mo[d] * p.ldy + n + e] = o;
          }
        }
      }
    }
  }
  if constexpr (kPool) {
    // MaxPooling2D((2, 2)) of what was just stored: the slabs now hold the FINAL values of the work-group's image rows
    // (OW == 32: one row per wave); waves (2i, 2i+1) over M pool their two rows, 128 threads per pair
    if (p.y_pool != nullptr && vec && p.OW == 32) {
      __syncthreads();
      const float* sE = smem + ((wm & ~1) * WN + wn) * 32 * LDC;
      const float* sO = sE + WN * 32 * LDC;
      const int r_img = m0 / 32 + (wm & ~1);                    // image row of the even wave (OW == 32)
      if ((r_img + 2) * 32 <= M) {
        const int fr = r_img / p.OH, oh = r_img - fr * p.OH;
        const size_t pp = ((size_t)fr * (p.OH / 2) + oh / 2) * 16;
        for (int idx = (wm & 1) * 64 + lane; idx < 16 * ROW4; idx += 128) {
          const int q = idx / ROW4, c4 = idx - q * ROW4;
          const int n = n0 + wn * TN * 32 + c4 * 4;
          if (n >= p.Cout) continue;
          const float4 a = *reinterpret_cast<const float4*>(&sE[(2 * q) * LDC + c4 * 4]);
          const float4 b = *reinterpret_cast<const float4*>(&sE[(2 * q + 1) * LDC + c4 * 4]);
          const float4 c = *reinterpret_cast<const float4*>(&sO[(2 * q) * LDC + c4 * 4]);
          const float4 d = *reinterpret_cast<const float4*>(&sO[(2 * q + 1) * LDC + c4 * 4]);
          float4 t;
          t.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)); t.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
          t.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)); t.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
          *reinterpret_cast<float4*>(p.y_pool + (pp + q) * p.ldyp + n) = t;
        }
      }
      __syncthreads();      // a kernel that runs this epilogue again through the same slabs (the split wide tiling's second
                            // column slice) must not stage over rows its partner wave is still pooling
    }
  }
  if constexpr (kPoolWave) {
    // [r06] the same at OW == 16 / 8 (SPNet's down path below 32 x 32, common.py:70-86; the hourglass's inner levels): the
    // wave's slab holds 2 / 4 whole image rows = 8 pooled pixels; wave-private, so no barrier -- only the slab writes above
    // have to have landed
    if (p.y_pool != nullptr && vec && p.OW < 32) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int mb = m0 + wm * 32;
      if (mb + 32 <= M) {
        const int pw_sh = __ffs(p.OW) - 2;                       // log2(OW / 2)
        const int row0 = mb >> (pw_sh + 1);                      // image row of the block's first pixel (even)
        const int fr = row0 / p.OH, oh = row0 - fr * p.OH;
        const size_t pp = ((size_t)fr * (p.OH >> 1) + (oh >> 1)) << pw_sh;     // first pooled pixel: the 8 follow row-major
        for (int idx = lane; idx < 8 * ROW4; idx += 64) {
          const int q = idx / ROW4, c4 = idx - q * ROW4;
          const int n = n0 + wn * TN * 32 + c4 * 4;
          if (n >= p.Cout) continue;
          const int pr = q >> pw_sh, pc = q & ((1 << pw_sh) - 1);
          const float* s0 = sC + ((2 * pr) * p.OW + 2 * pc) * LDC + c4 * 4;
          const float4 a = *reinterpret_cast<const float4*>(s0);
          const float4 b = *reinterpret_cast<const float4*>(s0 + LDC);
          const float4 c = *reinterpret_cast<const float4*>(s0 + p.OW * LDC);
          const float4 d = *reinterpret_cast<const float4*>(s0 + (p.OW + 1) * LDC);
          float4 t;
          t.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)); t.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
          t.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)); t.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
          *reinterpret_cast<float4*>(p.y_pool + (pp + q) * p.ldyp + n) = t;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the split wide tiling stages its second slice into this slab next)
    }
  }
}

}  // namespace dh
