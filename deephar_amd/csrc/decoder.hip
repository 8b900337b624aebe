// Heat-map -> key-point decoder kernels for gfx950 (the op the reference implements as a frozen
// depthwise convolution with a whole-map kernel: deephar/layers.py:160-200, activations.py:3-30,
// models/blocks.py:217-343, models/reception.py:167-222), the heat-map weighted feature pooling
// (layers.py:478-508) and the tiny action-head reductions (layers.py:428-442, action.py:14-17).
//
// Heat-maps are NHWC: a pixel's C channels are contiguous, so lanes run over channels (coalesced) and
// the spatial reduction runs over pixel-lanes: in-register partials -> wavefront __shfl_xor butterflies ->
// one LDS hop across the workgroup's waves.
#include "dh_kernels.h"

namespace dh {
namespace {

// The (frame, channel group) work-groups of ONE frame read the same cache lines (a pixel's channels are contiguous: a group
// of 4-16 channels touches every 128-byte line of the frame's maps), so they belong on one XCD's L2: the kernels below take
// their pair from dh::xcd_order (dh_kernels.h).  [r06: the per-launch PMC scan of the MPII forward
// (profiles/r06_pmc_all_launches.md) showed the fused decoder fetching 4.0x its maps -- its four joint quads of a frame sat on
// four XCDs.]  A pure re-mapping of work-groups: no result bit moves.

constexpr int CG = 16;             // channels per workgroup (depth_from_maps; soft-argmax when there is enough work)
constexpr int NTH = 256;           // threads per workgroup
constexpr int NW = NTH / 64;       // waves

// Reduce across the lanes of a wave that share the same (tid % G): xor G, 2G, .. 32.
template <int G>
__device__ __forceinline__ float wave_max_g(float v) {
#pragma unroll
  for (int d = G; d < 64; d <<= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}
template <int G>
__device__ __forceinline__ float wave_sum_g(float v) {
#pragma unroll
  for (int d = G; d < 64; d <<= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ float wave_max_cg(float v) { return wave_max_g<CG>(v); }
__device__ __forceinline__ float wave_sum_cg(float v) { return wave_sum_g<CG>(v); }
constexpr int PL = NTH / CG;       // pixel lanes per workgroup at CG channels (16)

// One workgroup = (frame f, group of CG channels).  Two passes over the group's maps:
//   pass 1: max of alpha*h (soft-max shift) and max 2x2-window sum of the raw map (confidence)
//   pass 2: e = exp(alpha*h - max); sums of e, e*gx, e*gy; max 2x2-window sum of e; optional prob store
// STAGED: the H*W*CG slab is first copied to LDS with every load in flight at once (one HBM/L2 round trip
// instead of one per loop iteration -- the kernel is latency-, not bandwidth-bound) and both passes read LDS.
// Thread mapping and summation order are identical in both variants, so are the results.
// G channels per work-group: 16 when F*C/16 work-groups already fill the chip, 4 otherwise (the kernel is
// issue-bound inside a work-group, so at small batch x channels the idle CUs are the resource to use).
template <int G, bool STAGED>
__global__ __launch_bounds__(NTH) void softargmax2d_kernel(const SamArgs p) {
  constexpr int CG = G;
  constexpr int PL = NTH / G;
  __shared__ float red[7][NW][CG];
  extern __shared__ __attribute__((aligned(16))) float slab[];   // STAGED: [H*W][CG], then gx[W], gy[H]
  const int tid = threadIdx.x;
  const int cc = tid % CG, pl = tid / CG;
  const int wave = tid >> 6;
  const int groups = (p.C + CG - 1) / CG;
  const int wg = xcd_order((int)blockIdx.x, (int)gridDim.x);
  const int f = wg / groups;
  const int c0 = (wg % groups) * CG;
  const int c = c0 + cc;
  const bool cok = c < p.C;
  const int HW = p.H * p.W;
  const float* base = p.h + (size_t)f * HW * p.ldh + c;

  float* sgx = slab + HW * CG;
  float* sgy = sgx + p.W;
  if constexpr (STAGED) {
    // (the coordinate grids' first values are requested BEFORE the maps and wait in registers: one round trip, not two)
    const float gx0 = tid < p.W ? p.gx[tid] : 0.f, gy0 = tid < p.H ? p.gy[tid] : 0.f;
    const float* src = p.h + (size_t)f * HW * p.ldh + c0;
    const bool vec = (p.ldh % 4 == 0) && (c0 + CG <= p.C) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    if (vec) {
      // 8 independent 16-byte loads in flight per thread and round
      const int total = HW * (CG / 4);
      for (int i0 = tid; i0 < total; i0 += 8 * NTH) {
        float4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          int i = i0 + k * NTH;
          i = i < total ? i : total - 1;
          const int px = i / (CG / 4), q4 = i - px * (CG / 4);
          t[k] = *reinterpret_cast<const float4*>(src + (size_t)px * p.ldh + q4 * 4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = i0 + k * NTH;
          if (i < total) *reinterpret_cast<float4*>(&slab[i * 4]) = t[k];      // slab index = px*CG + q4*4 = i*4
        }
      }
    } else {
      const int total = HW * CG;
      for (int i0 = tid; i0 < total; i0 += 8 * NTH) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          int i = i0 + k * NTH;
          i = i < total ? i : total - 1;
          const int px = i / CG, ch = i - px * CG;
          t[k] = src[(size_t)px * p.ldh + (c0 + ch < p.C ? ch : p.C - 1 - c0)];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = i0 + k * NTH;
          if (i < total) slab[i] = t[k];
        }
      }
    }
    // the coordinate grids are read once per pixel in pass 2: keep them next to the maps
    if (tid < p.W) sgx[tid] = gx0;
    if (tid < p.H) sgy[tid] = gy0;
    for (int i = tid + NTH; i < p.W; i += NTH) sgx[i] = p.gx[i];
    for (int i = tid + NTH; i < p.H; i += NTH) sgy[i] = p.gy[i];
    __syncthreads();
  }
  auto at = [&](int px) -> float {
    if constexpr (STAGED) return slab[px * CG + cc];
    else return base[(size_t)px * p.ldh];
  };

  float vmax = -INFINITY, cmax = -INFINITY, rmax = -INFINITY;
  if (cok) {
    for (int px = pl; px < HW; px += PL) {
      const float v = at(px);
      vmax = fmaxf(vmax, p.alpha * v);
      rmax = fmaxf(rmax, v);
      if (p.conf_raw != nullptr) {
        const int r = px / p.W, q = px - r * p.W;
        if (r + 1 < p.H && q + 1 < p.W) {
          const float s4 = ((v + at(px + 1)) + at(px + p.W)) + at(px + p.W + 1);
          cmax = fmaxf(cmax, p.conf_scale * s4);
        }
      }
    }
  }
  vmax = wave_max_g<G>(vmax);
  cmax = wave_max_g<G>(cmax);
  rmax = wave_max_g<G>(rmax);
  if ((tid & 63) < CG) { red[0][wave][cc] = vmax; red[1][wave][cc] = cmax; red[6][wave][cc] = rmax; }
  __syncthreads();
  vmax = red[0][0][cc]; cmax = red[1][0][cc]; rmax = red[6][0][cc];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    vmax = fmaxf(vmax, red[0][w][cc]); cmax = fmaxf(cmax, red[1][w][cc]); rmax = fmaxf(rmax, red[6][w][cc]);
  }
  __syncthreads();

  float s = 0.f, sx = 0.f, sy = 0.f, pmax = -INFINITY;
  if (cok) {
    for (int px = pl; px < HW; px += PL) {
      const int r = px / p.W, q = px - r * p.W;
      const float e = expf(p.alpha * at(px) - vmax);
      s += e;
      sx = fmaf(e, STAGED ? sgx[q] : p.gx[q], sx);
      sy = fmaf(e, STAGED ? sgy[r] : p.gy[r], sy);
      if ((p.conf_prob != nullptr || p.xy_times_conf) && r + 1 < p.H && q + 1 < p.W) {
        const float e1 = expf(p.alpha * at(px + 1) - vmax);
        const float e2 = expf(p.alpha * at(px + p.W) - vmax);
        const float e3 = expf(p.alpha * at(px + p.W + 1) - vmax);
        pmax = fmaxf(pmax, ((e + e1) + e2) + e3);
      }
    }
  }
  s = wave_sum_g<G>(s); sx = wave_sum_g<G>(sx); sy = wave_sum_g<G>(sy); pmax = wave_max_g<G>(pmax);
  if ((tid & 63) < CG) {
    red[2][wave][cc] = s; red[3][wave][cc] = sx; red[4][wave][cc] = sy; red[5][wave][cc] = pmax;
  }
  __syncthreads();
  s = red[2][0][cc]; sx = red[3][0][cc]; sy = red[4][0][cc]; pmax = red[5][0][cc];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    s += red[2][w][cc]; sx += red[3][w][cc]; sy += red[4][w][cc]; pmax = fmaxf(pmax, red[5][w][cc]);
  }
  s = fmaxf(s, 1e-7f);  // K.clip(sum, K.epsilon(), None), activations.py:12
  const float inv = 1.f / s;

  if (cok && pl == 0) {
    if (p.xy != nullptr) {
      float* o = p.xy + ((size_t)f * p.C + c) * p.ldxy;
      float ox = sx * inv, oy = sy * inv;
      if (p.xy_times_conf) {        // multiply([p, c]) of spnet.py:108 folded in: the two fp32 values a stand-alone launch would
        const float cf = pmax * inv;   // read back, one multiplication each
        ox *= cf;
        oy *= cf;
      }
      o[0] = ox;
      o[1] = oy;
    }
    if (p.conf_raw != nullptr) p.conf_raw[((size_t)f * p.C + c) * p.ldcr] = cmax;
    if (p.conf_prob != nullptr) p.conf_prob[((size_t)f * p.C + c) * p.ldcp] = pmax * inv;
    if (p.gmax != nullptr) p.gmax[(size_t)f * p.C + c] = rmax;
  }
  if (p.prob != nullptr && cok) {
    float* pb = p.prob + (size_t)f * HW * p.ldp + c;
    for (int px = pl; px < HW; px += PL) pb[(size_t)px * p.ldp] = expf(p.alpha * at(px) - vmax) * inv;
  }
}

// Soft-argmax + context aggregation in ONE launch (blocks.build_context_aggregation, blocks.py:217-285, on top of the two
// read-outs of reception.pose_regression_2d_context, reception.py:167-182): the joints' own maps are channels [0, J) of
// the heat-map tensor, joint j's nctx context maps are channels J + j*nctx + k.  Work-group = (frame, group of four
// joints): 4 + 4 nctx <= 16 channel lanes x NT / 16 pixel lanes, the same two passes as softargmax2d_kernel<16, true>; the
// twelve (x, y, confidence) triples meet in LDS and four threads write
//     y_j = alpha ys_j + (1 - alpha) sum_k yc_jk pc_jk / sum_k pc_jk        (same operation order as context_agg_kernel)
// and the joints' confidences.  Replaces three launches (two soft-argmax, one aggregation) of a few microseconds of
// work each: on one stream their launch gaps and ramps are paid in full.
template <int NT>
__global__ __launch_bounds__(NT) void softargmax2d_ctx_kernel(const SamArgs p, const int J, const int nctx,
                                                               const float agg_alpha, float* __restrict__ y,
                                                               const int ldy) {
  constexpr int NWV = NT / 64;      // waves
  constexpr int PLN = NT / CG;      // pixel lanes
  __shared__ float red[5][NWV][CG];
  __shared__ float fin[3][CG];
  extern __shared__ __attribute__((aligned(16))) float slab[];   // [H*W][CG], then gx[W], gy[H]
  const int tid = threadIdx.x;
  const int cc = tid % CG, pl = tid / CG;
  const int wave = tid >> 6;
  const int groups = (J + 3) / 4;
  const int wg = xcd_order((int)blockIdx.x, (int)gridDim.x);
  const int f = wg / groups;
  const int j0 = (wg % groups) * 4;
  const int nch = 4 + 4 * nctx;
  const int HW = p.H * p.W;
  // channel of lane cc: joints j0 .. j0+3, then their contexts
  const int jj = cc < 4 ? cc : (cc - 4) / nctx;
  const int c = cc < 4 ? j0 + cc : J + (j0 + jj) * nctx + (cc - 4) % nctx;
  const bool cok = cc < nch && j0 + jj < J;

  float* sgx = slab + HW * CG;
  float* sgy = sgx + p.W;
  {
    const float gx0 = tid < p.W ? p.gx[tid] : 0.f, gy0 = tid < p.H ? p.gy[tid] : 0.f;   // (requested before the maps, see above)
    const float* src = p.h + (size_t)f * HW * p.ldh;
    const int per = 1 + nctx;                           // float4 per pixel: the joints' quad, then nctx context quads
    const int total = HW * per;
    for (int i0 = tid; i0 < total; i0 += 8 * NT) {
      float4 t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int i = i0 + k * NT;
        i = i < total ? i : total - 1;
        const int px = i / per, q = i - px * per;
        const int ch = q == 0 ? j0 : J + j0 * nctx + (q - 1) * 4;
        t[k] = *reinterpret_cast<const float4*>(src + (size_t)px * p.ldh + ch);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + k * NT;
        if (i < total) {
          const int px = i / per, q = i - px * per;
          *reinterpret_cast<float4*>(&slab[px * CG + q * 4]) = t[k];
        }
      }
    }
    if (tid < p.W) sgx[tid] = gx0;
    if (tid < p.H) sgy[tid] = gy0;
    for (int i = tid + NT; i < p.W; i += NT) sgx[i] = p.gx[i];
    for (int i = tid + NT; i < p.H; i += NT) sgy[i] = p.gy[i];
    __syncthreads();
  }
  auto at = [&](int px) -> float { return slab[px * CG + cc]; };

  float vmax = -INFINITY, cmax = -INFINITY;
  if (cok) {
    for (int px = pl; px < HW; px += PLN) {
      const float v = at(px);
      vmax = fmaxf(vmax, p.alpha * v);
      const int r = px / p.W, q = px - r * p.W;
      if (r + 1 < p.H && q + 1 < p.W) {
        const float s4 = ((v + at(px + 1)) + at(px + p.W)) + at(px + p.W + 1);
        cmax = fmaxf(cmax, p.conf_scale * s4);
      }
    }
  }
  vmax = wave_max_cg(vmax);
  cmax = wave_max_cg(cmax);
  if ((tid & 63) < CG) { red[0][wave][cc] = vmax; red[1][wave][cc] = cmax; }
  __syncthreads();
  vmax = red[0][0][cc]; cmax = red[1][0][cc];
#pragma unroll
  for (int w = 1; w < NWV; ++w) { vmax = fmaxf(vmax, red[0][w][cc]); cmax = fmaxf(cmax, red[1][w][cc]); }

  float s = 0.f, sx = 0.f, sy = 0.f;
  if (cok) {
    for (int px = pl; px < HW; px += PLN) {
      const int r = px / p.W, q = px - r * p.W;
      const float e = expf(p.alpha * at(px) - vmax);
      s += e;
      sx = fmaf(e, sgx[q], sx);
      sy = fmaf(e, sgy[r], sy);
    }
  }
  s = wave_sum_cg(s); sx = wave_sum_cg(sx); sy = wave_sum_cg(sy);
  if ((tid & 63) < CG) { red[2][wave][cc] = s; red[3][wave][cc] = sx; red[4][wave][cc] = sy; }
  __syncthreads();
  if (tid < CG) {
    s = red[2][0][cc]; sx = red[3][0][cc]; sy = red[4][0][cc];
#pragma unroll
    for (int w = 1; w < NWV; ++w) { s += red[2][w][cc]; sx += red[3][w][cc]; sy += red[4][w][cc]; }
    s = fmaxf(s, 1e-7f);  // K.clip(sum, K.epsilon(), None), activations.py:12
    const float inv = 1.f / s;
    fin[0][cc] = sx * inv;
    fin[1][cc] = sy * inv;
    fin[2][cc] = cmax;
  }
  __syncthreads();
  if (tid < 4 && j0 + tid < J) {
    const int j = j0 + tid;
    float sp = 0.f, spx = 0.f, spy = 0.f;
    for (int k = 0; k < nctx; ++k) {
      const int ci = 4 + tid * nctx + k;
      const float pk = fin[2][ci];
      sp += pk;
      spx += fin[0][ci] * pk;
      spy += fin[1][ci] * pk;
    }
    float* o = y + ((size_t)f * J + j) * ldy;
    o[0] = agg_alpha * fin[0][tid] + (1.f - agg_alpha) * (spx / sp);
    o[1] = agg_alpha * fin[1][tid] + (1.f - agg_alpha) * (spy / sp);
    if (p.conf_raw != nullptr) p.conf_raw[((size_t)f * J + j) * p.ldcr] = fin[2][tid];
  }
}

__global__ __launch_bounds__(256) void context_agg_kernel(const float* __restrict__ ys,
                                                          const float* __restrict__ yc,
                                                          const float* __restrict__ pc, float* __restrict__ y,
                                                          int F, int J, int nctx, float alpha, int ldy) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= F * J) return;
  const int f = idx / J, j = idx - f * J;
  float sp = 0.f, spx = 0.f, spy = 0.f;
  for (int k = 0; k < nctx; ++k) {
    const size_t ci = (size_t)f * J * nctx + (size_t)j * nctx + k;
    const float pk = pc[ci];
    sp += pk;
    spx += yc[ci * 2 + 0] * pk;
    spy += yc[ci * 2 + 1] * pk;
  }
  const float xs = ys[(size_t)idx * 2 + 0], ysv = ys[(size_t)idx * 2 + 1];
  y[(size_t)idx * ldy + 0] = alpha * xs + (1.f - alpha) * (spx / sp);
  y[(size_t)idx * ldy + 1] = alpha * ysv + (1.f - alpha) * (spy / sp);
}

// hxy[f,p,j] = mean_d h[f,p,d*J+j]; hz[f,d,j] = mean_p h[f,p,d*J+j]   (reception.py:193-222)
//
// ONE summation order for every kernel below [r05], so the launcher may pick by batch size without moving a bit:
//   hxy: four accumulators over d (d & 3), then (a0 + a1) + (a2 + a3), times 1/D;
//   hz : the pixels are cut into chunks of 64 (missing pixels of a ragged last chunk count as 0).  Inside chunk k the
//        quad sums q[k][l] = (v[4l] + v[4l+1]) + (v[4l+2] + v[4l+3]), l = 0..15; each l accumulates its q over the
//        chunks in ascending k (P[l]); the sixteen P[l] are summed as a balanced tree pairing l with l^1, l^2, l^4, l^8;
//        times 1/HW.  Rounding depth: 2 + HW/64 + 4 additions (22 for 32 x 32 maps) -- round 4's one-pass kernel ran
//        256 serial additions per accumulator and cost the 3-D head two thirds of its parity margin.
__device__ __forceinline__ float depth_mean_d(const float* __restrict__ src, int D, int J, float invd) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int d = 0;
  for (; d + 3 < D; d += 4) {
    a0 += src[d * J];
    a1 += src[(d + 1) * J];
    a2 += src[(d + 2) * J];
    a3 += src[(d + 3) * J];
  }
  if (d < D) a0 += src[d * J];
  if (d + 1 < D) a1 += src[(d + 1) * J];
  if (d + 2 < D) a2 += src[(d + 2) * J];
  return ((a0 + a1) + (a2 + a3)) * invd;
}

__global__ __launch_bounds__(256) void depth_means_xy_kernel(const float* __restrict__ h, int ldh,
                                                             float* __restrict__ hxy, int F, int HW, int D,
                                                             int J) {
  const long long total = (long long)F * HW * J;
  const float invd = 1.f / (float)D;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % J);
    const long long px = idx / J;
    hxy[idx] = depth_mean_d(h + px * ldh + j, D, J, invd);
  }
}

// hz for small batches: work-group = (frame, group of CG channels), thread = (channel, quad lane l): F * DJ / 16
// work-groups where the one-pass kernel below has F.  The tree over l runs xor 16, xor 32 inside the wave (l = tid / 16:
// l^1 and l^2 are lanes of the same wave) and (r0 + r1) + (r2 + r3) across the four waves through LDS.
__global__ __launch_bounds__(NTH) void depth_means_z_kernel(const float* __restrict__ h, int ldh,
                                                            float* __restrict__ hz, int HW, int DJ) {
  __shared__ float red[NW][CG];
  const int tid = threadIdx.x;
  const int cc = tid % CG, pl = tid / CG;
  const int groups = (DJ + CG - 1) / CG;
  const int wg = xcd_order((int)blockIdx.x, (int)gridDim.x);
  const int f = wg / groups;
  const int c = (wg % groups) * CG + cc;
  float acc = 0.f;
  if (c < DJ) {
    const float* src = h + (size_t)f * HW * ldh + c;
    for (int p0 = 4 * pl; p0 < HW; p0 += 64) {
      const float v0 = src[(size_t)p0 * ldh];
      const float v1 = p0 + 1 < HW ? src[(size_t)(p0 + 1) * ldh] : 0.f;
      const float v2 = p0 + 2 < HW ? src[(size_t)(p0 + 2) * ldh] : 0.f;
      const float v3 = p0 + 3 < HW ? src[(size_t)(p0 + 3) * ldh] : 0.f;
      acc += (v0 + v1) + (v2 + v3);
    }
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  if ((tid & 63) < CG) red[tid >> 6][cc] = acc;
  __syncthreads();
  if (tid < CG && c < DJ) hz[(size_t)f * DJ + c] = ((red[0][cc] + red[1][cc]) + (red[2][cc] + red[3][cc])) * (1.f / (float)HW);
}

// Both means in ONE pass over the maps [r04]: the 3-D head reads its 16 x J depth-joint maps twice (once per mean; at
// batch 128 that tensor is 142 MB, the two launches were 2.6 % of the Human3.6M step).  Work-group = one frame, 512
// threads, walking chunks of 64 pixels: the chunk goes to LDS with 16-byte loads (the next chunk's loads are in flight
// during the arithmetic), hxy of its pixels is summed over depth out of LDS, and thread c keeps the sixteen running quad
// sums P[l] of channel c [r05: the order stated above, bit-identical to the two kernels above].
constexpr int DM_NT = 512, DM_PX = 64, DM_MAXC = 288, DM_MIN_FRAMES = 96;
__global__ __launch_bounds__(DM_NT) void depth_means_fused_kernel(const float* __restrict__ h, int ldh, float* __restrict__ hxy,
                                                                 float* __restrict__ hz, int HW, int D, int J) {
  extern __shared__ __attribute__((aligned(16))) float dm_lds[];      // [DM_PX][DJ]
  const int tid = threadIdx.x, f = blockIdx.x;
  const int DJ = D * J, c4n = DJ >> 2;
  const float* src = h + (size_t)f * HW * ldh;
  constexpr int NL = (DM_PX * (DM_MAXC / 4) + DM_NT - 1) / DM_NT;     // float4 loads per thread and chunk (9)
  const unsigned magic = (1u << 20) / (unsigned)c4n + 1u;             // i / c4n for i < 64 * 72
  float4 stage[NL];
  auto fetch = [&](int p0) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = tid + k * DM_NT;
      const int px = (int)(((unsigned)i * magic) >> 20), q = i - px * c4n;
      const bool ok = i < DM_PX * c4n && p0 + px < HW;
      stage[k] = ok ? *reinterpret_cast<const float4*>(src + (size_t)(p0 + px) * ldh + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float P[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) P[l] = 0.f;
  const float invd = 1.f / (float)D;
  fetch(0);
  for (int p0 = 0; p0 < HW; p0 += DM_PX) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = tid + k * DM_NT;
      if (i < DM_PX * c4n) reinterpret_cast<float4*>(dm_lds)[i] = stage[k];   // pixels past HW arrive as zeros
    }
    __syncthreads();
    if (p0 + DM_PX < HW) fetch(p0 + DM_PX);
    const int npx = HW - p0 < DM_PX ? HW - p0 : DM_PX;
    for (int it = tid; it < npx * J; it += DM_NT) {                   // hxy[f, p, j] = mean_d h[f, p, d * J + j]
      const int px = it / J, j = it - px * J;
      hxy[((size_t)f * HW + p0 + px) * J + j] = depth_mean_d(dm_lds + px * DJ + j, D, J, invd);
    }
    if (tid < DJ) {                                                   // P[l] += quad sum l of the chunk
      const float* col = dm_lds + tid;
#pragma unroll
      for (int l = 0; l < 16; ++l)
        P[l] += (col[(4 * l) * DJ] + col[(4 * l + 1) * DJ]) + (col[(4 * l + 2) * DJ] + col[(4 * l + 3) * DJ]);
    }
    __syncthreads();
  }
  if (tid < DJ) {
#pragma unroll
    for (int s = 1; s < 16; s <<= 1)
#pragma unroll
      for (int l = 0; l < 16; l += 2 * s) P[l] += P[l + s];
    hz[(size_t)f * DJ + tid] = P[0] * (1.f / (float)HW);
  }
}

__global__ __launch_bounds__(256) void softargmax1d_kernel(const float* __restrict__ hz,
                                                           const float* __restrict__ grid,
                                                           float* __restrict__ z, int ldz,
                                                           float* __restrict__ vz, int F, int D, int J) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= F * J) return;
  const int f = idx / J, j = idx - f * J;
  const float* src = hz + (size_t)f * D * J + j;
  float m = -INFINITY;
  for (int d = 0; d < D; ++d) m = fmaxf(m, src[d * J]);
  float s = 0.f, sz = 0.f;
  for (int d = 0; d < D; ++d) {
    const float e = expf(src[d * J] - m);
    s += e;
    sz = fmaf(e, grid[d], sz);
  }
  if (z != nullptr) z[(size_t)idx * ldz] = sz / s;
  if (vz != nullptr) vz[idx] = m;
}

// z[f, j] = sum_p sigmoid(d[f,p,j]) * h[f,p,j]   (spnet.py:201-205: depth as the probability-weighted mean of
// the sigmoid depth maps).  Workgroup = (frame, 16 channels) like the soft-argmax kernel.
__global__ __launch_bounds__(NTH) void depth_from_maps_kernel(const float* __restrict__ d, int ldd,
                                                              const float* __restrict__ h, int ldh,
                                                              float* __restrict__ z, int ldz, int HW, int J) {
  __shared__ float red[NW][CG];
  const int tid = threadIdx.x;
  const int cc = tid % CG, pl = tid / CG;
  const int groups = (J + CG - 1) / CG;
  const int wg = xcd_order((int)blockIdx.x, (int)gridDim.x);
  const int f = wg / groups;
  const int c = (wg % groups) * CG + cc;
  float acc = 0.f;
  if (c < J) {
    const float* dp = d + (size_t)f * HW * ldd + c;
    const float* hp = h + (size_t)f * HW * ldh + c;
    for (int px = pl; px < HW; px += PL) {
      const float sg = 1.f / (1.f + expf(-dp[(size_t)px * ldd]));
      acc = fmaf(sg, hp[(size_t)px * ldh], acc);
    }
  }
  acc = wave_sum_cg(acc);
  if ((tid & 63) < CG) red[tid >> 6][cc] = acc;
  __syncthreads();
  if (tid < CG && c < J) {
    float t = red[0][cc];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w][cc];
    z[((size_t)f * J + c) * ldz] = t;
  }
}

// f[b, j, c] = sum_p hm[b,p,j] * x[b,p,c].  Workgroup = (b, 64-channel slab); each thread owns one channel
// and JT joints at a time; the heat-map tile for a chunk of pixels is staged in LDS and broadcast.
constexpr int KR_PCH = 64;   // pixels per LDS chunk
constexpr int KR_JMAX = 32;  // joints handled per pass
__global__ __launch_bounds__(64) void kronecker_kernel(const float* __restrict__ hm, int ldh,
                                                       const float* __restrict__ x, int ldx,
                                                       float* __restrict__ f, int ldf, int P, int J, int C) {
  __shared__ float sh[KR_PCH][KR_JMAX + 1];
  const int b = blockIdx.y;
  const int c = blockIdx.x * 64 + threadIdx.x;
  const bool cok = c < C;
  for (int j0 = 0; j0 < J; j0 += KR_JMAX) {
    const int jn = min(KR_JMAX, J - j0);
    float acc[KR_JMAX];
#pragma unroll
    for (int j = 0; j < KR_JMAX; ++j) acc[j] = 0.f;
    for (int p0 = 0; p0 < P; p0 += KR_PCH) {
      const int pn = min(KR_PCH, P - p0);
      __syncthreads();
      for (int i = threadIdx.x; i < pn * jn; i += 64) {
        const int pp = i / jn, jj = i - pp * jn;
        sh[pp][jj] = hm[((size_t)b * P + p0 + pp) * ldh + j0 + jj];
      }
      __syncthreads();
      if (cok) {
        for (int pp = 0; pp < pn; ++pp) {
          const float xv = x[((size_t)b * P + p0 + pp) * ldx + c];
#pragma unroll
          for (int j = 0; j < KR_JMAX; ++j)
            if (j < jn) acc[j] = fmaf(sh[pp][j], xv, acc[j]);
        }
      }
    }
    if (cok) {
#pragma unroll
      for (int j = 0; j < KR_JMAX; ++j)
        if (j < jn) f[((size_t)b * J + j0 + j) * ldf + c] = acc[j];
    }
  }
}

// The same product for the shapes the models have (C % 4 == 0, 16-byte aligned rows): a [J x P] x [P x C] GEMM per
// frame with J = 16 ... 20 -- bound by reading x once.  Work-group = (frame, 64-channel slab), 256 threads = 16 channel
// quads x 16 pixel groups; a thread walks pixels g, g + 16, ... with float4 loads of its four channels (four pixels in
// flight) and JT x 4 accumulators, the heat-map values of a pixel are the same address for the 16 lanes that share it;
// the 16 pixel groups are then summed through LDS in two halves of JT / 2 joints (<= 40 KB: four work-groups per CU).
// 64 x 1024 x 16 x 576 (the Penn merge model, batch of 4 clips): 1 079 us with the kernel above, see DESIGN.md for this one.
template <int JT>
__global__ __launch_bounds__(256) void kronecker_tiled_kernel(const float* __restrict__ hm, int ldh,
                                                             const float* __restrict__ x, int ldx,
                                                             float* __restrict__ f, int ldf, int P, int J, int C,
                                                             int fvec) {
  constexpr int HP = JT + 4;                                       // LDS pitch of a pixel's heat-map values (floats)
  constexpr int JH = JT / 2;
  extern __shared__ __attribute__((aligned(16))) float4 kr_red[];  // [16 groups][JT / 2][16 quads], then the h chunk
  float* hs = reinterpret_cast<float*>(kr_red + 16 * JH * 16);     // [2 buffers][64 pixels][HP]
  const int tid = threadIdx.x;
  const int q = tid & 15, g = tid >> 4;
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * 64 + q * 4;
  const int cc = c0 < C ? c0 : C - 4;
  const float* xb = x + (size_t)b * P * ldx + cc;
  const float* hb = hm + (size_t)b * P * ldh;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = 0; j0 < J; j0 += JT) {
    const int jn = min(JT, J - j0);
    float4 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[j] = zero;
    // chunks of 64 pixels = four per pixel group.  [r06] Software-pipelined: the x rows and the heat-map values of chunk k + 1
    // are requested before chunk k is multiplied (they wait in registers), the heat-map values go through a DOUBLE-buffered LDS
    // tile (coalesced load, zero-filled beyond jn / P; each thread then reads its pixels' JT values as float4 -- the same
    // address for the 16 lanes of a group) -- one barrier and no exposed memory round trip per chunk; at a couple of clips per
    // call the loop used to be P / 64 serial round trips (33 us for the 32 x 32 level).  Same products, same order.
    constexpr int HV = JT / 4;                                     // heat-map values a thread stages per chunk (64 * JT / 256)
    auto load_x = [&](int p0, float4 (&xv)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + g + 16 * u;
        xv[u] = *reinterpret_cast<const float4*>(xb + (size_t)(p < P ? p : P - 1) * ldx);
      }
    };
    auto load_h = [&](int p0, float (&hv)[HV]) {
#pragma unroll
      for (int k = 0; k < HV; ++k) {
        const int i = tid + 256 * k;
        const int pp = i / JT, jj = i - pp * JT;
        hv[k] = (p0 + pp < P && jj < jn) ? hb[(size_t)(p0 + pp) * ldh + j0 + jj] : 0.f;
      }
    };
    float4 xv[4], xn[4];
    float hc[HV], hn[HV];
    load_x(0, xv);
    load_h(0, hc);
    __syncthreads();                                               // the previous joint tile's readers are done with both buffers
    int buf = 0;
    for (int p0 = 0; p0 < P; p0 += 64) {
      float* hbuf = hs + buf * (64 * HP);
#pragma unroll
      for (int k = 0; k < HV; ++k) {
        const int i = tid + 256 * k;
        const int pp = i / JT, jj = i - pp * JT;
        hbuf[pp * HP + jj] = hc[k];
      }
      if (p0 + 64 < P) {                                           // (uniform) next chunk's operands: in flight under this chunk
        load_x(p0 + 64, xn);
        load_h(p0 + 64, hn);
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4* hrow = reinterpret_cast<const float4*>(hbuf + (g + 16 * u) * HP);
#pragma unroll
        for (int j4 = 0; j4 < JT / 4; ++j4) {
          const float4 h = hrow[j4];
          const float hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float4& a = acc[j4 * 4 + e];
            a.x = fmaf(hv[e], xv[u].x, a.x); a.y = fmaf(hv[e], xv[u].y, a.y);
            a.z = fmaf(hv[e], xv[u].z, a.z); a.w = fmaf(hv[e], xv[u].w, a.w);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = xn[u];
#pragma unroll
      for (int k = 0; k < HV; ++k) hc[k] = hn[k];
      buf ^= 1;
    }
    // sum over the 16 pixel groups, JT / 2 joints at a time
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < JH; ++j) kr_red[(g * JH + j) * 16 + q] = acc[half * JH + j];
      __syncthreads();
      for (int o = tid; o < JH * 16; o += 256) {                     // o = (joint of the half, channel quad)
        const int j = o >> 4, qq = o & 15;
        float4 t = kr_red[o];
#pragma unroll
        for (int gg = 1; gg < 16; ++gg) {
          const float4 u = kr_red[(gg * JH + j) * 16 + qq];
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        const int jj = j0 + half * JH + j, co = blockIdx.x * 64 + qq * 4;
        if (half * JH + j < jn && co < C) {
          float* dst = f + ((size_t)b * J + jj) * ldf + co;       // the result often lands in a packed row (odd pitch)
          if (fvec) *reinterpret_cast<float4*>(dst) = t;
          else { dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w; }
        }
      }
    }
  }
}

// y[b, c] = softmax_c( max_p x[b,p,c] + min_p x[b,p,c] ); one workgroup per b, C <= 1024
__global__ __launch_bounds__(256) void global_maxmin_softmax_kernel(const float* __restrict__ x, int ldx,
                                                                    float* __restrict__ y, int P, int C,
                                                                    int softmax) {
  extern __shared__ float sv[];  // [C]
  __shared__ float sred[4];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float mx = -INFINITY, mn = INFINITY;
    for (int p = 0; p < P; ++p) {
      const float v = x[((size_t)b * P + p) * ldx + c];
      mx = fmaxf(mx, v);
      mn = fminf(mn, v);
    }
    sv[c] = mx + mn;
  }
  __syncthreads();
  if (!softmax) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) y[(size_t)b * C + c] = sv[c];
    return;
  }
  float m = -INFINITY;
  for (int c = threadIdx.x; c < C; c += blockDim.x) m = fmaxf(m, sv[c]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) s += expf(sv[c] - m);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
  __syncthreads();
  s = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  for (int c = threadIdx.x; c < C; c += blockDim.x) y[(size_t)b * C + c] = expf(sv[c] - m) / s;
}

}  // namespace

int launch_softargmax2d(const SamArgs& a, hipStream_t s) {
  if (a.F <= 0 || a.C <= 0 || a.H <= 0 || a.W <= 0 || a.h == nullptr || a.gx == nullptr || a.gy == nullptr)
    return DH_EINVAL;
  const long long blocks16 = (long long)a.F * ((a.C + 15) / 16);
  const long long blocks4 = (long long)a.F * ((a.C + 3) / 4);
  if (blocks4 > 0x7fffffffLL) return DH_EINVAL;
  constexpr size_t kMaxSlab = 128 * 1024;     // 32x32x16 maps need 64 KB: above the default dynamic-LDS limit
  static LdsLimit lim;
  lim.raise((const void*)softargmax2d_kernel<16, true>, (int)kMaxSlab);
  const bool wide = blocks16 >= 1024;
  const int g = wide ? 16 : 4;
  const size_t slab = ((size_t)a.H * a.W * g + a.W + a.H) * sizeof(float);
  const dim3 grid((unsigned)(wide ? blocks16 : blocks4));
  if (slab > (wide ? kMaxSlab : (size_t)60 * 1024)) {
    if (wide) hipLaunchKernelGGL((softargmax2d_kernel<16, false>), grid, dim3(NTH), 0, s, a);
    else hipLaunchKernelGGL((softargmax2d_kernel<4, false>), grid, dim3(NTH), 0, s, a);
  } else if (wide) {
    hipLaunchKernelGGL((softargmax2d_kernel<16, true>), grid, dim3(NTH), slab, s, a);
  } else {
    hipLaunchKernelGGL((softargmax2d_kernel<4, true>), grid, dim3(NTH), slab, s, a);
  }
  return check_launch();
}

int launch_softargmax2d_context(const SamArgs& a, int J, int nctx, float agg_alpha, float* y, int ldy, hipStream_t s) {
  if (a.F <= 0 || a.H <= 0 || a.W <= 0 || J <= 0 || nctx < 1 || nctx > 3 || y == nullptr || ldy < 2) return DH_EINVAL;
  if (a.h == nullptr || a.gx == nullptr || a.gy == nullptr || a.C != J * (1 + nctx)) return DH_EINVAL;
  if (a.xy != nullptr || a.conf_prob != nullptr || a.prob != nullptr || a.gmax != nullptr) return DH_EUNSUPPORTED;
  // float4 staging: the joints' and the contexts' channel quads are 16-byte aligned runs of the pixel
  if (J % 4 || a.ldh % 4 || (reinterpret_cast<uintptr_t>(a.h) & 15)) return DH_EUNSUPPORTED;
  const size_t slab = ((size_t)a.H * a.W * CG + a.W + a.H) * sizeof(float);
  if (slab > 128 * 1024) return DH_EUNSUPPORTED;
  // 16 channel lanes x 64 pixel lanes: sixteen pixels per thread and pass, like the four-channel soft-argmax variant
  constexpr int NT = 1024;
  static LdsLimit lim;
  lim.raise((const void*)softargmax2d_ctx_kernel<NT>, 128 * 1024);
  hipLaunchKernelGGL(softargmax2d_ctx_kernel<NT>, dim3((unsigned)(a.F * ((J + 3) / 4))), dim3(NT), slab, s, a, J, nctx,
                     agg_alpha, y, ldy);
  return check_launch();
}

int launch_context_agg(const float* ys, const float* yc, const float* pc, float* y, int F, int J, int nctx,
                       float alpha, int ldy, hipStream_t s) {
  if (F <= 0 || J <= 0 || nctx <= 0) return DH_EINVAL;
  hipLaunchKernelGGL(context_agg_kernel, dim3((F * J + 255) / 256), dim3(256), 0, s, ys, yc, pc, y, F, J, nctx,
                     alpha, ldy);
  return check_launch();
}

int launch_depth_means(const float* h, int ldh, float* hxy, float* hz, int F, int HW, int D, int J,
                       hipStream_t s) {
  if (F <= 0 || HW <= 0 || D <= 0 || J <= 0) return DH_EINVAL;
  const int DJ = D * J;
  // One work-group per frame: the one-pass kernel needs about as many frames as the chip has CUs; below
  // that the two kernels with F * groups work-groups run -- same bits (one summation order), so the choice is free.
  if (hxy != nullptr && hz != nullptr && DJ % 4 == 0 && DJ <= DM_MAXC && ldh % 4 == 0 && DJ <= DM_NT && F >= DM_MIN_FRAMES &&
      (reinterpret_cast<uintptr_t>(h) & 15) == 0) {                    // both means of the same maps: one pass
    const size_t lds = (size_t)DM_PX * DJ * sizeof(float);
    static LdsLimit lim;
    lim.raise((const void*)depth_means_fused_kernel, (int)(DM_PX * DM_MAXC * sizeof(float)));
    hipLaunchKernelGGL(depth_means_fused_kernel, dim3((unsigned)F), dim3(DM_NT), lds, s, h, ldh, hxy, hz, HW, D, J);
    return check_launch();
  }
  if (hxy != nullptr) {
    long long g = ((long long)F * HW * J + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(depth_means_xy_kernel, dim3((unsigned)g), dim3(256), 0, s, h, ldh, hxy, F, HW, D, J);
  }
  if (hz != nullptr)
    hipLaunchKernelGGL(depth_means_z_kernel, dim3((unsigned)(F * ((D * J + CG - 1) / CG))), dim3(NTH), 0, s, h, ldh, hz, HW,
                       D * J);
  return check_launch();
}

int launch_softargmax1d(const float* hz, const float* grid, float* z, int ldz, float* vz, int F, int D, int J,
                        hipStream_t s) {
  if (F <= 0 || D <= 0 || J <= 0) return DH_EINVAL;
  hipLaunchKernelGGL(softargmax1d_kernel, dim3((F * J + 255) / 256), dim3(256), 0, s, hz, grid, z, ldz, vz, F, D,
                     J);
  return check_launch();
}

int launch_depth_from_maps(const float* d, int ldd, const float* h, int ldh, float* z, int ldz, int F, int HW,
                           int J, hipStream_t s) {
  if (F <= 0 || HW <= 0 || J <= 0) return DH_EINVAL;
  hipLaunchKernelGGL(depth_from_maps_kernel, dim3((unsigned)(F * ((J + CG - 1) / CG))), dim3(NTH), 0, s, d, ldd, h,
                     ldh, z, ldz, HW, J);
  return check_launch();
}

int launch_kronecker(const float* hm, int ldh, const float* x, int ldx, float* f, int ldf, int B, int P, int J,
                     int C, hipStream_t s) {
  if (B <= 0 || P <= 0 || J <= 0 || C <= 0 || B > 65535) return DH_EINVAL;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (C % 4 == 0 && ldx % 4 == 0 && al16(x)) {
    const dim3 grid((C + 63) / 64, B);
    const int fvec = ldf % 4 == 0 && al16(f);
    if (J <= 16 || (J > 20 && J % 20 != 0 && J % 16 == 0))
      hipLaunchKernelGGL(kronecker_tiled_kernel<16>, grid, dim3(256), 16 * 8 * 16 * 16 + 2 * 64 * 20 * 4, s, hm, ldh, x, ldx, f, ldf, P, J, C, fvec);
    else
      hipLaunchKernelGGL(kronecker_tiled_kernel<20>, grid, dim3(256), 16 * 10 * 16 * 16 + 2 * 64 * 24 * 4, s, hm, ldh, x, ldx, f, ldf, P, J, C, fvec);
    return check_launch();
  }
  hipLaunchKernelGGL(kronecker_kernel, dim3((C + 63) / 64, B), dim3(64), 0, s, hm, ldh, x, ldx, f, ldf, P, J, C);
  return check_launch();
}

int launch_global_maxmin_softmax(const float* x, int ldx, float* y, int B, int P, int C, int softmax,
                                 hipStream_t s) {
  if (B <= 0 || P <= 0 || C <= 0 || C > 8192) return DH_EINVAL;
  hipLaunchKernelGGL(global_maxmin_softmax_kernel, dim3(B), dim3(256), C * sizeof(float), s, x, ldx, y, P, C,
                     softmax);
  return check_launch();
}

}  // namespace dh
