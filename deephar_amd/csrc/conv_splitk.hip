// Convolutions with a tiny output and a long reduction (the action heads: 3x3 over a [frames x joints] "image" with
// hundreds of channels, 64-1024 output pixels; reference deephar/models/action.py / blocks.py): the MFMA kernels of
// conv_igemm.hip / gemm1x1.hip give one 32 x 32 output tile to one wave, which then walks all of K alone -- 72 K-steps
// of DMA round trip + 16 MFMAs each, 42-68 us for 0.0-0.3 GFLOP, on 2-64 of the chip's 1024 SIMDs.
// Here a work-group of eight waves owns the 32 x 32 tile and splits K: wave w takes k-group pairs w, w + 8, ...; every
// lane loads its own operands straight from global memory (A: four consecutive channels of its pixel's tap, B: the
// packed weights' 16-byte unit of its column), eight pairs in flight, no LDS and no barrier in the loop; the eight
// partial tiles are summed through LDS in wave order, then BN / residuals / ReLU.
// The K order differs from the other conv kernels (eight interleaved partial sums), so this kernel is picked by a
// SHAPE rule (conv_is_skinny: per-frame geometry only), never by timing, batch size or alignment: a layer always
// computes the same bits.
#include "conv_common.h"

namespace dh {
namespace {

constexpr int SK_WAVES = 8;
constexpr int SK_DEPTH = 8;      // k-group pairs in flight per wave

__global__ __launch_bounds__(SK_WAVES * 64) void conv_splitk_kernel(const ConvArgs p, const int xvec, const int affvec) {
  __shared__ float red[SK_WAVES][16][64];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int M = p.N * p.OH * p.OW;
  const int tiles_n = (p.Cout + 31) / 32;
  const int m0 = (blockIdx.x / tiles_n) * 32;
  const int n0 = (blockIdx.x % tiles_n) * 32;

  // this lane's output pixel (A operand row) and its top-left input position
  int m = m0 + li;
  m = m < M ? m : M - 1;
  const int fr = m / (p.OH * p.OW);
  const int rem = m - fr * (p.OH * p.OW);
  const int oh = rem / p.OW, ow = rem - oh * p.OW;
  const int ih0 = oh * p.SH - p.PT, iw0 = ow * p.SW - p.PL;
  const float* xf = p.x + (size_t)fr * p.H * p.W * p.ldx;
  const float* wcol = p.w + (size_t)(n0 + li) * 4;        // packed [Kp / 4][Np][4]: unit (k-group, column)
  const bool aff = p.pre_scale != nullptr;
  const int pairs = p.Kp / 8;                             // k-group pairs: lanes lh = 0 / 1 take groups 2 kp / 2 kp + 1

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);

  auto load = [&](int kp, float4& a, float4& b) {
    const int kg = 2 * kp + lh;
    const int k0 = 4 * kg;
    b = *reinterpret_cast<const float4*>(wcol + (size_t)kg * p.Np * 4);
    const int tap = k0 / p.Cin;
    const int c = k0 - tap * p.Cin;                       // Cin % 4 == 0: the four k of a group share a tap
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int ih = ih0 + kh, iw = iw0 + kw;
    const bool ok = k0 < p.K && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    a = zero;
    if (ok) {
      const float* src = xf + ((size_t)ih * p.W + iw) * p.ldx + c;
      if (xvec) a = *reinterpret_cast<const float4*>(src);
      else a = make_float4(src[0], src[1], src[2], src[3]);       // same values, same arithmetic: same bits
      if (aff) {
        float4 sc, sh;
        if (affvec) {
          sc = *reinterpret_cast<const float4*>(p.pre_scale + c);
          sh = *reinterpret_cast<const float4*>(p.pre_shift + c);
        } else {
          sc = make_float4(p.pre_scale[c], p.pre_scale[c + 1], p.pre_scale[c + 2], p.pre_scale[c + 3]);
          sh = make_float4(p.pre_shift[c], p.pre_shift[c + 1], p.pre_shift[c + 2], p.pre_shift[c + 3]);
        }
        a.x = a.x * sc.x + sh.x; a.y = a.y * sc.y + sh.y; a.z = a.z * sc.z + sh.z; a.w = a.w * sc.w + sh.w;
      }
      if (p.pre_relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    }
  };

  float4 fa[SK_DEPTH], fb[SK_DEPTH];
#pragma unroll
  for (int d = 0; d < SK_DEPTH; ++d) {
    const int kp = wave + d * SK_WAVES;
    fa[d] = zero; fb[d] = zero;
    if (kp < pairs) load(kp, fa[d], fb[d]);
  }
  for (int kp = wave; kp < pairs; kp += SK_DEPTH * SK_WAVES) {
#pragma unroll
    for (int d = 0; d < SK_DEPTH; ++d) {
      const float4 a = fa[d], b = fb[d];
      const int nxt = kp + (d + SK_DEPTH) * SK_WAVES;
      fa[d] = zero; fb[d] = zero;
      if (nxt < pairs) load(nxt, fa[d], fb[d]);           // lands while the other pairs are multiplied
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);     // a pair beyond `pairs` is zeros
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  }

#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // C layout of the 32x32 tile: register r of lane (li, lh) = row (r & 3) + 8 (r >> 2) + 4 lh, column li
  const int ohw = p.OH * p.OW;
  for (int idx = tid; idx < 16 * 64; idx += SK_WAVES * 64) {
    const int r = idx >> 6, l = idx & 63;
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    const int mm = m0 + row, n = n0 + col;
    if (mm >= M || n >= p.Cout) continue;
    float t = red[0][r][l];
#pragma unroll
    for (int w = 1; w < SK_WAVES; ++w) t += red[w][r][l];
    if (p.post_scale != nullptr) t = t * p.post_scale[n] + p.post_shift[n];
    if (p.res1 != nullptr) t += p.res1[(size_t)mm * p.ldr1 + n];
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
      if (p.res2 != nullptr) t += p.res2[(size_t)mm * p.ldr2 + n];
      if (p.post_relu) t = fmaxf(t, 0.f);
      p.y[(size_t)mm * p.ldy + n] = t;
    }
  }
}

}  // namespace

// Shape rule -- deterministic, independent of timing, of the batch size and of buffer alignment (the bits of a layer
// must not depend on any of them): at most 256 output pixels per frame / clip, a reduction of at least 768, at most 256
// output channels, channels in groups of four.
bool conv_is_skinny(const ConvArgs& a) {
  if (a.x_u8 || a.w_split) return false;
  return a.OH * a.OW <= 256 && a.K >= 768 && a.Cout <= 256 && a.Cin % 4 == 0;
}

int launch_conv_splitk(const ConvArgs& a, hipStream_t s) {
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = ((M + 31) / 32) * ((a.Cout + 31) / 32);
  if (tiles <= 0 || tiles > 0x7fffffffLL) return DH_EINVAL;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int xvec = a.ldx % 4 == 0 && al16(a.x);
  const int affvec = a.pre_scale != nullptr && al16(a.pre_scale) && al16(a.pre_shift);
  hipLaunchKernelGGL(conv_splitk_kernel, dim3((unsigned)tiles), dim3(SK_WAVES * 64), 0, s, a, xvec, affvec);
  return check_launch();
}

}  // namespace dh
