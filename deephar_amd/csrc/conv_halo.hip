// Dense K x K convolution (stride 1 or 2, zero padded) with the input HALO TILE resident in LDS, for gfx950.
// Replaces, for the shapes it covers, the K x K variant of gemm1x1_kernel (every filter tap re-fetches its shifted A
// rows through LDS-DMA: KH*KW times the input from L2, 0.15 DMA pieces per MFMA) and the register-gather kernel
// conv_igemm_kernel (Cin not a multiple of 32): the stem of ReceptionNet (deephar/models/reception.py:61-98:
// conv_bn_act 3x3, 5x1, 1x5) and the 'normal' residual units of SPNet's entry flow (deephar/models/common.py:25-67 via
// spnet.py:317-352: 1x1 -> BN -> ReLU -> 3x3 with Cin = 48, 96, 144).
//
// Work-group = 4 waves = 128 consecutive output pixels (RT = 128 / OW whole image rows of one frame, or a 128-column
// run of a wide row) x BN = TN * 32 output channels; wave w owns pixels [32 w, 32 w + 32).  K runs CHUNK-MAJOR:
//     for chunk of 16 input channels:  for kh:  for kw:  16 k-values
// * the halo tile of one 16-channel chunk -- HR = (RT-1)*s + KH rows x HC = (CT-1)*s + KW columns, 80 bytes per pixel
//   (4 slots of 16 B + one pad slot: an odd pitch, so the 16 lanes ds_read_b128 serves per cycle, consecutive pixels,
//   fall on 16 different bank groups without any XOR arithmetic) -- is fetched ONCE by LDS-DMA (out-of-image pixels:
//   out-of-range buffer offset -> zeros) and every one of the KH*KW taps reads its A fragments from it at
//   (pixel + kh*HC + kw) * 80: one v_add per kernel row, every other address is an immediate;
// * only the weights stream: one stage per (chunk, kh) = KW*16 k-values x BN columns, double buffered, packed by the
//   host in this K order ([K/4][Np][4] like every other conv, dh_conv_args.w_split = 2);
// * per-lane DMA source offsets are fixed over the whole K loop (the chunk is the SCALAR soffset): the K loop holds
//   KW*8*TN MFMAs per stage beside ~3 VALU, 2*KW + 2*KW*TN ds_read_b128 and the stage's B pieces.
// The K order differs from the tap-major kernels', so this kernel is chosen by a rule on the layer's geometry only
// (conv_halo_eligible), never by timing: the result bits of a layer do not depend on batch size or tuning.  Its own
// tilings (TN = 1, 2, 3) are bit-identical.  Same epilogue as the other MFMA convs (conv_common.h).
#include <utility>
#include "conv_common.h"

namespace dh {
namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned OOB = 0xfffffff0u;     // beyond every descriptor used here: the load returns zeros
constexpr int HCH = 16;                   // channels per resident chunk
constexpr int HPIX = 80;                  // LDS bytes per halo pixel (64 + 16 pad)
constexpr int HMAXA = 12;                 // halo DMA instructions per wave (<= 48 KB per chunk)
constexpr int HMAXB = 8;                  // weight-stage DMA passes (KW * BN <= 512)

template <typename RSRC>
__device__ __forceinline__ void dma16(RSRC rs, char* dst, unsigned voff, int soff) {
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
__device__ __forceinline__ float relu1(float v) {      // one integer max (gemm1x1.hip)
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}
__device__ __forceinline__ float4 relu4(float4 v) { return make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)); }

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct HaloGeom {
  int ct_log2;      // output columns per tile = 1 << ct_log2 (OW, or 128 for wider rows)
  int hr, hc;       // halo rows / columns
  int a_bytes;      // one halo buffer, rounded up to whole 1 KB DMA instructions
  int nbuf_a;       // 1 or 2 halo buffers
  int b_bytes;      // one weight stage, rounded up to whole 4 KB DMA passes
  int bpass;
  int hc_magic;     // (1 << 20) / hc + 1: px / hc for px < 1024 as one multiply and a shift (an integer divide is ~40 VALU)
};

template <int KW, int TN, bool RELU>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(const ConvArgs p, const int epi_vec, const HaloGeom g) {
  constexpr int NT = 256, BM = 128, BN = TN * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  const int M = p.N * p.OH * p.OW;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM;
  const int n0 = (tile % tiles_n) * BN;

  // ---- tile geometry (uniform) ------------------------------------------------------------------------------
  const int s = p.SH;
  const int ct_mask = (1 << g.ct_log2) - 1;
  const int ohw = p.OH * p.OW;
  const int n_img = m0 / ohw;
  const int rem = m0 - n_img * ohw;
  const int oh0 = rem / p.OW, ow0 = rem - oh0 * p.OW;
  const int ih0 = oh0 * s - p.PT, iw0 = ow0 * s - p.PL;
  const int npix = g.hr * g.hc;

  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)(((unsigned)(p.N * p.H * p.W - 1) * p.ldx + (unsigned)p.Cin) * 4u), 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((unsigned)p.Kp * p.Np * 4u),
                                                      0x00020000);

  // ---- halo DMA: instruction i (of a_bytes / 1024) fills LDS bytes [1024 i, 1024 i + 1024) = 64 slots of 16 B; wave w
  // issues instructions w, w + 4, ...  Slot -> (pixel, 16-byte slot of the pixel); the pad slot and everything outside
  // the image read zeros.  The byte offsets do not depend on the chunk (that is the scalar soffset).
  const int ni_a = g.a_bytes >> 10;
  unsigned a_off[HMAXA];
#pragma unroll
  for (int j = 0; j < HMAXA; ++j) {
    const int slot = ((j * 4 + wave) << 6) + lane;
    const int px = slot / 5, sl = slot - px * 5;
    const int r = (int)(((unsigned)px * (unsigned)g.hc_magic) >> 20), c = px - r * g.hc;   // px / hc (px < 1024, hc <= 132: exact)
    const int ih = ih0 + r, iw = iw0 + c;
    const bool ok = sl < 4 && px < npix && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
    a_off[j] = ok ? ((unsigned)((n_img * p.H + ih) * p.W + iw) * p.ldx + sl * 4) * 4u : OOB;
  }
  auto issue_halo = [&](int chunk, int buf) {
    char* dst = lds + buf * g.a_bytes;
#pragma unroll
    for (int j = 0; j < HMAXA; ++j)
      if (j * 4 + wave_u < ni_a) dma16(rs_x, dst + ((j * 4 + wave_u) << 10), a_off[j], chunk * (HCH * 4));
  };

  // ---- weight stages: stage (chunk, kh) = rows [k0/4, k0/4 + 4 KW) of the packed weight, columns [n0, n0 + BN)
  char* const b_lds = lds + g.nbuf_a * g.a_bytes;
  unsigned b_off[HMAXB];
#pragma unroll
  for (int q = 0; q < HMAXB; ++q) {
    const int idx = tid + q * NT;
    const int kq = idx / BN, j = idx - kq * BN;
    const int col = n0 + j < p.Np ? n0 + j : 0;
    b_off[q] = kq < 4 * KW ? ((unsigned)kq * p.Np + col) * 16u : OOB;
  }
  const int b_stage_step = 4 * KW * p.Np * 16;          // bytes of packed weight per stage
  auto issue_b = [&](int st, int buf) {
    char* dst = b_lds + buf * g.b_bytes;
#pragma unroll
    for (int q = 0; q < HMAXB; ++q)
      if (q < g.bpass) dma16(rs_w, dst + ((q * NT + wave_u * 64) << 4), b_off[q], st * b_stage_step);
  };

  f32x16 acc[1][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  // ---- fragment addresses: lane li of wave w is output pixel w*32 + li of the tile = (row r, column c)
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const int pix = wave * 32 + li;
  const int pr = pix >> g.ct_log2, pc = pix & ct_mask;
  const unsigned a_pix = lds0 + (unsigned)((pr * s * g.hc + pc * s) * HPIX + lh * 16);
  const unsigned b_frag = lds0 + (unsigned)(g.nbuf_a * g.a_bytes) + (unsigned)((lh * BN + li) * 16);

  const int nchunk = p.Cin / HCH;
  const int nstage = nchunk * p.KH;

  issue_halo(0, 0);
  issue_b(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  EpiPrefetch<1, TN> pre;
  int chunk = 0, kh = 0;
  for (int st = 0; st < nstage; ++st) {
    if (st == nstage - 1) pre.template issue<4, 1>(p, m0, n0, M, epi_vec);
    const unsigned a_row = a_pix + (unsigned)((chunk & (g.nbuf_a - 1)) * g.a_bytes + kh * g.hc * HPIX);
    const unsigned b_st = b_frag + (unsigned)((st & 1) * g.b_bytes);

    // sub-step (kw, q) = 8 k-values: A fragment = 4 channels (q*2 + lh) of the tap's pixel, B fragments = the matching
    // rows of the stage.  Sub-step u+1's reads are in flight under sub-step u's MFMAs; the next stage's DMA goes behind
    // the first reads, which have nothing to hide under.
    float4 fa[2], fb[2][TN];
#define DH_HALO_FETCH(SET, KWI, Q)                                                                      \
    fa[SET] = lds_rd<(KWI) * HPIX + (Q) * 32>(a_row);                                                   \
    fb[SET][0] = lds_rd<((KWI) * 4 + (Q) * 2) * BN * 16>(b_st);                                          \
    if constexpr (TN > 1) fb[SET][1] = lds_rd<((KWI) * 4 + (Q) * 2) * BN * 16 + 512>(b_st);             \
    if constexpr (TN > 2) fb[SET][2] = lds_rd<((KWI) * 4 + (Q) * 2) * BN * 16 + 1024>(b_st);
#define DH_HALO_MFMA(SET)                                                                               \
    {                                                                                                   \
      float4 a = fa[SET];                                                                               \
      if constexpr (RELU) a = relu4(a);                                                                 \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                  \
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, fb[SET][j].x, acc[0][j], 0, 0, 0);        \
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, fb[SET][j].y, acc[0][j], 0, 0, 0);        \
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, fb[SET][j].z, acc[0][j], 0, 0, 0);        \
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, fb[SET][j].w, acc[0][j], 0, 0, 0);        \
      }                                                                                                 \
    }
    DH_HALO_FETCH(0, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
    {
      const int nst = st + 1;
      if (nst < nstage) {
        issue_b(nst, nst & 1);
        // the halo of the NEXT chunk: with two buffers it is requested at the first stage of this chunk (its buffer was
        // last read a whole chunk ago); with one buffer it has to wait until this chunk's last stage is done (below)
        if (g.nbuf_a == 2 && kh == 0 && chunk + 1 < nchunk) issue_halo(chunk + 1, (chunk + 1) & 1);
      }
    }
    lgkm_wait();
    // compile-time loop over the 2 KW sub-steps: every fetch and its wait sit in straight-line code (see gemm1x1.hip)
    static_for<2 * KW>([&](auto U) {
      constexpr int u = decltype(U)::value;
      constexpr int nu = u + 1;
      if constexpr (nu < 2 * KW) {
        DH_HALO_FETCH(nu & 1, nu >> 1, nu & 1)
        __builtin_amdgcn_sched_barrier(0);
      }
      DH_HALO_MFMA(u & 1)
      if constexpr (nu < 2 * KW) lgkm_wait();
    });
#undef DH_HALO_FETCH
#undef DH_HALO_MFMA

    // advance (chunk, kh); one halo buffer: the next chunk can only be fetched once every wave is done with this one
    ++kh;
    const bool chunk_done = kh == p.KH;
    if (chunk_done) { kh = 0; ++chunk; }
    if (g.nbuf_a == 1 && chunk_done && chunk < nchunk) {
      __syncthreads();
      issue_halo(chunk, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  conv_epilogue<4, 1, 1, TN, false, true, EpiNoHook, false>(p, acc, smem, m0, n0, M, epi_vec, pre);
}

bool geometry(const ConvArgs& a, int tn, HaloGeom* g) {
  const int bn = tn * 32;
  int ct;
  if (a.OW >= 128) {
    if (a.OW % 128) return false;
    ct = 128;
  } else {
    ct = a.OW;
    if (ct < 8 || (ct & (ct - 1)) || (a.OH * a.OW) % 128) return false;
  }
  int lg = 0;
  while ((1 << lg) < ct) ++lg;
  const int rt = 128 / ct;
  g->ct_log2 = lg;
  g->hr = (rt - 1) * a.SH + a.KH;
  g->hc = (ct - 1) * a.SW + a.KW;
  g->a_bytes = (g->hr * g->hc * HPIX + 1023) & ~1023;
  g->b_bytes = (4 * a.KW * bn * 16 + 4095) & ~4095;
  g->bpass = g->b_bytes >> 12;
  g->hc_magic = (1 << 20) / g->hc + 1;
  if ((g->a_bytes >> 10) > 4 * HMAXA || g->bpass > HMAXB) return false;
  // two work-groups per CU need <= 80 KB each; a second halo buffer (prefetch of the next chunk) when it fits
  const int epi = 4 * 32 * (bn + 4) * 4;
  const int one = g->a_bytes + 2 * g->b_bytes, two = 2 * g->a_bytes + 2 * g->b_bytes;
  g->nbuf_a = (two <= 80 * 1024 && a.Cin > HCH) ? 2 : 1;
  const int total = g->nbuf_a == 2 ? two : one;
  return (total > epi ? total : epi) <= 80 * 1024;
}

template <int KW, int TN>
int launch_tn(const ConvArgs& a, int epi, const HaloGeom& g, hipStream_t s) {
  constexpr int BN = TN * 32;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long tiles = (M / 128) * ((a.Cout + BN - 1) / BN);
  if (tiles <= 0 || tiles > 0x7fffffffLL) return DH_EINVAL;
  const int epi_bytes = 4 * 32 * (BN + 4) * 4;
  int lds = g.nbuf_a * g.a_bytes + 2 * g.b_bytes;
  if (lds < epi_bytes) lds = epi_bytes;
  auto kern = a.pre_relu ? conv_halo_kernel<KW, TN, true> : conv_halo_kernel<KW, TN, false>;
  if (lds > 64 * 1024) {
    static LdsLimit lim_t, lim_f;
    lim_t.raise((const void*)conv_halo_kernel<KW, TN, true>, 80 * 1024);
    lim_f.raise((const void*)conv_halo_kernel<KW, TN, false>, 80 * 1024);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), (size_t)lds, s, a, epi, g);
  return check_launch();
}

template <int KW>
int launch_kw(const ConvArgs& a, int tn, int epi, const HaloGeom& g, hipStream_t s) {
  switch (tn) {
    case 1: return launch_tn<KW, 1>(a, epi, g, s);
    case 2: return launch_tn<KW, 2>(a, epi, g, s);
    case 3: return launch_tn<KW, 3>(a, epi, g, s);
  }
  return DH_EINVAL;
}

}  // namespace

bool conv_is_skinny(const ConvArgs& a);

// What the kernel can run: dense K x K with K > 1 somewhere, stride 1 (stride 2 keeps the tap-major kernels: its halo is
// 4x the output tile), Cin a multiple of 16, float input, 16-byte aligned views, no BN prologue / fused up-sampling,
// whole-row output tiles of 128 pixels, LDS budget.
bool conv_halo_supported(const ConvArgs& a0) {
  ConvArgs a = a0;
  a.w_split = 0;
  if (a.x_u8 || a.up2 || a.pre_scale != nullptr || conv_is_skinny(a)) return false;
  if (a.KH * a.KW <= 1 || a.KH > 7 || !(a.KW == 1 || a.KW == 3 || a.KW == 5)) return false;
  if (a.SH != 1 || a.SW != 1 || a.PT < 0 || a.PL < 0) return false;
  if (a.Cin % HCH || a.Cin < 32 || a.ldx % 4 || (reinterpret_cast<uintptr_t>(a.x) & 15)) return false;
  if ((long long)a.N * a.H * a.W * a.ldx * 4 > 0xf0000000LL) return false;
  if (a.OH * a.OW < 1024) return false;      // small maps keep the tap-major kernels (a per-frame rule: never on N)
  HaloGeom g;
  return geometry(a, 1, &g);
}

// The layers a binding gives to this kernel (dh_conv2d_halo_eligible; a rule on the per-frame geometry only): what it can
// run AND what the LDS-DMA tap-major kernel cannot -- Cin not a multiple of 32 (SPNet's 48- and 144-channel 3x3 convs,
// which otherwise fall to the register-gather kernel: 1.17-1.21x faster here).  With Cin % 32 == 0 the two kernels are
// within +-5 % of each other (tools/bench_halo.py, profiles/r03_bench_halo.json), so those layers keep the tap-major
// family and its tiling freedom.
bool conv_halo_eligible(const ConvArgs& a) { return a.Cin % 32 != 0 && conv_halo_supported(a); }

int conv_halo_num_cfgs() { return 3; }

// cfg: 0, 1, 2 -> TN = 1, 2, 3 (output-channel tile 32, 64, 96); < 0: the widest that fits and divides the work well
int launch_conv_halo(const ConvArgs& a, int cfg, int epi, hipStream_t s) {
  if (!conv_halo_supported(a) || (reinterpret_cast<uintptr_t>(a.w) & 15)) return DH_EUNSUPPORTED;
  if (a.Kp % 32 || a.K != a.KH * a.KW * a.Cin) return DH_EINVAL;
  if (cfg >= 3) return DH_EINVAL;
  int tn = cfg + 1;
  HaloGeom g;
  if (cfg < 0) {
    const int np = (a.Cout + 31) / 32;
    tn = np % 3 == 0 ? 3 : (np % 2 == 0 ? 2 : (np >= 3 ? 3 : np));
    while (tn > 1 && !geometry(a, tn, &g)) --tn;
  }
  if (!geometry(a, tn, &g)) return DH_EUNSUPPORTED;
  switch (a.KW) {
    case 1: return launch_kw<1>(a, tn, epi, g, s);
    case 3: return launch_kw<3>(a, tn, epi, g, s);
    case 5: return launch_kw<5>(a, tn, epi, g, s);
  }
  return DH_EUNSUPPORTED;
}

}  // namespace dh
