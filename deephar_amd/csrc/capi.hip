// extern "C" surface of libdeephar_hip.so (see include/deephar_hip.h for the contract and the
// reference interfaces each entry point replaces).  Thin: validate, forward to the launcher.
#include <stddef.h>
#include <string.h>
#include <vector>
#include "dh_kernels.h"

using namespace dh;

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
// the fields added in round 6 (dh_conv_args.x_resample, dh_dw_args.up_in) sit in what was padding: serialised plans and
// foreign callers built against the older header keep their layout
static_assert(sizeof(dh_conv_seg) == 24, "dh_conv_seg layout");
static_assert(sizeof(dh_conv_args) == 200 && offsetof(dh_conv_args, y_pool) == 192 && offsetof(dh_conv_args, x_resample) == 188,
              "dh_conv_args layout");
static_assert(sizeof(dh_dw_args) == 88 && offsetof(dh_dw_args, up_in) == 84, "dh_dw_args layout");
static_assert(sizeof(dh_sam_args) == 112 && offsetof(dh_sam_args, xy_times_conf) == 108, "dh_sam_args layout");
static inline int rc_of(hipError_t e) { return e == hipSuccess ? DH_OK : DH_ELAUNCH; }

extern "C" {

int dh_version(void) { return 100; }

const char* dh_error_string(int rc) {
  switch (rc) {
    case DH_OK: return "ok";
    case DH_EINVAL: return "invalid argument or shape";
    case DH_EUNSUPPORTED: return "configuration not supported by the gfx950 kernels";
    case DH_ELAUNCH: return "HIP launch/runtime error";
    default: return "unknown error";
  }
}

int dh_device_info(int dev, char* arch_name, int arch_name_len, int* cu_count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || dev < 0 || dev >= n) return DH_ELAUNCH;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return DH_ELAUNCH;
  if (arch_name != nullptr && arch_name_len > 0) {
    strncpy(arch_name, p.gcnArchName, (size_t)arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  if (cu_count != nullptr) *cu_count = p.multiProcessorCount;
  return DH_OK;
}

int dh_conv2d_packed_dims(int KH, int KW, int Cin, int Cout, int* Kp, int* Np) {
  if (KH <= 0 || KW <= 0 || Cin <= 0 || Cout <= 0) return DH_EINVAL;
  const int K = KH * KW * Cin;
  if (Kp != nullptr) *Kp = (K + 31) / 32 * 32;
  if (Np != nullptr) *Np = (Cout + 31) / 32 * 32;
  return DH_OK;
}

int dh_conv2d_pack_weights_host(const float* w, float* packed, int KH, int KW, int Cin, int Cout) {
  int Kp, Np;
  if (w == nullptr || packed == nullptr || dh_conv2d_packed_dims(KH, KW, Cin, Cout, &Kp, &Np) != DH_OK)
    return DH_EINVAL;
  const int K = KH * KW * Cin;
  memset(packed, 0, sizeof(float) * (size_t)Kp * Np);
  // HWIO flattened k = (kh*KW + kw)*Cin + ci ; packed[(k/4)*Np*4 + n*4 + k%4]
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < Cout; ++n)
      packed[((size_t)(k >> 2) * Np + n) * 4 + (k & 3)] = w[(size_t)k * Cout + n];
  return DH_OK;
}

static inline uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int dh_conv2d_pack_weights_split_host(const float* w, uint16_t* packed, int KH, int KW, int Cin, int Cout) {
  int Kp, Np;
  if (w == nullptr || packed == nullptr || dh_conv2d_packed_dims(KH, KW, Cin, Cout, &Kp, &Np) != DH_OK)
    return DH_EINVAL;
  const int K = KH * KW * Cin;
  memset(packed, 0, sizeof(uint16_t) * (size_t)3 * Kp * Np);
  // unit (kg, part, n) = 8 bf16 at ((kg * 3 + part) * Np + n) * 8; element k % 8 inside
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < Cout; ++n) {
      const float x = w[(size_t)k * Cout + n];
      const uint16_t h1 = bf16_rne(x);
      const float r1 = x - bf16_to_f(h1);
      const uint16_t h2 = bf16_rne(r1);
      const float r2 = r1 - bf16_to_f(h2);
      const uint16_t h3 = bf16_rne(r2);
      const size_t base = ((size_t)(k >> 3) * 3 * Np + n) * 8 + (k & 7);
      packed[base] = h1;
      packed[base + (size_t)Np * 8] = h2;
      packed[base + (size_t)2 * Np * 8] = h3;
    }
  return DH_OK;
}

int dh_conv2d_num_tile_cfgs(void) { return conv_igemm_num_cfgs(); }
int dh_conv2d_num_split_tile_cfgs(void) { return gemm1x1_split_num_cfgs(); }
int dh_conv2d_uses_split_k(const dh_conv_args* a) { return a != nullptr && conv_is_skinny(*a) ? 1 : 0; }
int dh_conv2d_uses_first_layer_kernel(const dh_conv_args* a) {
  return a != nullptr && !conv_is_skinny(*a) && conv_stem_eligible(*a) ? 1 : 0;
}
int dh_conv2d_pick_tile_cfg(int M, int Cout) { return conv_igemm_pick_cfg(M, Cout); }
int dh_conv2d_split_eligible(const dh_conv_args* a) { return a != nullptr && gemm1x1_split_eligible(*a) ? 1 : 0; }
int dh_conv2d_halo_eligible(const dh_conv_args* a) { return a != nullptr && conv_halo_eligible(*a) ? 1 : 0; }
int dh_conv2d_num_halo_tile_cfgs(void) { return conv_halo_num_cfgs(); }

int dh_conv2d_f32(const dh_conv_args* a, int tile_cfg, void* stream) {
  if (a == nullptr || a->x == nullptr || a->w == nullptr || a->y == nullptr) return DH_EINVAL;
  if ((a->pre_scale == nullptr) != (a->pre_shift == nullptr)) return DH_EINVAL;
  if ((a->post_scale == nullptr) != (a->post_shift == nullptr)) return DH_EINVAL;
  if (a->SH <= 0 || a->SW <= 0 || a->KH <= 0 || a->KW <= 0) return DH_EINVAL;
  if (a->res2_down && (a->res2 == nullptr || a->up2 || (a->OH & 1) || (a->OW & 1))) return DH_EINVAL;
  return launch_conv_igemm(*a, tile_cfg, S(stream));
}

int dh_conv2d_dw_group_f32(const dh_conv_args* a, const dh_dw_args* d, void* stream) {
  if (a == nullptr || d == nullptr || a->x == nullptr || a->w == nullptr || a->y == nullptr || d->x == nullptr ||
      d->w == nullptr || d->y == nullptr)
    return DH_EINVAL;
  if ((a->pre_scale == nullptr) != (a->pre_shift == nullptr) || (a->post_scale == nullptr) != (a->post_shift == nullptr) ||
      (d->pre_scale == nullptr) != (d->pre_shift == nullptr))
    return DH_EINVAL;
  return launch_conv_dw_group(*a, *d, S(stream));
}

// the argument checks dh_conv2d_f32 and launch_conv_igemm make, for the entry points that go to the skinny-conv kernel directly
static int check_skinny_args(const dh_conv_args* c) {
  if (c == nullptr || c->x == nullptr || c->w == nullptr || c->y == nullptr) return DH_EINVAL;
  if ((c->pre_scale == nullptr) != (c->pre_shift == nullptr) || (c->post_scale == nullptr) != (c->post_shift == nullptr)) return DH_EINVAL;
  if (c->SH <= 0 || c->SW <= 0 || c->KH <= 0 || c->KW <= 0 || c->N <= 0 || c->Cin <= 0 || c->Cout <= 0 || c->OH <= 0 || c->OW <= 0)
    return DH_EINVAL;
  if (c->Kp % 32 != 0 || c->Np % 32 != 0 || c->Kp < c->K || c->Np < c->Cout || c->K != c->KH * c->KW * c->Cin) return DH_EINVAL;
  if ((long long)c->N * c->H * c->W > 0x7fffffffLL || (long long)c->N * c->OH * c->OW * (c->up2 ? 4 : 1) > 0x7fffffffLL) return DH_EINVAL;
  return DH_OK;
}

int dh_conv2d_pair_f32(const dh_conv_args* a, const dh_conv_args* b, void* stream) {
  if (check_skinny_args(a) != DH_OK || check_skinny_args(b) != DH_OK) return DH_EINVAL;
  return launch_conv_skinny_pair(*a, *b, S(stream));
}

int dh_conv2d_seg_f32(const dh_conv_args* a, const dh_conv_seg* seg, void* stream) {
  if (seg == nullptr || check_skinny_args(a) != DH_OK) return DH_EINVAL;
  return launch_conv_splitk_seg(*a, *seg, S(stream));
}

int dh_normalize_u8_f32(const uint8_t* x, const float* lut, float* y, int64_t n_pixels, int C, void* stream) {
  if (x == nullptr || lut == nullptr || y == nullptr) return DH_EINVAL;
  return launch_normalize_u8(x, lut, y, n_pixels, C, S(stream));
}

int dh_dwconv2d_f32(const dh_dw_args* a, void* stream) {
  if (a == nullptr || a->x == nullptr || a->w == nullptr || a->y == nullptr) return DH_EINVAL;
  if ((a->pre_scale == nullptr) != (a->pre_shift == nullptr)) return DH_EINVAL;
  return launch_dwconv(*a, S(stream));
}

int dh_pool2d_f32(const dh_pool_args* a, void* stream) {
  if (a == nullptr || a->x == nullptr || a->y == nullptr || a->SH <= 0 || a->SW <= 0) return DH_EINVAL;
  return launch_pool(*a, S(stream));
}

int dh_upsample2x_add_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int N, int H,
                          int W, int C, void* stream) {
  if (b == nullptr || y == nullptr) return DH_EINVAL;
  return launch_upsample2x_add(a, lda, b, ldb, y, ldy, N, H, W, C, S(stream));
}

int dh_eltwise_f32(const dh_elt_args* a, void* stream) {
  if (a == nullptr) return DH_EINVAL;
  if ((a->scale == nullptr) != (a->shift == nullptr)) return DH_EINVAL;
  return launch_eltwise(*a, S(stream));
}

int dh_softargmax2d_f32(const dh_sam_args* a, void* stream) {
  if (a == nullptr) return DH_EINVAL;
  return launch_softargmax2d(*a, S(stream));
}

int dh_softargmax2d_context_f32(const dh_sam_args* a, int J, int nctx, float agg_alpha, float* y, int ldy,
                                void* stream) {
  if (a == nullptr) return DH_EINVAL;
  return launch_softargmax2d_context(*a, J, nctx, agg_alpha, y, ldy, S(stream));
}

int dh_context_aggregation_f32(const float* ys, const float* yc, const float* pc, float* y, int F, int J,
                               int nctx, float alpha, int ldy, void* stream) {
  if (ys == nullptr || yc == nullptr || pc == nullptr || y == nullptr || ldy < 2) return DH_EINVAL;
  return launch_context_agg(ys, yc, pc, y, F, J, nctx, alpha, ldy, S(stream));
}

int dh_depth_means_f32(const float* h, int ldh, float* hxy, float* hz, int F, int HW, int D, int J,
                       void* stream) {
  if (h == nullptr || (hxy == nullptr && hz == nullptr)) return DH_EINVAL;
  return launch_depth_means(h, ldh, hxy, hz, F, HW, D, J, S(stream));
}

int dh_softargmax1d_f32(const float* hz, const float* grid, float* z, int ldz, float* vz, int F, int D, int J,
                        void* stream) {
  if (hz == nullptr || grid == nullptr) return DH_EINVAL;
  return launch_softargmax1d(hz, grid, z, ldz, vz, F, D, J, S(stream));
}

int dh_kronecker_f32(const float* hm, int ldh, const float* x, int ldx, float* f, int ldf, int B, int P, int J,
                     int C, void* stream) {
  if (hm == nullptr || x == nullptr || f == nullptr) return DH_EINVAL;
  return launch_kronecker(hm, ldh, x, ldx, f, ldf, B, P, J, C, S(stream));
}

int dh_global_maxmin_softmax_f32(const float* x, int ldx, float* y, int B, int P, int C, int softmax,
                                 void* stream) {
  if (x == nullptr || y == nullptr) return DH_EINVAL;
  return launch_global_maxmin_softmax(x, ldx, y, B, P, C, softmax, S(stream));
}

int dh_copy_channels_f32(const float* x, int ldx, float* y, int ldy, int64_t npix, int C, void* stream) {
  if (x == nullptr || y == nullptr) return DH_EINVAL;
  return launch_copy_channels(x, ldx, y, ldy, (long long)npix, C, S(stream));
}

int dh_zeropad2d_f32(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, int PT, int PL,
                     void* stream) {
  if (x == nullptr || y == nullptr) return DH_EINVAL;
  return launch_zeropad(x, y, B, H, W, C, OH, OW, PT, PL, S(stream));
}

int dh_depth_from_maps_f32(const float* d, int ldd, const float* h, int ldh, float* z, int ldz, int F, int HW,
                           int J, void* stream) {
  if (d == nullptr || h == nullptr || z == nullptr) return DH_EINVAL;
  return launch_depth_from_maps(d, ldd, h, ldh, z, ldz, F, HW, J, S(stream));
}

// ---- graphs / events ---------------------------------------------------------------------------
int dh_graph_begin_capture(void* stream) {
  return rc_of(hipStreamBeginCapture(S(stream), hipStreamCaptureModeThreadLocal));
}

int dh_graph_end_capture(void* stream, void** graph_exec_out) {
  if (graph_exec_out == nullptr) return DH_EINVAL;
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture(S(stream), &g) != hipSuccess || g == nullptr) return DH_ELAUNCH;
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) return DH_ELAUNCH;
  *graph_exec_out = ge;
  return DH_OK;
}

int dh_graph_launch(void* graph_exec, void* stream) {
  if (graph_exec == nullptr) return DH_EINVAL;
  return rc_of(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), S(stream)));
}

int dh_graph_destroy(void* graph_exec) {
  if (graph_exec == nullptr) return DH_OK;
  return rc_of(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)));
}

int dh_event_create(void** event_out) {
  if (event_out == nullptr) return DH_EINVAL;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return DH_ELAUNCH;
  *event_out = e;
  return DH_OK;
}
int dh_event_record(void* event, void* stream) {
  return rc_of(hipEventRecord(reinterpret_cast<hipEvent_t>(event), S(stream)));
}
int dh_event_synchronize(void* event) { return rc_of(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event))); }
int dh_event_elapsed_ms(void* start, void* stop, float* ms_out) {
  if (ms_out == nullptr) return DH_EINVAL;
  return rc_of(hipEventElapsedTime(ms_out, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
}
int dh_event_destroy(void* event) { return rc_of(hipEventDestroy(reinterpret_cast<hipEvent_t>(event))); }
int dh_stream_synchronize(void* stream) { return rc_of(hipStreamSynchronize(S(stream))); }

int dh_stream_create(void** stream_out) {
  if (stream_out == nullptr) return DH_EINVAL;
  hipStream_t st;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return DH_ELAUNCH;
  *stream_out = st;
  return DH_OK;
}
int dh_stream_destroy(void* stream) { return rc_of(hipStreamDestroy(S(stream))); }
int dh_event_create_sync(void** event_out) {
  if (event_out == nullptr) return DH_EINVAL;
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return DH_ELAUNCH;
  *event_out = e;
  return DH_OK;
}
int dh_stream_wait_event(void* stream, void* event) {
  return rc_of(hipStreamWaitEvent(S(stream), reinterpret_cast<hipEvent_t>(event), 0));
}

// one wave that holds its stream for `us` microseconds (constant-rate 100 MHz counter; bounded by an iteration cap)
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  for (int i = 0; i < (1 << 22) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(8);
}
int dh_stream_spin_us(void* stream, int us) {
  if (us < 0 || us > 20000) return DH_EINVAL;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, S(stream), (long long)us * 100);
  return check_launch();
}

}  // extern "C"
