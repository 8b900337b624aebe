// Internal launch interface between the C-ABI layer (capi.hip) and the gfx950 kernels.
// Argument structs are the public PODs of include/deephar_hip.h (documented there).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "deephar_hip.h"

namespace dh {

using ConvArgs = dh_conv_args;
using ConvSeg = dh_conv_seg;
using DwArgs = dh_dw_args;
using PoolArgs = dh_pool_args;
using EltArgs = dh_elt_args;
using SamArgs = dh_sam_args;

// Work-group b runs on XCD b % 8 (eight XCDs, an L2 each).  -> the position of work-group b in an order that gives every XCD a
// CONTIGUOUS run of [0, nwg): neighbours in that order -- the channel groups of one frame, the output rows of one image whose
// pooling windows overlap -- share an L2 instead of each fetching the lines they have in common.  A bijection on [0, nwg).
__device__ __forceinline__ int xcd_order(int b, int nwg) {
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

int launch_conv_igemm(const ConvArgs& a, int cfg, hipStream_t s);
int launch_normalize_u8(const unsigned char* x, const float* lut, float* y, long long n_pixels, int C, hipStream_t s);
int conv_igemm_pick_cfg(int M, int Cout);
int conv_igemm_num_cfgs();
bool conv_is_skinny(const ConvArgs& a);
bool conv_stem_eligible(const ConvArgs& a);
int gemm1x1_split_num_cfgs();
bool gemm1x1_split_eligible(const ConvArgs& a);
bool conv_halo_eligible(const ConvArgs& a);
int conv_halo_num_cfgs();
int launch_dwconv(const DwArgs& a, hipStream_t s);
int launch_conv_dw_group(const ConvArgs& a, const DwArgs& d, hipStream_t s);
int launch_conv_skinny_pair(const ConvArgs& a, const ConvArgs& b, hipStream_t s);
int launch_conv_splitk_seg(const ConvArgs& a, const ConvSeg& seg, hipStream_t s);
int launch_pool(const PoolArgs& a, hipStream_t s);
int launch_upsample2x_add(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int N, int H,
                          int W, int C, hipStream_t s);
int launch_eltwise(const EltArgs& a, hipStream_t s);
int launch_softargmax2d(const SamArgs& a, hipStream_t s);
int launch_softargmax2d_context(const SamArgs& a, int J, int nctx, float agg_alpha, float* y, int ldy, hipStream_t s);
int launch_context_agg(const float* ys, const float* yc, const float* pc, float* y, int F, int J, int nctx,
                       float alpha, int ldy, hipStream_t s);
int launch_depth_means(const float* h, int ldh, float* hxy, float* hz, int F, int HW, int D, int J,
                       hipStream_t s);
int launch_softargmax1d(const float* hz, const float* grid, float* z, int ldz, float* vz, int F, int D, int J,
                        hipStream_t s);
int launch_kronecker(const float* hm, int ldh, const float* x, int ldx, float* f, int ldf, int B, int P, int J,
                     int C, hipStream_t s);
int launch_global_maxmin_softmax(const float* x, int ldx, float* y, int B, int P, int C, int softmax,
                                 hipStream_t s);
int launch_copy_channels(const float* x, int ldx, float* y, int ldy, long long npix, int C, hipStream_t s);
int launch_zeropad(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, int PT, int PL,
                   hipStream_t s);
int launch_depth_from_maps(const float* d, int ldd, const float* h, int ldh, float* z, int ldz, int F, int HW,
                           int J, hipStream_t s);

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel's code object on ONE device: a process that drives
// several GPUs (Model(device=...), dh_plan executors on other devices) must raise it on each.  One flag word per call
// site, one bit per device ordinal; a race sets the attribute twice, which is harmless.
struct LdsLimit {
  unsigned long long done = 0;
  void raise(const void* kern, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return;
    hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done |= bit;
  }
};

inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DH_OK : DH_ELAUNCH;
}

}  // namespace dh
