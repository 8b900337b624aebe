/*
 * deephar_hip.h -- C-ABI of libdeephar_hip.so, the MI355X (gfx950) execution back-end for the
 * pose-regression hot path of dluvizon/deephar.
 *
 * The reference has no FFI of its own: its arithmetic is reached through Keras 2.1.4 layer objects that
 * lower to TensorFlow 1.6 kernels (reference deephar/layers.py:6-42, requirements.txt:2-3).  The seam is
 * therefore "what a Keras layer call computes".  Each entry point below names the Keras layer(s) /
 * reference function it replaces; the Python host (deephar_amd/) and any other host bind exactly these.
 *
 * Conventions
 *   - fp32, NHWC ("channels_last", reference deephar/config.py:4).  A tensor view is (pointer, ld) where
 *     ld = number of floats between consecutive pixels, so channel slices / concatenation targets are
 *     expressed by pointer offset + ld (no copies for keras.layers.concatenate / Lambda slicing).
 *   - all pointers are DEVICE pointers unless the name ends in _host; `stream` is a hipStream_t passed as
 *     void*; every call is asynchronous on that stream and captures cleanly into a hipGraph.
 *   - return value: 0 = OK, <0 = error (see dh_error_string).  Nothing throws, nothing allocates.
 */
#ifndef DEEPHAR_HIP_H_
#define DEEPHAR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_OK 0
#define DH_EINVAL (-1)       /* bad argument / shape */
#define DH_EUNSUPPORTED (-2) /* valid request the kernels do not cover */
#define DH_ELAUNCH (-3)      /* HIP reported a launch/runtime error */

int dh_version(void);
const char* dh_error_string(int rc);
/* name of the gfx target of device `dev`, CU count; rc<0 when no HIP device is visible */
int dh_device_info(int dev, char* arch_name, int arch_name_len, int* cu_count);

/* ---------------------------------------------------------------------------------------------------
 * Conv2D (+ folded BatchNormalization / ReLU / add / UpSampling2D):
 *   replaces layers.conv2d (layers.py:66-71) and its compositions conv_bn (:202), conv_act (:219),
 *   conv_bn_act (:230), act_conv_bn (:258), act_conv (:317); the pointwise half of sepconv2d (:74-80,
 *   288-301); common.residual_unit's BN->ReLU->conv chains (models/common.py:25-67).
 *   y[m,co] = relu?( (sum_k pro(x)[m,k] * w[k,co]) * post_scale[co] + post_shift[co] + res1 + res2 )
 *   pro(v)  = relu?( v * pre_scale[ci] + pre_shift[ci] ), applied to in-bounds taps only
 *   Padding is explicit (pt, pl) so TF-"SAME" asymmetric padding is the caller's arithmetic.
 *   up2 = 1 fuses keras UpSampling2D((2,2)) + add (reception.py:122-127): y is [N,2*OH,2*OW,Cout],
 *   res1 is read at conv resolution, res2 at the up-sampled resolution.
 * ------------------------------------------------------------------------------------------------- */
typedef struct dh_conv_args {
  const float* x;
  const float* w; /* packed by dh_conv2d_pack_weights_host */
  float* y;
  const float* pre_scale;
  const float* pre_shift;
  const float* post_scale;
  const float* post_shift;
  const float* res1;
  const float* res2;
  const float* in_lut; /* x_u8 only: [Cin][256] value of every byte after the loader's normalisation */
  int32_t N, H, W, Cin, ldx;
  int32_t OH, OW, Cout, ldy;
  int32_t KH, KW, SH, SW, PT, PL;
  int32_t K;      /* KH*KW*Cin */
  int32_t Kp, Np; /* padded dims of the packed weight */
  int32_t ldr1, ldr2;
  int32_t pre_relu, post_relu;
  int32_t up2;
  int32_t x_u8; /* 1: x points to uint8 frames [N,H,W,ldx]; every byte goes through in_lut before the (optional)
                   BN prologue, zero padding is applied after it.  This is utils/transform.normalize_channels
                   (transform.py:212-231: /255, power, -0.5, *2 in float32) fused into the first convolution; only
                   the general K x K path (tile_cfg < 0, or one of its tilings 0..8) and the first-layer kernel
                   (dh_conv2d_uses_first_layer_kernel) take it */
  int32_t w_split; /* weight layout.  0: fp32, [Kp/4][Np][4], K tap-major.  2: fp32, same container, K chunk-major for the
                      halo-resident K x K kernel (see dh_conv2d_halo_eligible).
                      1: `w` was packed by dh_conv2d_pack_weights_split_host (every weight split exactly into three bf16
                      parts) and the convolution runs on the bf16 matrix cores: fp32 activations are split the same way
                      on the fly, six of the nine partial products are accumulated in fp32 (gemm1x1s.hip).  Same
                      inputs / outputs / epilogue; per-product error <= 2^-23 relative, i.e. below the rounding of the
                      fp32 accumulation -- NOT bit-identical to w_split = 0.  Only shapes the LDS-DMA GEMM covers
                      (pointwise, or K x K with Cin % 32 == 0; 16-byte aligned x; no BN prologue), else DH_EUNSUPPORTED */
  int32_t res2_down; /* 1: res2 is at HALF the output resolution, [N, OH/2, OW/2, Cout]: out(oh, ow) += res2(oh/2, ow/2), i.e.
                        add([., UpSampling2D((2, 2))(res2)]) with the up-sampling folded into the residual read
                        (reception.py:122-127: `b = UpSampling2D((2, 2))(b); x = add([a, b])` fused into the convolution
                        that produces a).  Needs OH, OW even and up2 = 0 */
  int32_t ldyp;
  int32_t x_resample; /* [r06] skinny-conv layers only (dh_conv2d_uses_split_k; DH_EUNSUPPORTED elsewhere): x is not stored at the
                         resolution H x W the convolution sees --
                           1: x is [N, H/2, W/2, Cin] and is read as UpSampling2D((2, 2))(x)      (spnet.py:89-91: the action head's
                              conv3 on the up-sampled class maps);
                           2: x is [N, 2H, 2W, Cin] and is read through MaxPooling2D((2, 2));
                           3: ... through layers.max_min_pooling((2, 2)) = max + min of the window (layers.py:411-425; spnet.py:77-79).
                         BN / ReLU prologue and zero padding act on the resampled pixels: bit for bit the convolution of the
                         tensor a stand-alone up-sampling / pooling launch would have written.  The field sits in what was
                         padding in front of y_pool: a zero-initialised struct is unchanged */
  float* y_pool; /* optional second output: MaxPooling2D((2, 2)) of the convolution's FINAL output (after BN / residuals /
                    ReLU), [N, OH/2, OW/2, Cout] with pixel pitch ldyp -- the hourglass reads every level both at full and
                    at half resolution (reception.py:105-116: x = ...; MaxPooling2D((2, 2))(x)), and a stand-alone pool
                    has to read the whole tensor back.  Built for OW == 32, OH even, 16-byte aligned rows, no up2, on the
                    tilings whose waves own 32-row blocks in pairs (an image row per wave, the pair pools through the
                    epilogue's LDS slab), and [r06] for OW == 16 / OW == 8 with OH * OW a multiple of 32 on every tiling
                    with 32-row waves (a wave's block is two / four whole image rows: it pools its own slab -- SPNet's
                    down path below 32 x 32, common.py:70-86); anything else returns DH_EUNSUPPORTED */
} dh_conv_args;

/* padded dims of the packed weight for a [KH,KW,Cin,Cout] (Keras HWIO) kernel */
int dh_conv2d_packed_dims(int KH, int KW, int Cin, int Cout, int* Kp, int* Np);
/* host-side repack HWIO -> [Kp/4][Np][4]; `packed_host` holds Kp*Np floats */
int dh_conv2d_pack_weights_host(const float* w_hwio_host, float* packed_host, int KH, int KW, int Cin,
                                int Cout);
/* split packing for w_split = 1: `packed_host` holds 3 * Kp * Np uint16 (bf16 bit patterns) laid out
 * [Kp/8][3 parts][Np][8]; same Kp / Np as dh_conv2d_packed_dims */
int dh_conv2d_pack_weights_split_host(const float* w_hwio_host, uint16_t* packed_host, int KH, int KW, int Cin,
                                      int Cout);
/* tile_cfg < 0: library heuristic; 0..dh_conv2d_num_tile_cfgs()-1 forces a tiling (autotuning hook): 0..8 the general
 * implicit-GEMM kernel, 9..17 the same tile shapes on the LDS-DMA GEMM (pointwise, K x K with Cin % 32 == 0; also with a
 * BatchNormalization prologue when pointwise).  A tiling that does not cover the layer returns DH_EUNSUPPORTED.  All
 * tilings of a layer give the same bits. */
int dh_conv2d_num_tile_cfgs(void);
int dh_conv2d_num_split_tile_cfgs(void); /* tilings of the w_split = 1 kernels: tile_cfg in [0, this) */
int dh_conv2d_pick_tile_cfg(int M, int Cout);
/* Inputs must be FINITE.  K is padded to the kernels' step with zero weights, and on the LDS-DMA GEMM a padded k slot of
 * a pixel holds the following floats in memory (the next pixel's first channels, clamped inside the buffer): an Inf / NaN
 * there would reach this pixel's output as 0 * Inf.  The first-layer kernel (raw frames) and the skinny-conv
 * kernel zero the padded A operand instead; activations inside a model are finite by construction. */
/* 1 when dh_conv2d_f32 runs this convolution on the skinny-conv kernel (conv_splitk.hip: a tiny output map -- at most 256
 * positions per frame / clip, at most 256 output channels, K >= 64 [r05; was K >= 768, Cin % 4 == 0]: the action heads,
 * deephar/models/spnet.py:51-148, action.py:20-42, and the coarse heat-map heads, spnet.py:24-48): a rule on OH*OW, K, Cout and
 * Cin only, so that a layer's result bits never depend on tiling choice, batch size or alignment.  Such a layer takes
 * fp32-packed weights (w_split = 0) and ignores tile_cfg. */
int dh_conv2d_uses_split_k(const dh_conv_args* a);
/* 1 when dh_conv2d_f32 runs this convolution on the first-layer kernel (conv_stem.hip): 3 dense input channels (ldx = 3,
 * float or x_u8 frames), stride 2, 3x3 or 7x7, 128 output columns, an even number of output rows, 32 or 64 output
 * channels, fp32 tap-major weights (w_split = 0), BN / ReLU epilogue only -- layers.conv_bn_act of reception.py:61-66 and
 * the 7x7 entry conv of spnet.py:317-322 on 256 x 256 frames.  Like the split-K rule it looks at the layer's geometry
 * only; such a layer ignores tile_cfg. */
int dh_conv2d_uses_first_layer_kernel(const dh_conv_args* a);
/* 1 when dh_conv2d_f32 would accept this convolution with w_split = 1 (every field but `w` / `w_split` filled in as for
 * the launch): an LDS-DMA GEMM shape (pointwise, or K x K with Cin % 32 == 0, no fused up-sampling), 16-byte aligned
 * float input, no BN prologue, not a split-K layer, operands within the 32-bit buffer offsets of the kernel.  A
 * binding asks this BEFORE it packs the weights, so that a layer is never bound with a packing its launch rejects. */
int dh_conv2d_split_eligible(const dh_conv_args* a);
/* 1 when this convolution belongs to the halo-resident K x K kernel (conv_halo.hip): dense K x K, stride 1, Cin % 16
 * == 0 but Cin % 32 != 0 (the layers the LDS-DMA tap-major kernel cannot take), maps of >= 1024 pixels per frame whose
 * rows tile into runs of 128 output pixels, 16-byte aligned float input, no BN prologue / fused up-sampling.  (The
 * kernel itself also runs Cin % 32 == 0 when asked with w_split = 2.)  A rule on the per-frame geometry only -- never on the batch size or on timing --
 * because that kernel sums K chunk-major ([16-channel chunk][kh][kw][16]) and its results differ in the last bits
 * from the tap-major kernels'.  Such a layer is launched with w_split = 2 and weights packed in that K order: HWIO
 * re-ordered to [Cin/16][KH][KW][16] rows, then dh_conv2d_pack_weights_host with (1, 1, KH*KW*Cin, Cout);
 * tile_cfg in [0, dh_conv2d_num_halo_tile_cfgs()) or < 0. */
int dh_conv2d_halo_eligible(const dh_conv_args* a);
int dh_conv2d_num_halo_tile_cfgs(void);
int dh_conv2d_f32(const dh_conv_args* a, int tile_cfg, void* stream);

/* Stand-alone version of the same normalisation for inputs that do not feed a convolution directly:
 * y[i, c] = lut[c*256 + x[i, c]] for n_pixels x C bytes (transform.py:212-231). */
int dh_normalize_u8_f32(const uint8_t* x, const float* lut, float* y, int64_t n_pixels, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Depthwise KxK conv, stride 1, explicit padding: the depthwise half of keras SeparableConv2D
 * (layers.py:74-80, 288-301; depth_multiplier 1, no bias).  w is [KH,KW,C] (= Keras [KH,KW,C,1]).
 * ------------------------------------------------------------------------------------------------- */
typedef struct dh_dw_args {
  const float* x;
  const float* w;
  float* y;
  const float* pre_scale;
  const float* pre_shift;
  int32_t N, H, W, C, ldx, ldy;
  int32_t KH, KW, PT, PL;
  int32_t pre_relu;
  int32_t up_in; /* [r06] 1: x is stored at HALF resolution, [N, H/2, W/2, C] (H, W even), and the convolution reads
                    UpSampling2D((2, 2))(x): input pixel (h, w) = x(h/2, w/2) -- the nearest up-sampling in front of SPNet's
                    up-scaling unit (deephar/models/common.py:89-108: residual_unit(UpSampling2D(x))) is never written out;
                    BN / ReLU prologue and zero padding as without it (both act on the up-sampled pixels).  The field
                    occupies what was tail padding of the struct: a caller that zero-initialises it is unchanged */
} dh_dw_args;
int dh_dwconv2d_f32(const dh_dw_args* a, void* stream);

/* [r06] The two independent first layers of a pre-activation residual unit in ONE launch (deephar/models/common.py:25-67:
 * `shortcut = conv2d(relu(BN(x)), out, (1, 1))` beside `sepconv2d(relu(BN(x)), ...)`, whose depthwise half is `dw`): the
 * 1x1 convolution `conv` (BatchNormalization + ReLU prologue, w_split = 0, any epilogue of dh_conv_args but up2 / y_pool)
 * and the 5x5 depthwise convolution `dw` (BatchNormalization + ReLU prologue).  Work-groups of the one grid run either
 * kernel's own code, so the results are bit for bit those of dh_conv2d_f32 + dh_dwconv2d_f32; what is saved is one
 * dependent launch (~5 us plus the shorter of the two kernels) -- the latency regime of a couple of clips per call
 * (exp/pennaction/eval_speed2d.py).  DH_EUNSUPPORTED for any pair outside that description (call the two entry points). */
int dh_conv2d_dw_group_f32(const dh_conv_args* conv, const dh_dw_args* dw, void* stream);

/* [r06] A skinny-conv layer (dh_conv2d_uses_split_k) whose input is a concatenation that is never written out:
 *   concatenate([MaxPooling2D((2, 2), strides=(pool_sh, 2), padding='same')(x), x2])
 * -- SPNet's action head pools its pose and appearance features and concatenates them with the features handed on by the
 * previous head in front of its second residual unit (deephar/models/spnet.py:126-141: x1, x2 = maxpooling2d(...);
 * concat_tensorlist([x1, x2, xa]); residual(...)).  `a` describes the convolution as dh_conv2d_f32 would see the concatenated
 * tensor (H, W: the pooled extent; Cin: all channels; ldx: the pixel pitch of x; x_resample = 0); `seg` says where the second
 * run of channels lives.  x is [N, H * pool_sh, 2 W, c_split] (pool_sh = 1: windows of two rows starting at EVERY row, the last
 * one a single row -- the reference's time_stride = 1 for clips shorter than 16 frames; pool_sh = 2: disjoint windows).  The
 * BatchNormalization / ReLU prologue and the zero padding act on the concatenated pixels: bit for bit dh_conv2d_f32 on the
 * tensor a pooling launch and the concatenation would have written. */
typedef struct dh_conv_seg {
  const float* x2;   /* channels [c_split, Cin): [N, H, W, Cin - c_split], pixel pitch ldx2; may be NULL when c_split == Cin */
  int32_t ldx2;
  int32_t c_split;   /* channels [0, c_split) are the pooled x */
  int32_t pool_sh;   /* 1 or 2: row stride of the pooling window (its column stride is 2) */
  int32_t reserved;  /* 0 */
} dh_conv_seg;
int dh_conv2d_seg_f32(const dh_conv_args* a, const dh_conv_seg* seg, void* stream);

/* [r06] Two INDEPENDENT convolutions with a tiny output map in ONE launch -- in SPNet's action head (deephar/models/spnet.py:
 * 113-133) the residual unit on the pose features and `conv2d(af, num_visual_features, (1, 1))` on the appearance features
 * meet only at the concatenation behind them.  Both must be layers of the skinny-conv kernel (dh_conv2d_uses_split_k), read
 * their inputs as stored (x_resample = 0) and neither may read or overwrite what the other writes (the caller's business:
 * they run concurrently).  Work-groups [0, tiles of a) run `a`, the rest `b`, each with the code its own launch would run:
 * the results are bit for bit those of two dh_conv2d_f32 calls.  DH_EUNSUPPORTED for any other pair. */
int dh_conv2d_pair_f32(const dh_conv_args* a, const dh_conv_args* b, void* stream);

/* MaxPooling2D (reception.py:74,86,108,115; layers.py:92-97), padding cells ignored;
 * mode 1 = layers.max_min_pooling (layers.py:411-425): maxpool(x) - maxpool(-x) */
typedef struct dh_pool_args {
  const float* x;
  float* y;
  int32_t N, H, W, C, ldx;
  int32_t OH, OW, ldy;
  int32_t KH, KW, SH, SW, PT, PL;
  int32_t mode;
} dh_pool_args;
int dh_pool2d_f32(const dh_pool_args* a, void* stream);

/* UpSampling2D((2,2)) [+ add]: y[n,h,w,c] = a[n,h,w,c] + b[n,h/2,w/2,c]; a may be NULL */
int dh_upsample2x_add_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int N, int H,
                          int W, int C, void* stream);

/* element-wise glue: keras add / multiply / Activation('sigmoid') / standalone BatchNormalization+ReLU.
 *   op 0: y = relu?( (a*scale+shift) + b + c )    op 1: y = (a*scale+shift) * b
 *   op 2: y = sigmoid( (a*scale+shift) + b )      bcast_b: b has one channel, broadcast over C */
typedef struct dh_elt_args {
  const float* a;
  const float* b;
  const float* c;
  float* y;
  const float* scale;
  const float* shift;
  int32_t lda, ldb, ldc, ldy;
  int64_t npix;
  int32_t C;
  int32_t relu;
  int32_t op;
  int32_t bcast_b;
} dh_elt_args;
int dh_eltwise_f32(const dh_elt_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Soft-argmax decoder: activations.channel_softmax_2d (activations.py:3-16) + layers.lin_interpolation_2d /
 * softargmax2d (layers.py:122-129,160-200) + keypoint_confidence / build_joints_probability
 * (layers.py:107-119, blocks.py:328-343) in one pass over the maps.
 *   xy[f,c]        = sum_hw softmax(alpha*h)[h,w,c] * (gx[w], gy[h])
 *   conf_raw[f,c]  = max over 2x2 windows of conf_scale * (sum of the 4 raw values)
 *   conf_prob[f,c] = the same on the soft-max probabilities (spnet.py:183)
 *   prob           = the probability maps themselves (needed by kronecker_prod), may be NULL
 *   gmax[f,c]      = max_hw h  (GlobalMaxPooling2D on the raw maps, reception.py:216), may be NULL
 * ------------------------------------------------------------------------------------------------- */
typedef struct dh_sam_args {
  const float* h;
  const float* gx;
  const float* gy;
  float* xy;
  float* conf_raw;
  float* conf_prob;
  float* prob;
  float* gmax;
  int32_t F, H, W, C, ldh, ldxy, ldcr, ldcp, ldp;
  float alpha;
  float conf_scale;
  int32_t xy_times_conf; /* [r06] 1: xy receives (x, y) * conf_prob -- `multiply([p, c])` in front of the action head's pose
                            convolutions (spnet.py:108) folded into the read-out that produces both factors (the same two
                            fp32 values, one multiplication each: bit-identical); conf_prob itself may be NULL.  The field sits
                            in what was tail padding of the struct */
} dh_sam_args;
int dh_softargmax2d_f32(const dh_sam_args* a, void* stream);

/* The 2-D decoder with context of reception.pose_regression_2d_context (reception.py:167-182) in ONE launch: soft-argmax
 * of the J joint maps (channels [0, J) of `a->h`) and of the J*nctx context maps (channels J + j*nctx + k), their
 * confidences (keypoint_confidence of the raw maps) and blocks.build_context_aggregation (blocks.py:217-285):
 *   y[f, j] = agg_alpha * xy_j + (1 - agg_alpha) * sum_k xy_jk * conf_jk / sum_k conf_jk      -> y [F, J, 2], pitch ldy
 * a->conf_raw (optional, pitch ldcr) receives the J joint confidences; a->C = J * (1 + nctx); the other outputs of
 * dh_sam_args must be NULL.  Needs J % 4 == 0, nctx <= 3, ldh % 4 == 0 and a 16-byte aligned `h` (DH_EUNSUPPORTED
 * otherwise: use dh_softargmax2d_f32 twice + dh_context_aggregation_f32). */
int dh_softargmax2d_context_f32(const dh_sam_args* a, int J, int nctx, float agg_alpha, float* y, int ldy,
                                void* stream);

/* blocks.build_context_aggregation (blocks.py:217-285); ys [F,J,2], yc [F,J*nctx,2], pc [F,J*nctx] */
int dh_context_aggregation_f32(const float* ys, const float* yc, const float* pc, float* y, int F, int J,
                               int nctx, float alpha, int ldy, void* stream);

/* reception.pose_regression_3d (reception.py:193-222): depth/spatial means of the D*J maps ... */
int dh_depth_means_f32(const float* h, int ldh, float* hxy, float* hz, int F, int HW, int D, int J,
                       void* stream);
/* ... and blocks.build_softargmax_1d (blocks.py:288-303, layers.py:132-157, activations.py:18-30);
 * vz = max_d hz (GlobalMaxPooling1D, reception.py:217) */
int dh_softargmax1d_f32(const float* hz, const float* grid, float* z, int ldz, float* vz, int F, int D, int J,
                        void* stream);

/* layers.kronecker_prod (layers.py:478-508): f[b,j,c] = sum_p hm[b,p,j] * x[b,p,c] */
int dh_kronecker_f32(const float* hm, int ldh, const float* x, int ldx, float* f, int ldf, int B, int P, int J,
                     int C, void* stream);

/* layers.global_max_min_pooling (+ Activation('softmax')) (layers.py:428-442, action.py:14-17) */
int dh_global_maxmin_softmax_f32(const float* x, int ldx, float* y, int B, int P, int C, int softmax,
                                 void* stream);

/* keras concatenate / Lambda channel slicing fallback, ZeroPadding2D (spnet.py:98-107) */
int dh_copy_channels_f32(const float* x, int ldx, float* y, int ldy, int64_t npix, int C, void* stream);
/* y is [B,OH,OW,C]; x [B,H,W,C] lands at row offset PT, column offset PL, zeros elsewhere */
int dh_zeropad2d_f32(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, int PT, int PL,
                     void* stream);

/* SPNet depth read-out (spnet.py:201-205): z[f,j] = sum_hw sigmoid(d[f,h,w,j]) * prob[f,h,w,j]
 * (Activation('sigmoid') -> multiply -> Lambda K.sum over (H,W)) */
int dh_depth_from_maps_f32(const float* d, int ldd, const float* h, int ldh, float* z, int ldz, int F, int HW,
                           int J, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Plan / execute pair: a whole model forward behind three calls, for hosts without Python (what a C or
 * C++ replacement of Keras' Model.predict -- exp/common/mpii_tools.py:25,86, h36m_tools.py:46, penn_tools.py:124 -- binds).
 *   blob   : written by deephar_amd.Model.export_plan(path, batch) (deephar_amd/engine/serialize.py): the bound,
 *            autotuned launch list of Model.predict for `batch` items, device pointers relative to the activation
 *            arena / the weight image, plus the weight image itself (packed kernels, BN affines, grids).
 *   create : allocates arena + weights on the current device, uploads the weights, patches the pointers.  One plan
 *            per device; a plan is not re-entrant (one forward at a time), different plans are independent.
 *   forward: m <= batch items; inputs[i] / outputs[i] are DEVICE pointers to dense float32 [m, ...] tensors in the
 *            model's input / output order (outputs[i] may be NULL: not wanted); everything is enqueued on `stream`,
 *            nothing synchronises.  Results are bit-identical to Model.predict of the exporting process.
 *   forward_host: the same with HOST pointers (H2D, forward, D2H on an internal stream; returns when done).
 * uint8-input plans (Model.export_plan(path, batch, uint8=True): raw frames, normalised on the GPU like
 * utils/transform.normalize_channels, inside the first convolution where possible): dh_plan_input_is_u8(plan, i) == 1 and
 * inputs[i] points to m * dh_plan_input_items(plan, i) BYTES (cast the uint8 pointer to const float*).
 * rc != 0 -> the host raises its own error.
 * ------------------------------------------------------------------------------------------------- */
typedef struct dh_plan dh_plan;
int dh_plan_create(const void* blob, size_t blob_bytes, dh_plan** plan_out);
int dh_plan_destroy(dh_plan* plan);
int dh_plan_batch(const dh_plan* plan);
int dh_plan_num_inputs(const dh_plan* plan);
int dh_plan_num_outputs(const dh_plan* plan);
int64_t dh_plan_input_items(const dh_plan* plan, int i);  /* elements (floats, or bytes of a uint8 input) per batch item */
int dh_plan_input_is_u8(const dh_plan* plan, int i);      /* 1: input i takes raw uint8 frames, 0: float32, -1: no such input */
int64_t dh_plan_output_items(const dh_plan* plan, int i); /* floats per batch item of output i */
int dh_forward(dh_plan* plan, const float* const* inputs_dev, int m, float* const* outputs_dev, void* stream);
int dh_forward_host(dh_plan* plan, const float* const* inputs_host, int m, float* const* outputs_host);

/* ---------------------------------------------------------------------------------------------------
 * Stream-ordered runtime helpers (no torch types): graphs for launch-bound replay, events for timing.
 * ------------------------------------------------------------------------------------------------- */
int dh_graph_begin_capture(void* stream);
int dh_graph_end_capture(void* stream, void** graph_exec_out);
int dh_graph_launch(void* graph_exec, void* stream);
int dh_graph_destroy(void* graph_exec);

int dh_event_create(void** event_out);
int dh_event_record(void* event, void* stream);
int dh_event_synchronize(void* event);
int dh_event_elapsed_ms(void* start, void* stop, float* ms_out);
int dh_event_destroy(void* event);
int dh_stream_synchronize(void* stream);
/* extra non-blocking streams + cross-stream ordering, so independent branches of a model can be captured as
 * parallel branches of one hipGraph (fork: side stream waits on an event of the origin stream; join: origin waits
 * on an event recorded at the tail of every side stream) */
int dh_stream_create(void** stream_out);
int dh_stream_destroy(void* stream);
int dh_event_create_sync(void** event_out); /* hipEventDisableTiming */
int dh_stream_wait_event(void* stream, void* event);
/* test aid for multi-stream schedules: one idle wave occupies `stream` for about `us` microseconds (<= 20 000), so a
 * test can delay one stream of a plan against the others and check that every cross-stream dependency is an event wait
 * (tests/test_gpu_speed2d.py: the 'tail' policy of the latency regime under random delays). */
int dh_stream_spin_us(void* stream, int us);

#ifdef __cplusplus
}
#endif
#endif /* DEEPHAR_HIP_H_ */
