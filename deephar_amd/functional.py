"""Op-level Python API over the C-ABI, on torch CUDA tensors (NHWC fp32, contiguous).

One function per kernel entry point of include/deephar_hip.h; used by the op parity tests and available to
callers that want a single fused op rather than a whole Model.  Launches go to torch's current stream.
No CPU path: every function raises if handed a non-CUDA tensor.
"""
import ctypes as C

import numpy as np

from . import _lib
from .engine import packing
from .engine.executor import grid_x, grid_depth
from .layers import same_pad


def _t():
    import torch
    return torch


def _stream():
    return _t().cuda.current_stream().cuda_stream


def _chk(*tensors):
    torch = _t()
    for t in tensors:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError('deephar_amd.functional expects contiguous float32 CUDA tensors')


def _p(t):
    return t.data_ptr() if t is not None else None


def pack_conv_weight(w_hwio, device='cuda'):
    """HWIO numpy kernel -> (packed device tensor, Kp, Np)."""
    torch = _t()
    packed, kp, np_ = packing.pack_conv(np.asarray(w_hwio, np.float32))
    return torch.from_numpy(packed).to(device), kp, np_


def conv2d(x, w_hwio, strides=(1, 1), padding='same', pre_scale=None, pre_shift=None, pre_relu=False,
           post_scale=None, post_shift=None, post_relu=False, res1=None, res2=None, up2=False, tile_cfg=-1,
           packed=None, in_lut=None, split=False, halo=False, res2_down=False, pool2=False, x_resample=0, seg=None):
    """Fused conv (see dh_conv2d_f32).  x [N,H,W,Cin]; w_hwio numpy [kh,kw,Cin,Cout].  A uint8 `x` needs
    `in_lut` (float32 [Cin,256] device tensor, engine.executor.normalization_lut): bytes are normalised on load."""
    torch = _t()
    if x.dtype == torch.uint8:
        if in_lut is None or tuple(in_lut.shape) != (x.shape[-1], 256):
            raise ValueError('uint8 input needs in_lut of shape [Cin, 256]')
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError('conv2d expects contiguous CUDA tensors')
        _chk(in_lut, pre_scale, pre_shift, post_scale, post_shift, res1, res2)
    else:
        _chk(x, pre_scale, pre_shift, post_scale, post_shift, res1, res2)
    lib = _lib.load()
    kh, kw, cin, cout = w_hwio.shape
    n, h, w_, c = x.shape
    if seg is not None:        # dh_conv2d_seg_f32: seg = (x2 or None, pool_sh): the input is [maxpool(x, strides (pool_sh, 2)) | x2]
        x2, pool_sh = seg
        _chk(x2)
        assert c + (x2.shape[-1] if x2 is not None else 0) == cin and h % pool_sh == 0 and w_ % 2 == 0
        h, w_ = h // pool_sh, w_ // 2
    else:
        assert c == cin
    if x_resample:             # dh_conv_args.x_resample: x is stored at half (1) / double (2, 3) the resolution the conv sees
        h, w_ = (2 * h, 2 * w_) if x_resample == 1 else (h // 2, w_ // 2)
    if padding == 'same':
        pt, _, oh = same_pad(h, kh, strides[0])
        pl, _, ow = same_pad(w_, kw, strides[1])
    else:
        pt = pl = 0
        oh, ow = (h - kh) // strides[0] + 1, (w_ - kw) // strides[1] + 1
    if split and packed is None:
        pk, kp, np_ = packing.pack_conv_split(np.asarray(w_hwio, np.float32))
        packed = (torch.from_numpy(pk).to(x.device), kp, np_)
    if halo and packed is None:                  # chunk-major fp32 packing for the halo-resident K x K kernel
        pk, kp, np_ = packing.pack_conv_halo(np.asarray(w_hwio, np.float32))
        packed = (torch.from_numpy(pk).to(x.device), kp, np_)
    wt, kp, np_ = packed if packed is not None else pack_conv_weight(w_hwio, x.device)
    up = 2 if up2 else 1
    y = torch.empty((n, oh * up, ow * up, cout), dtype=torch.float32, device=x.device)
    a = _lib.ConvArgs()
    a.w_split = 2 if halo else int(split)
    a.x, a.w, a.y = _p(x), _p(wt), _p(y)
    a.pre_scale, a.pre_shift, a.post_scale, a.post_shift = _p(pre_scale), _p(pre_shift), _p(post_scale), _p(post_shift)
    a.res1, a.res2 = _p(res1), _p(res2)
    a.N, a.H, a.W, a.Cin, a.ldx = n, h, w_, cin, c
    a.OH, a.OW, a.Cout, a.ldy = oh, ow, cout, cout
    a.KH, a.KW, a.SH, a.SW, a.PT, a.PL = kh, kw, strides[0], strides[1], pt, pl
    a.K, a.Kp, a.Np = kh * kw * cin, kp, np_
    a.ldr1 = res1.shape[-1] if res1 is not None else 0
    a.ldr2 = res2.shape[-1] if res2 is not None else 0
    a.pre_relu, a.post_relu, a.up2 = int(pre_relu), int(post_relu), int(up2)
    a.res2_down = int(res2_down)
    a.x_resample = int(x_resample)
    if x.dtype == torch.uint8:
        a.in_lut, a.x_u8 = _p(in_lut), 1
    yp = None
    if pool2:                                    # second output: MaxPooling2D((2, 2)) of y (dh_conv_args.y_pool)
        yp = torch.empty((n, oh // 2, ow // 2, cout), dtype=torch.float32, device=x.device)
        a.y_pool, a.ldyp = _p(yp), cout
    if seg is not None:
        sg = _lib.ConvSeg()
        if x2 is not None:
            sg.x2, sg.ldx2 = _p(x2), x2.shape[-1]
        sg.c_split, sg.pool_sh = c, pool_sh
        _lib.check(lib.dh_conv2d_seg_f32(C.byref(a), C.byref(sg), _stream()), 'dh_conv2d_seg_f32')
        return y
    _lib.check(lib.dh_conv2d_f32(C.byref(a), tile_cfg, _stream()), 'dh_conv2d_f32')
    return (y, yp) if pool2 else y


def normalize_u8(x, lut):
    """uint8 [.., C] -> float32 through lut [C,256] (dh_normalize_u8_f32)."""
    torch = _t()
    if x.dtype != torch.uint8 or not x.is_cuda or not x.is_contiguous():
        raise ValueError('normalize_u8 expects a contiguous CUDA uint8 tensor')
    _chk(lut)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dh_normalize_u8_f32(_p(x), _p(lut), _p(y), x.numel() // x.shape[-1], x.shape[-1],
                                               _stream()), 'dh_normalize_u8_f32')
    return y


def dwconv2d(x, dw_kernel, pre_scale=None, pre_shift=None, pre_relu=False, up_in=False):
    """Depthwise conv, stride 1, TF-SAME.  dw_kernel numpy [kh,kw,C,1].  up_in: x is at half resolution and the
    convolution runs on UpSampling2D((2, 2))(x) without writing it out (dh_dw_args.up_in)."""
    torch = _t()
    _chk(x, pre_scale, pre_shift)
    lib = _lib.load()
    kh, kw, c, _ = dw_kernel.shape
    n, h, w_, cx = x.shape
    assert cx == c
    if up_in:
        h, w_ = 2 * h, 2 * w_
    pt, _, _ = same_pad(h, kh, 1)
    pl, _, _ = same_pad(w_, kw, 1)
    wt = torch.from_numpy(np.ascontiguousarray(dw_kernel.reshape(kh * kw, c), np.float32)).to(x.device)
    y = torch.empty((n, h, w_, c), dtype=torch.float32, device=x.device)
    a = _lib.DwArgs()
    a.x, a.w, a.y, a.pre_scale, a.pre_shift = _p(x), _p(wt), _p(y), _p(pre_scale), _p(pre_shift)
    a.N, a.H, a.W, a.C, a.ldx, a.ldy = n, h, w_, c, c, c
    a.KH, a.KW, a.PT, a.PL, a.pre_relu, a.up_in = kh, kw, pt, pl, int(pre_relu), int(up_in)
    _lib.check(lib.dh_dwconv2d_f32(C.byref(a), _stream()), 'dh_dwconv2d_f32')
    return y


def pool2d(x, pool=(2, 2), strides=None, padding='valid', mode=0):
    torch = _t()
    _chk(x)
    lib = _lib.load()
    strides = strides or pool
    n, h, w_, c = x.shape
    if padding == 'same':
        pt, _, oh = same_pad(h, pool[0], strides[0])
        pl, _, ow = same_pad(w_, pool[1], strides[1])
    else:
        pt = pl = 0
        oh, ow = (h - pool[0]) // strides[0] + 1, (w_ - pool[1]) // strides[1] + 1
    y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    a = _lib.PoolArgs()
    a.x, a.y = _p(x), _p(y)
    a.N, a.H, a.W, a.C, a.ldx, a.OH, a.OW, a.ldy = n, h, w_, c, c, oh, ow, c
    a.KH, a.KW, a.SH, a.SW, a.PT, a.PL, a.mode = pool[0], pool[1], strides[0], strides[1], pt, pl, mode
    _lib.check(lib.dh_pool2d_f32(C.byref(a), _stream()), 'dh_pool2d_f32')
    return y


def upsample2x_add(b, a=None):
    torch = _t()
    _chk(a, b)
    n, h, w_, c = b.shape
    y = torch.empty((n, 2 * h, 2 * w_, c), dtype=torch.float32, device=b.device)
    _lib.check(_lib.load().dh_upsample2x_add_f32(_p(a), c, _p(b), c, _p(y), c, n, 2 * h, 2 * w_, c, _stream()),
               'dh_upsample2x_add_f32')
    return y


def softargmax2d(h, alpha=1.0, conf_scale=1.0, want_prob=False):
    """Returns dict(xy [F,C,2], conf_raw [F,C,1], conf_prob [F,C,1], gmax [F,C], prob [F,H,W,C] | None)."""
    torch = _t()
    _chk(h)
    f, hh, ww, c = h.shape
    dev = h.device
    out = dict(xy=torch.empty((f, c, 2), device=dev), conf_raw=torch.empty((f, c, 1), device=dev),
               conf_prob=torch.empty((f, c, 1), device=dev), gmax=torch.empty((f, c), device=dev),
               prob=torch.empty_like(h) if want_prob else None)
    gx = torch.from_numpy(grid_x(ww)).to(dev)
    gy = torch.from_numpy(grid_x(hh)).to(dev)
    a = _lib.SamArgs()
    a.h, a.gx, a.gy = _p(h), _p(gx), _p(gy)
    a.xy, a.conf_raw, a.conf_prob, a.prob, a.gmax = (_p(out['xy']), _p(out['conf_raw']), _p(out['conf_prob']),
                                                     _p(out['prob']), _p(out['gmax']))
    a.F, a.H, a.W, a.C, a.ldh, a.ldxy, a.ldcr, a.ldcp, a.ldp = f, hh, ww, c, c, 2, 1, 1, c
    a.alpha, a.conf_scale = float(alpha), float(conf_scale)
    _lib.check(_lib.load().dh_softargmax2d_f32(C.byref(a), _stream()), 'dh_softargmax2d_f32')
    torch.cuda.current_stream().synchronize()  # gx/gy are temporaries
    return out


def softargmax2d_context(h, joints, num_context, agg_alpha, alpha=1.0, conf_scale=1.0, pitch=None):
    """dh_softargmax2d_context_f32: h [F, H, W, >= J*(1+nctx)] (channel pitch = h.shape[-1], the first J*(1+nctx)
    channels are read) -> (pose [F, J, 2], joint confidences [F, J, 1])."""
    torch = _t()
    _chk(h)
    f, hh, ww, ld = h.shape
    dev = h.device
    y = torch.empty((f, joints, 2), device=dev)
    conf = torch.empty((f, joints, 1), device=dev)
    gx = torch.from_numpy(grid_x(ww)).to(dev)
    gy = torch.from_numpy(grid_x(hh)).to(dev)
    a = _lib.SamArgs()
    a.h, a.gx, a.gy, a.conf_raw = _p(h), _p(gx), _p(gy), _p(conf)
    a.F, a.H, a.W, a.C, a.ldh, a.ldcr = f, hh, ww, joints * (1 + num_context), ld, 1
    a.alpha, a.conf_scale = float(alpha), float(conf_scale)
    _lib.check(_lib.load().dh_softargmax2d_context_f32(C.byref(a), joints, num_context, float(agg_alpha), _p(y), 2,
                                                       _stream()), 'dh_softargmax2d_context_f32')
    torch.cuda.current_stream().synchronize()  # gx/gy are temporaries
    return y, conf


def context_aggregation(ys, yc, pc, num_context, alpha):
    torch = _t()
    _chk(ys, yc, pc)
    f, j, _ = ys.shape
    y = torch.empty_like(ys)
    _lib.check(_lib.load().dh_context_aggregation_f32(_p(ys), _p(yc), _p(pc), _p(y), f, j, num_context,
                                                      float(alpha), 2, _stream()), 'dh_context_aggregation_f32')
    return y


def depth_means(h, depth, joints):
    torch = _t()
    _chk(h)
    f, hh, ww, c = h.shape
    assert c == depth * joints
    hxy = torch.empty((f, hh, ww, joints), device=h.device)
    hz = torch.empty((f, depth, joints), device=h.device)
    _lib.check(_lib.load().dh_depth_means_f32(_p(h), c, _p(hxy), _p(hz), f, hh * ww, depth, joints, _stream()),
               'dh_depth_means_f32')
    return hxy, hz


def softargmax1d(hz):
    torch = _t()
    _chk(hz)
    f, d, j = hz.shape
    z = torch.empty((f, j, 1), device=hz.device)
    vz = torch.empty((f, j), device=hz.device)
    grid = torch.from_numpy(grid_depth(d)).to(hz.device)
    _lib.check(_lib.load().dh_softargmax1d_f32(_p(hz), _p(grid), _p(z), 1, _p(vz), f, d, j, _stream()),
               'dh_softargmax1d_f32')
    torch.cuda.current_stream().synchronize()
    return z, vz


def kronecker(hm, x, out_pitch=None):
    """layers.kronecker_prod.  `out_pitch` > C writes the rows into a wider (packed) buffer, as a plan does."""
    torch = _t()
    _chk(hm, x)
    b, hh, ww, j = hm.shape
    c = x.shape[-1]
    ld = c if out_pitch is None else int(out_pitch)
    f = torch.zeros((b, j, ld), device=hm.device)
    _lib.check(_lib.load().dh_kronecker_f32(_p(hm), j, _p(x), c, _p(f), ld, b, hh * ww, j, c, _stream()),
               'dh_kronecker_f32')
    return f if ld == c else f[..., :c]


def global_maxmin_softmax(x, softmax=True):
    torch = _t()
    _chk(x)
    b, t, j, c = x.shape
    y = torch.empty((b, c), device=x.device)
    _lib.check(_lib.load().dh_global_maxmin_softmax_f32(_p(x), c, _p(y), b, t * j, c, int(softmax), _stream()),
               'dh_global_maxmin_softmax_f32')
    return y
