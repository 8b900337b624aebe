"""Op-level restatement of the reference's layer vocabulary (deephar/layers.py, deephar/activations.py)
on PyTorch-CPU.  TEST INFRASTRUCTURE (see oracle/__init__.py: graph wiring pinned by reference-code goldens, Keras/TF layer numerics restated).

Tensors are NHWC torch tensors of the dtype chosen by the caller (float32 = the reference's dtype,
float64 = accuracy arbiter).  Keras/TF defaults restated from SURVEY.md A.3:
  - padding='same' is TF-SAME (asymmetric, extra pad at bottom/right)
  - BatchNormalization inference: y = x*inv + (beta - mean*inv), inv = rsqrt(var + 1e-3) * gamma
  - K.epsilon() = 1e-7
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3       # keras.layers.BatchNormalization default epsilon
K_EPSILON = 1e-7    # keras.backend.epsilon()


def same_pad(size, k, s):
    """TF 'SAME' padding for one dimension -> (before, after, out)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    before = total // 2
    return before, total - before, out


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def conv2d(x, kernel, strides=(1, 1), padding='same'):
    """keras Conv2D(use_bias=False) -- layers.py:66-71.  kernel is Keras HWIO [kh,kw,cin,cout]."""
    kh, kw, cin, cout = kernel.shape
    assert x.shape[-1] == cin, (x.shape, kernel.shape)
    xc = _nchw(x)
    if padding == 'same':
        pt, pb, _ = same_pad(x.shape[1], kh, strides[0])
        pl, pr, _ = same_pad(x.shape[2], kw, strides[1])
        xc = F.pad(xc, (pl, pr, pt, pb))
    w = kernel.permute(3, 2, 0, 1).contiguous()
    return _nhwc(F.conv2d(xc, w, stride=strides))


def depthwise_conv2d(x, dw_kernel, strides=(1, 1), padding='same'):
    """Depthwise half of keras SeparableConv2D; dw_kernel [kh,kw,c,1]."""
    kh, kw, c, mult = dw_kernel.shape
    assert mult == 1 and x.shape[-1] == c
    xc = _nchw(x)
    if padding == 'same':
        pt, pb, _ = same_pad(x.shape[1], kh, strides[0])
        pl, pr, _ = same_pad(x.shape[2], kw, strides[1])
        xc = F.pad(xc, (pl, pr, pt, pb))
    w = dw_kernel.permute(2, 3, 0, 1).contiguous()  # [c,1,kh,kw]
    return _nhwc(F.conv2d(xc, w, stride=strides, groups=c))


def sepconv2d(x, dw_kernel, pw_kernel, strides=(1, 1), padding='same'):
    """keras SeparableConv2D(use_bias=False) -- layers.py:74-80: depthwise then 1x1, nothing in between."""
    return conv2d(depthwise_conv2d(x, dw_kernel, strides, padding), pw_kernel, (1, 1), 'valid')


def batchnorm(x, beta, mean, var, gamma=None):
    """keras BatchNormalization(axis=-1) inference (scale=False when gamma is None) -- layers.py:209."""
    inv = torch.rsqrt(var + BN_EPS)
    if gamma is not None:
        inv = inv * gamma
    return x * inv + (beta - mean * inv)


def relu(x):
    return torch.clamp_min(x, 0)


def maxpool2d(x, pool=(2, 2), strides=None, padding='valid'):
    """keras MaxPooling2D; TF ignores padded cells (== -inf padding)."""
    strides = strides or pool
    xc = _nchw(x)
    if padding == 'same':
        pt, pb, _ = same_pad(x.shape[1], pool[0], strides[0])
        pl, pr, _ = same_pad(x.shape[2], pool[1], strides[1])
        xc = F.pad(xc, (pl, pr, pt, pb), value=-math.inf)
    return _nhwc(F.max_pool2d(xc, pool, strides))


def upsample2d(x, size=(2, 2)):
    """keras UpSampling2D = nearest-neighbour repeat."""
    return x.repeat_interleave(size[0], dim=1).repeat_interleave(size[1], dim=2)


def channel_softmax_2d(x, alpha=1.0):
    """activations.py:3-16 (4-D case)."""
    if alpha != 1:
        x = alpha * x
    e = torch.exp(x - torch.amax(x, dim=(1, 2), keepdim=True))
    s = torch.clamp_min(torch.sum(e, dim=(1, 2), keepdim=True), K_EPSILON)
    return e / s


def linspace_2d(rows, cols, dim):
    """utils/math.py:6-19: float32 grid, np.linspace(0,1) along x (dim=0) or y (dim=1)."""
    if dim == 1:
        lin = np.linspace(0.0, 1.0, num=rows)
        g = np.empty((cols, rows), dtype=np.float32)
        g[:] = lin
        return g.T.copy()
    lin = np.linspace(0.0, 1.0, num=cols)
    g = np.empty((rows, cols), dtype=np.float32)
    g[:] = lin
    return g


def lin_interpolation_2d(p, axis):
    """layers.py:160-200: frozen depthwise conv whose kernel is the whole map (vmin/vmax ignored)."""
    rows, cols = p.shape[1], p.shape[2]
    g = torch.from_numpy(linspace_2d(rows, cols, axis)).to(p.dtype)
    return torch.sum(p * g[None, :, :, None], dim=(1, 2)).unsqueeze(-1)  # [N, C, 1]


def softargmax2d_from_prob(p):
    """layers.py:122-129 / blocks.py:318-320: concat [x, y]."""
    return torch.cat([lin_interpolation_2d(p, 0), lin_interpolation_2d(p, 1)], dim=-1)


def softargmax2d(h, alpha=1.0):
    """blocks.build_softargmax_2d (blocks.py:306-325)."""
    return softargmax2d_from_prob(channel_softmax_2d(h, alpha))


def joints_probability(x):
    """blocks.build_joints_probability (blocks.py:328-343) == layers.keypoint_confidence (layers.py:107-119):
    4 * AveragePooling2D((2,2), strides 1, valid) -> GlobalMaxPooling2D -> expand_dims."""
    a = F.avg_pool2d(_nchw(x), (2, 2), stride=(1, 1))
    a = 4 * a
    return torch.amax(a, dim=(2, 3)).unsqueeze(-1)  # [N, C, 1]


def context_aggregation(ys, yc, pc, num_joints, num_context, alpha):
    """blocks.build_context_aggregation (blocks.py:217-285), num_frames == 1 branch."""
    n = ys.shape[0]
    xi = yc[:, :, 0:1]
    yi = yc[:, :, 1:2]
    pxi = xi * pc
    pyi = yi * pc

    def ctx_sum(t):  # frozen Dense with a 0/1 block matrix (blocks.py:221-233)
        return t.reshape(n, num_joints, num_context, 1).sum(dim=2)

    pc_sum = ctx_sum(pc)
    pxi_div = ctx_sum(pxi) / pc_sum
    pyi_div = ctx_sum(pyi) / pc_sum
    yc_div = torch.cat([pxi_div, pyi_div], dim=-1)
    return alpha * ys + (1 - alpha) * yc_div


def channel_softmax_1d(x):
    """activations.py:18-30: soft-max over axis 1 of [N, D, J]."""
    e = torch.exp(x - torch.amax(x, dim=1, keepdim=True))
    return e / torch.sum(e, dim=1, keepdim=True)


def lin_interpolation_1d(p):
    """layers.py:132-157: frozen Conv1D with kernel = linspace(1/(2D), 1-1/(2D), D) per channel."""
    depth = p.shape[1]
    start = 1 / (2 * depth)
    lin = np.linspace(start, 1 - start, num=depth).astype(np.float32)  # Conv1D weights are float32
    g = torch.from_numpy(lin).to(p.dtype)
    return torch.sum(p * g[None, :, None], dim=1).unsqueeze(-1)  # [N, J, 1]


def softargmax1d(hz):
    """blocks.build_softargmax_1d (blocks.py:288-303)."""
    return lin_interpolation_1d(channel_softmax_1d(hz))


def kronecker_prod(hm, x):
    """layers.kronecker_prod (layers.py:478-508): f[..., j, c] = sum_hw hm[..., h, w, j] * x[..., h, w, c]."""
    return torch.einsum('...hwj,...hwc->...jc', hm, x)


def max_min_pooling(x, strides=(2, 2), padding='same'):
    """layers.max_min_pooling (layers.py:411-425): MaxPooling2D(strides, padding)(x) - MaxPooling2D(..)(-x).
    NB: the reference passes `strides` as the pool size (first positional arg)."""
    return maxpool2d(x, strides, None, padding) - maxpool2d(-x, strides, None, padding)


def global_max_min_pooling(x):
    """layers.global_max_min_pooling (layers.py:428-442)."""
    return torch.amax(x, dim=(1, 2)) - torch.amax(-x, dim=(1, 2))
