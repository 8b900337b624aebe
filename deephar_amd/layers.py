"""Layer/op vocabulary of the reference (deephar/layers.py), emitting graph-IR nodes instead of Keras layers.

Same function names, argument meaning and composition order as the reference helpers, so the model
builders in deephar_amd/models read like deephar/models/*.py.  Nothing here computes: nodes are fused and
lowered to gfx950 kernels by deephar_amd/engine.  Helpers the reference defines but no experiment calls
(localconv1d, deconv, bn_act_conv, dense, ... -- SURVEY.md section 2) are intentionally absent.
"""
import numpy as np

from . import graph as G
from .graph import Input  # noqa: F401  (re-exported like keras.layers.Input)


def same_pad(size, k, s):
    """TF 'SAME' padding for one dim -> (before, after, out_size)  (SURVEY.md A.3)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2, out


def _pair(v):
    return (int(v), int(v)) if isinstance(v, (int, np.integer)) else (int(v[0]), int(v[1]))


def _conv_out(shape, size, strides, padding):
    h, w = shape[-3], shape[-2]
    if padding == 'same':
        pt, _, oh = same_pad(h, size[0], strides[0])
        pl, _, ow = same_pad(w, size[1], strides[1])
    elif padding == 'valid':
        pt = pl = 0
        oh = (h - size[0]) // strides[0] + 1
        ow = (w - size[1]) // strides[1] + 1
    else:
        raise ValueError('padding must be "same" or "valid"')
    return oh, ow, pt, pl


# ---- primitive layers ------------------------------------------------------------------------------------

def relu(x, leakyrelu=False, name=None):
    """layers.relu (layers.py:51-55); LeakyReLU is never requested by the experiment scripts."""
    if leakyrelu:
        raise NotImplementedError('LeakyReLU is unused on the hot path')
    return G.emit('relu', [x], [x.shape], name=name)[0]


def scale(x, k, name=None):
    """Lambda(lambda x: k * x) (action.py:291)."""
    return G.emit('scale', [x], [x.shape], dict(k=float(k)), name=name)[0]


def sigmoid(x, name=None):
    return G.emit('sigmoid', [x], [x.shape], name=name)[0]


def conv2d(x, filters, kernel_size, strides=(1, 1), padding='same', name=None):
    """layers.conv2d (layers.py:66-71): Conv2D(use_bias=False); TimeDistributed is implicit (leading dims)."""
    size, strides = _pair(kernel_size), _pair(strides)
    cin = x.shape[-1]
    oh, ow, pt, pl = _conv_out(x.shape, size, strides, padding)
    layer = G.make_layer('Conv2D', name, [('kernel', (size[0], size[1], cin, filters), 'conv')])
    layer.params[0].fan_in = size[0] * size[1] * cin
    attrs = dict(kh=size[0], kw=size[1], sh=strides[0], sw=strides[1], pt=pt, pl=pl, filters=filters)
    return G.emit('conv', [x], [x.shape[:-3] + (oh, ow, filters)], attrs, dict(conv=layer), name=layer.name)[0]


conv = conv2d


def sepconv2d(x, filters, kernel_size, strides=(1, 1), padding='same', name=None):
    """layers.sepconv2d (layers.py:74-80): SeparableConv2D(use_bias=False), depth_multiplier 1."""
    size, strides = _pair(kernel_size), _pair(strides)
    if strides != (1, 1):
        raise NotImplementedError('strided SeparableConv2D is not used by any experiment script')
    cin = x.shape[-1]
    oh, ow, pt, pl = _conv_out(x.shape, size, strides, padding)
    layer = G.make_layer('SeparableConv2D', name, [
        ('depthwise_kernel', (size[0], size[1], cin, 1), 'depthwise'),
        ('pointwise_kernel', (1, 1, cin, filters), 'conv')])
    layer.params[0].fan_in = size[0] * size[1]
    layer.params[1].fan_in = cin
    attrs = dict(kh=size[0], kw=size[1], pt=pt, pl=pl, filters=filters)
    return G.emit('sepconv', [x], [x.shape[:-3] + (oh, ow, filters)], attrs, dict(sepconv=layer),
                  name=layer.name)[0]


def BatchNormalization(x, axis=-1, scale=True, name=None):
    """keras BatchNormalization (inference).  Weight order follows Keras: [gamma,] beta, mean, variance."""
    assert axis == -1
    c = x.shape[-1]
    specs = ([('gamma', (c,), 'gamma')] if scale else []) + [
        ('beta', (c,), 'beta'), ('moving_mean', (c,), 'mean'), ('moving_variance', (c,), 'var')]
    layer = G.make_layer('BatchNormalization', name, specs)
    return G.emit('bn', [x], [x.shape], {}, dict(bn=layer), name=layer.name)[0]


def add(tensors, name=None):
    shape = tensors[0].shape
    for t in tensors:
        if t.shape != shape:
            raise ValueError('add: shape mismatch %s vs %s' % (t.shape, shape))
    return G.emit('add', list(tensors), [shape], name=name)[0]


def multiply(tensors, name=None):
    a, b = tensors
    if a.shape != b.shape and not (b.shape[:-1] == a.shape[:-1] and b.shape[-1] == 1):
        raise ValueError('multiply: shapes %s and %s' % (a.shape, b.shape))
    return G.emit('mul', [a, b], [a.shape], name=name)[0]


def concatenate(tensors, axis=-1, name=None):
    assert axis == -1, 'only channel concatenation is used on the hot path'
    if len(tensors) == 1:
        return tensors[0]
    lead = tensors[0].shape[:-1]
    for t in tensors:
        if t.shape[:-1] != lead:
            raise ValueError('concatenate: shape mismatch %s vs %s' % (t.shape, tensors[0].shape))
    return G.emit('concat', list(tensors), [lead + (sum(t.shape[-1] for t in tensors),)], name=name)[0]


def MaxPooling2D(x, pool_size=(2, 2), strides=None, padding='valid', name=None):
    """keras MaxPooling2D (default stride = pool, padding 'valid': reception.py:86,108,115)."""
    pool = _pair(pool_size)
    strides = pool if strides is None else _pair(strides)
    oh, ow, pt, pl = _conv_out(x.shape, pool, strides, padding)
    attrs = dict(kh=pool[0], kw=pool[1], sh=strides[0], sw=strides[1], pt=pt, pl=pl, mode=0)
    return G.emit('pool', [x], [x.shape[:-3] + (oh, ow, x.shape[-1])], attrs, name=name)[0]


def maxpooling2d(x, kernel_size=(2, 2), strides=(2, 2), padding='same', name=None):
    """layers.maxpooling2d (layers.py:92-97)."""
    return MaxPooling2D(x, kernel_size, strides, padding, name)


def UpSampling2D(x, size=(2, 2), name=None):
    if _pair(size) != (2, 2):
        raise NotImplementedError('only x2 up-sampling is used')
    return G.emit('upsample', [x], [x.shape[:-3] + (2 * x.shape[-3], 2 * x.shape[-2], x.shape[-1])],
                  name=name)[0]


def upsampling2d(x, kernel_size=(2, 2), name=None):
    """layers.upsampling2d (layers.py:100-104)."""
    return UpSampling2D(x, kernel_size, name)


def ZeroPadding2D(x, padding, name=None):
    """keras ZeroPadding2D(((top, bottom), (left, right))) (spnet.py:124-133)."""
    (pt, pb), (pl, pr) = padding
    shape = x.shape[:-3] + (x.shape[-3] + pt + pb, x.shape[-2] + pl + pr, x.shape[-1])
    return G.emit('zeropad', [x], [shape], dict(pt=int(pt), pl=int(pl)), name=name)[0]


def depth_from_maps(d, h, name=None):
    """spnet.py:201-205: Activation('sigmoid')(d) -> multiply([d, h]) -> K.sum over (H, W) -> expand_dims:
    [.., H, W, J] x [.., H, W, J] -> [.., J, 1]."""
    if d.shape != h.shape:
        raise ValueError('depth_from_maps: %s vs %s' % (d.shape, h.shape))
    return G.emit('depthsum', [d, h], [d.shape[:-3] + (d.shape[-1], 1)], name=name)[0]


# ---- soft-argmax / confidence ------------------------------------------------------------------------

def act_channel_softmax(x, alpha=1.0, name=None):
    """layers.act_channel_softmax (layers.py:363-365) with activations.channel_softmax_2d(alpha)."""
    return G.emit('softmax2d', [x], [x.shape], dict(alpha=float(alpha)), name=name)[0]


def softargmax2d(x, limits=(0, 0, 1, 1), name=None):
    """layers.softargmax2d (layers.py:122-129) on probability maps: [.., H, W, C] -> [.., C, 2].
    `limits` is accepted and ignored exactly like the reference (vmin/vmax unused, layers.py:160-187)."""
    return G.emit('expect2d', [x], [x.shape[:-3] + (x.shape[-1], 2)], name=name)[0]


def keypoint_confidence(x, scale=1.0, name=None):
    """layers.keypoint_confidence (layers.py:107-119) == blocks.build_joints_probability: max over 2x2 window
    sums; [.., H, W, C] -> [.., C, 1].  `scale` folds a preceding Lambda(k*x) (action.py:200)."""
    return G.emit('jointprob', [x], [x.shape[:-3] + (x.shape[-1], 1)], dict(scale=float(scale)), name=name)[0]


def act_depth_softmax_interp(hz, name=None):
    """blocks.build_softargmax_1d (blocks.py:288-303): depth soft-max + lin_interpolation_1d;
    [.., D, J] -> [.., J, 1]."""
    return G.emit('softargmax1d', [hz], [hz.shape[:-2] + (hz.shape[-1], 1)], name=name)[0]


def kronecker_prod(h, f, name='Kronecker_prod'):
    """layers.kronecker_prod (layers.py:478-508): [..,H,W,J] x [..,H,W,C] -> [..,J,C]."""
    if h.shape[:-1] != f.shape[:-1]:
        raise ValueError('kronecker_prod: %s vs %s' % (h.shape, f.shape))
    return G.emit('kronecker', [h, f], [h.shape[:-3] + (h.shape[-1], f.shape[-1])], name=name)[0]


def max_min_pooling(x, strides=(2, 2), padding='same', name=None):
    """layers.max_min_pooling (layers.py:411-425).  The reference passes `strides` as the pool size."""
    pool = _pair(strides)
    oh, ow, pt, pl = _conv_out(x.shape, pool, pool, padding)
    attrs = dict(kh=pool[0], kw=pool[1], sh=pool[0], sw=pool[1], pt=pt, pl=pl, mode=1)
    return G.emit('pool', [x], [x.shape[:-3] + (oh, ow, x.shape[-1])], attrs, name=name)[0]


def global_max_min_pooling(x, name=None):
    """layers.global_max_min_pooling (layers.py:428-442): [.., T, J, C] -> [.., C]."""
    return G.emit('globalmaxmin', [x], [x.shape[:-3] + (x.shape[-1],)], name=name)[0]


def softmax(x, name=None):
    """Activation('softmax') over the last axis (action.py:16, spnet.py:67)."""
    return G.emit('softmax', [x], [x.shape], name=name)[0]


def reshape(x, shape, name=None):
    """Lambda expand_dims/squeeze/reshape: a zero-copy view (row-major contiguous)."""
    if int(np.prod(shape)) != int(np.prod(x.shape)):
        raise ValueError('reshape %s -> %s' % (x.shape, shape))
    return G.emit('reshape', [x], [tuple(shape)], name=name)[0]


# ---- compositions (same order of operations as the reference) ---------------------------------------

def conv_bn(x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.conv_bn (layers.py:202-210)"""
    x = conv(x, filters, size, strides, padding, name + '_conv' if name is not None else None)
    return BatchNormalization(x, axis=-1, scale=False, name=name)


def conv_act(x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.conv_act (layers.py:219-227)"""
    x = conv(x, filters, size, strides, padding, name + '_conv' if name is not None else None)
    return relu(x, name=name)


def conv_bn_act(x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.conv_bn_act (layers.py:230-241)"""
    x = conv(x, filters, size, strides, padding, name + '_conv' if name is not None else None)
    x = BatchNormalization(x, axis=-1, scale=False, name=name + '_bn' if name is not None else None)
    return relu(x, name=name)


def act_conv_bn(x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.act_conv_bn (layers.py:258-269)"""
    x = relu(x, name=name + '_act' if name is not None else None)
    x = conv(x, filters, size, strides, padding, name + '_conv' if name is not None else None)
    return BatchNormalization(x, axis=-1, scale=False, name=name)


def separable_act_conv_bn(x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.separable_act_conv_bn (layers.py:288-301)"""
    x = relu(x, name=name + '_act' if name is not None else None)
    x = sepconv2d(x, filters, size, strides, padding, name + '_conv' if name is not None else None)
    return BatchNormalization(x, axis=-1, scale=False, name=name)


def act_conv(x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.act_conv (layers.py:317-325)"""
    x = relu(x, name=name + '_act' if name is not None else None)
    return conv(x, filters, size, strides, padding, name)
