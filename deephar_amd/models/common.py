"""Residual / down- / up-scaling units of SPNet (reference deephar/models/common.py:9-108).
Training helpers (set_trainable_layers, copy_replica_layers, compile_model, :111-160) are out of scope."""
from .. import layers as L
from ..utils import appstr


def concat_tensorlist(t):
    assert isinstance(t, list), 't should be a list, got ({})'.format(t)
    return L.concatenate(t) if len(t) > 1 else t[0]


def add_tensorlist(t):
    assert isinstance(t, list), 't should be a list, got ({})'.format(t)
    return L.add(t) if len(t) > 1 else t[0]


def residual_unit(x, kernel_size, strides=(1, 1), out_size=None, convtype='depthwise', shortcut_act=True,
                  features_div=2, name=None):
    """Pre-activation residual unit with the default (gamma+beta) BatchNormalization (common.py:25-67).
    When the width or stride changes the shortcut is a 1x1 conv of relu(BN(x)); otherwise it is x itself and
    only the main path sees BN.  'depthwise' main path: BN-ReLU-SeparableConv; 'normal': BN-ReLU-1x1
    (out/features_div) -BN-ReLU-kxk."""
    assert convtype in ['depthwise', 'normal'], 'Invalid convtype ({}).'.format(convtype)
    cin = x.shape[-1]
    out_size = cin if out_size is None else out_size
    strides = tuple(strides)
    project = (cin != out_size) or (strides != (1, 1))

    normed = L.BatchNormalization(x, name=appstr(name, '_bn1'))
    if project:
        s = L.relu(normed, name=appstr(name, '_shortcut_act')) if shortcut_act else normed
        shortcut = L.conv2d(s, out_size, (1, 1), strides=strides, name=appstr(name, '_shortcut_conv'))
    else:
        shortcut = x
    y = L.relu(normed, name=appstr(name, '_act1'))
    if convtype == 'depthwise':
        y = L.sepconv2d(y, out_size, kernel_size, strides=strides, name=appstr(name, '_conv1'))
    else:
        y = L.conv2d(y, int(out_size / features_div), (1, 1), name=appstr(name, '_conv1'))
        y = L.relu(L.BatchNormalization(y, name=appstr(name, '_bn2')), name=appstr(name, '_act2'))
        y = L.conv2d(y, out_size, kernel_size, strides=strides, name=appstr(name, '_conv2'))
    return L.add([shortcut, y])


def downscaling_unit(x, cfg, out_size=None, name=None):
    """common.py:70-86 (max-pooling flavour; strided-conv flavour keeps the reference's s1=(2,2))."""
    out_size = x.shape[-1] if out_size is None else out_size
    s1 = (2, 2) if cfg.downsampling_type == 'conv' else (1, 1)
    if cfg.downsampling_type == 'maxpooling':
        x = L.maxpooling2d(x, (2, 2))
    return residual_unit(x, cfg.kernel_size, out_size=out_size, strides=s1, name=appstr(name, '_r0'))


def upscaling_unit(x, cfg, out_size=None, name=None):
    """common.py:89-108; the transposed-conv flavour (downsampling_type='conv') is used by no experiment."""
    out_size = x.shape[-1] if out_size is None else out_size
    if cfg.downsampling_type != 'maxpooling':
        raise NotImplementedError("Conv2DTranspose up-scaling (downsampling_type='conv') is not on the hot path")
    return residual_unit(L.upsampling2d(x, (2, 2)), cfg.kernel_size, out_size=out_size, name=appstr(name, '_r0'))


# Aliases (common.py:158-160)
residual = residual_unit
downscaling = downscaling_unit
upscaling = upscaling_unit
