"""CPU restatement of the ReceptionNet forward pass (reference deephar/models/reception.py).
TEST INFRASTRUCTURE (see oracle/__init__.py: graph wiring pinned by reference-code goldens, Keras/TF layer numerics restated).

Written as straight functional code in the reference's source order, so that layer creation order (and with
it the weight naming of oracle/naming.py) follows the reference.  Everything is NHWC.
"""
import numpy as np
import torch

from . import ops
from .naming import Weights


# ---- layer helpers (reference deephar/layers.py) ---------------------------------------------------

def _bn(W, x, name):
    """BatchNormalization(axis=-1, scale=False, name=name) -- layers.py:209,239,268,300."""
    lname = name or W.auto('batch_normalization')
    c = x.shape[-1]
    return ops.batchnorm(x, W.get(lname, 'beta', (c,)), W.get(lname, 'moving_mean', (c,)),
                         W.get(lname, 'moving_variance', (c,)))


def _conv(W, x, filters, size, strides, padding, name):
    """layers.conv2d (layers.py:66-71)."""
    lname = name or W.auto('conv2d')
    k = W.get(lname, 'kernel', (size[0], size[1], x.shape[-1], filters))
    return ops.conv2d(x, k, strides, padding)


def _sepconv(W, x, filters, size, strides, padding, name):
    """keras SeparableConv2D(use_bias=False) as used by layers.py:288-301."""
    lname = name or W.auto('separable_conv2d')
    dw = W.get(lname, 'depthwise_kernel', (size[0], size[1], x.shape[-1], 1))
    pw = W.get(lname, 'pointwise_kernel', (1, 1, x.shape[-1], filters))
    return ops.sepconv2d(x, dw, pw, strides, padding)


def conv_bn(W, x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.py:202-210"""
    x = _conv(W, x, filters, size, strides, padding, name + '_conv' if name else None)
    return _bn(W, x, name)


def conv_bn_act(W, x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.py:230-241"""
    x = _conv(W, x, filters, size, strides, padding, name + '_conv' if name else None)
    x = _bn(W, x, name + '_bn' if name else None)
    return ops.relu(x)


def act_conv_bn(W, x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.py:258-269"""
    x = ops.relu(x)
    x = _conv(W, x, filters, size, strides, padding, name + '_conv' if name else None)
    return _bn(W, x, name)


def act_conv(W, x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.py:317-325"""
    return _conv(W, ops.relu(x), filters, size, strides, padding, name)


def separable_act_conv_bn(W, x, filters, size, strides=(1, 1), padding='same', name=None):
    """layers.py:288-301"""
    x = ops.relu(x)
    x = _sepconv(W, x, filters, size, strides, padding, name + '_conv' if name else None)
    return _bn(W, x, name)


# ---- reception.py ------------------------------------------------------------------------------------

def sepconv_residual(W, x, out_size, name, kernel_size=(3, 3)):
    """reception._sepconv_residual (reception.py:43-59)"""
    num_filters = x.shape[-1]
    if num_filters == out_size:
        ident = x
    else:
        ident = act_conv_bn(W, x, out_size, (1, 1), name=name + '_shortcut')
    if out_size < num_filters:
        x = act_conv_bn(W, x, out_size, (1, 1), name=name + '_reduce')
    x = separable_act_conv_bn(W, x, out_size, kernel_size, name=name)
    return ident + x


def stem(W, inp):
    """reception._stem, old_model=False (reception.py:61-98)"""
    W.push('Stem')
    x = conv_bn_act(W, inp, 32, (3, 3), strides=(2, 2))
    x = conv_bn_act(W, x, 32, (3, 3))
    x = conv_bn_act(W, x, 64, (3, 3))

    a = conv_bn_act(W, x, 96, (3, 3), strides=(2, 2))
    b = ops.maxpool2d(x, (3, 3), (2, 2), 'same')
    x = torch.cat([a, b], dim=-1)

    a = conv_bn_act(W, x, 64, (1, 1))
    a = conv_bn(W, a, 96, (3, 3))
    b = conv_bn_act(W, x, 64, (1, 1))
    b = conv_bn_act(W, b, 64, (5, 1))
    b = conv_bn_act(W, b, 64, (1, 5))
    b = conv_bn(W, b, 96, (3, 3))
    x = torch.cat([a, b], dim=-1)

    a = act_conv_bn(W, x, 192, (3, 3), strides=(2, 2))
    b = ops.maxpool2d(x, (2, 2), (2, 2), 'valid')
    x = torch.cat([a, b], dim=-1)

    x = sepconv_residual(W, x, 3 * 192, name='sepconv1')
    W.pop()
    return x


def reception_block(W, xi, name, ksize):
    """reception.build_reception_block (reception.py:101-131)"""
    W.push(name)
    size = xi.shape[-1]
    a = sepconv_residual(W, xi, size, 'sepconv_l1', ksize)

    low1 = ops.maxpool2d(xi, (2, 2))
    low1 = act_conv_bn(W, low1, size // 2, (1, 1))
    low1 = sepconv_residual(W, low1, size // 2, 'sepconv_l2_1', ksize)
    b = sepconv_residual(W, low1, size // 2, 'sepconv_l2_2', ksize)

    c = ops.maxpool2d(low1, (2, 2))
    c = sepconv_residual(W, c, size // 2, 'sepconv_l3_1', ksize)
    c = sepconv_residual(W, c, size // 2, 'sepconv_l3_2', ksize)
    c = sepconv_residual(W, c, size // 2, 'sepconv_l3_3', ksize)
    c = ops.upsample2d(c)

    b = b + c
    b = sepconv_residual(W, b, size, 'sepconv_l2_3', ksize)
    b = ops.upsample2d(b)
    x = a + b
    W.pop()
    return x


def sconv_block(W, xi, name, ksize):
    """reception.build_sconv_block (reception.py:134-142)"""
    W.push(name)
    x = separable_act_conv_bn(W, xi, xi.shape[-1], ksize)
    W.pop()
    return x


def regmap_block(W, xi, num_maps, name):
    """reception.build_regmap_block (reception.py:145-153)"""
    W.push(name)
    x = act_conv(W, xi, num_maps, (1, 1))
    W.pop()
    return x


def fremap_block(W, xi, num_filters, name):
    """reception.build_fremap_block (reception.py:156-164)"""
    W.push(name)
    x = act_conv_bn(W, xi, num_filters, (1, 1))
    W.pop()
    return x


def pose_regression_2d_context(h, num_joints, num_context, alpha):
    """reception.pose_regression_2d_context (reception.py:167-182) with sSAM/cSAM/sjProb/cjProb/Agg."""
    hs = h[..., :num_joints]
    hc = h[..., num_joints:]
    ps = ops.softargmax2d(hs)
    pc = ops.softargmax2d(hc)
    vc = ops.joints_probability(hc)
    pose = ops.context_aggregation(ps, pc, vc, num_joints, num_context, alpha)
    visible = ops.joints_probability(hs)
    return pose, visible, hs, dict(ps=ps, pc=pc, vc=vc)


def pose_regression_2d(h):
    """reception.pose_regression_2d (reception.py:185-190)"""
    return ops.softargmax2d(h), ops.joints_probability(h), h


def pose_regression_3d(h, num_joints, depth_maps):
    """reception.pose_regression_3d (reception.py:193-222); channel c = d*num_joints + j."""
    n, rows, cols, ch = h.shape
    assert ch == depth_maps * num_joints
    h5 = h.reshape(n, rows, cols, depth_maps, num_joints)
    hxy = h5.mean(dim=3)
    hz = h5.mean(dim=(1, 2))
    pxy = ops.softargmax2d(hxy)
    pz = ops.softargmax1d(hz)
    pose = torch.cat([pxy, pz], dim=-1)
    vxy = torch.amax(hxy, dim=(1, 2))
    vz = torch.amax(hz, dim=1)
    visible = torch.sigmoid((vxy + vz).unsqueeze(-1))
    return pose, visible, hxy


def forward(weights, x, num_joints, dim, num_context_per_joint=None, alpha=0.8, num_blocks=4,
            depth_maps=16, ksize=(3, 3), export_heatmaps=False, export_vfeat_block=None,
            concat_pose_confidence=True, dtype=torch.float32, taps=None):
    """reception.build(...) + Model.predict (reception.py:225-319).

    weights: {key: np.ndarray}; x: [N,H,W,3] array.  Returns the list of outputs as numpy arrays in
    model.outputs order.  `taps`, if a dict, receives named intermediates for debugging/parity.
    """
    if dim == 2:
        if num_context_per_joint is None:
            num_context_per_joint = 2
        num_heatmaps = (num_context_per_joint + 1) * num_joints
    elif dim == 3:
        assert num_context_per_joint is None
        num_heatmaps = depth_maps * num_joints
    else:
        raise ValueError('"dim" must be 2 or 3 and not (%d)' % dim)

    W = weights if isinstance(weights, Weights) else Weights(weights, dtype)
    W.reset()
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(x)).to(dtype)
        outputs = []
        vfeat = None
        x = stem(W, x)
        if taps is not None:
            taps['stem'] = x
        for bidx in range(num_blocks):
            b = bidx + 1
            num_filters = x.shape[-1]
            x = reception_block(W, x, 'rBlock%d' % b, ksize)
            if export_vfeat_block == b:
                vfeat = x
            ident_map = x
            x = sconv_block(W, x, 'SepConv%d' % b, ksize)
            h = regmap_block(W, x, num_heatmaps, 'RegMap%d' % b)
            if taps is not None:
                taps['rblock%d' % b] = ident_map
                taps['heatmaps%d' % b] = h
            if dim == 2:
                if num_context_per_joint is not None and num_context_per_joint > 0:
                    pose, visible, hm, aux = pose_regression_2d_context(h, num_joints, num_context_per_joint,
                                                                         alpha)
                    if taps is not None:
                        for k, v in aux.items():
                            taps['%s%d' % (k, b)] = v
                else:
                    pose, visible, hm = pose_regression_2d(h)
            else:
                pose, visible, hm = pose_regression_3d(h, num_joints, depth_maps)
            if concat_pose_confidence:
                outputs.append(torch.cat([pose, visible], dim=-1))
            else:
                outputs.append(pose)
                outputs.append(visible)
            if export_heatmaps:
                outputs.append(hm)
            if bidx < num_blocks - 1:
                h = fremap_block(W, h, num_filters, 'fReMap%d' % b)
                x = ident_map + x + h
        if vfeat is not None:
            outputs.append(vfeat)
        if taps is not None:
            for k in list(taps):
                taps[k] = taps[k].numpy()
        return [o.numpy() for o in outputs]
