"""Parameter-free decoder sub-models (reference deephar/models/blocks.py:217-344).

The legacy stems/hourglass of blocks.py:9-214 are dead code in the reference (they call a `residual(int_size=`
that does not exist) and are intentionally not provided.
"""
from .. import graph as G
from ..layers import (Input, act_channel_softmax, softargmax2d, keypoint_confidence,
                      act_depth_softmax_interp)
from ..model import Model


def build_context_aggregation(num_joints, num_context, alpha, num_frames=1, name=None):
    """blocks.build_context_aggregation (blocks.py:217-285):
    y = alpha*ys + (1-alpha) * [sum_c(x_c p_c)/sum_c(p_c), sum_c(y_c p_c)/sum_c(p_c)] with the contexts of
    joint j being channels j*num_context .. j*num_context+num_context-1 (the frozen Dense of :221-233)."""
    lead = (num_frames,) if num_frames > 1 else ()
    ys = Input(lead + (num_joints, 2))
    yc = Input(lead + (num_joints * num_context, 2))
    pc = Input(lead + (num_joints * num_context, 1))
    y = G.emit('context_agg', [ys, yc, pc], [lead + (num_joints, 2)],
               dict(nctx=int(num_context), alpha=float(alpha)), name=name)[0]
    model = Model([ys, yc, pc], y, name=name)
    model.trainable = False
    return model


def build_softargmax_1d(input_shape, name=None):
    """blocks.build_softargmax_1d (blocks.py:288-303): depth soft-max + lin_interpolation_1d."""
    inp = Input(input_shape)
    x = act_depth_softmax_interp(inp, name=name)
    model = Model(inp, x, name=name)
    model.trainable = False
    return model


def build_softargmax_2d(input_shape, rho=0., name=None):
    """blocks.build_softargmax_2d (blocks.py:306-325).  rho>0 (KL regulariser) is training-only."""
    if rho > 0:
        raise NotImplementedError('kl_divergence_regularizer is a training-time activity regulariser')
    inp = Input(input_shape)
    x = act_channel_softmax(inp, name=(name + '_softmax') if name else None)
    x = softargmax2d(x)
    model = Model(inp, x, name=name)
    model.trainable = False
    return model


def build_joints_probability(input_shape, name=None, verbose=0):
    """blocks.build_joints_probability (blocks.py:328-343)."""
    inp = Input(input_shape)
    x = keypoint_confidence(inp)
    model = Model(inp, x, name=name)
    if verbose:
        model.summary()
    return model
