"""ReceptionNet graph builder for the gfx950 engine.

Public entry point `build(...)` has the signature and output ordering of the reference
deephar/models/reception.py:225-319; the nested sub-model names (Stem, rBlock%d, SepConv%d, RegMap%d,
fReMap%d, sSAM, cSAM, sjProb, cjProb, Agg, zSAM) are the ones deephar/models/action.py:117-179 looks up with
get_layer().  Layer creation order inside each sub-model follows the reference so that weight files map
one-to-one.  Training helpers (reception.compile, :322) are out of scope.

Per frame (MPII config):
  256x256x3 -> Stem -> 32x32x576 -> B x [ rBlock (3-level hourglass of separable residual units)
     -> SepConv kxk -> RegMap 1x1 -> heat-maps -> soft-argmax decoder ; fReMap 1x1 re-injection ]
"""
from .. import graph as G
from .. import layers as L
from ..model import Model
from . import blocks


def _sep_residual(x, width, tag, k=(3, 3)):
    """Separable residual unit (reception.py:43-59): identity (1x1 projection when the width changes)
    plus ReLU -> depthwise kxk -> 1x1 -> BN; a 1x1 'reduce' comes first when narrowing."""
    cin = x.shape[-1]
    skip = x if cin == width else L.act_conv_bn(x, width, (1, 1), name=tag + '_shortcut')
    if width < cin:
        x = L.act_conv_bn(x, width, (1, 1), name=tag + '_reduce')
    return L.add([skip, L.separable_act_conv_bn(x, width, k, name=tag)])


def _as_submodel(name, in_shape, body):
    """Build `body` on a fresh Input inside naming scope `name` and wrap it as a nested Model."""
    with G.name_scope(name):
        xi = L.Input(in_shape)
        return Model(xi, body(xi), name=name)


def _stem_body(old_model):
    """Inception-v4-like stem (reception.py:61-98): 256^2x3 -> 32^2x576 (new model)."""
    def body(x):
        x = L.conv_bn_act(x, 32, (3, 3), strides=(2, 2))
        if not old_model:
            x = L.conv_bn_act(x, 32, (3, 3))
        x = L.conv_bn_act(x, 64, (3, 3))
        # s4: strided conv || max-pool
        x = L.concatenate([L.conv_bn_act(x, 32 if old_model else 96, (3, 3), strides=(2, 2)),
                           L.MaxPooling2D(x, (3, 3), strides=(2, 2), padding='same')])
        # s5: two factorised branches
        a = L.conv_bn(L.conv_bn_act(x, 64, (1, 1)), 96, (3, 3))
        b = L.conv_bn_act(x, 64, (1, 1))
        for size in ((5, 1), (1, 5)):
            b = L.conv_bn_act(b, 64, size)
        x = L.concatenate([a, L.conv_bn(b, 96, (3, 3))])
        # s6: strided pre-activated conv || max-pool
        x = L.concatenate([L.act_conv_bn(x, 192, (3, 3), strides=(2, 2)),
                           L.MaxPooling2D(x, (2, 2), strides=(2, 2))])
        if not old_model:
            x = _sep_residual(x, 3 * 192, 'sepconv1')
        return x
    return body


def _hourglass_body(k):
    """Three-level hourglass of separable residual units (reception.py:101-131)."""
    def body(x):
        full = x.shape[-1]
        half = int(full / 2)
        top = _sep_residual(x, full, 'sepconv_l1', k)

        mid = L.act_conv_bn(L.MaxPooling2D(x, (2, 2)), half, (1, 1))
        mid = _sep_residual(mid, half, 'sepconv_l2_1', k)
        mid_skip = _sep_residual(mid, half, 'sepconv_l2_2', k)

        low = L.MaxPooling2D(mid, (2, 2))
        for i in (1, 2, 3):
            low = _sep_residual(low, half, 'sepconv_l3_%d' % i, k)

        mid = L.add([mid_skip, L.UpSampling2D(low, (2, 2))])
        mid = _sep_residual(mid, full, 'sepconv_l2_3', k)
        return L.add([top, L.UpSampling2D(mid, (2, 2))])
    return body


def _decode_2d_context(h, nj, sam_s, sam_c, prob_c, agg, prob_s):
    """reception.py:167-182: specialised maps h[..., :nj], contextual maps h[..., nj:]; confidences are taken
    on the RAW maps."""
    hs, hc = h.channels(0, nj), h.channels(nj, h.shape[-1])
    pose = agg([sam_s(hs), sam_c(hc), prob_c(hc)])
    return pose, prob_s(hs), hs


def _decode_3d(h, nj, depth, sam_s, sam_z):
    """reception.py:193-222: maps are (H, W, depth, joints) flattened as c = d*joints + j; xy from the
    depth-mean maps, z from the spatial-mean profile, visibility = sigmoid(max_hw + max_d)."""
    assert h.shape[-1] == depth * nj
    lead, (rows, cols) = h.shape[:-3], h.shape[-3:-1]
    meta = dict(D=int(depth), J=int(nj))
    hxy = G.emit('depthmean', [h], [lead + (rows, cols, nj)], dict(meta, axis='d'))[0]
    hz = G.emit('depthmean', [h], [lead + (depth, nj)], dict(meta, axis='hw'))[0]
    pose = L.concatenate([sam_s(hxy), sam_z(hz)])
    peak = L.add([G.emit('globalmax2d', [hxy], [lead + (nj,)])[0],
                  G.emit('globalmax1d', [hz], [lead + (nj,)])[0]])
    return pose, L.sigmoid(L.reshape(peak, lead + (nj, 1))), hxy


def build(input_shape, num_joints, dim,
          num_context_per_joint=None,
          alpha=0.8,
          num_blocks=4,
          depth_maps=16,
          ksize=(3, 3),
          export_heatmaps=False,
          export_vfeat_block=None,
          old_model=False,
          concat_pose_confidence=True):
    """Drop-in for reception.build (reception.py:225-319)."""
    if dim == 2:
        if num_context_per_joint is None:
            num_context_per_joint = 2
        num_heatmaps = (num_context_per_joint + 1) * num_joints
    elif dim == 3:
        assert num_context_per_joint is None, \
            'For 3D pose estimation, contextual heat maps are not allowed.'
        num_heatmaps = depth_maps * num_joints
    else:
        raise ValueError('"dim" must be 2 or 3 and not (%d)' % dim)

    inp = L.Input(tuple(input_shape))
    x = _as_submodel('Stem', inp.shape, _stem_body(old_model))(inp)
    if old_model:
        x = _sep_residual(x, 512, 'sepconv1')
    rows, cols, width = x.shape[-3:]

    # parameter-free decoder sub-models, shared by all blocks (reception.py:257-275)
    sam_s = blocks.build_softargmax_2d((rows, cols, num_joints), rho=0, name='sSAM')
    prob_s = blocks.build_joints_probability((rows, cols, num_joints), name='sjProb')
    if dim == 2 and num_context_per_joint is not None:
        ctx_shape = (rows, cols, num_heatmaps - num_joints)
        sam_c = blocks.build_softargmax_2d(ctx_shape, rho=0, name='cSAM')
        prob_c = blocks.build_joints_probability(ctx_shape, name='cjProb')
        agg = blocks.build_context_aggregation(num_joints, num_context_per_joint, alpha, name='Agg')
    if dim == 3:
        sam_z = blocks.build_softargmax_1d((depth_maps, num_joints), name='zSAM')

    outputs, vfeat = [], None
    for b in range(1, num_blocks + 1):
        x = _as_submodel('rBlock%d' % b, x.shape, _hourglass_body(ksize))(x)
        if export_vfeat_block == b:
            vfeat = x
        trunk = x
        x = _as_submodel('SepConv%d' % b, x.shape,
                         lambda t: L.separable_act_conv_bn(t, t.shape[-1], ksize))(x)
        h = _as_submodel('RegMap%d' % b, x.shape, lambda t: L.act_conv(t, num_heatmaps, (1, 1)))(x)

        if dim == 3:
            pose, visible, hm = _decode_3d(h, num_joints, depth_maps, sam_s, sam_z)
        elif num_context_per_joint is not None:
            pose, visible, hm = _decode_2d_context(h, num_joints, sam_s, sam_c, prob_c, agg, prob_s)
        else:
            pose, visible, hm = sam_s(h), prob_s(h), h

        outputs += [L.concatenate([pose, visible])] if concat_pose_confidence else [pose, visible]
        if export_heatmaps:
            outputs.append(hm)

        if b < num_blocks:
            back = _as_submodel('fReMap%d' % b, h.shape, lambda t: L.act_conv_bn(t, width, (1, 1)))(h)
            x = L.add([trunk, x, back])

    if vfeat is not None:
        outputs.append(vfeat)
    return Model(inputs=inp, outputs=outputs)
