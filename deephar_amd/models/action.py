"""Merge action model (pose-based + appearance-based action recognition on top of a ReceptionNet) for the
gfx950 engine.  Public builders keep the reference signatures: deephar/models/action.py:45 (build_pose_model),
:93 (build_visual_model), :319 (build_merge_model).  build_guided_visual_model (:300) is broken in the
reference (passes num_context_per_joint twice) and `compile` (:403) is training-only: both out of scope.

Per clip:  T frames -> [Stem, rBlock1] per frame -> xb1 ; blocks 1..B chained, only the LAST block's heat-maps
kept -> soft-argmax pose y[T,J,dim], confidence p[T,J,1] ; probability maps x xb1 -> kronecker-pooled
appearance features f[T,J,576] ; PoseAR(y,p) and GuidedVisAR(f) over the (T,J) plane -> 4+4 action maps ->
soft-max(global max+min) heads + weighted merge 'm'.  Everything before PoseAR/GuidedVisAR is frame-independent
(this is what deephar_amd.parallel shards across GPUs).
"""
from .. import graph as G
from .. import layers as L
from ..model import Model
from . import blocks


def action_top(x, name=None):
    """action.action_top (action.py:14-17)"""
    return L.softmax(L.global_max_min_pooling(x), name=name)


def build_act_pred_block(x, num_out, name=None, last=False, include_top=True):
    """action.build_act_pred_block (action.py:20-42): bottleneck residual, then a max+min-pooled 3x3 head whose
    (pre-soft-max) action maps are up-sampled and re-injected unless this is the last block."""
    width = x.shape[-1]
    x = L.add([x, L.act_conv_bn(L.act_conv_bn(x, int(width / 2), (1, 1)), width, (3, 3))])
    trunk = x
    x1 = L.act_conv_bn(x, width, (3, 3))
    maps = L.act_conv(L.max_min_pooling(x1, (2, 2)), num_out, (3, 3))
    y = action_top(maps) if include_top else maps
    if not last:
        back = L.act_conv_bn(L.UpSampling2D(maps, (2, 2)), width, (3, 3))
        x = L.add([trunk, x1, back])
    return x, y


def _four_blocks(x, num_actions, include_top):
    outs = []
    for i in range(4):
        x, y = build_act_pred_block(x, num_actions, name='y%d' % (i + 1), include_top=include_top, last=(i == 3))
        outs.append(y)
    return outs


def build_pose_model(num_joints, num_actions, num_temp_frames=None, pose_dim=2, name=None, include_top=True,
                     network_version='v1'):
    """action.build_pose_model (action.py:45-90): convs over the (T, J) plane of confidence-masked poses."""
    widths = {'v1': (8, 16, 24, 56, 32), 'v2': (12, 24, 36, 112, 64)}
    if network_version not in widths:
        raise Exception('Unkown network version "{}"'.format(network_version))
    w1, w2, w3, w4, w5 = widths[network_version]
    with G.name_scope(name):
        y = L.Input((num_temp_frames, num_joints, pose_dim))
        p = L.Input((num_temp_frames, num_joints, 1))
        x = L.multiply([y, p])                               # y * tile(p)
        x = L.concatenate([L.conv_bn_act(x, w1, (3, 1)), L.conv_bn_act(x, w2, (3, 3)),
                           L.conv_bn_act(x, w3, (3, 5))])
        a = L.conv_bn(x, w4, (3, 3))
        b = L.conv_bn(L.conv_bn(x, w5, (1, 1)), w4, (3, 3))
        x = L.max_min_pooling(L.concatenate([a, b]), (2, 2))
        return Model(inputs=[y, p], outputs=_four_blocks(x, num_actions, include_top), name=name)


def build_visual_model(num_joints, num_actions, num_features, num_temp_frames=None, name=None,
                       include_top=True):
    """action.build_visual_model (action.py:93-109)"""
    with G.name_scope(name):
        inp = L.Input((num_temp_frames, num_joints, num_features))
        x = L.MaxPooling2D(L.conv_bn(inp, 256, (1, 1)), (2, 2))
        return Model(inp, _four_blocks(x, num_actions, include_top), name=name)


def _pose_regressor(model_pe, xb1_shape, num_blocks):
    """'PoseReg' (action.py:127-153): re-wire the ReceptionNet blocks so that only the last RegMap is output."""
    def layer(kind, i):
        return model_pe.get_layer('%s%d' % (kind, i))

    inp = L.Input(xb1_shape)
    x2 = layer('SepConv', 1)(inp)
    x = L.add([inp, x2, layer('fReMap', 1)(layer('RegMap', 1)(x2))])
    for i in range(2, num_blocks):
        x1 = layer('rBlock', i)(x)
        x2 = layer('SepConv', i)(x1)
        x = L.add([x1, x2, layer('fReMap', i)(layer('RegMap', i)(x2))])
    x = layer('RegMap', num_blocks)(layer('SepConv', num_blocks)(layer('rBlock', num_blocks)(x)))
    return Model(inp, x, name='PoseReg')


def _frames_to_maps(inp, model_pe, num_blocks):
    x1 = model_pe.get_layer('Stem')(inp)                      # TimeDistributed: leading T dim is batch
    xb1 = model_pe.get_layer('rBlock1')(x1)
    h = _pose_regressor(model_pe, xb1.shape[-3:], num_blocks)(xb1)
    return h, xb1


def _get_2d_pose_estimation_from_model(inp, model_pe, num_joints, num_blocks, num_context_per_joint,
                                       full_trainable=False):
    """action.py:112-205 -> (y [T,J,2], p [T,J,1], soft-maxed maps hs [T,H,W,J], xb1)."""
    num_frames = inp.shape[0]
    h, xb1 = _frames_to_maps(inp, model_pe, num_blocks)
    sam_s = model_pe.get_layer('sSAM')
    if num_context_per_joint > 0:
        hs, hc = h.channels(0, num_joints), h.channels(num_joints, h.shape[-1])
        agg = blocks.build_context_aggregation(num_joints, num_context_per_joint, 0.8, num_frames=num_frames,
                                               name='Agg')
        y = agg([sam_s(hs), model_pe.get_layer('cSAM')(hc), model_pe.get_layer('cjProb')(hc)])
    else:
        hs = h
        y = sam_s(hs)
    p = L.keypoint_confidence(hs, scale=4.0)                  # sjProb(4 * hs), action.py:200
    hs = L.act_channel_softmax(hs, name='td_ChannelSoftmax')
    return y, p, hs, xb1


def _get_3d_pose_estimation_from_model(inp, model_pe, num_joints, num_blocks, depth_maps, full_trainable=False):
    """action.py:208-297 -> (pose [T,J,3], visible [T,J,1], soft-maxed xy maps, xb1)."""
    h, xb1 = _frames_to_maps(inp, model_pe, num_blocks)
    assert h.shape[-1] == depth_maps * num_joints
    lead, (rows, cols) = h.shape[:-3], h.shape[-3:-1]
    meta = dict(D=int(depth_maps), J=int(num_joints))
    hxy = G.emit('depthmean', [h], [lead + (rows, cols, num_joints)], dict(meta, axis='d'))[0]
    hz = G.emit('depthmean', [h], [lead + (depth_maps, num_joints)], dict(meta, axis='hw'))[0]
    pose = L.concatenate([model_pe.get_layer('sSAM')(hxy), model_pe.get_layer('zSAM')(hz)])
    peak = L.add([G.emit('globalmax2d', [hxy], [lead + (num_joints,)])[0],
                  G.emit('globalmax1d', [hz], [lead + (num_joints,)])[0]])
    visible = L.sigmoid(L.scale(L.reshape(peak, lead + (num_joints, 1)), 2.0))     # sigmoid(2 * v), :291-292
    hxy = L.act_channel_softmax(hxy, name='td_ChannelSoftmax')
    return pose, visible, hxy, xb1


def _heatmap_weighting(x):
    """action.py:377-389: a trainable 1x1 SeparableConv2D (identity at initialisation in the reference)."""
    return L.sepconv2d(x, x.shape[-1], (1, 1))


def build_merge_model(model_pe,
                      num_actions,
                      input_shape,
                      num_frames,
                      num_joints,
                      num_blocks,
                      pose_dim=2,
                      depth_maps=8,
                      num_context_per_joint=2,
                      pose_net_version='v1',
                      output_poses=False,
                      weighted_merge=True,
                      ar_pose_weights=None,
                      ar_visual_weights=None,
                      full_trainable=False):
    """Drop-in for action.build_merge_model (action.py:319-400).  Outputs: [y, p]? + p1..p4 + v1..v4 + m."""
    inp = L.Input((num_frames,) + tuple(input_shape))
    outputs = []

    if pose_dim == 2:
        y, p, hs, xb1 = _get_2d_pose_estimation_from_model(inp, model_pe, num_joints, num_blocks,
                                                           num_context_per_joint, full_trainable=full_trainable)
    elif pose_dim == 3:
        y, p, hs, xb1 = _get_3d_pose_estimation_from_model(inp, model_pe, num_joints, num_blocks, depth_maps,
                                                           full_trainable=full_trainable)
    else:
        raise ValueError('pose_dim must be 2 or 3')

    if output_poses:
        outputs += [y, p]

    model_pose = build_pose_model(num_joints, num_actions, num_frames, pose_dim=pose_dim, include_top=False,
                                  name='PoseAR', network_version=pose_net_version)
    if ar_pose_weights is not None:
        model_pose.load_weights(ar_pose_weights)
    out_pose = model_pose([y, p])

    f = L.kronecker_prod(hs, xb1)
    model_vis = build_visual_model(num_joints, num_actions, f.shape[-1], num_temp_frames=num_frames,
                                   include_top=False, name='GuidedVisAR')
    if ar_visual_weights is not None:
        model_vis.load_weights(ar_visual_weights)
    out_vis = model_vis(f)

    outputs += [action_top(o, name='p%d' % (i + 1)) for i, o in enumerate(out_pose)]
    outputs += [action_top(o, name='v%d' % (i + 1)) for i, o in enumerate(out_vis)]

    p, v = out_pose[-1], out_vis[-1]
    if weighted_merge:
        p, v = _heatmap_weighting(p), _heatmap_weighting(v)
    outputs.append(action_top(L.add([p, v]), name='m'))

    return Model(inp, outputs)
