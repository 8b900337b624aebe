"""SPNet (Sequential Pyramid Network, TPAMI'20) graph builder for the gfx950 engine.

Drop-in for the reference deephar/models/spnet.py: build(cfg) (:355), get_num_predictions (:413), split_model
(:417).  Training-side helpers (compile_split_models :451, stop_grad_stem) are out of scope.  Layer names follow
the reference exactly -- SPNet weights are loaded by name (exp/pennaction/eval_penn_multitask.py:76).

Per frame: entry flow (7x7 s2 -> res0 -> pool -> res1,res2 -> pool -> res3,res4 = 32x32x288), then alternating
down / up pyramids; every level ends in a prediction block that emits pose [J, dim+1] from soft-argmax'ed
heat-maps (+ sigmoid depth maps) and re-injects its maps.  Action recognition (early fusion) runs on the (T, J)
plane of poses and kronecker-pooled appearance features; the action stream never feeds the pose stream, so all
per-frame work is finished before any temporal op.
"""
import numpy as np

from .. import layers as L
from ..config import ModelConfig
from ..model import Model
from ..utils import appstr
from .common import residual, downscaling, upscaling, add_tensorlist, concat_tensorlist

_act_cnt = 0   # the reference's process-global act%d counter (spnet.py:210-214); reset per build()


def prediction_branch(x, cfg, pred_activate=True, replica=None, forward_maps=True, name=None):
    """spnet.prediction_branch (spnet.py:24-48): maps = 1x1(relu(x)); re-injection = 1x1(relu?(concat[fw, maps]))."""
    width = x.shape[-1]
    x = L.relu(x, name=appstr(name, '_act1'))
    maps = L.conv2d(x, cfg.num_joints, (1, 1), name=appstr(name, '_conv1'))
    twin = L.conv2d(x, cfg.num_joints, (1, 1), name=appstr(name, '_conv1_replica')) if replica else replica
    back = L.concatenate([L.conv2d(x, cfg.num_joints, (1, 1), name=appstr(name, '_fw_maps')), maps]) \
        if forward_maps else maps
    if pred_activate:
        back = L.relu(back, name=appstr(name, '_act2'))
    return L.conv2d(back, width, (1, 1), name=appstr(name, '_conv2')), maps, twin


def action_prediction_early_fusion(xa, p, c, af, cfg, name=None):
    """spnet.action_prediction_early_fusion (spnet.py:51-148) -> ([soft-maxed action scores], carried features)."""
    npf, nvf = cfg.num_pose_features, cfg.num_visual_features
    short = name[0:7] if name is not None else None
    scores = []

    def predict(x, name, short):
        """_prediction (spnet.py:69-94)"""
        trunk = x
        x1 = L.conv2d(L.relu(L.BatchNormalization(x, name=appstr(name, '_bn1')), name=appstr(name, '_act1')),
                      x.shape[-1], (3, 3), name=appstr(name, '_conv1'))
        x = L.relu(L.BatchNormalization(L.max_min_pooling(x1, (2, 2)), name=appstr(name, '_bn2')),
                   name=appstr(name, '_act2'))
        heads = [L.conv2d(x, nact, (3, 3), name=appstr(name, '_conv2h%d' % i))
                 for i, nact in enumerate(cfg.num_actions)]
        for i, h in enumerate(heads):                       # one soft-max per dataset / action set
            scores.append(L.softmax(L.global_max_min_pooling(h), name=appstr(short, '%d' % i)))
        x = L.conv2d(L.relu(L.UpSampling2D(concat_tensorlist(heads), (2, 2)), name=appstr(name, '_act3')),
                     trunk.shape[-1], (3, 3), name=appstr(name, '_conv3'))
        return L.add([trunk, x1, x])

    # padding so that (T, J) pool cleanly: J to a multiple of 4, T to a multiple of 2*time_stride (spnet.py:96-107)
    num_frames, num_joints = p.shape[0], p.shape[1]
    time_stride = 2 if num_frames >= 16 else 1
    pad_of = lambda div, n: int(div * np.ceil(n / div) - n)
    jp, fp = pad_of(4, num_joints), pad_of(2 * time_stride, num_frames)
    pads = ((fp // 2, (fp + 1) // 2), (jp // 2, (jp + 1) // 2))

    def pad_pool(x):
        if sum(pads[0]) + sum(pads[1]) > 0:
            x = L.ZeroPadding2D(x, pads)
        return L.maxpooling2d(x, (2, 2), strides=(time_stride, 2))

    # pose features: confidence-masked coordinates -> three bare convs over the (T, J) plane
    x = L.multiply([p, c])
    x = L.concatenate([L.conv2d(x, npf // 16, (3, 1), name=appstr(name, '_p_conv0a')),
                       L.conv2d(x, npf // 8, (3, 3), name=appstr(name, '_p_conv0b')),
                       L.conv2d(x, npf // 4, (3, 5), name=appstr(name, '_p_conv0c'))])
    x1 = pad_pool(residual(x, (3, 3), out_size=npf, convtype='normal', features_div=2, name=appstr(name, '_r1')))
    # appearance features
    x2 = pad_pool(L.conv2d(af, nvf, (1, 1), name=appstr(name, '_v_conv0')))
    fusion = [x1, x2] + ([xa] if xa is not None else [])
    x = residual(concat_tensorlist(fusion), (3, 3), out_size=max(npf, nvf), convtype='normal', features_div=4,
                 name=appstr(name, '_r2'))
    return scores, predict(x, appstr(name, '_pred'), appstr(short, '_a'))


def prediction_block(xp, xa, zp, outlist, cfg, do_action, name=None):
    """spnet.prediction_block (spnet.py:151-248)."""
    global _act_cnt
    if cfg.dbg_decoupled_pose or cfg.dbg_decoupled_h:
        raise NotImplementedError('debug outputs (dbg_decoupled_*) are not part of the hot path')
    limits = (cfg.xmin, cfg.ymin, 1 - cfg.xmin, 1 - cfg.ymin)      # ignored downstream, like the reference
    width = xp.shape[-1]
    replica = cfg.pose_replica and do_action

    xp = residual(xp, cfg.kernel_size, name=appstr(name, '_r1'))
    reinject = [xp]
    xp = L.sepconv2d(L.relu(L.BatchNormalization(xp, name=appstr(name, '_bn1')), name=appstr(name, '_act1')),
                     width, cfg.kernel_size, name=appstr(name, '_conv1'))
    reinject.append(xp)
    xp = L.BatchNormalization(xp, name=appstr(name, '_bn2'))

    # 2-D pose from soft-argmax'ed heat-maps; confidence on the PROBABILITY maps (spnet.py:178-183)
    x1, org_h, rep_h = prediction_branch(xp, cfg, pred_activate=True, replica=replica, name=appstr(name, '_heatmaps'))
    reinject.append(x1)
    h = L.act_channel_softmax(org_h, alpha=cfg.sam_alpha, name=appstr(name, '_probmaps'))
    p = L.softargmax2d(h, limits=limits, name=appstr(name, '_xy'))
    c = L.keypoint_confidence(h, name=appstr(name, '_vis'))

    if cfg.dim == 3:
        x1, org_d, rep_d = prediction_branch(xp, cfg, pred_activate=False, replica=replica, forward_maps=False,
                                             name=appstr(name, '_depthmaps'))
        reinject.append(x1)
        p = L.concatenate([p, L.depth_from_maps(org_d, h)], name=appstr(name, '_xyz'))

    action = []
    if do_action:
        _act_cnt += 1
        act = 'act%d' % _act_cnt
        act_h = L.act_channel_softmax(rep_h if replica else org_h, alpha=cfg.sam_alpha,
                                      name=appstr(act, '_probmaps2'))
        act_p = L.softargmax2d(act_h, limits=limits, name=appstr(act, '_xy2'))
        act_c = L.keypoint_confidence(act_h, name=appstr(act, '_vis2'))
        if cfg.dim == 3:
            act_p = L.concatenate([act_p, L.depth_from_maps(rep_d if replica else org_d, act_h)],
                                  name=appstr(act, '_xyz2'))
        af = L.kronecker_prod(act_h, zp, name=appstr(act, '_kron'))
        action, xa = action_prediction_early_fusion(xa, act_p, act_c, af, cfg, name=appstr(act, '_action'))

    xp = add_tensorlist(reinject)
    outlist[0].append(L.concatenate([p, c], name=name))
    if do_action:
        outlist[1] += action
    return xp, xa


def _pyramid(levels, step, scale_unit, tag, lp, la, lzp, outlist, cfg, do_action, name):
    """Shared body of downscaling_pyramid / upscaling_pyramid (spnet.py:251-314)."""
    assert len(lp) == len(la), 'Pose and action must have the same number of levels!'
    xp, xa = (lp[0], la[0]) if step > 0 else (lp[-1], la[-1])
    if lzp[0] is None:
        lzp[0] = xp
    for i in levels:
        xp = scale_unit(xp, cfg, out_size=xp.shape[-1] + step * cfg.growth, name=appstr(name, '_%s%d' % (tag, i)))
        if lzp[i] is None:
            lzp[i] = xp                      # first visit of the level: appearance features for kronecker pooling
        if lp[i] is not None:
            xp = L.add([xp, lp[i]])
        if xa is not None and do_action:
            xa = residual(xa, (3, 3), name=appstr(name, '_%s%d_action_r0' % (tag, i)))
            if la[i] is not None:
                xa = L.add([xa, la[i]])
        xp, xa = prediction_block(xp, xa, lzp[i], outlist, cfg, do_action, name=appstr(name, '_pb%d' % i))
        lp[i], la[i] = xp, xa                # lateral connections


def downscaling_pyramid(lp, la, lzp, outlist, cfg, do_action, name=None):
    _pyramid(range(1, len(lp)), +1, downscaling, 'du', lp, la, lzp, outlist, cfg, do_action, name)


def upscaling_pyramid(lp, la, lzp, outlist, cfg, do_action, name=None):
    _pyramid(range(len(lp) - 1)[::-1], -1, upscaling, 'uu', lp, la, lzp, outlist, cfg, do_action, name)


def entry_flow(x, cfg):
    """spnet.entry_flow (spnet.py:317-352): 256^2x3 -> (256/image_div)^2 x (image_div/4 + 1)*growth."""
    growth, image_div = cfg.growth, cfg.image_div
    assert (image_div & (image_div - 1) == 0) and image_div >= 4, 'Invalid image_div ({}).'.format(image_div)
    assert cfg.downsampling_type in ['maxpooling', 'conv'], \
        'Invalid downsampling_type ({}).'.format(cfg.downsampling_type)
    x = L.conv2d(x, 64, (7, 7), strides=(2, 2), name='conv1')
    x = residual(x, (3, 3), out_size=growth, convtype='normal', name='res0')
    x = L.maxpooling2d(x, (3, 3), strides=(2, 2))
    x = residual(x, (3, 3), out_size=2 * growth, convtype='normal', name='res1')
    x = residual(x, (3, 3), out_size=2 * growth, convtype='normal', name='res2')
    width, idx, div = 2 * growth, 2, 4
    s1 = (2, 2) if cfg.downsampling_type == 'conv' else (1, 1)
    while div < image_div:
        width += growth
        if cfg.downsampling_type == 'maxpooling':
            x = L.maxpooling2d(x, (2, 2), strides=(2, 2))
        x = residual(x, (3, 3), out_size=width, strides=s1, convtype='normal', name='res%d' % (idx + 1))
        x = residual(x, (3, 3), out_size=width, convtype='normal', name='res%d' % (idx + 2))
        idx += 2
        div *= 2
    return x


def build(cfg, stop_grad_stem=False):
    """Drop-in for spnet.build (spnet.py:355-410).  Outputs: all poses [.., J, dim+1] first, then all action
    scores (one per action-enabled prediction block and action set)."""
    global _act_cnt
    assert type(cfg) == ModelConfig, 'type(cfg) ({}) is not ModelConfig'.format(type(cfg))
    input_shape = tuple(cfg.input_shape)
    assert len(input_shape) in [3, 4], 'Invalid input_shape ({})'.format(input_shape)
    _act_cnt = 0

    inp = L.Input(input_shape)
    outlist = [[] for _ in range(len(cfg.num_actions) + 1)]
    rows, cols = input_shape[-3], input_shape[-2]
    cfg.xmin = 1 / (2 * cols)
    cfg.ymin = 1 / (2 * rows)

    lp, la, lzp = ([None] * cfg.num_levels for _ in range(3))
    lp[0] = entry_flow(inp, cfg)
    for pyr in range(cfg.num_pyramids):
        do_action = (pyr + 1) in cfg.action_pyramids
        if pyr % 2 == 0:
            downscaling_pyramid(lp, la, lzp, outlist, cfg, do_action, name='dp%d' % (pyr + 1))
        else:
            upscaling_pyramid(lp, la, lzp, outlist, cfg, do_action, name='up%d' % (pyr + 1))

    return Model(inputs=inp, outputs=[o for group in outlist for o in group], name='SPNet')


def get_num_predictions(num_pyramids, num_levels):
    return num_pyramids * (num_levels - 1)


def split_model(full_model, cfg, interlaced=False, model_names=[None, None]):
    """spnet.split_model (spnet.py:417-448): [pose model, action model] sharing the full model's graph."""
    num_pose_pred = get_num_predictions(cfg.num_pyramids, cfg.num_levels)
    num_act_pred = get_num_predictions(len(cfg.action_pyramids), cfg.num_levels)
    assert len(full_model.outputs) == num_pose_pred + len(cfg.num_actions) * num_act_pred, \
        'The given model and config are not compatible!'
    assert num_act_pred > 0, 'You are trying to split a "pose only" model.'
    outs = full_model.outputs
    if interlaced:
        out_p, out_a, idx = [], [], 0
        for _ in range(num_pose_pred):
            out_p.append(outs[idx])
            idx += 1
            if len(out_a) < len(cfg.num_actions) * num_act_pred:
                out_a += outs[idx:idx + len(cfg.num_actions)]
                idx += len(cfg.num_actions)
    else:
        out_p, out_a = outs[:num_pose_pred], outs[num_pose_pred:]
    return [Model(full_model.input, out_p, name=model_names[0]), Model(full_model.input, out_a, name=model_names[1])]
