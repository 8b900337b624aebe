"""CPU restatement of SPNet (reference deephar/models/spnet.py + models/common.py).
TEST INFRASTRUCTURE (see oracle/__init__.py: graph wiring pinned by reference-code goldens, Keras/TF layer numerics restated).

Frame-level tensors are kept as [N*T, H, W, C] (TimeDistributed == fold T into the batch); the action stream
works on [N, T, J, C] planes.  Every weight-carrying layer of SPNet is explicitly named in the reference, so
weights are looked up as '/<name>/<weight>' (top-level scope).
"""
import numpy as np
import torch

from . import ops
from .naming import Weights


def _bn(W, x, name):
    """Default keras BatchNormalization (gamma + beta), common.py:40,50,60."""
    c = x.shape[-1]
    return ops.batchnorm(x, W.get(name, 'beta', (c,)), W.get(name, 'moving_mean', (c,)),
                         W.get(name, 'moving_variance', (c,)), gamma=W.get(name, 'gamma', (c,)))


def _conv(W, x, filters, size, name, strides=(1, 1)):
    k = W.get(name, 'kernel', (size[0], size[1], x.shape[-1], filters))
    return ops.conv2d(x, k, strides, 'same')


def _sepconv(W, x, filters, size, name):
    dw = W.get(name, 'depthwise_kernel', (size[0], size[1], x.shape[-1], 1))
    pw = W.get(name, 'pointwise_kernel', (1, 1, x.shape[-1], filters))
    return ops.sepconv2d(x, dw, pw)


def residual_unit(W, x, kernel_size, name, strides=(1, 1), out_size=None, convtype='depthwise',
                  features_div=2):
    """common.residual_unit (common.py:25-67)"""
    nf = x.shape[-1]
    out_size = nf if out_size is None else out_size
    skip_conv = (nf != out_size) or (tuple(strides) != (1, 1))
    if skip_conv:
        x = _bn(W, x, name + '_bn1')
    shortcut = x
    if skip_conv:
        shortcut = _conv(W, ops.relu(shortcut), out_size, (1, 1), name + '_shortcut_conv', strides)
    if not skip_conv:
        x = _bn(W, x, name + '_bn1')
    x = ops.relu(x)
    if convtype == 'depthwise':
        x = _sepconv(W, x, out_size, kernel_size, name + '_conv1')
    else:
        x = _conv(W, x, int(out_size / features_div), (1, 1), name + '_conv1')
        x = ops.relu(_bn(W, x, name + '_bn2'))
        x = _conv(W, x, out_size, kernel_size, name + '_conv2', strides)
    return shortcut + x


def entry_flow(W, x, growth=96, image_div=8):
    """spnet.entry_flow (spnet.py:317-352), downsampling_type='maxpooling'."""
    x = _conv(W, x, 64, (7, 7), 'conv1', (2, 2))
    x = residual_unit(W, x, (3, 3), 'res0', out_size=growth, convtype='normal')
    x = ops.maxpool2d(x, (3, 3), (2, 2), 'same')
    x = residual_unit(W, x, (3, 3), 'res1', out_size=2 * growth, convtype='normal')
    x = residual_unit(W, x, (3, 3), 'res2', out_size=2 * growth, convtype='normal')
    nf, cnt, div = 2 * growth, 2, 4
    while div < image_div:
        nf += growth
        x = ops.maxpool2d(x, (2, 2), (2, 2), 'same')
        x = residual_unit(W, x, (3, 3), 'res%d' % (cnt + 1), out_size=nf, convtype='normal')
        x = residual_unit(W, x, (3, 3), 'res%d' % (cnt + 2), out_size=nf, convtype='normal')
        cnt += 2
        div *= 2
    return x


def prediction_branch(W, x, num_joints, name, pred_activate=True, forward_maps=True, reinject=True,
                      replica=False, taps=None):
    """spnet.prediction_branch (spnet.py:24-48).  `replica` (spnet.py:36-38): a second, independently weighted
    1x1 conv '<name>_conv1_replica' on the same activated input; its maps feed the ACTION stream only
    (spnet.py:216,224), the pose stream and the re-injection keep using '<name>_conv1'.
    `reinject=False` for the very last block of the model: its re-injection convs are created by the reference
    but are not reachable from any model output, so Keras drops them from the Model (no weights exist for them).
    Returns (re-injected features | None, prediction maps, replica maps | None)."""
    nf = x.shape[-1]
    x = ops.relu(x)
    if taps is not None and taps.get('want_head_inputs'):
        taps[name + '/in'] = x.numpy().copy()        # what the 1x1 heads read (tests/wellcond.py fits heads on it)
        if taps.get('stop_at') == name + '/in':      # the fit of this head needs nothing beyond this point
            raise StopForward(name)
    pred_maps = _conv(W, x, num_joints, (1, 1), name + '_conv1')
    rep = _conv(W, x, num_joints, (1, 1), name + '_conv1_replica') if replica else None
    if not reinject:
        return None, pred_maps, rep
    if forward_maps:
        x = torch.cat([_conv(W, x, num_joints, (1, 1), name + '_fw_maps'), pred_maps], dim=-1)
    else:
        x = pred_maps
    if pred_activate:
        x = ops.relu(x)
    return _conv(W, x, nf, (1, 1), name + '_conv2'), pred_maps, rep


def keypoint_confidence(h):
    """layers.keypoint_confidence (layers.py:107-119)"""
    return ops.joints_probability(h)


def action_early_fusion(W, xa, p, c, af, cfg, name, carry=True):
    """spnet.action_prediction_early_fusion (spnet.py:51-148).  p [N,T,J,dim], c [N,T,J,1], af [N,T,J,C]."""
    npf, nvf = cfg['num_pose_features'], cfg['num_visual_features']
    actions = []

    def _prediction(x, name):
        nf = x.shape[-1]
        ident = x
        x = ops.relu(_bn(W, x, name + '_bn1'))
        x1 = _conv(W, x, nf, (3, 3), name + '_conv1')
        x = ops.max_min_pooling(x1, (2, 2))
        x = ops.relu(_bn(W, x, name + '_bn2'))
        hlist = [_conv(W, x, nact, (3, 3), name + '_conv2h%d' % i) for i, nact in enumerate(cfg['num_actions'])]
        for h in hlist:
            actions.append(torch.softmax(ops.global_max_min_pooling(h), dim=-1))
        if not carry:      # last action block of the model: the carried features are dead code in Keras
            return None
        h = torch.cat(hlist, dim=-1) if len(hlist) > 1 else hlist[0]
        x = ops.relu(ops.upsample2d(h))
        x = _conv(W, x, nf, (3, 3), name + '_conv3')
        return ident + x1 + x

    num_frames, num_joints = p.shape[1], p.shape[2]
    time_stride = 2 if num_frames >= 16 else 1
    get_pad = lambda div, n: int(div * np.ceil(n / div) - n)
    joints_pad = get_pad(4, num_joints)
    frames_pad = get_pad(2 * time_stride, num_frames)
    top, bottom = frames_pad // 2, (frames_pad + 1) // 2
    left, right = joints_pad // 2, (joints_pad + 1) // 2

    def pad_pool(x):
        if top + bottom + left + right > 0:
            x = torch.nn.functional.pad(x, (0, 0, left, right, top, bottom))   # NHWC: pad W then H
        return ops.maxpool2d(x, (2, 2), (time_stride, 2), 'same')

    x = p * c
    a = _conv(W, x, npf // 16, (3, 1), name + '_p_conv0a')
    b = _conv(W, x, npf // 8, (3, 3), name + '_p_conv0b')
    cc = _conv(W, x, npf // 4, (3, 5), name + '_p_conv0c')
    x = torch.cat([a, b, cc], dim=-1)
    x = residual_unit(W, x, (3, 3), name + '_r1', out_size=npf, convtype='normal', features_div=2)
    x1 = pad_pool(x)
    x2 = pad_pool(_conv(W, af, nvf, (1, 1), name + '_v_conv0'))
    fusion = [x1, x2] + ([xa] if xa is not None else [])
    x = torch.cat(fusion, dim=-1)
    x = residual_unit(W, x, (3, 3), name + '_r2', out_size=max(npf, nvf), convtype='normal', features_div=4)
    return actions, _prediction(x, name + '_pred')


class _State:
    pass


class StopForward(Exception):
    """Raised by forward() when taps['stop_at'] names the head input just recorded (test infrastructure: the head fit
    of tests/wellcond.py runs one pass per prediction block and reads nothing behind that block)."""


def forward(weights, clips, cfg, dtype=torch.float32, taps=None):
    """spnet.build(cfg) + predict.  cfg: dict(num_joints, dim, num_actions, num_pyramids, action_pyramids,
    num_levels, kernel_size, growth, image_div, num_pose_features, num_visual_features, sam_alpha
    [, pose_replica=False]).
    taps: optional dict, filled with '<prediction block>/logits' = heat-map logits [N*T, h, w, J] and, for 3-D
    models, '<prediction block>/dlogits' = depth-map logits (numpy); when it already holds a true
    'want_head_inputs', also '<prediction block>_heatmaps/in' = the activated tensor the 1x1 heads read.
    clips: [N, T, H, W, 3] (or [N, H, W, 3]).  Returns poses [N,(T,)J,dim+1] ... then action scores [N, A] ..."""
    W = weights if isinstance(weights, Weights) else Weights(weights, dtype)
    W.reset()
    J, dim, ks, growth = cfg['num_joints'], cfg['dim'], cfg['kernel_size'], cfg['growth']
    alpha = cfg.get('sam_alpha', 1)
    st = _State()
    st.act_cnt = 0
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(clips)).to(dtype)
        clip = x.dim() == 5
        n, t = (x.shape[0], x.shape[1]) if clip else (x.shape[0], 1)
        frames = x.reshape((n * t,) + tuple(x.shape[-3:]))
        poses, actions = [], []

        def prediction_block(xp, xa, zp, do_action, name, last_pose, last_action):
            """spnet.prediction_block (spnet.py:151-248)"""
            nf = xp.shape[-1]
            xp = residual_unit(W, xp, ks, name + '_r1')
            reinject = [xp]
            xp = ops.relu(_bn(W, xp, name + '_bn1'))
            xp = _sepconv(W, xp, nf, ks, name + '_conv1')
            reinject.append(xp)
            xp = _bn(W, xp, name + '_bn2')
            replica = bool(cfg.get('pose_replica', False)) and do_action          # spnet.py:160
            x1, org_h, rep_h = prediction_branch(W, xp, J, name + '_heatmaps', pred_activate=True,
                                                 reinject=not last_pose, replica=replica, taps=taps)
            reinject.append(x1)
            if taps is not None:
                taps[name + '/logits'] = org_h.numpy().copy()
            h = ops.channel_softmax_2d(org_h, alpha)
            p = ops.softargmax2d_from_prob(h)
            c = keypoint_confidence(h)
            if dim == 3:
                x1, org_d, rep_d = prediction_branch(W, xp, J, name + '_depthmaps', pred_activate=False,
                                                     forward_maps=False, reinject=not last_pose, replica=replica)
                reinject.append(x1)
                if taps is not None:
                    taps[name + '/dlogits'] = org_d.numpy().copy()
                z = (torch.sigmoid(org_d) * h).sum(dim=(1, 2)).unsqueeze(-1)
                p = torch.cat([p, z], dim=-1)
            if do_action:
                st.act_cnt += 1
                act = 'act%d' % st.act_cnt
                act_h = ops.channel_softmax_2d(rep_h if replica else org_h, alpha)       # spnet.py:216-218
                act_p = ops.softargmax2d_from_prob(act_h)
                act_c = keypoint_confidence(act_h)
                if dim == 3:
                    act_d = rep_d if replica else org_d                                  # spnet.py:224
                    act_z = (torch.sigmoid(act_d) * act_h).sum(dim=(1, 2)).unsqueeze(-1)
                    act_p = torch.cat([act_p, act_z], dim=-1)
                af = ops.kronecker_prod(act_h, zp)
                unfold = lambda v: v.reshape((n, t) + tuple(v.shape[1:]))
                acts, xa = action_early_fusion(W, xa, unfold(act_p), unfold(act_c), unfold(af), cfg,
                                               act + '_action', carry=not last_action)
                actions.extend(acts)
            out = torch.cat([p, c], dim=-1)
            poses.append(out.reshape((n, t) + tuple(out.shape[1:])) if clip else out)
            if last_pose:
                return None, xa
            xp = reinject[0]
            for r in reinject[1:]:
                xp = xp + r
            return xp, xa

        L = cfg['num_levels']
        lp, la, lzp = [None] * L, [None] * L, [None] * L
        lp[0] = entry_flow(W, frames, growth, cfg.get('image_div', 8))
        for pyr in range(cfg['num_pyramids']):
            do_action = (pyr + 1) in cfg['action_pyramids']
            down = pyr % 2 == 0
            name = ('dp%d' if down else 'up%d') % (pyr + 1)
            xp, xa = (lp[0], la[0]) if down else (lp[-1], la[-1])
            if lzp[0] is None:
                lzp[0] = xp
            levels = list(range(1, L) if down else range(L - 1)[::-1])
            for i in levels:
                if down:   # common.downscaling_unit (common.py:70-86)
                    xp = ops.maxpool2d(xp, (2, 2), (2, 2), 'same')
                    xp = residual_unit(W, xp, ks, name + '_du%d_r0' % i, out_size=xp.shape[-1] + growth)
                else:      # common.upscaling_unit (common.py:89-108)
                    xp = ops.upsample2d(xp)
                    xp = residual_unit(W, xp, ks, name + '_uu%d_r0' % i, out_size=xp.shape[-1] - growth)
                if lzp[i] is None:
                    lzp[i] = xp
                if lp[i] is not None:
                    xp = xp + lp[i]
                if xa is not None and do_action:
                    xa = residual_unit(W, xa, (3, 3), name + ('_du%d_action_r0' if down else '_uu%d_action_r0') % i)
                    if la[i] is not None:
                        xa = xa + la[i]
                final = i == levels[-1]
                xp, xa = prediction_block(xp, xa, lzp[i], do_action, name + '_pb%d' % i,
                                          last_pose=final and pyr == cfg['num_pyramids'] - 1,
                                          last_action=final and (pyr + 1) == max(cfg['action_pyramids'], default=0))
                lp[i], la[i] = xp, xa
        return [o.numpy() for o in poses + actions]
