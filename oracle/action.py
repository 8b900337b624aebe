"""CPU restatement of the merge action model (reference deephar/models/action.py).
TEST INFRASTRUCTURE (see oracle/__init__.py: graph wiring pinned by reference-code goldens, Keras/TF layer numerics restated).

TimeDistributed(f)(x) is restated as "fold T into the batch, apply f, unfold" (SURVEY.md A.3).
"""
import numpy as np
import torch

from . import ops
from . import reception as R
from .naming import Weights


def action_top(x):
    """action.action_top (action.py:14-17): global max+min pooling over (T,J) then soft-max."""
    return torch.softmax(ops.global_max_min_pooling(x), dim=-1)


def act_pred_block(W, x, num_out, last=False):
    """action.build_act_pred_block (action.py:20-42), include_top=False: returns (x, raw action maps)."""
    nf = x.shape[-1]
    ident = x
    x = R.act_conv_bn(W, x, nf // 2, (1, 1))
    x = R.act_conv_bn(W, x, nf, (3, 3))
    x = ident + x

    ident = x
    x1 = R.act_conv_bn(W, x, nf, (3, 3))
    x = ops.max_min_pooling(x1, (2, 2))
    action_hm = R.act_conv(W, x, num_out, (3, 3))
    y = action_hm
    if not last:
        action_hm = ops.upsample2d(action_hm)
        action_hm = R.act_conv_bn(W, action_hm, nf, (3, 3))
        x = ident + x1 + action_hm
    return x, y


def pose_model(W, y, p, num_actions, version='v1'):
    """action.build_pose_model (action.py:45-90), include_top=False."""
    W.push('PoseAR')
    x = y * p                      # K.tile(p, pose_dim) * y
    c1, c2, c3, c4, c5 = (8, 16, 24, 56, 32) if version == 'v1' else (12, 24, 36, 112, 64)
    if version not in ('v1', 'v2'):
        raise Exception('Unkown network version "{}"'.format(version))
    a = R.conv_bn_act(W, x, c1, (3, 1))
    b = R.conv_bn_act(W, x, c2, (3, 3))
    c = R.conv_bn_act(W, x, c3, (3, 5))
    x = torch.cat([a, b, c], dim=-1)
    a = R.conv_bn(W, x, c4, (3, 3))
    b = R.conv_bn(W, x, c5, (1, 1))
    b = R.conv_bn(W, b, c4, (3, 3))
    x = torch.cat([a, b], dim=-1)
    x = ops.max_min_pooling(x, (2, 2))
    outs = []
    for i in range(4):
        x, yi = act_pred_block(W, x, num_actions, last=(i == 3))
        outs.append(yi)
    W.pop()
    return outs


def visual_model(W, f, num_actions):
    """action.build_visual_model (action.py:93-109), include_top=False."""
    W.push('GuidedVisAR')
    x = R.conv_bn(W, f, 256, (1, 1))
    x = ops.maxpool2d(x, (2, 2))
    outs = []
    for i in range(4):
        x, yi = act_pred_block(W, x, num_actions, last=(i == 3))
        outs.append(yi)
    W.pop()
    return outs


def pose_regressor(W, xb1, num_blocks, ksize, num_heatmaps):
    """The 'PoseReg' sub-model (action.py:127-153 / 225-250): blocks 1..num_blocks, only the last RegMap kept."""
    width = xb1.shape[-1]
    x2 = R.sconv_block(W, xb1, 'SepConv1', ksize)
    x3 = R.fremap_block(W, R.regmap_block(W, x2, num_heatmaps, 'RegMap1'), width, 'fReMap1')
    x = xb1 + x2 + x3
    for i in range(2, num_blocks):
        x1 = R.reception_block(W, x, 'rBlock%d' % i, ksize)
        x2 = R.sconv_block(W, x1, 'SepConv%d' % i, ksize)
        x3 = R.fremap_block(W, R.regmap_block(W, x2, num_heatmaps, 'RegMap%d' % i), width, 'fReMap%d' % i)
        x = x1 + x2 + x3
    x = R.reception_block(W, x, 'rBlock%d' % num_blocks, ksize)
    x = R.sconv_block(W, x, 'SepConv%d' % num_blocks, ksize)
    return R.regmap_block(W, x, num_heatmaps, 'RegMap%d' % num_blocks)


def heatmap_weighting(W, x):
    """action._heatmap_weighting (action.py:377-389): a trainable 1x1 SeparableConv2D."""
    lname = W.auto('separable_conv2d')
    c = x.shape[-1]
    dw = W.get(lname, 'depthwise_kernel', (1, 1, c, 1))
    pw = W.get(lname, 'pointwise_kernel', (1, 1, c, c))
    return ops.sepconv2d(x, dw, pw, (1, 1), 'valid')


def merge_frames(W, frames, num_joints, num_blocks, pose_dim=2, depth_maps=8, num_context_per_joint=2,
                 ksize=(5, 5)):
    """Frame-independent stage of the merge model (action.py:112-205 / 208-297 + the kronecker pooling of :359):
    frames [F, H, W, 3] -> y [F, J, dim], p [F, J, 1], pooled appearance features f [F, J, C]."""
    x1 = R.stem(W, frames)
    xb1 = R.reception_block(W, x1, 'rBlock1', ksize)
    if pose_dim == 2:
        num_heatmaps = (num_context_per_joint + 1) * num_joints
        h = pose_regressor(W, xb1, num_blocks, ksize, num_heatmaps)
        if num_context_per_joint > 0:
            hs, hc = h[..., :num_joints], h[..., num_joints:]
        else:
            hs = h
        ys = ops.softargmax2d(hs)
        if num_context_per_joint > 0:
            yc = ops.softargmax2d(hc)
            pc = ops.joints_probability(hc)
            y = ops.context_aggregation(ys, yc, pc, num_joints, num_context_per_joint, 0.8)
        else:
            y = ys
        p = ops.joints_probability(4 * hs)
        hmaps = ops.channel_softmax_2d(hs)
    elif pose_dim == 3:
        h = pose_regressor(W, xb1, num_blocks, ksize, depth_maps * num_joints)
        f, rows, cols, ch = h.shape
        assert ch == depth_maps * num_joints
        h5 = h.reshape(f, rows, cols, depth_maps, num_joints)
        hxy = h5.mean(dim=3)
        hz = h5.mean(dim=(1, 2))
        y = torch.cat([ops.softargmax2d(hxy), ops.softargmax1d(hz)], dim=-1)
        v = torch.amax(hxy, dim=(1, 2)) + torch.amax(hz, dim=1)
        p = torch.sigmoid(2 * v.unsqueeze(-1))
        hmaps = ops.channel_softmax_2d(hxy)
    else:
        raise ValueError('pose_dim must be 2 or 3')
    return y, p, ops.kronecker_prod(hmaps, xb1)


def merge_head(W, y, p, feat, num_actions, pose_net_version='v1', weighted_merge=True):
    """Clip-coupled stage (action.py:351-396): y [N,T,J,dim], p [N,T,J,1], feat [N,T,J,C] -> 9 score vectors."""
    out_pose = pose_model(W, y, p, num_actions, pose_net_version)
    out_vis = visual_model(W, feat, num_actions)
    outputs = [action_top(o) for o in out_pose] + [action_top(o) for o in out_vis]
    pm, vm = out_pose[-1], out_vis[-1]
    if weighted_merge:
        pm = heatmap_weighting(W, pm)
        vm = heatmap_weighting(W, vm)
    outputs.append(action_top(pm + vm))
    return outputs


def forward_merge(weights, clips, num_actions, num_joints, num_blocks, pose_dim=2, depth_maps=8,
                  num_context_per_joint=2, pose_net_version='v1', output_poses=False, weighted_merge=True,
                  ksize=(5, 5), dtype=torch.float32, taps=None):
    """action.build_merge_model(...) + predict (action.py:319-400).  clips: [N, T, H, W, 3].
    Returns outputs in model order: [y, p]? + p1..p4 + v1..v4 + m  (soft-maxed action scores)."""
    W = weights if isinstance(weights, Weights) else Weights(weights, dtype)
    W.reset()
    with torch.no_grad():
        clips = torch.from_numpy(np.ascontiguousarray(clips)).to(dtype)
        n, t = clips.shape[:2]
        frames = clips.reshape((n * t,) + tuple(clips.shape[2:]))
        y, p, feat = merge_frames(W, frames, num_joints, num_blocks, pose_dim, depth_maps, num_context_per_joint,
                                  ksize)
        J = num_joints
        y = y.reshape(n, t, J, pose_dim)
        p = p.reshape(n, t, J, 1)
        feat = feat.reshape(n, t, J, feat.shape[-1])
        outputs = [y, p] if output_poses else []
        if taps is not None:
            taps.update(y=y, p=p, f=feat)
        outputs += merge_head(W, y, p, feat, num_actions, pose_net_version, weighted_merge)
        if taps is not None:
            for k in list(taps):
                taps[k] = taps[k].numpy()
        return [o.numpy() for o in outputs]
