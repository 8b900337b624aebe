#!/usr/bin/env python
"""Where does SPNet's coordinate error against the fp64 arbiter enter: the conv stack or the decoder?

For every prediction block of an SPNet (heat-maps at 16x16, 8x8, 4x4 -- the coarse pyramid levels) this
compares, on the same synthetic weights and frames,
    fp32 CPU oracle (PyTorch-CPU)          vs  fp64 oracle
    HIP engine (only with a GPU present)   vs  fp64 oracle
and splits the coordinate error into
    upstream   = | decode64(logits_X) - decode64(logits_64) |   X's fp32 logits pushed through an fp64 decoder:
                 what the conv stack's fp32 rounding alone does to the coordinates
    decoder    = | decode_X(logits_X)  - decode64(logits_X) |   X's decoder vs an fp64 decoder on X's own logits
together with the logit errors themselves and the first-order sensitivity bound
    |dx| <= max|dlogit| * sum_i p_i |g_i - x|      (soft-argmax, d x / d logit_i = p_i (g_i - x)).
Writes JSON (default profiles/r02_spnet_noise.json when run from the repo root on a GPU box: gpurun_out/).

    python tools/spnet_noise_analysis.py [--frames 4] [--out path] [--no-hip]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def decode64(logits):
    """fp64 channel soft-max + (x, y) expectation of [F, h, w, J] logits (activations.py:3-16, layers.py:160-200)."""
    from oracle import ops
    t = torch.from_numpy(np.asarray(logits, dtype=np.float64))
    p = ops.channel_softmax_2d(t, 1.0)
    return ops.softargmax2d_from_prob(p).numpy(), p.numpy()


def sensitivity(p64, xy64):
    """max over (frame, joint) of sum_i p_i |g_i - x| for both axes (first-order error amplification)."""
    from oracle import ops
    h, w = p64.shape[1], p64.shape[2]
    gx = ops.linspace_2d(h, w, 0).astype(np.float64)[None, :, :, None]
    gy = ops.linspace_2d(h, w, 1).astype(np.float64)[None, :, :, None]
    sx = (p64 * np.abs(gx - xy64[:, None, None, :, 0])).sum(axis=(1, 2))
    sy = (p64 * np.abs(gy - xy64[:, None, None, :, 1])).sum(axis=(1, 2))
    return float(max(sx.max(), sy.max()))


def analyse(a, calibrate):
    from deephar_amd import graph, weights, utils, Model
    from deephar_amd.config import ModelConfig
    from deephar_amd.models import spnet
    from oracle import spnet as osp
    T = a.frames
    graph.reset_naming()
    lay = utils.pa17j3d
    cfg = ModelConfig((T, 256, 256, 3), lay, num_actions=[60], num_pyramids=2, action_pyramids=[1, 2], num_levels=4,
                      pose_replica=False, num_pose_features=192, num_visual_features=192)
    m = spnet.build(cfg)
    weights.init_synthetic(m, seed=0)
    wd = weights.as_dict(m)
    ocfg = dict(num_joints=lay.num_joints, dim=lay.dim, num_actions=[60], num_pyramids=2, action_pyramids=[1, 2],
                num_levels=4, kernel_size=(5, 5), growth=96, image_div=8, num_pose_features=192,
                num_visual_features=192, sam_alpha=1)
    x = np.random.default_rng(11).uniform(-1, 1, (1, T, 256, 256, 3)).astype(np.float32)
    if calibrate:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import paritylog
        paritylog.calibrate_spnet_heads(m, ocfg, x)
        wd = weights.as_dict(m)
    t32, t64 = {}, {}
    o32 = osp.forward(wd, x, ocfg, dtype=torch.float32, taps=t32)
    o64 = osp.forward(wd, x, ocfg, dtype=torch.float64, taps=t64)
    blocks = [k[:-len('/logits')] for k in t64 if k.endswith('/logits')]
    npose = len(blocks)

    hip_logits, hip_out = None, None
    if not a.no_hip and torch.cuda.is_available():
        # the heat-map logits are the outputs of the '<block>_heatmaps_conv1' convolutions: re-wrap the graph
        taps = []
        for b in blocks:
            node = [n for n in m._nodes if any(l.name == b + '_heatmaps_conv1' for l in n.layers.values())]
            assert len(node) == 1, b
            taps.append(node[0].outputs[0])
        side = Model(m.input, m.outputs[:npose] + taps, name='spnet_with_logits')
        res = side.predict(x, batch_size=1)
        hip_out = res[:npose]
        hip_logits = [r.reshape((-1,) + r.shape[-3:]) for r in res[npose:]]

    rows = []
    for k, b in enumerate(blocks):
        l64, l32 = t64[b + '/logits'], t32[b + '/logits']
        xy64, p64 = decode64(l64)
        ref = o64[k].reshape((-1,) + o64[k].shape[-2:])[..., :2]
        assert np.abs(ref - xy64).max() < 1e-12
        row = dict(block=b, map=list(l64.shape[1:3]), logit_std=float(l64.std()), logit_absmax=float(np.abs(l64).max()),
                   sensitivity=sensitivity(p64, xy64))
        for tag, lg, out in (('cpu32', l32, o32[k]), ('hip', hip_logits[k] if hip_logits else None,
                                                      hip_out[k] if hip_out else None)):
            if lg is None:
                continue
            xy = out.reshape((-1,) + out.shape[-2:])[..., :2].astype(np.float64)
            up, _ = decode64(lg)
            row[tag] = dict(
                logit_err_max=float(np.abs(lg - l64).max()),
                logit_err_rms=float(np.sqrt(np.mean((lg - l64) ** 2))),
                total_px=256 * float(np.abs(xy - xy64).max()),
                upstream_px=256 * float(np.abs(up - xy64).max()),
                decoder_px=256 * float(np.abs(xy - up).max()),
                first_order_bound_px=256 * float(np.abs(lg - l64).max()) * row['sensitivity'])
        rows.append(row)
        print(json.dumps(row))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--out', default=None)
    ap.add_argument('--no-hip', action='store_true')
    a = ap.parse_args()
    T = a.frames
    print('--- heads as init_synthetic leaves them')
    rows = analyse(a, False)
    print('--- heads calibrated to logit std 6 (tests/paritylog.py: calibrate_spnet_heads)')
    rows_cal = analyse(a, True)
    out = a.out or os.path.join(ROOT, 'gpurun_out' if torch.cuda.is_available() else 'profiles', 'r02_spnet_noise.json')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as fh:
        json.dump(dict(config='SPNet NTU-like (pa17j3d, 2 pyramids, actions on 1,2), T=%d frames, 256x256, seed 0/11' % T,
                       tolerance_px=1e-3, blocks=rows, blocks_calibrated=rows_cal), fh, indent=1)
    print('wrote', out)


if __name__ == '__main__':
    main()
