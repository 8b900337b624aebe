#!/usr/bin/env python
"""Throughput of the other BASELINE.json configurations (parity cases, not the headline bench line): hipGraph replay
with inputs resident in HBM, same timing rules as bench.py.  python tools/bench_configs.py [--steps K]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import graph, weights, utils
from deephar_amd.config import ModelConfig
from deephar_amd.models import reception, action, spnet

ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=10); args = ap.parse_args()


def h36m():
    m = reception.build((256, 256, 3), 17, dim=3, num_blocks=8, depth_maps=16, ksize=(5, 5))
    return m, (128, 256, 256, 3), 128, 'cfg2 H36M 3-D pose, 8 blocks, batch 128', 'frames'


def penn():
    pe = reception.build((256, 256, 3), 16, dim=2, num_blocks=4, num_context_per_joint=2, ksize=(5, 5))
    m = action.build_merge_model(pe, 15, (256, 256, 3), 16, 16, 4, pose_dim=2, pose_net_version='v1')
    return m, (4, 16, 256, 256, 3), 64, 'cfg3 PennAction merge model, 16-frame clips, batch 4 clips', 'frames'


def ntu():
    cfg = ModelConfig((32, 256, 256, 3), utils.pa17j3d, num_actions=[60], num_pyramids=2, action_pyramids=[1, 2],
                      num_levels=4, pose_replica=False, num_pose_features=192, num_visual_features=192)
    return spnet.build(cfg), (8, 32, 256, 256, 3), 256, 'cfg4 NTU SPNet, 32-frame clips, batch 8 clips', 'frames'


for make in (h36m, penn, ntu):
    graph.reset_naming()
    m, shape, frames, name, unit = make()
    weights.init_synthetic(m, seed=0)
    ex = m.executor
    bp = ex.bind(shape[0])
    x = np.random.default_rng(0).uniform(-1, 1, shape).astype(np.float32)
    with torch.cuda.stream(ex.stream):
        x_dev = torch.from_numpy(x).to(ex.device)
    def step():
        with torch.cuda.stream(ex.stream):
            bp.tensor(m.plan.inputs[0]).copy_(x_dev, non_blocking=True)
            ex.forward(bp)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    fl = m.plan.total_flops(shape[0])
    print('%-62s %7.1f ms/step  %8.1f %s/s  %5.1f %% of fp32 MFMA peak  (%d launches)' %
          (name, dt * 1e3, frames / dt, unit, 100 * fl / dt / 157.3e12, len(m.plan.steps)))
