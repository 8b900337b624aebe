#!/usr/bin/env python
"""In-graph cost of ONE node of a chain of identical small layers (round 5, the small-batch regime of
exp/pennaction/eval_speed2d.py): a model that is nothing but `depth` copies of one layer is bound, captured and replayed;
replay time / depth = what such a launch costs when nothing else is in its way (instruction cache, kernel arguments and
translation all hot).  Compared with the same kernel's average inside the SPNet forward (rocprofv3) it separates the
kernel's own time from what the surrounding forward does to it.

    gpurun -- 'python tools/exp_node_cost.py > gpurun_out/r05_node_cost.txt'
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
from deephar_amd import Model, graph, weights   # noqa: E402
from deephar_amd import layers as L             # noqa: E402


def chain(shape, depth, make):
    graph.reset_naming()
    inp = L.Input(shape)
    x = inp
    for i in range(depth):
        x = make(x, i)
    m = Model(inp, [x])
    weights.init_synthetic(m, seed=0)
    return m


def per_node(m, n, reps=30):
    ex = m.executor
    bp = ex.bind(n)
    x = np.random.default_rng(0).uniform(-1, 1, (n,) + tuple(m.inputs[0].shape)).astype(np.float32)
    with torch.cuda.stream(ex.stream):
        ex.set_inputs(bp, [x])
        for _ in range(3):
            ex.forward(bp)
        ex.stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ex.stream)
        for _ in range(reps):
            ex.forward(bp)
        e1.record(ex.stream)
        e1.synchronize()
    kinds = {}
    for s in m.plan.steps:
        kinds[s.kind] = kinds.get(s.kind, 0) + 1
    return 1e3 * e0.elapsed_time(e1) / reps / len(m.plan.steps), kinds


CASES = [
    ('1x1 160->160 on [8,16,160] x 2 clips (skinny, 8 waves)', (8, 16, 160), 2, lambda x, i: L.conv2d(x, 160, (1, 1), name='c%d' % i)),
    ('3x3 160->160 on [8,8,160] x 2 clips, K=1440 (skinny, 16 waves)', (8, 8, 160), 2, lambda x, i: L.conv2d(x, 160, (3, 3), name='c%d' % i)),
    ('3x3 80->80 on [8,16,80] x 2 clips, K=720 (skinny, 16 waves)', (8, 16, 80), 2, lambda x, i: L.conv2d(x, 80, (3, 3), name='c%d' % i)),
    ('1x1 480->480 on [8,8,480] x 16 frames (gemm1x1, M=1024)', (8, 8, 480), 16, lambda x, i: L.conv2d(x, 480, (1, 1), name='c%d' % i)),
    ('1x1 384->384 on [16,16,384] x 16 frames (gemm1x1, M=4096)', (16, 16, 384), 16, lambda x, i: L.conv2d(x, 384, (1, 1), name='c%d' % i)),
    ('1x1 576->576 on [4,4,576] x 16 frames (gemm1x1, M=256)', (4, 4, 576), 16, lambda x, i: L.conv2d(x, 576, (1, 1), name='c%d' % i)),
    ('sepconv 5x5 480 on [8,8,480] x 16 frames (dw + pw)', (8, 8, 480), 16, lambda x, i: L.sepconv2d(x, 480, (5, 5), name='s%d' % i)),
    ('maxpool 3x3 s1 same on [8,8,160] x 2 clips', (8, 8, 160), 2, lambda x, i: L.MaxPooling2D(x, (3, 3), strides=(1, 1), padding='same')),
]

if __name__ == '__main__':
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for what, shape, n, make in CASES:
        try:
            t0 = time.time()
            us, kinds = per_node(chain(shape, depth, make), n)
            print('%-72s %6.2f us / node   %s  (%.0f s)' % (what, us, kinds, time.time() - t0), flush=True)
        except Exception as e:      # a layer signature this probe got wrong must not hide the other rows
            print('%-72s failed: %r' % (what, e), flush=True)
