#!/usr/bin/env python
"""Per-node cost of a replayed hipGraph chain on this box: 200 dependent tiny kernels (x += 1 on 256 floats) and 200
dependent 20-us kernels, captured on one stream, replayed 50 times.  (Is launch spacing a visible part of a 215-launch
forward?)"""
import torch
dev = torch.device('cuda:0')
s = torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, numel in (('tiny (1 KB)', 256), ('~20 us (64 MB r+w)', 16 << 20)):
    x = torch.zeros(numel, device=dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(200):
                x.add_(1.0)
        for _ in range(5):
            g.replay()
        s.synchronize()
        e0.record(s)
        for _ in range(50):
            g.replay()
        e1.record(s)
        e1.synchronize()
        per_node = e0.elapsed_time(e1) * 1e3 / (50 * 200)
        # the same kernel alone, back to back without a graph
        e0.record(s)
        for _ in range(2000):
            x.add_(1.0)
        e1.record(s)
        e1.synchronize()
        eager = e0.elapsed_time(e1) * 1e3 / 2000
    print('%-20s graph replay: %.2f us per node   eager stream launches: %.2f us per launch' % (name, per_node, eager))
