#!/usr/bin/env python
"""What a streaming read+write kernel reaches on this box (the ceiling the depthwise / pooling kernels are measured
against): torch copy_ and add of 151 MB fp32 tensors (the 64 x 32 x 32 x 576 activation), median of 30."""
import numpy as np, torch
dev = torch.device('cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn):
    for _ in range(20): fn()
    ts = []
    for _ in range(30):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))
n = 64 * 32 * 32 * 576
a, b, c = torch.randn(n, device=dev), torch.empty(n, device=dev), torch.randn(n, device=dev)
t = timed(lambda: b.copy_(a)); print('copy  (1 read + 1 write): %.1f us  %.2f TB/s' % (t, 2 * n * 4 / t / 1e6))
t = timed(lambda: torch.add(a, c, out=b)); print('add   (2 reads + 1 write): %.1f us  %.2f TB/s' % (t, 3 * n * 4 / t / 1e6))
t = timed(lambda: a.sum()); print('sum   (1 read): %.1f us  %.2f TB/s' % (t, n * 4 / t / 1e6))
