"""Experiment: does running the hourglass low-res path concurrently with the full-res branch pay?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import functional as F
dev = torch.device('cuda:0'); N = 64
rng = np.random.default_rng(0)
def mk(H, cin, cout, k=1):
    x = torch.randn(N, H, H, cin, device=dev)
    w = (rng.standard_normal((k, k, cin, cout)) * 0.05).astype(np.float32)
    return x, w, F.pack_conv_weight(w, dev)
big = mk(32, 576, 576)
xdw = torch.randn(N, 32, 32, 576, device=dev); dw5 = (rng.standard_normal((5,5,576,1))*0.2).astype(np.float32)
s16 = mk(16, 288, 288); s8 = mk(8, 288, 288)
x16 = torch.randn(N,16,16,288,device=dev); dw16=(rng.standard_normal((5,5,288,1))*0.2).astype(np.float32)
x8 = torch.randn(N,8,8,288,device=dev)
def branch_a():
    F.dwconv2d(xdw, dw5, pre_relu=True)
    F.conv2d(big[0], big[1], packed=big[2], tile_cfg=2)
def branch_b():
    for _ in range(3):
        F.dwconv2d(x16, dw16, pre_relu=True); F.conv2d(s16[0], s16[1], packed=s16[2], tile_cfg=4)
    for _ in range(3):
        F.dwconv2d(x8, dw16, pre_relu=True); F.conv2d(s8[0], s8[1], packed=s8[2], tile_cfg=4)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def serial():
    branch_a(); branch_b()
def conc():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(sa):
        sa.wait_event(ev); branch_a(); ea = torch.cuda.Event(); ea.record()
    with torch.cuda.stream(sb):
        sb.wait_event(ev); branch_b(); eb = torch.cuda.Event(); eb.record()
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print('branch a alone  %.1f us' % timeit(branch_a))
print('branch b alone  %.1f us' % timeit(branch_b))
print('serial          %.1f us' % timeit(serial))
print('two streams     %.1f us' % timeit(conc))
