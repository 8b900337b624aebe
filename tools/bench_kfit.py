#!/usr/bin/env python
"""Fixed cost vs per-K-step cost of the small (16 x 16 map) GEMMs: time the same M x N for K = 32 ... 1152 and fit a line.
Round 2, batch 64, 16 x 16 x (K -> 288): split-bf16 1.85 us per 32 k (matrix floor 0.96 us at the sustained bf16 rate)
+ 10 us fixed; fp32 2.7 us per 32 k (floor 1.96) + 10 us fixed, of which ~6 us is the launch + event overhead of the
measurement itself.  The per-K-step excess is tile quantisation: 1536 wave tiles of 32 x 96 on 1024 SIMDs."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib, functional as F
from deephar_amd.engine import packing
lib = _lib.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn):
    for _ in range(50): fn()
    ts = []
    for _ in range(30):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))
rng = np.random.default_rng(0)
n, h, cout = 64, 16, 288
for split, cfgs in ((1, (9, 2, 5, 3)), (0, (11, 14, 12))):
    for cfg in cfgs:
        row = []
        for cin in (32, 64, 128, 288, 576, 1152):
            x = torch.randn(n, h, h, cin, device=dev)
            w = (rng.standard_normal((1, 1, cin, cout)) * 0.05).astype(np.float32)
            if split:
                pk, kp, np_ = packing.pack_conv_split(w); wt = torch.from_numpy(pk).to(dev)
            else:
                wt, kp, np_ = F.pack_conv_weight(w, dev)
            y = torch.empty(n, h, h, cout, device=dev); r1 = torch.randn(n, h, h, cout, device=dev)
            sc, sb = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
            a = _lib.ConvArgs()
            a.x, a.y, a.post_scale, a.post_shift, a.res1 = x.data_ptr(), y.data_ptr(), sc.data_ptr(), sb.data_ptr(), r1.data_ptr()
            a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, h, cin, cin, h, h, cout, cout
            a.KH = a.KW = 1; a.SH = a.SW = 1; a.PT = a.PL = 0
            a.K, a.Kp, a.Np, a.ldr1, a.pre_relu = cin, kp, np_, cout, 0
            a.w, a.w_split = wt.data_ptr(), split
            row.append((cin, round(timed(lambda: lib.dh_conv2d_f32(C.byref(a), cfg, st)), 1)))
        ks = np.array([r[0] for r in row[2:]], float); ts = np.array([r[1] for r in row[2:]])
        slope, icpt = np.polyfit(ks, ts, 1)
        print('split' if split else 'f32  ', 'cfg', cfg, row, ' -> %.2f us per 32 k, intercept %.1f us' % (slope * 32, icpt))
# an empty kernel launch for scale
x = torch.zeros(16, device=dev)
print('torch tiny op', round(timed(lambda: x.add_(1.0)), 1), 'us')
