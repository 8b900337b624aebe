#!/usr/bin/env python
"""Pointwise GEMM shapes of the hourglass on every LDS-DMA tiling (cfg 9..), HIP events around 20 back-to-back launches,
median of 5 (A/B helper).  `python tools/bench_gemm_shapes.py [frames]`."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib
from deephar_amd.engine import packing
lib = _lib.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
# (map side, Cin, Cout, ReLU on load)
SHAPES = [(32, 576, 576, 0), (16, 288, 288, 0), (16, 288, 576, 0), (16, 288, 576, 1), (16, 576, 288, 1), (8, 288, 288, 0),
          (32, 48, 576, 1), (32, 576, 48, 0)]
rng = np.random.default_rng(0)
ncfg = lib.dh_conv2d_num_tile_cfgs()
for h, cin, cout, relu in SHAPES:
    x = torch.randn(n, h, h, cin, device=dev); y = torch.empty(n, h, h, cout, device=dev); r1 = torch.randn(n, h, h, cout, device=dev)
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    pk, kp, np_ = packing.pack_conv(w); wd = torch.from_numpy(pk).to(dev)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    a = _lib.ConvArgs()
    a.x, a.w, a.y, a.res1, a.post_scale, a.post_shift = x.data_ptr(), wd.data_ptr(), y.data_ptr(), r1.data_ptr(), sc.data_ptr(), sh.data_ptr()
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy, a.ldr1 = n, h, h, cin, cin, h, h, cout, cout, cout
    a.KH = a.KW = a.SH = a.SW = 1; a.K, a.Kp, a.Np = cin, kp, np_; a.pre_relu = relu
    out = {}
    for cfg in range(9, ncfg):
        if lib.dh_conv2d_f32(C.byref(a), cfg, st) != 0: continue
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): lib.dh_conv2d_f32(C.byref(a), cfg, st)
            e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        out[cfg] = float(np.median(ts))
    fl = 2.0 * n * h * h * cin * cout
    best2 = min((c for c in out if c < 18), key=out.get); best = min(out, key=out.get)
    print('%dx%d %d->%d relu=%d n=%d: ideal %.1f us | best 2-stage cfg %d %.1f us (%.0f TF) | best cfg %d %.1f us (%.0f TF) | %s' % (
        h, h, cin, cout, relu, n, fl / 157.3e6, best2, out[best2], fl / out[best2] / 1e6, best, out[best], fl / out[best] / 1e6,
        ' '.join('%d:%.1f' % kv for kv in out.items())), flush=True)
