#!/usr/bin/env python
"""Pointwise GEMM shapes of the hourglass on chosen tilings, HIP events, median of reps (A/B helper)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib
from deephar_amd.engine import packing
lib = _lib.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
SHAPES = [(32, 576, 576, (11, 12)), (16, 288, 576, (11, 12, 13)), (16, 288, 288, (11, 13, 14)), (16, 576, 288, (11, 13)),
          (8, 288, 288, (13, 15, 16, 17))]
rng = np.random.default_rng(0)
for h, cin, cout, cfgs in SHAPES:
    n = 64
    x = torch.randn(n, h, h, cin, device=dev); y = torch.empty(n, h, h, cout, device=dev); r1 = torch.randn(n, h, h, cout, device=dev)
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    pk, kp, np_ = packing.pack_conv(w); wd = torch.from_numpy(pk).to(dev)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    a = _lib.ConvArgs()
    a.x, a.w, a.y, a.res1, a.post_scale, a.post_shift = x.data_ptr(), wd.data_ptr(), y.data_ptr(), r1.data_ptr(), sc.data_ptr(), sh.data_ptr()
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy, a.ldr1 = n, h, h, cin, cin, h, h, cout, cout, cout
    a.KH = a.KW = a.SH = a.SW = 1; a.K, a.Kp, a.Np = cin, kp, np_
    out = []
    for cfg in cfgs:
        if lib.dh_conv2d_f32(C.byref(a), cfg, st) != 0: continue
        torch.cuda.synchronize(); ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.dh_conv2d_f32(C.byref(a), cfg, st); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        out.append('cfg %d: %.1f us (%.0f TF)' % (cfg, np.median(ts), 2.0 * n * h * h * cin * cout / np.median(ts) / 1e6))
    print('prio', os.environ.get('DEEPHAR_GEMM_PRIO', '0'), (h, cin, cout), ' | '.join(out))
