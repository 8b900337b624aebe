#!/usr/bin/env python
"""Depthwise conv shapes of the models, HIP events around 20 back-to-back launches, median of 5 (A/B helper; a second
library via DEEPHAR_HIP_LIB).  Prints us per launch and the algorithmic TB/s (8 B per element)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib
lib = _lib.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
# (frames, map side, channels, kernel, ReLU on load, BN prologue)
SHAPES = [(64, 32, 576, 5, 1, 0), (64, 16, 288, 5, 1, 0), (64, 8, 288, 5, 1, 0), (64, 32, 384, 5, 1, 0), (64, 64, 128, 5, 1, 0),
          (64, 32, 576, 3, 1, 1), (256, 32, 288, 3, 1, 1), (256, 16, 576, 3, 1, 1), (64, 32, 576, 5, 0, 0), (64, 32, 576, 5, 1, 0)]
for n, h, c, ks, relu, aff in SHAPES:
    torch.manual_seed(n + h + c + ks)
    x = torch.randn(n, h, h, c, device=dev); y = torch.empty_like(x)
    w = torch.randn(ks * ks, c, device=dev)
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    a = _lib.DwArgs()
    a.x, a.w, a.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    if aff: a.pre_scale, a.pre_shift = sc.data_ptr(), sh.data_ptr()
    a.N, a.H, a.W, a.C, a.ldx, a.ldy = n, h, h, c, c, c
    a.KH = a.KW = ks; a.PT = a.PL = (ks - 1) // 2; a.pre_relu = relu
    assert lib.dh_dwconv2d_f32(C.byref(a), st) == 0
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.dh_dwconv2d_f32(C.byref(a), st)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    t = float(np.median(ts))
    print('dw %dx%d n=%d %dx%dx%d relu=%d bn=%d: %.1f us  %.2f TB/s  checksum %.6e' % (
        ks, ks, n, h, h, c, relu, aff, t, 8.0 * n * h * h * c / t / 1e6, float(y.double().sum())), flush=True)
