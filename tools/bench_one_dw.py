#!/usr/bin/env python
"""One depthwise shape, a few launches (for rocprofv3 --pmc passes): python tools/bench_one_dw.py [frames side channels ks reps]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib
lib = _lib.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
n, h, c, ks, reps = (int(v) for v in (sys.argv[1:6] + ['64', '32', '576', '5', '4'][len(sys.argv) - 1:]))
torch.manual_seed(0)
x = torch.randn(n, h, h, c, device=dev); y = torch.empty_like(x); w = torch.randn(ks * ks, c, device=dev)
a = _lib.DwArgs()
a.x, a.w, a.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
a.N, a.H, a.W, a.C, a.ldx, a.ldy = n, h, h, c, c, c
a.KH = a.KW = ks; a.PT = a.PL = (ks - 1) // 2; a.pre_relu = 1
for _ in range(reps):
    assert lib.dh_dwconv2d_f32(C.byref(a), st) == 0
torch.cuda.synchronize()
