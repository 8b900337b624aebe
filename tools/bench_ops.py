#!/usr/bin/env python
"""Micro-benchmark of the hot kernels at the MPII-config shapes (batch 64) with HIP events on the launch
stream: TFLOP/s for the MFMA conv, GB/s for the HBM-bound kernels.  Usage: python tools/bench_ops.py [--cfgs]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib, functional as F  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    dev = torch.device('cuda:0')
    N = 64
    rng = np.random.default_rng(0)
    shapes = [  # (H, Cin, Cout, k, stride, res, up2, name)
        (32, 576, 576, 1, 1, 1, 0, 'pw 576->576 @32 +res'),
        (32, 576, 576, 1, 1, 0, 0, 'pw 576->576 @32'),
        (32, 576, 48, 1, 1, 0, 0, 'RegMap 576->48'),
        (32, 48, 576, 1, 1, 2, 0, 'fReMap 48->576 +2res'),
        (16, 288, 288, 1, 1, 1, 0, 'pw 288->288 @16 +res'),
        (16, 576, 288, 1, 1, 0, 0, 'pw 576->288 @16'),
        (16, 288, 576, 1, 1, 1, 1, 'pw 288->576 @16 up2'),
        (8, 288, 288, 1, 1, 1, 0, 'pw 288->288 @8 +res'),
        (8, 288, 288, 1, 1, 1, 1, 'pw 288->288 @8 up2'),
        (64, 192, 192, 3, 2, 0, 0, 'stem 3x3 s2 192->192'),
        (128, 32, 64, 3, 1, 0, 0, 'stem 3x3 32->64 @128'),
        (256, 3, 32, 3, 2, 0, 0, 'stem 3x3 s2 3->32'),
        (64, 64, 96, 3, 1, 0, 0, 'stem 3x3 64->96 @64'),
    ]
    cfgs = [-1] + (list(range(20)) if '--cfgs' in sys.argv else [])
    for (H, cin, cout, k, s, nres, up2, name) in shapes:
        x = torch.from_numpy(rng.standard_normal((N, H, H, cin)).astype(np.float32)).to(dev)
        w = (rng.standard_normal((k, k, cin, cout)) * 0.05).astype(np.float32)
        packed = F.pack_conv_weight(w, dev)
        oh = -(-H // s)
        qs = torch.ones(cout, device=dev)
        qb = torch.zeros(cout, device=dev)
        r1 = torch.randn(N, oh, oh, cout, device=dev) if nres >= 1 else None
        r2 = torch.randn(N, oh * (2 if up2 else 1), oh * (2 if up2 else 1), cout, device=dev) if (nres >= 2 or up2) else None
        flops = 2.0 * N * oh * oh * k * k * cin * cout
        for cfg in cfgs:
            try:
                t = timeit(lambda: F.conv2d(x, w, (s, s), 'same', pre_relu=True, post_scale=qs, post_shift=qb,
                                            res1=r1, res2=r2, up2=bool(up2), tile_cfg=cfg, packed=packed))
            except Exception as e:  # unsupported cfg
                print('%-28s cfg %2d: %s' % (name, cfg, str(e)[:60]))
                continue
            byt = 4.0 * N * (H * H * cin + oh * oh * cout * (4 if up2 else 1) * (1 + (1 if r2 is not None else 0)) +
                             (oh * oh * cout if r1 is not None else 0))
            print('%-28s cfg %2d: %8.1f us  %6.1f TFLOP/s  %7.0f GB/s' % (name, cfg, t * 1e6, flops / t / 1e12,
                                                                         byt / t / 1e9))
    for (H, c, k) in [(32, 576, 5), (16, 288, 5), (8, 288, 5), (32, 384, 3)]:
        x = torch.randn(N, H, H, c, device=dev)
        dw = (rng.standard_normal((k, k, c, 1)) * 0.2).astype(np.float32)
        t = timeit(lambda: F.dwconv2d(x, dw, pre_relu=True))
        print('dw %dx%d C=%d @%d: %8.1f us  %7.0f GB/s' % (k, k, c, H, t * 1e6, 8.0 * N * H * H * c / t / 1e9))
    x = torch.randn(N, 32, 32, 576, device=dev)
    t = timeit(lambda: F.pool2d(x))
    print('pool 2x2 576 @32: %8.1f us %7.0f GB/s' % (t * 1e6, 5.0 * N * 32 * 32 * 576 / t / 1e9))
    h = torch.randn(N, 32, 32, 48, device=dev)
    t = timeit(lambda: F.softargmax2d(h))
    print('softargmax2d 48 maps: %8.1f us' % (t * 1e6))


if __name__ == '__main__':
    main()
