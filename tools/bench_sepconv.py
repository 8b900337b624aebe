#!/usr/bin/env python
"""Fused SeparableConv2D (dh_sepconv2d_f32) vs the two-launch pair (dh_dwconv2d_f32 + dh_conv2d_f32) on the
separable-conv shapes of the models at a given batch; HIP events on the launch stream, median of `reps`.

    python tools/bench_sepconv.py [--batch 64] [--reps 20]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # H, W, Cin, Cout, K, res1, up2 -- MPII ReceptionNet (8 blocks) counts per forward in the comment
    (32, 32, 576, 576, 5, True, False),    # x8 (+7 without residual, +1 with post-ReLU)
    (32, 32, 384, 576, 3, True, False),    # x1 (stem)
    (16, 16, 288, 288, 5, True, False),    # x16
    (16, 16, 288, 576, 5, True, True),     # x8
    (8, 8, 288, 288, 5, True, False),      # x16
    (8, 8, 288, 288, 5, True, True),       # x8
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'bench_sepconv.json'))
    a = ap.parse_args()
    from deephar_amd import _lib, functional as F
    from deephar_amd.layers import same_pad
    lib = _lib.load()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(a.reps):
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(ts))

    rows = []
    for (h, w, cin, cout, ks, res, up2) in SHAPES:
        n = a.batch
        x = torch.from_numpy(rng.standard_normal((n, h, w, cin)).astype(np.float32)).to(dev)
        dw = rng.standard_normal((ks, ks, cin, 1)).astype(np.float32) * 0.2
        pw = rng.standard_normal((1, 1, cin, cout)).astype(np.float32) * 0.05
        dwt = torch.from_numpy(np.ascontiguousarray(dw.reshape(ks * ks, cin))).to(dev)
        wt, kp, np_ = F.pack_conv_weight(pw, dev)
        sc = torch.ones(cout, device=dev)
        sh = torch.zeros(cout, device=dev)
        r1 = torch.randn((n, h, w, cout), device=dev) if res else None
        up = 2 if up2 else 1
        r2 = torch.randn((n, h * up, w * up, cout), device=dev) if up2 else None
        mid = torch.empty_like(x)
        y = torch.empty((n, h * up, w * up, cout), device=dev)
        pad = same_pad(h, ks, 1)[0]
        da = _lib.DwArgs()
        da.x, da.w, da.y = x.data_ptr(), dwt.data_ptr(), mid.data_ptr()
        da.N, da.H, da.W, da.C, da.ldx, da.ldy = n, h, w, cin, cin, cin
        da.KH, da.KW, da.PT, da.PL, da.pre_relu = ks, ks, pad, pad, 1
        s = _lib.SepConvArgs()
        for args, src in ((s.pw, x), ):
            args.x, args.w, args.y = src.data_ptr(), wt.data_ptr(), y.data_ptr()
            args.post_scale, args.post_shift = sc.data_ptr(), sh.data_ptr()
            args.res1 = r1.data_ptr() if r1 is not None else None
            args.res2 = r2.data_ptr() if r2 is not None else None
            args.N, args.H, args.W, args.Cin, args.ldx = n, h, w, cin, cin
            args.OH, args.OW, args.Cout, args.ldy = h, w, cout, cout
            args.KH = args.KW = args.SH = args.SW = 1
            args.K, args.Kp, args.Np = cin, kp, np_
            args.ldr1, args.ldr2 = cout, cout
            args.pre_relu, args.up2 = 1, int(up2)
        s.dw_w, s.DKH, s.DKW, s.DPT, s.DPL = dwt.data_ptr(), ks, ks, pad, pad
        ca = _lib.ConvArgs()
        C.memmove(C.byref(ca), C.byref(s.pw), C.sizeof(ca))
        ca.x, ca.pre_relu = mid.data_ptr(), 0
        t_dw = timed(lambda: lib.dh_dwconv2d_f32(C.byref(da), st))
        best_pw = min((timed(lambda c=c: lib.dh_conv2d_f32(C.byref(ca), c, st)), c)
                      for c in range(9, lib.dh_conv2d_num_tile_cfgs()) if lib.dh_conv2d_f32(C.byref(ca), c, st) == 0)
        fused = {}
        for c in range(lib.dh_sepconv2d_num_tile_cfgs()):
            if lib.dh_sepconv2d_f32(C.byref(s), c, st) == 0:
                fused[c] = timed(lambda c=c: lib.dh_sepconv2d_f32(C.byref(s), c, st))
        flop = 2.0 * n * h * w * cin * cout
        bf = min(fused.values()) if fused else None
        row = dict(shape=[n, h, w, cin, cout, ks], res=res, up2=up2, dw_us=t_dw, pw_us=best_pw[0], pw_cfg=best_pw[1],
                   pair_us=t_dw + best_pw[0], fused_us=fused, fused_best_us=bf,
                   fused_tflops=flop / bf / 1e6 if bf else None, pair_tflops=flop / (t_dw + best_pw[0]) / 1e6)
        rows.append(row)
        print(json.dumps(row))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as fh:
        json.dump(rows, fh, indent=1)


if __name__ == '__main__':
    main()
