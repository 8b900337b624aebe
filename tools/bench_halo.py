#!/usr/bin/env python
"""Halo-resident K x K kernel (conv_halo.hip, w_split = 2) against the best tap-major tiling (cfg 0..17) on the dense
K x K layers of the two workloads, batch as in bench.py.  HIP events, median of `reps`.
    python tools/bench_halo.py [--reps 10]"""
import argparse, ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib, functional as F
from deephar_amd.engine import packing

SHAPES = [  # tag, N, H, W, Cin, Cout, KH, KW, relu
    ('mpii stem 32->64 @128', 64, 128, 128, 32, 64, 3, 3, 0), ('mpii stem 32->32 @128', 64, 128, 128, 32, 32, 3, 3, 0),
    ('mpii stem 64->96 @64', 64, 64, 64, 64, 96, 3, 3, 0), ('mpii stem 64->64 5x1 @64', 64, 64, 64, 64, 64, 5, 1, 0),
    ('mpii stem 64->64 1x5 @64', 64, 64, 64, 64, 64, 1, 5, 0),
    ('spnet res0 48->96 @128', 256, 128, 128, 48, 96, 3, 3, 1), ('spnet res1/2 96->192 @64', 256, 64, 64, 96, 192, 3, 3, 1),
    ('spnet res3/4 144->288 @32', 256, 32, 32, 144, 288, 3, 3, 1),
]

def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--reps', type=int, default=10); ap.add_argument('--out', default=None)
    args = ap.parse_args()
    lib = _lib.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(0); rows = []
    for tag, n, h, w, cin, cout, kh, kw, relu in SHAPES:
        x = torch.randn(n, h, w, cin, device=dev); y = torch.empty(n, h, w, cout, device=dev)
        wk = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
        sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev) * 0.1
        pk = {0: packing.pack_conv(wk), 2: packing.pack_conv_halo(wk)}
        wd = {k: torch.from_numpy(v[0]).to(dev) for k, v in pk.items()}
        def mk(layout):
            a = _lib.ConvArgs()
            a.x, a.w, a.y = x.data_ptr(), wd[layout].data_ptr(), y.data_ptr()
            a.post_scale, a.post_shift = sc.data_ptr(), sh.data_ptr()
            a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, w, cin, cin, h, w, cout, cout
            a.KH, a.KW, a.SH, a.SW, a.PT, a.PL = kh, kw, 1, 1, (kh - 1) // 2, (kw - 1) // 2
            a.K, a.Kp, a.Np = kh * kw * cin, pk[layout][1], pk[layout][2]
            a.pre_relu, a.post_relu, a.w_split = relu, 1, layout
            return a
        def timed(a, cfg):
            if lib.dh_conv2d_f32(C.byref(a), cfg, st) != 0: return None
            torch.cuda.synchronize(); ts = []
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); lib.dh_conv2d_f32(C.byref(a), cfg, st); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return float(np.median(ts))
        gf = 2.0 * n * h * w * kh * kw * cin * cout / 1e9
        tap = {c: timed(mk(0), c) for c in range(lib.dh_conv2d_num_tile_cfgs())}
        tap = {c: t for c, t in tap.items() if t}
        halo = {c: timed(mk(2), c) for c in range(lib.dh_conv2d_num_halo_tile_cfgs())}
        halo = {c: t for c, t in halo.items() if t}
        bt = min(tap, key=tap.get); bh = min(halo, key=halo.get) if halo else None
        rows.append(dict(shape=tag, gflop=round(gf, 2), tap_cfg=bt, tap_us=round(tap[bt], 1), tap_tflops=round(gf / tap[bt] * 1e3, 1),
                         halo_us={c: round(t, 1) for c, t in halo.items()},
                         halo_tflops=round(gf / halo[bh] * 1e3, 1) if halo else None,
                         speedup=round(tap[bt] / halo[bh], 3) if halo else None))
        print(json.dumps(rows[-1]))
    if args.out:
        json.dump(rows, open(args.out, 'w'), indent=1)

if __name__ == '__main__':
    main()
