#!/usr/bin/env python
"""Every tiling of dh_conv2d_f32 on the pointwise GEMMs of SPNet's pose stream at 16 frames per call (the latency regime of
exp/pennaction/eval_speed2d.py), timed as nodes of a replayed hipGraph (50 dependent launches per graph: host launch cost is
out of the picture).  `python tools/bench_small_gemm.py [frames]` -> one line per shape: us per node for each tiling."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib                      # noqa: E402
from deephar_amd.engine import packing            # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
stream = torch.cuda.Stream()
st = stream.cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# (map side, Cin, Cout, ReLU on load, residual)
SHAPES = [(16, 384, 384, 0, 1), (16, 480, 384, 1, 0), (8, 480, 480, 0, 1), (8, 384, 480, 1, 0), (4, 576, 576, 0, 1),
          (4, 480, 576, 1, 0), (32, 288, 288, 0, 1), (16, 384, 16, 1, 0), (8, 16, 480, 1, 1)]
rng = np.random.default_rng(0)
ncfg = lib.dh_conv2d_num_tile_cfgs()
NODES, REPS = 50, 20
for h, cin, cout, relu, res in SHAPES:
    x = torch.randn(n, h, h, cin, device=dev)
    y = torch.empty(n, h, h, cout, device=dev)
    r1 = torch.randn(n, h, h, cout, device=dev)
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    pk, kp, np_ = packing.pack_conv(w)
    wd = torch.from_numpy(pk).to(dev)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    a = _lib.ConvArgs()
    a.x, a.w, a.y, a.post_scale, a.post_shift = x.data_ptr(), wd.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
    if res:
        a.res1, a.ldr1 = r1.data_ptr(), cout
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, h, cin, cin, h, h, cout, cout
    a.KH = a.KW = a.SH = a.SW = 1
    a.K, a.Kp, a.Np, a.pre_relu = cin, kp, np_, relu
    out, ref = {}, None
    with torch.cuda.stream(stream):
        for cfg in range(0, ncfg):
            if lib.dh_conv2d_f32(C.byref(a), cfg, st) != 0:
                continue
            stream.synchronize()
            got = y.clone()
            ref = got if ref is None else ref
            assert torch.equal(got, ref), 'tiling %d differs' % cfg
            assert lib.dh_graph_begin_capture(st) == 0
            for _ in range(NODES):
                lib.dh_conv2d_f32(C.byref(a), cfg, st)
            g = C.c_void_p()
            assert lib.dh_graph_end_capture(st, C.byref(g)) == 0
            for _ in range(3):
                lib.dh_graph_launch(g, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(REPS):
                lib.dh_graph_launch(g, st)
            e1.record(stream)
            e1.synchronize()
            out[cfg] = e0.elapsed_time(e1) * 1e3 / (REPS * NODES)
            lib.dh_graph_destroy(g)
    fl = 2.0 * n * h * h * cin * cout
    best = min(out, key=out.get)
    print('%2dx%-2d %3d->%-3d relu=%d res=%d M=%5d: at the fp32 peak %4.1f us | best cfg %2d %5.1f us | %s' % (
        h, h, cin, cout, relu, res, n * h * h, fl / 157.3e6, best, out[best],
        ' '.join('%d:%.1f' % kv for kv in sorted(out.items()) if kv[0] >= 9)), flush=True)
