#!/usr/bin/env python
"""fp32-MFMA vs split-bf16 GEMM on the dominant conv shapes (batch 64); HIP events, median of 20, best tiling each."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib, functional as F
from deephar_amd.engine import packing
lib = _lib.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
SH = [(32, 576, 576, 1, True, True), (32, 576, 576, 1, False, True), (16, 288, 288, 1, True, True), (16, 288, 576, 1, False, True),
      (8, 288, 288, 1, True, True), (32, 576, 48, 1, True, False), (64, 64, 96, 3, False, False), (32, 384, 576, 1, False, True)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))
rng = np.random.default_rng(0); rows = []
for (h, cin, cout, ks, relu, res) in SH:
    n = 64
    x = torch.randn(n, h, h, cin, device=dev)
    w = (rng.standard_normal((ks, ks, cin, cout)) * 0.05).astype(np.float32)
    wf = F.pack_conv_weight(w, dev)
    pk, kp, np_ = packing.pack_conv_split(w); ws = torch.from_numpy(pk).to(dev)
    y = torch.empty(n, h, h, cout, device=dev); r1 = torch.randn(n, h, h, cout, device=dev) if res else None
    sc, sb = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    a = _lib.ConvArgs()
    a.x, a.y, a.post_scale, a.post_shift = x.data_ptr(), y.data_ptr(), sc.data_ptr(), sb.data_ptr()
    a.res1 = r1.data_ptr() if res else None
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, h, cin, cin, h, h, cout, cout
    a.KH = a.KW = ks; a.SH = a.SW = 1; a.PT = a.PL = (ks - 1) // 2
    a.K, a.Kp, a.Np, a.ldr1, a.pre_relu = ks * ks * cin, kp, np_, cout, int(relu)
    res_ = {}
    allc = {}
    for tag, wt, split, cfgs in (('f32', wf[0], 0, range(9, lib.dh_conv2d_num_tile_cfgs())), ('bf16x3', ws, 1, range(lib.dh_conv2d_num_split_tile_cfgs()))):
        a.w, a.w_split = wt.data_ptr(), split
        ts = {c: timed(lambda c=c: lib.dh_conv2d_f32(C.byref(a), c, st)) for c in cfgs
              if lib.dh_conv2d_f32(C.byref(a), c, st) == 0}
        allc[tag] = {str(c): round(v, 1) for c, v in ts.items()}
        best = min((v, c) for c, v in ts.items())
        res_[tag] = best
    flop = 2.0 * n * h * h * ks * ks * cin * cout
    row = dict(shape=[n, h, h, cin, cout, ks], relu=relu, res=res, f32_us=res_['f32'][0], f32_cfg=res_['f32'][1],
               split_us=res_['bf16x3'][0], split_cfg=res_['bf16x3'][1], speedup=res_['f32'][0] / res_['bf16x3'][0],
               f32_tflops=flop / res_['f32'][0] / 1e6, split_tflops_fp32_equiv=flop / res_['bf16x3'][0] / 1e6, split_all=allc['bf16x3'])
    rows.append(row); print(json.dumps(row))
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'bench_split.json'), 'w'), indent=1)
