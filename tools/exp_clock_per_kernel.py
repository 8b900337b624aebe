#!/usr/bin/env python
"""Which kernels make the firmware lower the engine clock?  Each kernel shape of the MPII forward is launched in a loop
for ~1.5 s while rocm-smi is sampled (clock, socket power); prints the median sclk / W per kernel.  A second library
via DEEPHAR_HIP_LIB.  (profiles/r03_dvfs_study.md)"""
import ctypes as C, os, re, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib
from deephar_amd.engine import packing
lib = _lib.load(); dev = torch.device('cuda:0'); st = torch.cuda.current_stream().cuda_stream
n = 64
rng = np.random.default_rng(0)


def sampler(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--csv'], capture_output=True, text=True, timeout=5).stdout
            line = [l for l in txt.splitlines() if l.startswith('card0')][0]
            mhz = re.findall(r'\((\d+)Mhz\)', line)
            out.append((int(mhz[2]), float(line.strip().split(',')[-1] or 0)))
        except Exception:
            pass
        time.sleep(0.15)


def conv_case(h, cin, cout, relu, ks=1):
    x = torch.randn(n, h, h, cin, device=dev); y = torch.empty(n, h, h, cout, device=dev); r1 = torch.randn(n, h, h, cout, device=dev)
    w = (rng.standard_normal((ks, ks, cin, cout)) / np.sqrt(ks * ks * cin)).astype(np.float32)
    pk, kp, np_ = packing.pack_conv(w); wd = torch.from_numpy(pk).to(dev)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    a = _lib.ConvArgs()
    a.x, a.w, a.y, a.res1, a.post_scale, a.post_shift = x.data_ptr(), wd.data_ptr(), y.data_ptr(), r1.data_ptr(), sc.data_ptr(), sh.data_ptr()
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy, a.ldr1 = n, h, h, cin, cin, h, h, cout, cout, cout
    a.KH = a.KW = ks; a.SH = a.SW = 1; a.PT = a.PL = (ks - 1) // 2; a.K, a.Kp, a.Np = ks * ks * cin, kp, np_; a.pre_relu = relu
    keep = (x, y, r1, wd, sc, sh)
    return (lambda: lib.dh_conv2d_f32(C.byref(a), -1, st)), keep


def dw_case(h, c, ks=5):
    x = torch.randn(n, h, h, c, device=dev); y = torch.empty_like(x); w = torch.randn(ks * ks, c, device=dev)
    a = _lib.DwArgs()
    a.x, a.w, a.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    a.N, a.H, a.W, a.C, a.ldx, a.ldy = n, h, h, c, c, c
    a.KH = a.KW = ks; a.PT = a.PL = (ks - 1) // 2; a.pre_relu = 1
    return (lambda: lib.dh_dwconv2d_f32(C.byref(a), st)), (x, y, w)


def copy_case(mb):
    x = torch.randn(mb * 1024 * 1024 // 4, device=dev); y = torch.empty_like(x)
    return (lambda: (y.copy_(x), 0)[1]), (x, y)


CASES = [('gemm 32x32 576->576 (dominant)', conv_case(32, 576, 576, 0)), ('gemm 32x32 48->576 relu (fReMap)', conv_case(32, 48, 576, 1)),
         ('gemm 32x32 576->48', conv_case(32, 576, 48, 0)), ('gemm 16x16 288->288', conv_case(16, 288, 288, 0)),
         ('conv 3x3 64x64 64->64 (stem)', conv_case(64, 64, 64, 0, 3)), ('dw 5x5 32x32x576', dw_case(32, 576)), ('dw 5x5 16x16x288', dw_case(16, 288)),
         ('torch copy 302 MB', copy_case(151)), ('idle', (lambda: (time.sleep(0.001), 0)[1], None))]
for name, (fn, keep) in CASES:
    assert fn() == 0
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, out)); th.start()
    t0 = time.time(); k = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 1.6:
        for _ in range(50): fn()
        k += 50
        torch.cuda.synchronize()
    e1.record(); e1.synchronize()
    stop.set(); th.join()
    out = out[2:] or out                      # the first samples see the ramp
    clk = sorted(c for c, _ in out); pw = sorted(p for _, p in out)
    print('%-36s %8.1f us/launch   sclk median %4d MHz (min %4d)   %4.0f W   (%d samples)' % (
        name, e0.elapsed_time(e1) * 1e3 / k, clk[len(clk) // 2] if clk else 0, clk[0] if clk else 0, pw[len(pw) // 2] if pw else 0, len(out)), flush=True)
