"""Time the pointwise-GEMM / conv kernel on the shapes that dominate the MPII forward (batch 64), one line per
(shape, tile cfg).  A/B two builds on the same box with DEEPHAR_HIP_LIB=<other.so> (tools/build_variant.py);
the first line of a process is a warm-up (clock ramp) and is repeated."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import functional as F
dev = torch.device('cuda:0'); N = 64
rng = np.random.default_rng(0)
shapes = [(32, 576, 576, 12), (32, 576, 576, 12), (32, 576, 576, 11), (32, 576, 576, 9), (16, 288, 288, 12),
          (16, 288, 576, 13), (8, 288, 288, 13)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for spec in shapes:
    H, cin, cout, cfg = spec[:4]
    up2 = len(spec) > 4 and spec[4] == 1
    x = torch.randn(N, H, H, cin, device=dev)
    w = (rng.standard_normal((1, 1, cin, cout)) * 0.05).astype(np.float32)
    packed = F.pack_conv_weight(w, dev)
    qs, qb = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    r1 = torch.randn(N, H, H, cout, device=dev)
    r2 = torch.randn(N, 2 * H, 2 * H, cout, device=dev) if up2 else None
    run = (lambda: F.conv2d(x, w, (1, 1), 'same', pre_relu=True, post_scale=qs, post_shift=qb, res2=r2, up2=True,
                            tile_cfg=cfg, packed=packed)) if up2 else \
        (lambda: F.conv2d(x, w, (1, 1), 'same', pre_relu=True, post_scale=qs, post_shift=qb, res1=r1, tile_cfg=cfg,
                          packed=packed))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print('H%d %d->%d cfg%d%s: %.1f us  %.1f TF' % (H, cin, cout, cfg, ' up2' if up2 else '', best * 1e3,
                                                   2.0 * N * H * H * cin * cout / best / 1e9))
