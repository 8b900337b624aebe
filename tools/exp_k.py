import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import functional as F
dev = torch.device('cuda:0'); N = 64
rng = np.random.default_rng(0)
def run(H, cin, cout, cfg, res=True, reps=10):
    x = torch.randn(N, H, H, cin, device=dev)
    w = (rng.standard_normal((1, 1, cin, cout)) * 0.05).astype(np.float32)
    packed = F.pack_conv_weight(w, dev)
    r1 = torch.randn(N, H, H, cout, device=dev) if res else None
    f = lambda: F.conv2d(x, w, packed=packed, res1=r1, tile_cfg=cfg)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / reps * 1e-3
    print('H=%d K=%d N=%d cfg=%d res=%d: %.1f us %.1f TF' % (H, cin, cout, cfg, res, t*1e6, 2.0*N*H*H*cin*cout/t/1e12))
for cfg in (0, 2, 3):
    for K in (288, 576, 1152, 2304, 4608):
        run(32, K, 576, cfg, res=False)
run(32, 576, 1152, 2, res=False); run(32, 576, 2304, 2, res=False)
run(64, 576, 576, 2, res=False)
