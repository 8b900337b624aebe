import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import functional as F
dev = torch.device('cuda:0'); N = 64
rng = np.random.default_rng(0)
H, cin, cout = 32, 576, 576
x = torch.randn(N, H, H, cin, device=dev)
w = (rng.standard_normal((1, 1, cin, cout)) * 0.05).astype(np.float32)
packed = F.pack_conv_weight(w, dev)
def run(cfg, mode, reps=10):
    f = lambda: F.conv2d(x, w, packed=packed, tile_cfg=cfg + 16 * mode)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / reps * 1e-3
    print('cfg=%d mode=%d (noload=%d nobarrier=%d nomfma=%d): %.1f us %.1f TF-equivalent' % (cfg, mode, mode & 1, (mode >> 1) & 1, (mode >> 2) & 1, t*1e6, 2.0*N*H*H*cin*cout/t/1e12))
for cfg in (2, 0, 1):
    for mode in (0, 1, 2, 3, 4, 5):
        run(cfg, mode)
