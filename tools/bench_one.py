"""Run one conv shape repeatedly (for rocprofv3 --pmc):
    python tools/bench_one.py H Cin Cout k stride cfg [reps] [pre_relu=1] [split=0]
BN + residual epilogue always on (the dominant GEMMs of the model carry both)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import functional as F
H, cin, cout, k, s, cfg = [int(v) for v in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
relu = bool(int(sys.argv[8])) if len(sys.argv) > 8 else True
split = bool(int(sys.argv[9])) if len(sys.argv) > 9 else False
dev = torch.device('cuda:0'); N = 64
rng = np.random.default_rng(0)
x = torch.randn(N, H, H, cin, device=dev)
w = (rng.standard_normal((k, k, cin, cout)) * 0.05).astype(np.float32)
packed = F.pack_conv_weight(w, dev)
if split:
    from deephar_amd.engine import packing
    pk, kp, np_ = packing.pack_conv_split(w)
    packed = (torch.from_numpy(pk).to(dev), kp, np_)
oh = -(-H // s)
qs, qb = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
r1 = torch.randn(N, oh, oh, cout, device=dev)
for _ in range(reps):
    F.conv2d(x, w, (s, s), 'same', pre_relu=relu, post_scale=qs, post_shift=qb, res1=r1, tile_cfg=cfg, packed=packed, split=split)
torch.cuda.synchronize()
