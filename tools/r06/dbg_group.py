import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
os.environ['DEEPHAR_GROUP_LAUNCHES'] = '0'
from test_gpu_models import _spnet
m, _, _, _ = _spnet(8, 'pa16j2d', 15, 2, [1, 2], 160, replica=True, res=128)
x = np.random.default_rng(0).uniform(-1, 1, (2, 8, 128, 128, 3)).astype(np.float32)
m.predict(x, batch_size=2)
ex = m.executor
bp = next(iter(ex.bound.values()))
lib = bp.lib
for i in range(len(bp.calls) - 1):
    fc, ac, sc = bp.calls[i]
    fd, ad, sd = bp.calls[i + 1]
    if sc.kind == 'conv' and sd.kind == 'dwconv' and (sc.name or '').endswith('shortcut_conv'):
        ca, da = ac[0]._obj, ad[0]._obj
        rc = lib.dh_conv2d_dw_group_f32(ac[0], ad[0], ex.stream_ptr)
        print(sc.name, 'rc', rc, 'same buf', sc.ins['x'].buf is sd.ins['x'].buf, 'rows', ca.N * ca.OH * ca.OW, 'dw elems', da.N * da.H * da.W * da.C,
              'conv HxW', ca.H, ca.W, 'Cin', ca.Cin, 'Cout', ca.Cout, 'pre', bool(ca.pre_scale), ca.pre_relu, 'dw', da.H, da.W, da.C, da.KW, bool(da.pre_scale), da.pre_relu, da.up_in,
              'ldx', ca.ldx, da.ldx, da.ldy, 'wait', sd.wait, 'streams', sc.stream, sd.stream)
torch.cuda.synchronize()
