import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from deephar_amd.engine import executor as E
orig = E.BoundPlan.group_launches
def traced(self, sp):
    lib = self.lib
    for i in range(self.npre, len(self.calls) - 1):
        fc, ac, sc = self.calls[i]; fd, ad, sd = self.calls[i + 1]
        if sc.kind == 'conv' and (sc.name or '').endswith('shortcut_conv'):
            print('pair', i, sc.name, sd.kind, sd.name, 'streams', sc.stream, sd.stream, 'wait', sd.wait, 'grouped attr', sc.attrs.get('grouped'),
                  'same', sc.ins['x'].buf is sd.ins['x'].buf if sd.kind == 'dwconv' else None)
    r = orig(self, sp)
    print('grouped ->', r)
    return r
E.BoundPlan.group_launches = traced
from test_gpu_models import _spnet
m, _, _, _ = _spnet(8, 'pa16j2d', 15, 2, [1, 2], 160, replica=True, res=128)
x = np.random.default_rng(0).uniform(-1, 1, (2, 8, 128, 128, 3)).astype(np.float32)
m.predict(x, batch_size=2)
