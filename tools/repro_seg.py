import os, sys, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import graph, weights
from deephar_amd.models import reception, action
mode = sys.argv[1]
keep = []
def h36(batch):
    graph.reset_naming()
    m = reception.build((256, 256, 3), 17, dim=3, num_blocks=8, depth_maps=16, ksize=(5, 5))
    weights.init_synthetic(m, seed=0)
    x = np.random.default_rng(21).uniform(-1, 1, (batch, 256, 256, 3)).astype(np.float32)
    m.predict(x, batch_size=batch); m.predict(x, batch_size=batch)
    if 'two' in mode: m.predict(x, batch_size=batch // 4); m.predict(x, batch_size=batch)
    if 'keep' in mode: keep.append(m)
def merge():
    graph.reset_naming()
    pe = reception.build((256, 256, 3), 16, dim=2, num_blocks=4, num_context_per_joint=2, ksize=(5, 5))
    m = action.build_merge_model(pe, 15, (256, 256, 3), 16, 16, 4, pose_dim=2, pose_net_version='v1', output_poses=True)
    weights.init_synthetic(m, seed=0)
    x = np.random.default_rng(22).uniform(-1, 1, (4, 16, 256, 256, 3)).astype(np.float32)
    m.predict(x, batch_size=4); m.predict(x, batch_size=4)
if mode.startswith('rev'):
    merge(); gc.collect(); h36(128)
else:
    h36(32 if 'small' in mode else 128); gc.collect(); merge()
print('OK', mode)
