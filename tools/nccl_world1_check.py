"""First contact of the sharded clip path with RCCL: a world of ONE rank on one GPU runs ShardedClipModel with the
nccl backend (process-group init, all_gather of the packed frame tensor, head stage) and must reproduce the plain
model bit for bit.  (Two ranks cannot share one GPU under RCCL; the 2-rank logic runs under gloo in the CPU tests.)"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
from deephar_amd import graph, weights, parallel
from deephar_amd.models import reception, action
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
graph.reset_naming()
pe = reception.build((128, 128, 3), 16, dim=2, num_blocks=2, num_context_per_joint=2, ksize=(5, 5))
m = action.build_merge_model(pe, 15, (128, 128, 3), 4, 16, 2, pose_dim=2, output_poses=True)
weights.init_synthetic(m, seed=0)
x = np.random.default_rng(3).uniform(-1, 1, (2, 4, 128, 128, 3)).astype(np.float32)
plain = m.predict(x, batch_size=2)
runner = parallel.ShardedClipModel(m)
sharded = runner.predict(x)
assert len(plain) == len(sharded)
for a, b in zip(plain, sharded):
    assert np.array_equal(a, np.asarray(b)), float(np.abs(a - np.asarray(b)).max())
t = torch.ones(4, device='cuda'); dist.all_reduce(t); dist.barrier()
print('nccl world-1 OK: %d outputs identical, packed channels %d' % (len(plain), runner.info['packed_channels']))
dist.destroy_process_group()
