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

# [r05] SPNet-NTU at T = 32 (BASELINE configs[4]; consumers of the gathered tensors: spnet.py:219-235): the 2 634-channel
# packed buffer goes through a REAL all_gather_into_tensor call (always_collective: a world of one would otherwise
# short-cut to a view), the persistent gather buffer is re-used over three steps, and the head stage reads its strided
# channel runs -- bit-identical to the plain model.
if os.environ.get('DEEPHAR_NCCL_CHECK_SPNET', '1') != '0':
    from deephar_amd import utils
    from deephar_amd.config import ModelConfig
    from deephar_amd.models import spnet
    graph.reset_naming()
    cfg = ModelConfig((32, 256, 256, 3), utils.pa17j3d, num_actions=[60], num_pyramids=2, action_pyramids=[1, 2],
                      num_levels=4, pose_replica=False, num_pose_features=192, num_visual_features=192)
    sp = spnet.build(cfg)
    weights.init_synthetic(sp, seed=0)
    xs = np.random.default_rng(5).uniform(-1, 1, (3, 1, 32, 256, 256, 3)).astype(np.float32)
    runner = parallel.ShardedClipModel(sp, always_collective=True)
    for step in range(3):
        plain = sp.predict(xs[step], batch_size=1)
        sharded = runner.predict(xs[step])
        assert runner._gather_buf is not None and tuple(runner._gather_buf.shape[:3]) == (1, 1, 32)
        for a, b in zip(plain, sharded):
            assert np.array_equal(a, np.asarray(b)), (step, float(np.abs(a - np.asarray(b)).max()))
    print('nccl world-1 SPNet-NTU T=32 OK: %d outputs identical over 3 steps, packed channels %d, gather buffer %s' % (
        len(plain), runner.info['packed_channels'], tuple(runner._gather_buf.shape)))

    # [r06] the pipelined streams (two send / gather slots, the collective on its own stream): five steps issued BACK TO
    # BACK with no synchronisation in between -- step i + 1's frame stage runs while step i's all-gather and head stage are
    # still in flight -- must give, step for step, the serial form's bits; and the serial form still works
    assert runner.overlap and runner._comm_stream is not None
    xd = [torch.from_numpy(xs[i % 3]).to('cuda') for i in range(5)]
    want = [[np.asarray(o) for o in sp.predict(xs[i % 3], batch_size=1)] for i in range(5)]
    kept = []
    H = runner.head_model.executor.stream
    for i in range(5):
        outs = runner.forward_device(xd[i])
        with torch.cuda.stream(H):                 # the outputs are views of the head plan's arena, valid until its next
            kept.append([o.clone() for o in outs])  # forward: copied out on the head's own stream, no host wait in between
    runner.synchronize()
    for i in range(5):
        for a, b in zip(want[i], kept[i]):
            assert np.array_equal(a, b.cpu().numpy()), ('pipelined', i)
    serial = parallel.ShardedClipModel(sp, always_collective=True, overlap=False)
    for a, b in zip(want[1], serial.predict(xs[1])):
        assert np.array_equal(a, np.asarray(b))
    print('nccl world-1 pipelined OK: 5 back-to-back steps identical, 2 slots, collective on its own stream')
dist.destroy_process_group()
