"""N>1 path on CPU: two gloo processes exercise the frame sharding, the packed all-gather and the head
replication of deephar_amd.parallel with the CPU oracle standing in for the two HIP stages."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

T, J, BLOCKS, NACT = 4, 16, 2, 15


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build():
    from deephar_amd import graph, weights
    from deephar_amd.models import reception, action
    graph.reset_naming()
    pe = reception.build((64, 64, 3), J, dim=2, num_blocks=BLOCKS, num_context_per_joint=2, ksize=(5, 5))
    m = action.build_merge_model(pe, NACT, (64, 64, 3), T, J, BLOCKS, pose_dim=2, output_poses=True)
    weights.init_synthetic(m, seed=0)
    return m, weights.as_dict(m)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        from deephar_amd import parallel
        from oracle import action as oact
        from oracle.naming import Weights
        m, wd = _build()
        W = Weights(wd)
        clips = np.random.default_rng(5).uniform(-1, 1, (2, T, 64, 64, 3)).astype(np.float32)

        def frame_fn(x_local):        # stand-in for the HIP frame stage: packed [N, T/G, J, 2+1+2+C]
            n, tl = x_local.shape[:2]
            W.reset()
            with torch.no_grad():
                y, p, f = oact.merge_frames(W, torch.from_numpy(x_local).reshape((n * tl,) + x_local.shape[2:]),
                                            J, BLOCKS)
            y, p, f = (v.reshape((n, tl) + tuple(v.shape[1:])) for v in (y, p, f))
            return torch.cat([y, p, y * p, f], dim=-1)

        def head_fn(parts):           # parts follow info['cut']: y, p, y*p, f
            with torch.no_grad():
                return [o.numpy() for o in oact.merge_head(W, parts[0], parts[1], parts[3], NACT)]

        runner = parallel.ShardedClipModel(m, frame_fn=frame_fn, head_fn=head_fn)
        assert runner.info['Tl'] == T // world and runner.info['packed_channels'] == 2 + 1 + 2 + 576
        outs = runner.predict(clips)
        ref = oact.forward_merge(wd, clips, NACT, J, BLOCKS, output_poses=True)
        err = max(float(np.abs(a - b).max()) for a, b in zip(outs, ref))
        lo, hi = parallel.shard_slice(7, rank, world)
        q.put((rank, err, [o.shape for o in outs], (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_frame_sharded_clip_model_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    for rank, err, shapes, sl in res:
        assert err < 2e-5, 'rank %d: sharded result differs from the un-sharded oracle by %g' % (rank, err)
        assert shapes[0] == (2, T, J, 2) and shapes[1] == (2, T, J, 1) and shapes[-1] == (2, NACT)
    assert [r[3] for r in res] == [(0, 4), (4, 7)]


def _host_cost_worker(rank, world, port, q):
    """Host time of one sharded step with the device work taken out: stages that return a preallocated tensor; timed with
    the real gloo collective, the collective on its own, and with the collective stubbed -- the last is the Python of
    ShardedClipModel.forward_device (views of the gathered buffer, channel slices per cut tensor, the re-ordering copies
    of this CPU stand-in path, output bookkeeping)."""
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from deephar_amd import parallel
        m, _ = _build()
        runner = parallel.ShardedClipModel(m, frame_fn=None, head_fn=None)
        n, tl, cp = 1, runner.info["Tl"], runner.info["packed_channels"]    # (one clip: the stand-in path copies the parts)
        packed = torch.zeros((n, tl, J, cp))
        head_out = [torch.zeros((n, NACT)) for _ in runner.info['head_outputs']]
        runner.frame_fn = lambda x: packed
        runner.head_fn = lambda parts: head_out
        x = np.zeros((n, tl, 8, 8, 3), np.float32)
        steps = 300

        def timed(fn):
            for _ in range(20):
                fn()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            return (time.perf_counter() - t0) / steps * 1e6

        buf = [None]

        def gather_only():
            buf[0] = parallel.all_gather_rank_major(packed, None, world, out=buf[0])
        t_step = timed(lambda: runner.forward_device(x))
        t_coll = timed(gather_only)
        real = parallel.all_gather_rank_major          # (c) the same step with the collective stubbed out: what is left
        parallel.all_gather_rank_major = lambda *a, **k: buf[0]
        try:                                           # best of three: a loaded host must not fail the bound
            t_host = min(timed(lambda: runner.forward_device(x)) for _ in range(3))
        finally:
            parallel.all_gather_rank_major = real
        q.put((rank, t_step, t_coll, t_host))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_host_cost_of_a_sharded_step_two_ranks():
    """the floor the N > 1 numbers will stand on -- per sharded step the host spends <= 150 us outside
    the collective (world 2, gloo, merge model's cut: 4 tensors in a 581-channel packed buffer).  On the GPU path the two
    stages add one hipGraph launch each (~10 us of host time per launch, measured by bench.py's launch loop)."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_host_cost_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t_step, t_coll, t_host in res:
        print('rank %d: step %.1f us, gloo all-gather alone %.1f us, step without the collective %.1f us' % (
            rank, t_step, t_coll, t_host))
        assert t_host <= 150.0, 'rank %d: %.1f us of host time per sharded step outside the collective' % (rank, t_host)


def test_split_frames_partitions_merge_and_spnet():
    from deephar_amd import graph, parallel
    from deephar_amd.config import ModelConfig
    from deephar_amd.models import spnet
    from deephar_amd.utils import pa17j3d
    m, _ = _build()
    fm, hm, info = parallel.split_frames(m, 2)
    assert fm.input_shape == (None, 2, 64, 64, 3) and fm.output_shape == (None, 2, J, 581)
    assert info['passthrough'] == {0: 0, 1: 1} and info['head_outputs'] == list(range(2, 11))
    kinds = {s.kind for s in hm.plan.steps}
    assert 'dwconv' in kinds and 'kronecker' not in kinds          # pooling stays in the frame stage
    assert all(s.kind != 'globalmaxmin' for s in fm.plan.steps)     # temporal heads stay in the head stage
    # flops are conserved by the cut (frame stage scales with T/G)
    full = m.plan.total_flops()
    assert abs(2 * fm.plan.total_flops() + hm.plan.total_flops() - full) / full < 1e-6
    with pytest.raises(ValueError):
        parallel.split_frames(m, 3)
    graph.reset_naming()
    cfg = ModelConfig((8, 64, 64, 3), pa17j3d, num_actions=[60], num_pyramids=2, action_pyramids=[1, 2],
                      num_levels=3, num_pose_features=64, num_visual_features=64)
    sp = spnet.build(cfg)
    fm, hm, info = parallel.split_frames(sp, 4)
    assert fm.input_shape == (None, 2, 64, 64, 3)
    assert sorted(info['passthrough']) == [0, 1, 2, 3] and info['head_outputs'] == [4, 5, 6, 7]
    assert len(hm.inputs) == len(info['cut'])


def test_stage_models_inherit_engine_options():
    """The frame / head stages of a sharded clip model run with the clip model's engine options (GEMM precision,
    stream count): bench.py --gemm bf16x3 --workload penn_merge must not silently run fp32 stages."""
    from deephar_amd import parallel
    m, _ = _build()
    m.gemm_precision, m.num_streams = 'bf16x3', 2
    sh = parallel.ShardedClipModel(m, rank=0, world=2, frame_fn=lambda x: x, head_fn=lambda t: t)
    for stage in (sh.frame_model, sh.head_model):
        assert (stage.gemm_precision, stage.num_streams) == ('bf16x3', 2)
        assert stage.plan.gemm_precision == 'bf16x3'


def test_shard_slice_covers_range():
    from deephar_amd.parallel import shard_slice
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_slice(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
