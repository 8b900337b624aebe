"""`python bench.py --gpus N` must work on its own: with WORLD_SIZE unset it starts the N ranks
itself; under torch.distributed.run it uses the ranks it is given.  Both forms are exercised here on CPU through
bench.py's --dry-run mode (gloo backend, no HIP device): rendezvous on 127.0.0.1, the rank-major all-gather + frame
re-ordering view of deephar_amd/parallel.py, max-over-ranks timing, exactly ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_json_line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_bench_spawns_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['steps'] == 3 and line['warmup'] == 1
    assert line['dry_run'] is True and line['value'] > 0 and line['collective_us'] > 0
    # [r05] the per-rank decomposition a first multi-GPU run will be read by: every rank's own clock, host enqueue time
    # and collective time
    assert [r['rank'] for r in line['per_rank']] == [0, 1]
    assert all(r['ms_per_step'] > 0 and r['collective_us'] > 0 and r['host_enqueue_us_per_step'] > 0 for r in line['per_rank'])
    for key in ('metric', 'unit', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert key in line
    # [r06] per-rank spread and the strong-scaling leg (a fixed clip batch, T / N frames of it per rank) beside the weak one
    lo, hi = line['per_rank_ms_min_max']
    assert 0 < lo <= hi
    assert line['scaling'] == 'weak' and line['strong_scaling']['scaling'] == 'strong'
    assert line['strong_scaling']['frames_per_rank'] == 4 and line['strong_scaling']['value'] > 0


@pytest.mark.timeout(300)
def test_bench_under_torch_distributed_run():
    from test_parallel_gloo import _free_port
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), 'bench.py', '--gpus', '2',
                        '--steps', '2', '--warmup', '1', '--dry-run'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2


def test_rank_major_gather_view_orders_frames():
    """world of one rank: all_gather_rank_major is a view, frames_view + flatten restores [N, T, ...]; the strided
    [N, G, T/G, J, c] slices are what Executor.run_device copies into the head stage's contiguous inputs."""
    import torch
    from deephar_amd import parallel
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    g = parallel.all_gather_rank_major(x, world=1)
    assert g.shape == (1, 2, 3, 4, 5) and g.data_ptr() == x.data_ptr()
    assert torch.equal(parallel.frames_view(g).reshape(2, 3, 4, 5), x)
    assert torch.equal(parallel.all_gather_frames(x, world=1), x)
    # `always`: the collective is really issued for a world of one (what tools/nccl_world1_check.py does under RCCL)
    import torch.distributed as dist
    from test_parallel_gloo import _free_port
    dist.init_process_group('gloo', rank=0, world_size=1, init_method='tcp://127.0.0.1:%d' % _free_port())
    try:
        buf = torch.full((1, 2, 3, 4, 5), -1.0)
        g2 = parallel.all_gather_rank_major(x, out=buf, always=True)
        assert g2.data_ptr() == buf.data_ptr() and torch.equal(g2[0], x)
    finally:
        dist.destroy_process_group()
    # two "ranks" written rank-major by hand: rank r holds frames [r*Tl, (r+1)*Tl) of every clip
    full = torch.arange(2 * 6 * 4 * 5, dtype=torch.float32).reshape(2, 6, 4, 5)
    rank_major = torch.stack([full[:, 0:3], full[:, 3:6]])
    v = parallel.frames_view(rank_major)
    assert v.shape == (2, 2, 3, 4, 5) and not v.is_contiguous()
    dst = torch.empty(2, 6, 4, 2)
    dst.view(2, 2, 3, 4, 2).copy_(v[..., 1:3])                   # what run_device does with a strided head input
    assert torch.equal(dst, full[..., 1:3])
