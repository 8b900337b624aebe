"""The RCCL leg of the frame-sharded clip path on the one GPU a test box has: a world of one rank goes through
process-group init, the packed all_gather and the replicated head stage with backend "nccl" and must equal the plain
model bit for bit (tools/nccl_world1_check.py, run in a subprocess so that the process group does not outlive it).
The 2-rank bookkeeping runs under gloo in tests/test_parallel_gloo.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_sharded_clip_model_over_rccl_world_of_one(hip_lib, cuda):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'nccl_world1_check.py')], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'nccl world-1 OK: 11 outputs identical' in out.stdout
    # [r05] SPNet-NTU at T = 32 through a real all_gather_into_tensor call (world of one, always_collective)
    assert 'nccl world-1 SPNet-NTU T=32 OK: 12 outputs identical over 3 steps, packed channels 2634' in out.stdout, out.stdout[-1500:]
    # [r06] the pipelined form: frame stage of step i + 1 beside the all-gather and head stage of step i
    assert 'nccl world-1 pipelined OK: 5 back-to-back steps identical' in out.stdout, out.stdout[-1500:]
