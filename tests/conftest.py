import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    try:   # the CPU oracle (PyTorch-CPU) is slower, not faster, with hundreds of threads on the GPU box
        import torch
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
    except ImportError:
        pass
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hip_lib():
    """Make sure the in-tree HIP library exists (hipcc cross-compiles without a GPU)."""
    from deephar_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from deephar_amd.csrc import build
        build.build(verbose=False)
    return _lib.load()


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a HIP device; none visible')
    return torch.device('cuda:0')


def pytest_sessionfinish(session, exitstatus):
    """-m gpu runs: write the per-output parity table (tests/paritylog.py) to gpurun_out/parity_r06.json."""
    try:
        import paritylog
        path = paritylog.dump()
        if path:
            print('\nparity table: %s (%d comparisons)' % (path, len(paritylog.RECORDS)))
    except Exception as e:      # never turn a green run red over the report
        print('parity table not written:', e)
