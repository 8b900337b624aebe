"""The CPU oracle vs golden vectors computed by the REFERENCE'S OWN builder code (deephar/layers.py,
activations.py, models/*.py imported unmodified and executed on oracle/refrun/minikeras.py; generated in the build
container by tests/golden/make_reference_golden.py).  This is what pins the oracle's graph wiring, layer order,
constants and output ordering to the reference; Keras/TF layer semantics remain restated."""
import numpy as np
import pytest
import torch

from refgolden import CASES, build_case, golden


@pytest.mark.parametrize('tag', CASES)
def test_oracle_matches_reference_code(tag):
    _, _, run = build_case(tag)
    g32, g64 = golden(tag)
    o64 = run(torch.float64)
    assert [o.shape for o in o64] == [g.shape for g in g64]
    for k, (a, b) in enumerate(zip(o64, g64)):
        # same fp64 arithmetic up to summation order: agreement to ~1e-10 means identical wiring and constants
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()), (tag, k, np.abs(a - b).max())
    o32 = run(torch.float32)
    for k, (a, b) in enumerate(zip(o32, g32)):
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max()), (tag, k, np.abs(a - b).max())
