"""The CPU oracle vs golden vectors computed by the REFERENCE'S OWN builder code (deephar/layers.py,
activations.py, models/*.py imported unmodified and executed on oracle/refrun/minikeras.py; generated in the build
container by tests/golden/make_reference_golden.py).  This is what pins the oracle's graph wiring, layer order,
constants and output ordering to the reference; Keras/TF layer semantics remain restated."""
import os

import numpy as np
import pytest
import torch

from refgolden import CASES, REAL_CASES, SMOOTH_CASES, build_case, golden


@pytest.mark.parametrize('tag', CASES + SMOOTH_CASES)
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


@pytest.mark.parametrize('tag', REAL_CASES)
def test_oracle_matches_reference_code_at_real_size(tag):
    """[r05] The BASELINE configurations at their REAL size -- ReceptionNet 8 blocks 2-D (configs[1]) and 3-D (configs[2]) at
    256 px, the merge model exactly as exp/pennaction/eval_penn_ar_pe_merge.py:51-57 builds it (T = 16, 4 blocks, nine
    heads), SPNet-NTU at T = 32 / 256 px (time_stride 2, spnet.py:100) -- against the reference's own model code run on
    mini-Keras: pins reception.py:277-312 at loop indices 3 .. 8 and action.py:127-153 at 4 blocks, which no reduced golden
    reaches.  fp64 only (the fp32 leg of the small cases above already covers dtype handling; this keeps the CPU suite
    within minutes)."""
    _, _, run = build_case(tag)
    _, g64 = golden(tag)
    o64 = run(torch.float64)
    assert [o.shape for o in o64] == [g.shape for g in g64]
    for k, (a, b) in enumerate(zip(o64, g64)):
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()), (tag, k, np.abs(a - b).max())


# [r06] the two real-size SPNet goldens must be resolvable in plain fp32 with a factor of two to spare: the round-5 fit of
# 'spnet3d_32_s' left the reference code's own fp32 run 1.0e-3 px from its fp64 run
TIGHT = {'spnet3d_32_s': 5e-4, 'spnet2d_speed_s': 5e-4}


@pytest.mark.parametrize('tag', SMOOTH_CASES + sorted(TIGHT))
def test_smooth_goldens_are_well_conditioned(tag):
    """The '<tag>_s' goldens (reference code on tests/wellcond.py vectors) are what the flat 1e-3 px SPNet tests stand
    on: every prediction block's read-out sensitivity S <= 0.05, maps not one-hot, and plain fp32 arithmetic (the
    reference code's own fp32 run) within 1e-3 px of its fp64 run (5e-4 px for the real-size cases) -- i.e. these vectors
    can resolve the bar."""
    import wellcond
    _, _, run = build_case(tag)
    t64 = {}
    run(torch.float64, taps=t64)
    stats = wellcond.assert_well_conditioned(t64, tag)
    assert len(stats) in (6, 18)
    g32, g64 = golden(tag)
    for a, b in zip(g32, g64):
        if b.ndim == 4:          # poses [1, T, J, dim + 1]
            d = b.shape[-1] - 1
            assert 256.0 * np.abs(a[..., :d] - b[..., :d]).max() <= TIGHT.get(tag, 1e-3), tag


@pytest.mark.skipif(not os.path.isdir('/root/reference/deephar'), reason='needs the reference checkout')
def test_keras_parity_kit_round_trip(tmp_path):
    """tools/make_keras_parity_kit.py: the product's weights in Keras' file layout, loaded BY ORDER into the
    reference's own model (here on mini-keras standing in for Keras 2.1.4) by the kit's runner script, reproduce
    the committed golden outputs exactly -- i.e. the kit is ready for a machine with the real Keras / TF."""
    import importlib.util
    import runpy
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tests', 'golden'))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import make_keras_parity_kit as kit
    import make_reference_golden as G
    kit.main(str(tmp_path), ['rec2d'])
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == 'keras' or k.startswith('keras.') or
             k == 'deephar' or k.startswith('deephar.') or k == 'tensorflow'}
    try:
        G.load_reference()
        spec = importlib.util.spec_from_file_location('deephar.utils.pose', '/root/reference/deephar/utils/pose.py')
        pose = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pose)
        u = types.ModuleType('deephar.utils')
        u.pa16j2d, u.pa17j3d = pose.pa16j2d, pose.pa17j3d
        sys.modules['deephar.utils'] = u
        runpy.run_path(str(tmp_path / 'run_in_keras.py'), run_name='__main__')
    finally:
        for k in [k for k in sys.modules if k == 'keras' or k.startswith('keras.') or k == 'deephar' or
                  k.startswith('deephar.') or k == 'tensorflow']:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})
    out = np.load(str(tmp_path / 'keras_outputs.npz'))
    g32, _ = golden('rec2d')
    for i, a in enumerate(g32):
        assert np.array_equal(out['rec2d/%d' % i], a)


def test_oracle_runs_a_pose_only_spnet_on_single_frames():
    """exp/pennaction/predict_bboxes.py:35-41, exp/ntu/predict_bboxes.py:35-41: SPNet built on single frames with
    `action_pyramids=[]` (pose only).  The oracle takes that configuration (it used to take max() of the empty list), its
    output count and shapes are the builder's, frames are independent, and the last pose is what `Model(full.input,
    full.outputs[-1])` re-wraps."""
    from deephar_amd import Model, graph, utils, weights
    from deephar_amd.config import ModelConfig
    from deephar_amd.models import spnet
    from oracle import spnet as osp
    graph.reset_naming()
    cfg = ModelConfig((64, 64, 3), utils.pa16j2d, num_pyramids=2, action_pyramids=[], num_levels=3)
    full = spnet.build(cfg)
    weights.init_synthetic(full, seed=0)
    one = Model(full.input, full.outputs[-1])
    assert len(one.outputs) == 1 and one.outputs[0].shape == (16, 3)
    ocfg = dict(num_joints=16, dim=2, num_actions=[], num_pyramids=2, action_pyramids=[], num_levels=3, kernel_size=(5, 5),
                growth=96, image_div=8, num_pose_features=0, num_visual_features=0, sam_alpha=1)
    x = np.random.default_rng(3).uniform(-1, 1, (3, 64, 64, 3)).astype(np.float32)
    outs = osp.forward(weights.as_dict(full), x, ocfg)
    assert len(outs) == len(full.outputs) == spnet.get_num_predictions(2, 3)
    assert all(o.shape == (3,) + t.shape for o, t in zip(outs, full.outputs))
    again = osp.forward(weights.as_dict(full), x[::-1].copy(), ocfg)
    assert np.allclose(again[-1][::-1], outs[-1], atol=1e-6)
    assert np.all(np.isfinite(outs[-1])) and outs[-1][..., :2].min() >= 0 and outs[-1][..., :2].max() <= 1
