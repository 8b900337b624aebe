"""BASELINE.json configs 2-4 at their REAL sizes, through properties that need no oracle (the oracle parity of the
same model families runs at reduced depth / clip length in tests/test_gpu_models.py):

  cfg 2  Human3.6M 3-D pose: ReceptionNet dim=3, 8 blocks, J=17, 16 depth maps, batch 128 (exp/h36m/eval_h36m.py:42-48)
  cfg 3  PennAction pose+action: merge model, 16-frame 256x256 clips, 4 blocks, J=16 (exp/pennaction/eval_penn_ar_pe_merge.py:42-57)
  cfg 4  NTU multitask: SPNet, 32-frame clips, batch 8, pa17j3d, 60 actions (exp/ntu/eval_ntu_multitask.py:34-38 builds it
         with 8 frames; BASELINE.json asks for 32)

Properties: frames / clips are independent (batch permutation permutes the outputs bit-exactly, duplicated items
give duplicated rows), results do not depend on batch_size chunking or on hipGraph replay, coordinates and
probabilities are in range, arg-max labels are invariant under all of the above.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _properties(m, x, bs_full, bs_small, check):
    rng = np.random.default_rng(17)
    n = len(x)
    x[n - 1] = x[0]
    a = m.predict(x, batch_size=bs_full)
    perm = rng.permutation(n)
    b = m.predict(x[perm], batch_size=bs_full)
    c = m.predict(x, batch_size=bs_small)
    d = m.predict(x, batch_size=bs_full)            # replays the captured hipGraph
    a, b, c, d = [v if isinstance(v, list) else [v] for v in (a, b, c, d)]
    for k in range(len(a)):
        assert np.all(np.isfinite(a[k]))
        assert np.array_equal(a[k][perm], b[k]), 'output %d: batch items are not independent' % k
        assert np.array_equal(a[k], c[k]), 'output %d depends on the batch size' % k
        assert np.array_equal(a[k], d[k]), 'output %d: graph replay differs' % k
        assert np.array_equal(a[k][n - 1], a[k][0])
        check(k, a[k])
    return a


def test_cfg2_h36m_3d_batch128(hip_lib, cuda):
    from deephar_amd import graph, weights
    from deephar_amd.models import reception
    graph.reset_naming()
    m = reception.build((256, 256, 3), 17, dim=3, num_blocks=8, depth_maps=16, ksize=(5, 5))
    weights.init_synthetic(m, seed=0)
    x = np.random.default_rng(21).uniform(-1, 1, (128, 256, 256, 3)).astype(np.float32)

    def check(k, y):
        assert y.shape == (128, 17, 4)
        assert y[..., :3].min() >= 0.0 and y[..., :3].max() <= 1.0         # x, y, z expectations of grids in [0, 1]
        assert y[..., 3].min() > 0.0 and y[..., 3].max() <= 1.0            # sigmoid visibility (saturates in fp32)
    out = _properties(m, x, 128, 32, check)
    assert len(out) == 8
    assert np.std(out[-1][..., :2]) > 1e-3                                  # not a degenerate constant output


def test_cfg3_penn_merge_16_frame_clips(hip_lib, cuda):
    from deephar_amd import graph, weights
    from deephar_amd.models import reception, action
    graph.reset_naming()
    pe = reception.build((256, 256, 3), 16, dim=2, num_blocks=4, num_context_per_joint=2, ksize=(5, 5))
    m = action.build_merge_model(pe, 15, (256, 256, 3), 16, 16, 4, pose_dim=2, pose_net_version='v1', output_poses=True)
    weights.init_synthetic(m, seed=0)
    x = np.random.default_rng(22).uniform(-1, 1, (4, 16, 256, 256, 3)).astype(np.float32)

    def check(k, y):
        if k == 0:
            assert y.shape == (4, 16, 16, 2) and y.min() >= 0.0 and y.max() <= 1.0
        elif k == 1:
            assert y.shape == (4, 16, 16, 1)
        else:
            assert y.shape == (4, 15) and y.min() >= 0.0
            np.testing.assert_allclose(y.sum(-1), 1.0, rtol=1e-5)
    out = _properties(m, x, 4, 1, check)
    assert len(out) == 11


def test_cfg4_ntu_spnet_32_frame_clips_batch8(hip_lib, cuda):
    from deephar_amd import graph, weights, utils
    from deephar_amd.config import ModelConfig
    from deephar_amd.models import spnet
    graph.reset_naming()
    cfg = ModelConfig((32, 256, 256, 3), utils.pa17j3d, num_actions=[60], num_pyramids=2, action_pyramids=[1, 2],
                      num_levels=4, pose_replica=False, num_pose_features=192, num_visual_features=192)
    m = spnet.build(cfg)
    weights.init_synthetic(m, seed=0)
    x = np.random.default_rng(23).uniform(-1, 1, (8, 32, 256, 256, 3)).astype(np.float32)
    npose = spnet.get_num_predictions(2, 4)

    def check(k, y):
        if k < npose:
            assert y.shape == (8, 32, 17, 4)
            assert y[..., :2].min() >= 0.0 and y[..., :2].max() <= 1.0
        else:
            assert y.shape == (8, 60) and y.min() >= 0.0
            np.testing.assert_allclose(y.sum(-1), 1.0, rtol=1e-5)
    out = _properties(m, x, 8, 2, check)
    assert len(out) == npose + spnet.get_num_predictions(2, 4)
