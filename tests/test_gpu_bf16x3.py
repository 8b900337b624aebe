"""The opt-in split-bf16 GEMM mode (Model.gemm_precision = 'bf16x3', csrc/gemm1x1s.hip) against the SAME bars as the
default fp32-MFMA path: joint coordinates within 1e-3 px of the fp64 oracle, identical arg-max action labels, bit-exact
batch invariance.  Every fp32 operand is split exactly into three bf16 parts; six of the nine partial products run on
the bf16 matrix cores with fp32 accumulation -- the measured error against fp64 equals the fp32 path's
(profiles/parity_r02_bf16x3.json next to profiles/parity_r02.json)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import paritylog                                   # noqa: E402
from paritylog import PX_TOL, check                # noqa: E402

pytestmark = pytest.mark.gpu


def _split_count(m):
    bp = next(iter(m.executor.bound.values()))
    return sum(1 for s in m.plan.steps if s.kind == 'conv' and s.attrs.get('w_split')), \
        sum(1 for s in m.plan.steps if s.kind == 'conv')


def test_mpii_8_blocks_bf16x3(hip_lib, cuda):
    from test_gpu_models import _build, _oracle
    kw = dict(num_context_per_joint=2, concat_pose_confidence=False)
    m, wd = _build(2, 8, 16, **kw)
    m.gemm_precision = 'bf16x3'
    x = np.random.default_rng(0).uniform(-1, 1, (3, 256, 256, 3)).astype(np.float32)
    hip = m.predict(x, batch_size=3)
    nsplit, nconv = _split_count(m)
    assert nsplit >= 100 and nconv - nsplit <= 8, (nsplit, nconv)       # all but the Cin = 3 stem conv and friends
    o32, _ = _oracle(wd, x, 2, 8, 16, torch.float32, **kw)
    o64, _ = _oracle(wd, x, 2, 8, 16, torch.float64, **kw)
    for b in range(8):
        check('bf16x3.pose%d' % (b + 1), hip[2 * b], o32[2 * b], o64[2 * b], PX_TOL)
        check('bf16x3.vis%d' % (b + 1), hip[2 * b + 1], o32[2 * b + 1], o64[2 * b + 1], 1e-5, rel=True)
    # against the default fp32-MFMA path: same accuracy class, not the same bits
    f, _ = _build(2, 8, 16, **kw)
    f.gemm_precision = 'f32'
    ref = f.predict(x, batch_size=3)
    d = max(float(np.abs(a[..., :2] - b[..., :2]).max()) for a, b in zip(hip[::2], ref[::2]))
    print('bf16x3 vs fp32-MFMA path: max |dxy| = %.2e px' % (256 * d))
    assert 0 < 256 * d < 1e-3


def test_h36m_8_blocks_bf16x3(hip_lib, cuda):
    from test_gpu_models import _build, _oracle
    m, wd = _build(3, 8, 17, depth_maps=16)
    m.gemm_precision = 'bf16x3'
    x = np.random.default_rng(31).uniform(-1, 1, (2, 256, 256, 3)).astype(np.float32)
    hip = m.predict(x, batch_size=2)
    o32, _ = _oracle(wd, x, 3, 8, 17, torch.float32, depth_maps=16)
    o64, _ = _oracle(wd, x, 3, 8, 17, torch.float64, depth_maps=16)
    for b in range(8):
        check('bf16x3.xyz%d' % (b + 1), hip[b][..., :3], o32[b][..., :3], o64[b][..., :3], PX_TOL)
        check('bf16x3.vis%d' % (b + 1), hip[b][..., 3:], o32[b][..., 3:], o64[b][..., 3:], 1e-6)


def test_penn_merge_T16_bf16x3_labels(hip_lib, cuda):
    from test_gpu_models import _merge
    from oracle import action as oact
    T, blocks, nact, joints = 16, 4, 15, 16
    m, wd = _merge(2, T, joints, blocks, pose_net_version='v1', num_actions=nact)
    m.gemm_precision = 'bf16x3'
    x = np.random.default_rng(32).uniform(-1, 1, (1, T, 256, 256, 3)).astype(np.float32)
    hip = m.predict(x, batch_size=1)
    okw = dict(pose_dim=2, pose_net_version='v1', output_poses=True)
    o32 = oact.forward_merge(wd, x, nact, joints, blocks, dtype=torch.float32, **okw)
    o64 = oact.forward_merge(wd, x, nact, joints, blocks, dtype=torch.float64, **okw)
    check('bf16x3.pose', hip[0], o32[0], o64[0], PX_TOL)
    for k in range(2, len(hip)):
        check('bf16x3.action%d' % (k - 1), hip[k], o32[k], o64[k], 1e-5)
        assert np.array_equal(hip[k].argmax(-1), o64[k].argmax(-1))


def test_spnet_ntu_bf16x3(hip_lib, cuda):
    from test_gpu_models import _spnet, spnet_parity
    x = np.random.default_rng(11).uniform(-1, 1, (1, 8, 256, 256, 3)).astype(np.float32)
    m, cfg, wd, ocfg = _spnet(8, 'pa17j3d', 60, 2, [1, 2], 192, calibrate=x)
    m.gemm_precision = 'bf16x3'
    spnet_parity(m, cfg, wd, ocfg, x, 2, [1, 2])


def test_bf16x3_batch_invariance_is_bit_exact(hip_lib, cuda):
    """The split kernels sum K in one fixed order in every tiling: results do not depend on the batch size the plan was
    bound (and tuned) for, on the position inside the batch, or on graph replay."""
    from test_gpu_models import _build
    m, _ = _build(2, 4, 16, num_context_per_joint=2)
    m.gemm_precision = 'bf16x3'
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (32, 256, 256, 3)).astype(np.float32)
    a = m.predict(x, batch_size=32)
    perm = rng.permutation(32)
    b = m.predict(x[perm], batch_size=32)
    c = m.predict(x, batch_size=8)
    d = m.predict(x, batch_size=32)
    for k in range(len(a)):
        assert np.array_equal(a[k][perm], b[k]) and np.array_equal(a[k], c[k]) and np.array_equal(a[k], d[k])
