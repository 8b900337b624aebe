"""Model-level parity on the GPU: deephar_amd Model.predict (HIP kernels through the C-ABI) vs the CPU oracle
on identical seeded weights and inputs, plus size-independent properties at the BASELINE batch sizes.

Tolerance (BASELINE.json north_star): joint coordinates within 1e-3 px of a 256-px crop, i.e.
|d| <= 3.9e-6 in the model's normalised [0,1] output, asserted as max|hip - oracle_fp64| <= 1e-3 px with no
relative clause (tests/paritylog.py; every comparison also lands in gpurun_out/parity_r05.json together with
|oracle_fp32 - oracle_fp64| and |hip - oracle_fp32|).  SPNet on per-pixel-noise inputs is the one place where the
synthetic read-out itself is ill-conditioned: those cases are kept here as a stress test under
paritylog.conditioned_tolerance; SPNet at the flat 1e-3 px bar lives in tests/test_gpu_spnet_flat.py.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import paritylog                                   # noqa: E402
from paritylog import PX_TOL, check as _check      # noqa: E402

pytestmark = pytest.mark.gpu


def _build(dim, num_blocks, joints, **kw):
    from deephar_amd import graph, weights
    from deephar_amd.models import reception
    graph.reset_naming()
    m = reception.build((256, 256, 3), joints, dim=dim, num_blocks=num_blocks, ksize=(5, 5), **kw)
    weights.init_synthetic(m, seed=0)
    return m, weights.as_dict(m)


def _oracle(wd, x, dim, num_blocks, joints, dtype, **kw):
    from oracle import reception as oref
    taps = {}
    outs = oref.forward(wd, x, joints, dim, num_blocks=num_blocks, ksize=(5, 5), dtype=dtype, taps=taps, **kw)
    return outs, taps


def test_reception_mpii_2d_context_parity(hip_lib, cuda):
    """cfg 2 model (8 blocks, J=16, 2 contexts, k=5) on seeds {0, 1, 2}, four frames each; also the
    pre-aggregation tensors."""
    kw = dict(num_context_per_joint=2, concat_pose_confidence=False)
    m, wd = _build(2, 8, 16, **kw)
    for seed in (0, 1, 2):
        x = np.random.default_rng(seed).uniform(-1, 1, (4, 256, 256, 3)).astype(np.float32)
        hip = m.predict(x, batch_size=4)
        o32, t32 = _oracle(wd, x, 2, 8, 16, torch.float32, **kw)
        o64, t64 = _oracle(wd, x, 2, 8, 16, torch.float64, **kw)
        assert len(hip) == 16
        for b in range(8):
            # the reference divides by sum_c(vc) without epsilon (blocks.py:273-274): make sure the test is
            # not sitting on a pole, then check pose and visibility
            vc = t64['vc%d' % (b + 1)].reshape(4, 16, 2).sum(axis=2)
            assert np.all(vc > 1.0), 'synthetic context confidences too close to 0'
            _check('pose%d.seed%d' % (b + 1, seed), hip[2 * b], o32[2 * b], o64[2 * b], PX_TOL)
            _check('vis%d.seed%d' % (b + 1, seed), hip[2 * b + 1], o32[2 * b + 1], o64[2 * b + 1], 1e-5, rel=True)
        # heat-maps must be neither flat nor one-hot, else the px test is vacuous
        hm = t64['heatmaps8']
        assert 1.0 < hm.std() < 30.0


def test_reception_mpii_intermediates(hip_lib, cuda):
    """Localises a mismatch: stem / rBlock / heat-map tensors of a 2-block model vs the oracle."""
    from deephar_amd import Model
    kw = dict(num_context_per_joint=2, export_heatmaps=True, export_vfeat_block=1)
    m, wd = _build(2, 2, 16, **kw)
    x = np.random.default_rng(3).uniform(-1, 1, (2, 256, 256, 3)).astype(np.float32)
    hip = m.predict(x, batch_size=2)
    o32, _ = _oracle(wd, x, 2, 2, 16, torch.float32, **kw)
    o64, _ = _oracle(wd, x, 2, 2, 16, torch.float64, **kw)
    assert [h.shape for h in hip] == [o.shape for o in o64]
    names = ['out1', 'hm1', 'out2', 'hm2', 'vfeat1']
    for n, h, a, b in zip(names, hip, o32, o64):
        if n.startswith('out'):
            _check(n + '.xy', h[..., :2], a[..., :2], b[..., :2], PX_TOL)
            _check(n + '.vis', h[..., 2:], a[..., 2:], b[..., 2:], 1e-5, rel=True)
        else:
            _check(n, h, a, b, 2e-5, rel=True)


def test_reception_h36m_3d_parity(hip_lib, cuda):
    """cfg 3 model family (dim=3, J=17, 16 depth maps); 4 blocks keep the CPU oracle to a few seconds."""
    m, wd = _build(3, 4, 17, depth_maps=16)
    x = np.random.default_rng(2).uniform(-1, 1, (3, 256, 256, 3)).astype(np.float32)
    hip = m.predict(x, batch_size=3)
    o32, _ = _oracle(wd, x, 3, 4, 17, torch.float32, depth_maps=16)
    o64, _ = _oracle(wd, x, 3, 4, 17, torch.float64, depth_maps=16)
    assert len(hip) == 4 and hip[0].shape == (3, 17, 4)
    for b in range(4):
        _check('xyz%d' % (b + 1), hip[b][..., :3], o32[b][..., :3], o64[b][..., :3], PX_TOL)
        _check('vis%d' % (b + 1), hip[b][..., 3:], o32[b][..., 3:], o64[b][..., 3:], 1e-6)


def test_rewrapped_outputs_match(hip_lib, cuda):
    """The eval scripts' idiom Model(model.input, [concatenate([pose_b, vis_b]) ...])
    (exp/mpii/eval_mpii_singleperson.py:56-61) returns exactly the un-wrapped model's numbers."""
    from deephar_amd import Model, concatenate
    m, _ = _build(2, 2, 16, num_context_per_joint=2, concat_pose_confidence=False)
    x = np.random.default_rng(4).uniform(-1, 1, (2, 256, 256, 3)).astype(np.float32)
    plain = m.predict(x, batch_size=2)
    outs = [concatenate([m.outputs[2 * b], m.outputs[2 * b + 1]], name='blk%d' % (b + 1)) for b in range(2)]
    wrapped = Model(m.input, outputs=outs, name='wrapped').predict(x, batch_size=2)
    for b in range(2):
        assert np.array_equal(wrapped[b], np.concatenate([plain[2 * b], plain[2 * b + 1]], axis=-1))
    last = Model(m.input, m.outputs[-1]).predict(x, batch_size=2)
    assert isinstance(last, np.ndarray) and np.array_equal(last, plain[-1])


def test_full_size_properties_batch64(hip_lib, cuda):
    """BASELINE cfg 2 at its real size (batch 64): properties that need no oracle.
    - frames are independent: permuting the batch permutes the outputs bit-exactly
    - predict(batch_size=64) == predict(batch_size=16) == graph replay of the same plan, bit-exactly
    - coordinates in [0,1]; duplicated frames give duplicated rows."""
    m, _ = _build(2, 8, 16, num_context_per_joint=2, concat_pose_confidence=True)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (64, 256, 256, 3)).astype(np.float32)
    x[63] = x[0]
    a = m.predict(x, batch_size=64)
    perm = rng.permutation(64)
    b = m.predict(x[perm], batch_size=64)
    c = m.predict(x, batch_size=16)
    d = m.predict(x, batch_size=64)     # second call replays the captured hipGraph
    for k in range(8):
        assert a[k].shape == (64, 16, 3)
        assert np.array_equal(a[k][perm], b[k])
        assert np.array_equal(a[k], c[k]) and np.array_equal(a[k], d[k])
        assert np.array_equal(a[k][63], a[k][0])
        assert np.all(np.isfinite(a[k]))
        assert a[k][..., :2].min() >= 0.0 and a[k][..., :2].max() <= 1.0


def test_multi_stream_plans_are_bit_identical(hip_lib, cuda):
    """Model.num_streams > 1 (independent branches on parallel hipGraph branches; opt-in since round 3) changes the
    schedule and the arena layout, never a result bit -- eager and captured."""
    x = np.random.default_rng(10).uniform(-1, 1, (3, 256, 256, 3)).astype(np.float32)
    m, _ = _build(2, 2, 16, num_context_per_joint=2)
    assert m.num_streams == 1 and m.plan.nstreams == 1
    ref = m.predict(x, batch_size=3)
    same = lambda got, rows: all(np.array_equal(g, r[:rows]) for g, r in zip(got, ref)) and len(got) == len(ref)
    for ns in (2, 3):
        m.num_streams = ns                                # re-plans
        assert 2 <= m.plan.nstreams <= ns and len({s.stream for s in m.plan.steps}) == m.plan.nstreams
        assert same(m.predict(x, batch_size=3), 3)
        assert same(m.predict(x, batch_size=3), 3)                      # second call: hipGraph replay
        m.executor.use_graph = False
        assert same(m.predict(x[:2], batch_size=2), 2)                  # eager multi-stream launch
    sp, _, _, _ = _spnet(4, 'pa17j3d', 60, 2, [1, 2], 192)
    clips = np.random.default_rng(11).uniform(-1, 1, (1, 4, 256, 256, 3)).astype(np.float32)
    one = sp.predict(clips, batch_size=1)
    sp.num_streams = 2
    for a, b in zip(one, sp.predict(clips, batch_size=1)):
        assert np.array_equal(a, b)
    # [r05] the 'tail' policy of the latency regime: the action stream on a second stream, re-ordered by readiness
    sp.stream_policy = 'tail'
    assert sp.plan.nstreams == 2 and all(len(s.wait) <= 1 for s in sp.plan.steps)
    for rep in range(2):                                            # capture, then replay
        for a, b in zip(one, sp.predict(clips, batch_size=1)):
            assert np.array_equal(a, b)


def test_heat_map_head_rules_are_bit_identical(hip_lib, cuda, monkeypatch):
    """[r05] Planner rules R4b (`pred_maps` written straight into the slab of concatenate([fw_maps, pred_maps]) although the
    channel soft-max reads it too: spnet.py:24-48) and R10 (`_fw_maps` and `_conv1`, two 1x1 convolutions of one tensor into
    neighbouring slabs, as ONE launch over both weight matrices): fewer launches, not one bit moved -- 2-D and 3-D SPNet,
    with and without the replica head; weights changed after the first predict reach the merged launch."""
    from deephar_amd import weights
    monkeypatch.setenv('DEEPHAR_MERGE_SIBLINGS', '0')      # (rule R10c has its own test; here: R4b and R10 alone)
    for layout, nact, rep in (('pa16j2d', 15, True), ('pa17j3d', 60, False)):
        clips = np.random.default_rng(21).uniform(-1, 1, (2, 4, 128, 128, 3)).astype(np.float32)
        monkeypatch.setenv('DEEPHAR_MERGE_HEADS', '0')
        monkeypatch.setenv('DEEPHAR_CONCAT_SHARED', '0')
        base, _, _, _ = _spnet(4, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        want = base.predict(clips, batch_size=2)
        nbase = len(base.plan.steps)
        monkeypatch.setenv('DEEPHAR_MERGE_HEADS', '1')
        monkeypatch.setenv('DEEPHAR_CONCAT_SHARED', '1')
        m, _, _, _ = _spnet(4, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        merged = [s for s in m.plan.steps if s.kind == 'conv' and '+' in (s.name or '') and 'heatmaps' in s.name]
        # five heads: one copy and one conv less each; [r06] the switch also governs rule R10b (two launches less per action head)
        assert len(merged) == 5 and len(m.plan.steps) == nbase - 10 - 12
        for a, b in zip(want, m.predict(clips, batch_size=2)):
            assert np.array_equal(a, b)
        # a weight of ONE part changed after the first predict: the merged launch must see it
        layer = next(l for n in m._nodes for l in n.layers.values() if l.name.endswith('pb1_heatmaps_fw_maps'))
        layer.params[0].set(0.5 * layer.params[0].value)
        next(l for n in base._nodes for l in n.layers.values() if l.name == layer.name).params[0].set(layer.params[0].value)
        for a, b in zip(base.predict(clips, batch_size=2), m.predict(clips, batch_size=2)):
            assert np.array_equal(a, b)


def test_merged_kxk_siblings_are_bit_identical(hip_lib, cuda, monkeypatch):
    """[r06] Planner rule R10b: the three bare convolutions over the (T, J) plane that open every action head (spnet.py:
    109-112: 3x1, 3x3, 3x5 of one tensor, concatenated) as ONE 3x5 convolution whose smaller kernels sit centred between
    zero taps -- two launches fewer per head, not one bit moved (the extra products are exact zeros, the taps keep their
    order); weights set on one part after the first predict reach the merged launch."""
    clips = np.random.default_rng(31).uniform(-1, 1, (2, 8, 128, 128, 3)).astype(np.float32)
    for layout, nact, rep in (('pa16j2d', 15, True), ('pa17j3d', 60, False)):
        monkeypatch.setenv('DEEPHAR_MERGE_KXK', '0')
        base, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        want = base.predict(clips, batch_size=2)
        nbase = len(base.plan.steps)
        monkeypatch.setenv('DEEPHAR_MERGE_KXK', '1')
        m, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        merged = [s for s in m.plan.steps if s.kind == 'conv' and (s.name or '').count('p_conv0') == 3]
        assert len(merged) == 6 and len(m.plan.steps) == nbase - 12
        assert all((s.attrs['kh'], s.attrs['kw'], s.attrs['Cout']) == (3, 5, 70) for s in merged)
        for a, b in zip(want, m.predict(clips, batch_size=2)):
            assert np.array_equal(a, b)
        layer = next(l for n in m._nodes for l in n.layers.values() if l.name.endswith('act1_action_p_conv0a'))
        layer.params[0].set(0.5 * layer.params[0].value)
        next(l for n in base._nodes for l in n.layers.values() if l.name == layer.name).params[0].set(layer.params[0].value)
        for a, b in zip(base.predict(clips, batch_size=2), m.predict(clips, batch_size=2)):
            assert np.array_equal(a, b)


def test_siblings_merged_into_joint_buffers_are_bit_identical(hip_lib, cuda, monkeypatch):
    """[r06] Planner rule R10c: sibling 1x1 convolutions of one tensor whose results are separate tensors -- shortcut and
    first convolution of the action heads' 'normal' residual units (common.py:33-52), the replica heat-map head beside the
    forward / heat-map pair (spnet.py:32-38) -- as ONE launch into a joint buffer, with a per-column affine where only one
    part has a BatchNormalization and the odd ReLU moved into its reader's prologue: fewer launches, not one bit moved;
    2-D replica model and 3-D model, one and two streams; weights set after the first predict reach the merged launch."""
    from deephar_amd.engine.planner import ConcatAffine
    monkeypatch.setenv('DEEPHAR_POOL_SEGMENTS', '0')          # (rule R14 needs the merged unit as the concatenation's ONLY reader:
                                                              #  it would take six more launches out of the merged plans alone)
    clips = np.random.default_rng(33).uniform(-1, 1, (2, 8, 128, 128, 3)).astype(np.float32)
    for layout, nact, rep in (('pa16j2d', 15, True), ('pa17j3d', 60, False)):
        monkeypatch.setenv('DEEPHAR_MERGE_SIBLINGS', '0')
        base, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        want = base.predict(clips, batch_size=2)
        nbase = len(base.plan.steps)
        monkeypatch.setenv('DEEPHAR_MERGE_SIBLINGS', '1')
        for streams, policy in ((1, 'list'), (2, 'tail')):
            m, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
            m.num_streams, m.stream_policy = streams, policy
            res_units = [s for s in m.plan.steps if isinstance(s.params.get('post_affine'), ConcatAffine)]
            heads3 = [s for s in m.plan.steps if s.kind == 'conv' and 'replica' in (s.name or '') and '+' in s.name]
            assert len(res_units) == 12 and len(heads3) == (6 if rep else 0)       # six action heads x (r1, r2); six replica heads
            assert len(m.plan.steps) == nbase - 12 - len(heads3)
            for a, b in zip(want, m.predict(clips, batch_size=2)):
                assert np.array_equal(a, b), (layout, streams)
        layer = next(l for n in m._nodes for l in n.layers.values() if l.name.endswith('act1_action_r1_conv1'))
        layer.params[0].set(0.5 * layer.params[0].value)
        next(l for n in base._nodes for l in n.layers.values() if l.name == layer.name).params[0].set(layer.params[0].value)
        bn = next(l for n in m._nodes for l in n.layers.values() if l.name.endswith('act1_action_r1_bn2'))
        new_beta = bn.params[[p.role for p in bn.params].index('beta')]
        new_beta.set(new_beta.value + 0.25)
        bb = next(l for n in base._nodes for l in n.layers.values() if l.name == bn.name)
        bb.params[[p.role for p in bb.params].index('beta')].set(new_beta.value)
        for a, b in zip(base.predict(clips, batch_size=2), m.predict(clips, batch_size=2)):
            assert np.array_equal(a, b)


def test_resample_on_load_rule_is_bit_identical(hip_lib, cuda, monkeypatch):
    """[r06] Planner rule R12: the up-sampling in front of the action head's conv3 and the max+-min pooling in front of its
    conv2h (spnet.py:77-91) are not written out -- the skinny-conv kernel resamples while it loads: two launches fewer per
    head, not one bit moved (2-D and 3-D model)."""
    clips = np.random.default_rng(35).uniform(-1, 1, (2, 8, 128, 128, 3)).astype(np.float32)
    for layout, nact, rep in (('pa16j2d', 15, True), ('pa17j3d', 60, False)):
        monkeypatch.setenv('DEEPHAR_RESAMPLE_ON_LOAD', '0')
        base, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        want = base.predict(clips, batch_size=2)
        nbase = len(base.plan.steps)
        monkeypatch.setenv('DEEPHAR_RESAMPLE_ON_LOAD', '1')
        m, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        modes = sorted(s.attrs['x_resample'] for s in m.plan.steps if s.kind == 'conv' and s.attrs.get('x_resample'))
        assert modes == [1] * 5 + [3] * 6 and len(m.plan.steps) == nbase - 11, (modes, len(m.plan.steps), nbase)
        for a, b in zip(want, m.predict(clips, batch_size=2)):
            assert np.array_equal(a, b), layout


def test_sibling_pools_merged_are_bit_identical(hip_lib, cuda, monkeypatch):
    """[r06] Planner rule R13: the action head's two poolings (pose features, appearance features: spnet.py:126-133) read one
    joint buffer their producers fill and run as ONE launch into the concatenation: one launch less per head, not one bit
    moved -- 2-D replica model, one and two streams, 8-frame clips (window stride (1, 2)) and 16-frame clips (stride (2, 2));
    the 17-joint 3-D model zero-pads its features to 20 joints in front of the pooling (spnet.py:124-132): ZeroPadding2D writes
    dense rows, the rule leaves those heads alone."""
    monkeypatch.setenv('DEEPHAR_POOL_SEGMENTS', '0')          # (rule R14 would read the joint pooling through its consumer)
    for frames, seed in ((8, 41), (16, 43)):
        clips = np.random.default_rng(seed).uniform(-1, 1, (2, frames, 128, 128, 3)).astype(np.float32)
        for layout, nact, rep in (('pa16j2d', 15, True), ('pa17j3d', 60, False)):
            monkeypatch.setenv('DEEPHAR_MERGE_POOLS', '0')
            base, _, _, _ = _spnet(frames, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
            want = base.predict(clips, batch_size=2)
            nbase = len(base.plan.steps)
            monkeypatch.setenv('DEEPHAR_MERGE_POOLS', '1')
            for streams, policy in ((1, 'list'), (2, 'tail')):
                m, _, _, _ = _spnet(frames, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
                m.num_streams, m.stream_policy = streams, policy
                joint = [s for s in m.plan.steps if s.kind == 'pool' and '+' in (s.name or '')]
                want_joint = 6 if layout == 'pa16j2d' else 0
                assert len(joint) == want_joint and len(m.plan.steps) == nbase - want_joint, (len(joint), len(m.plan.steps), nbase)
                assert all(s.ins['x'].C == 320 and s.ins['x'].ld == 320 and s.outs['y'].C == 320 for s in joint)
                for a, b in zip(want, m.predict(clips, batch_size=2)):
                    assert np.array_equal(a, b), (frames, layout, streams)


def test_paired_skinny_convs_are_bit_identical(hip_lib, cuda, monkeypatch):
    """[r06] BoundPlan._pair_skinny_convs: at a couple of clips per call two independent skinny-conv layers that follow each
    other on a stream (an action head's residual unit on the pose features beside v_conv0 on the appearance features,
    spnet.py:113-133) are ONE launch (dh_conv2d_pair_f32) -- 2-D replica model and 3-D model, one and two streams: the same
    bits as with the switch off; at a throughput batch nothing is paired; the exported plan replays the paired form."""
    clips = np.random.default_rng(47).uniform(-1, 1, (2, 8, 128, 128, 3)).astype(np.float32)
    for layout, nact, rep in (('pa16j2d', 15, True), ('pa17j3d', 60, False)):
        monkeypatch.setenv('DEEPHAR_PAIR_CONVS', '0')
        base, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        want = base.predict(clips, batch_size=2)
        assert next(iter(base.executor.bound.values())).paired == []
        monkeypatch.setenv('DEEPHAR_PAIR_CONVS', '1')
        for streams, policy in ((1, 'list'), (2, 'tail')):
            m, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
            m.num_streams, m.stream_policy = streams, policy
            got = m.predict(clips, batch_size=2)
            bp = next(iter(m.executor.bound.values()))
            # (the one-stream launch order puts every convolution behind the launch it depends on: nothing to pair there;
            #  the two-stream order interleaves the heads)
            if streams == 2:
                assert len(bp.paired) >= 1, (layout, streams, bp.paired)
            print('paired launches', layout, streams, len(bp.paired))
            for i, k in bp.paired:
                a, b = bp.calls[i][2], bp.calls[k][2]
                assert i < k and a.stream == b.stream and k in bp.noop_calls and a.kind == b.kind == 'conv'
            for x, y in zip(want, got):
                assert np.array_equal(x, y), (layout, streams, policy)
    big = np.random.default_rng(48).uniform(-1, 1, (72, 8, 128, 128, 3)).astype(np.float32)
    m.predict(big, batch_size=72)
    assert len(m.executor.bound[72].paired) < len(bp.paired)      # (rows of the bound batch decide: beyond the latency regime, apart)
    from deephar_amd.engine import serialize
    assert 'dh_conv2d_pair_f32' in serialize.FUNCTIONS


def test_pool_read_by_segmented_conv_is_bit_identical(hip_lib, cuda, monkeypatch):
    """[r06] Planner rule R14: the pooled features of an action head (one joint launch after R13) are never written -- the
    head's second residual unit reads concatenate([pool(U), xa]) through dh_conv2d_seg_f32: one launch less per head, not one
    bit moved; 2-D replica model and 17-joint 3-D model (ZeroPadding2D in front of its poolings: R13 does not apply, the first
    pooling alone is read through) at 8- and 16-frame clips (window stride (1, 2) / (2, 2)), one and two streams."""
    for frames, seed in ((8, 51), (16, 53)):
        clips = np.random.default_rng(seed).uniform(-1, 1, (2, frames, 128, 128, 3)).astype(np.float32)
        for layout, nact, rep in (('pa16j2d', 15, True), ('pa17j3d', 60, False)):
            monkeypatch.setenv('DEEPHAR_POOL_SEGMENTS', '0')
            base, _, _, _ = _spnet(frames, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
            want = base.predict(clips, batch_size=2)
            nbase = len(base.plan.steps)
            monkeypatch.setenv('DEEPHAR_POOL_SEGMENTS', '1')
            for streams, policy in ((1, 'list'), (2, 'tail')):
                m, _, _, _ = _spnet(frames, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
                m.num_streams, m.stream_policy = streams, policy
                seg = [s for s in m.plan.steps if s.kind == 'conv' and s.attrs.get('seg')]
                # 2-D: the joint pooling of both feature sets (320 channels); 3-D: the pooling of the (zero-padded) pose features
                # alone -- the appearance features' pooling still writes its run of the concatenation, which is then part of x2
                csplit = 320 if layout == 'pa16j2d' else 160
                assert len(seg) == 6 and len(m.plan.steps) == nbase - 6, (layout, len(seg), len(m.plan.steps), nbase)
                assert all(s.attrs['seg'] == dict(c_split=csplit, pool_sh=1 if frames == 8 else 2) for s in seg)
                assert sum(1 for s in seg if 'x2' in s.ins) == (5 if layout == 'pa16j2d' else 6)      # the first 2-D head has no xa
                for a, b in zip(want, m.predict(clips, batch_size=2)):
                    assert np.array_equal(a, b), (frames, layout, streams)


def test_two_models_predict_from_two_threads(hip_lib, cuda):
    """[r06] Every Model of a process launches on ONE compute stream (engine/executor.py: shared_stream -- a stream per Model
    ran into ROCm's four hardware queues, profiles/r06_hw_queue_collision.md); a lock per stream keeps one thread's hipGraph
    capture from swallowing another thread's launches.  Two different models, bound, tuned, captured and run from two threads
    at the same time -- a 2-D ReceptionNet through predict on host arrays, a two-stream SPNet on clips: the bits of the same
    calls made one after the other."""
    import threading
    rng = np.random.default_rng(61)
    x = rng.uniform(-1, 1, (12, 256, 256, 3)).astype(np.float32)
    clips = rng.uniform(-1, 1, (4, 8, 128, 128, 3)).astype(np.float32)

    def rec():
        m, _ = _build(2, 2, 16, num_context_per_joint=2)
        return m

    def sp():
        m, _, _, _ = _spnet(8, 'pa16j2d', 15, 2, [1, 2], 160, replica=True, res=128)
        m.num_streams, m.stream_policy = 2, 'tail'
        return m
    want_a = rec().predict(x, batch_size=4)
    want_b = sp().predict(clips, batch_size=2)
    ma, mb = rec(), sp()                                  # fresh models: binding, tuning and capture happen inside the threads
    got, errs = {}, []

    def run(tag, m, arr, bs):
        try:
            for _ in range(3):
                got[tag] = m.predict(arr, batch_size=bs)
        except Exception as e:                            # noqa: BLE001  (reported below, with the thread's tag)
            errs.append((tag, repr(e)))
    ta = threading.Thread(target=run, args=('a', ma, x, 4))
    tb = threading.Thread(target=run, args=('b', mb, clips, 2))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    assert ma.executor.stream is mb.executor.stream       # the point of the test: they DO share the stream
    for w, g in zip(want_a, got['a']):
        assert np.array_equal(w, g)
    for w, g in zip(want_b, got['b']):
        assert np.array_equal(w, g)


def test_pose_times_confidence_folded_into_the_read_out(hip_lib, cuda, monkeypatch):
    """[r06] multiply([p, c]) in front of an action head (spnet.py:108) on a replica read-out whose coordinates and confidence
    have no other reader is folded into the soft-argmax launch (dh_sam_args.xy_times_conf): one launch less per head, the same
    bits (the same two fp32 factors, one multiplication); the 3-D model (pose = concat(xy, z), also a model output) keeps its
    multiply."""
    clips = np.random.default_rng(37).uniform(-1, 1, (2, 8, 128, 128, 3)).astype(np.float32)
    for layout, nact, rep, folded in (('pa16j2d', 15, True, 6), ('pa17j3d', 60, False, 0)):
        monkeypatch.setenv('DEEPHAR_FOLD_POSE_MUL', '0')
        base, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        want = base.predict(clips, batch_size=2)
        monkeypatch.setenv('DEEPHAR_FOLD_POSE_MUL', '1')
        m, _, _, _ = _spnet(8, layout, nact, 2, [1, 2], 160, replica=rep, res=128)
        assert sum(1 for s in m.plan.steps if s.kind == 'sam' and s.attrs.get('xy_times_conf')) == folded
        assert len(m.plan.steps) == len(base.plan.steps) - folded
        for a, b in zip(want, m.predict(clips, batch_size=2)):
            assert np.array_equal(a, b), layout


def test_grouped_launches_are_bit_identical(hip_lib, cuda, monkeypatch):
    """[r06] BoundPlan.group_launches: at a couple of clips per call every (1x1 shortcut convolution, depthwise convolution)
    pair of SPNet's down- / up-scaling units is ONE launch (dh_conv2d_dw_group_f32) -- 2-D replica model, one and two
    streams: the same bits as with the switch off, and the planned pairs are really merged; at a throughput batch nothing
    is merged."""
    clips = np.random.default_rng(29).uniform(-1, 1, (2, 8, 128, 128, 3)).astype(np.float32)
    monkeypatch.setenv('DEEPHAR_GROUP_LAUNCHES', '0')
    base, _, _, _ = _spnet(8, 'pa16j2d', 15, 2, [1, 2], 160, replica=True, res=128)
    want = base.predict(clips, batch_size=2)
    assert next(iter(base.executor.bound.values())).grouped == 0
    monkeypatch.setenv('DEEPHAR_GROUP_LAUNCHES', '1')
    for streams, policy in ((1, 'list'), (2, 'tail')):
        m, _, _, _ = _spnet(8, 'pa16j2d', 15, 2, [1, 2], 160, replica=True, res=128)
        m.num_streams, m.stream_policy = streams, policy
        got = m.predict(clips, batch_size=2)
        bp = next(iter(m.executor.bound.values()))
        units = sum(1 for s in m.plan.steps if s.kind == 'conv' and (s.name or '').endswith('_r0_shortcut_conv'))
        # 3 down- + 3 up-scaling units; at 128 px three of them have their depthwise half on maps of >= 8 columns (the LDS
        # kernel the group is built from), the others run as two launches
        assert units == 6 and bp.grouped == 3, (units, bp.grouped)
        for a, b in zip(want, got):
            assert np.array_equal(a, b), (streams, policy)
    big = np.random.default_rng(30).uniform(-1, 1, (24, 8, 128, 128, 3)).astype(np.float32)
    m.predict(big, batch_size=24)
    assert m.executor.bound[24].grouped < bp.grouped                        # beyond the latency regime the pairs stay apart
    # the C-level plan of the grouped form: the same bits from a blob (one function id more, the no-op entries left out)
    import tempfile
    from deephar_amd.engine import serialize
    blob = serialize.dump_plan(m, 2)
    assert 'dh_conv2d_dw_group_f32' in serialize.FUNCTIONS and len(blob) > 0


def test_speed_protocol_truncated_models(hip_lib, cuda):
    """exp/pennaction/eval_speed2d.py:60-68: `Model(full_model.input, full_model.outputs[2*b:2*b+2])` for every prediction
    block b, each with one warm-up `predict(x[0:1])` and then `predict(x, batch_size=2)`.  On a reduced configuration (two
    pyramids, 128 px; actions on both, pose_replica): every truncated model returns exactly the two outputs of the full
    model it was cut from -- bit for bit, also with the latency-regime engine setting (two streams, 'tail' policy), and the
    plan of a pose-only truncation holds no action-stream launch."""
    from deephar_amd import Model
    full, _, _, _ = _spnet(8, 'pa16j2d', 15, 2, [1, 2], 160, replica=True, res=128)
    x = np.random.default_rng(23).uniform(-1, 1, (4, 8, 128, 128, 3)).astype(np.float32)
    want = full.predict(x, batch_size=2)
    nb = len(full.outputs) // 2
    assert nb == 6 and len(want) == 12
    for b in range(nb):
        for streams, policy in ((1, 'list'), (2, 'tail')):
            m = Model(full.input, full.outputs[2 * b:2 * b + 2])
            m.num_streams, m.stream_policy = streams, policy
            m.predict(x[0:1])                                        # "Warming up the new model."
            got = m.predict(x, batch_size=2)
            assert len(got) == 2
            for a, w in zip(got, want[2 * b:2 * b + 2]):
                assert a.shape == w.shape and np.array_equal(a, w), (b, streams)
            if b < nb // 2:                                          # outputs 2b, 2b + 1 are poses: no action head needed
                assert not any('action' in (s.name or '') for s in m.plan.steps)
            if streams == 2 and b == nb - 1:
                assert m.plan.nstreams == 2


def test_predict_accepts_float64_and_partial_batches(hip_lib, cuda):
    m, _ = _build(2, 1, 16, num_context_per_joint=2)
    x = np.random.default_rng(6).uniform(-1, 1, (5, 256, 256, 3))     # float64, like loader.py:139-140
    a = m.predict(x, batch_size=2)
    b = m.predict(x.astype(np.float32), batch_size=5)
    assert a.dtype == np.float32 and a.shape == (5, 16, 3) and np.array_equal(a, b)
    with pytest.raises(ValueError):
        m.predict(np.zeros((2, 128, 128, 3), np.float32))


def test_weights_changed_after_first_predict_are_used(hip_lib, cuda):
    """device weights must follow Param.set / Layer.set_weights / Model.set_weights made AFTER the first
    predict, also for batch sizes that were bound before the change and for a second Model sharing the layers."""
    from deephar_amd import Model, weights
    m, _ = _build(2, 1, 16, num_context_per_joint=2)
    x = np.random.default_rng(8).uniform(-1, 1, (2, 256, 256, 3)).astype(np.float32)
    first = m.predict(x, batch_size=2)
    tail = Model(m.input, m.outputs[-1])              # shares every layer with m, own executor
    assert np.array_equal(tail.predict(x, batch_size=2), first)
    weights.init_synthetic(m, seed=5)                 # new values for every Param, same objects
    second = m.predict(x, batch_size=2)               # batch size 2 was bound with the OLD weights
    assert not np.array_equal(first, second)
    o64, _ = _oracle(weights.as_dict(m), x, 2, 1, 16, torch.float64, num_context_per_joint=2)
    o32, _ = _oracle(weights.as_dict(m), x, 2, 1, 16, torch.float32, num_context_per_joint=2)
    _check('reloaded.xy', second[..., :2], o32[0][..., :2], o64[0][..., :2], PX_TOL)
    assert np.array_equal(tail.predict(x, batch_size=2), second)     # the sharing model sees the change too
    assert np.array_equal(m.predict(x[:1], batch_size=1), second[:1])  # a newly bound batch size as well
    vals = [v.copy() for v in m.get_weights()]
    vals[0] = vals[0] * np.float32(0.5)
    m.set_weights(vals)
    assert not np.array_equal(m.predict(x, batch_size=2), second)


def _merge(pose_dim, T, joints, blocks, **kw):
    from deephar_amd import graph, weights
    from deephar_amd.models import reception, action
    graph.reset_naming()
    if pose_dim == 2:
        pe = reception.build((256, 256, 3), joints, dim=2, num_blocks=blocks, num_context_per_joint=2, ksize=(5, 5))
    else:
        pe = reception.build((256, 256, 3), joints, dim=3, num_blocks=blocks, depth_maps=kw.get('depth_maps', 8),
                             ksize=(5, 5))
    m = action.build_merge_model(pe, kw.pop('num_actions', 15), (256, 256, 3), T, joints, blocks, pose_dim=pose_dim,
                                 output_poses=True, **kw)
    weights.init_synthetic(m, seed=0)
    return m, weights.as_dict(m)


@pytest.mark.parametrize('pose_dim,joints,version', [(2, 16, 'v1'), (3, 20, 'v2')])
def test_merge_action_model_parity(pose_dim, joints, version, hip_lib, cuda):
    """cfg 4 model family (action.build_merge_model): poses within 1e-3 px, action soft-max scores close and
    identical arg-max labels on every one of the 9 heads (p1..p4, v1..v4, m)."""
    from oracle import action as oact
    T, blocks, nact = 8, 2, 15
    kw = dict(pose_net_version=version, num_actions=nact)
    if pose_dim == 3:
        kw['depth_maps'] = 8
    m, wd = _merge(pose_dim, T, joints, blocks, **dict(kw))
    x = np.random.default_rng(7).uniform(-1, 1, (2, T, 256, 256, 3)).astype(np.float32)
    hip = m.predict(x, batch_size=2)
    okw = dict(pose_dim=pose_dim, pose_net_version=version, output_poses=True)
    if pose_dim == 3:
        okw.update(depth_maps=8, num_context_per_joint=0)
    o32 = oact.forward_merge(wd, x, nact, joints, blocks, dtype=torch.float32, **okw)
    o64 = oact.forward_merge(wd, x, nact, joints, blocks, dtype=torch.float64, **okw)
    assert [h.shape for h in hip] == [o.shape for o in o64] and len(hip) == 11
    _check('pose', hip[0], o32[0], o64[0], PX_TOL)
    _check('conf', hip[1], o32[1], o64[1], 1e-5, rel=True)
    for k in range(2, 11):
        _check('action%d' % (k - 1), hip[k], o32[k], o64[k], 1e-5)
        assert np.array_equal(hip[k].argmax(-1), o64[k].argmax(-1)), 'action label differs on head %d' % (k - 1)
        np.testing.assert_allclose(hip[k].sum(-1), 1.0, rtol=1e-5)


def _spnet(T, layout, num_actions, pyramids, action_pyramids, feats, replica=False, calibrate=None, res=256):
    """calibrate: frames [N, T, 256, 256, 3] -> heat-map heads are brought to logit std ~ 6 before the weights are
    read out (paritylog.calibrate_spnet_heads)."""
    from deephar_amd import graph, weights, utils
    from deephar_amd.config import ModelConfig
    from deephar_amd.models import spnet
    graph.reset_naming()
    lay = getattr(utils, layout)
    cfg = ModelConfig((T, res, res, 3), lay, num_actions=[num_actions], num_pyramids=pyramids,
                      action_pyramids=action_pyramids, num_levels=4, pose_replica=replica, num_pose_features=feats,
                      num_visual_features=feats)
    m = spnet.build(cfg)
    weights.init_synthetic(m, seed=0)
    ocfg = dict(num_joints=lay.num_joints, dim=lay.dim, num_actions=[num_actions], num_pyramids=pyramids,
                action_pyramids=action_pyramids, num_levels=4, kernel_size=(5, 5), growth=96, image_div=8,
                num_pose_features=feats, num_visual_features=feats, sam_alpha=1, pose_replica=replica)
    if calibrate is not None:
        paritylog.calibrate_spnet_heads(m, ocfg, calibrate)
    return m, cfg, weights.as_dict(m), ocfg


def spnet_parity(m, cfg, wd, ocfg, x, pyr, apyr, case=None):
    """Shared body of the SPNet parity tests: poses through the conditioned 1e-3 px check, action scores + labels."""
    from deephar_amd.models import spnet
    from oracle import spnet as osp
    hip = m.predict(x, batch_size=len(x))
    t64 = {}
    o32 = osp.forward(wd, x, ocfg, dtype=torch.float32)
    o64 = osp.forward(wd, x, ocfg, dtype=torch.float64, taps=t64)
    npose = spnet.get_num_predictions(pyr, 4)
    nact_out = spnet.get_num_predictions(len(apyr), 4)
    assert len(hip) == npose + nact_out and [h.shape for h in hip] == [o.shape for o in o64]
    dim = ocfg['dim']
    blocks = [k[:-len('/logits')] for k in t64 if k.endswith('/logits')]
    assert len(blocks) == npose
    for k, b in enumerate(blocks):
        std = float(t64[b + '/logits'].std())
        assert 1.0 < std < 30.0, 'heat-map logits of %s are flat or one-hot (std %.2f): vacuous test' % (b, std)
        tol_xy, tol_z, tol_c = paritylog.conditioned_tolerance(t64[b + '/logits'], t64.get(b + '/dlogits'))
        flat = lambda a: a.reshape((-1,) + a.shape[-2:])
        h, a32, a64 = flat(hip[k]), flat(o32[k]), flat(o64[k])
        paritylog.check_conditioned('%s.xy' % b, h[..., :2], a32[..., :2], a64[..., :2], tol_xy, case=case)
        if dim == 3:
            paritylog.check_conditioned('%s.z' % b, h[..., 2], a32[..., 2], a64[..., 2], tol_z, case=case)
        paritylog.check_conditioned('%s.conf' % b, h[..., dim], a32[..., dim], a64[..., dim], tol_c, case=case, px=False)
    for k in range(npose, npose + nact_out):
        _check('action%d' % (k - npose), hip[k], o32[k], o64[k], 1e-5, case=case)
        assert np.array_equal(hip[k].argmax(-1), o64[k].argmax(-1))
    return hip


@pytest.mark.parametrize('T,layout,nact,pyr,apyr,feats,replica', [
    (8, 'pa17j3d', 60, 2, [1, 2], 192, False),     # exp/ntu/eval_ntu_multitask.py:35-38 (cfg 5 family), time_stride 1
    (16, 'pa16j2d', 15, 2, [2], 160, False),       # Penn-like 2-D, T>=16 -> time_stride 2, action only on pyramid 2
    (8, 'pa16j2d', 15, 6, [5, 6], 160, True),      # exp/pennaction/eval_penn_multitask.py:36-40 as shipped: 6 pyramids,
                                                   # actions on 5 and 6, pose_replica=True (18 pose + 6 action outputs)
])
def test_spnet_multitask_parity(T, layout, nact, pyr, apyr, feats, replica, hip_lib, cuda):
    """SPNet pose + action outputs vs the oracle; split_model() slices the same numbers."""
    from deephar_amd.models import spnet, split_model
    x = np.random.default_rng(11).uniform(-1, 1, (1, T, 256, 256, 3)).astype(np.float32)
    m, cfg, wd, ocfg = _spnet(T, layout, nact, pyr, apyr, feats, replica=replica, calibrate=x)
    hip = spnet_parity(m, cfg, wd, ocfg, x, pyr, apyr)
    npose = spnet.get_num_predictions(pyr, 4)
    nact_out = spnet.get_num_predictions(len(apyr), 4)
    pose_model, act_model = split_model(m, cfg)
    a = act_model.predict(x, batch_size=1)
    a = a if isinstance(a, list) else [a]
    for k in range(nact_out):
        assert np.array_equal(a[k], hip[npose + k])
    last = pose_model.predict(x, batch_size=1)[-1]
    assert np.array_equal(last, hip[npose - 1])


def test_spnet_replica_feeds_only_the_action_stream(hip_lib, cuda):
    """pose_replica=True (spnet.py:36-38,160,216,224): the '<pb>_heatmaps_conv1_replica' maps drive the action heads
    and nothing else -- perturbing a replica kernel changes action scores but no pose output bit; perturbing the
    matching '_conv1' kernel changes the poses."""
    x = np.random.default_rng(12).uniform(-1, 1, (1, 4, 256, 256, 3)).astype(np.float32)
    m, cfg, _, _ = _spnet(4, 'pa16j2d', 15, 2, [2], 160, replica=True)
    base = m.predict(x, batch_size=1)
    npose = 6
    names = {l.name: l for n in m._nodes for l in n.layers.values()}
    assert 'up2_pb2_heatmaps_conv1_replica' in names and 'dp1_pb1_heatmaps_conv1_replica' not in names
    lay = names['up2_pb2_heatmaps_conv1_replica']
    lay.set_weights([lay.get_weights()[0] * np.float32(1.5)])
    pert = m.predict(x, batch_size=1)
    for k in range(npose):
        assert np.array_equal(base[k], pert[k]), 'pose output %d moved with a replica kernel' % k
    assert any(not np.array_equal(base[k], pert[k]) for k in range(npose, len(base)))
    lay = names['up2_pb2_heatmaps_conv1']
    lay.set_weights([lay.get_weights()[0] * np.float32(1.5)])
    pert2 = m.predict(x, batch_size=1)
    assert not np.array_equal(pert2[3], pert[3])


def test_frame_sharded_stages_match_full_model(hip_lib, cuda):
    """cfg 4 (frame-shard + all-gather): running the FRAME stage on T/2 frames per 'rank' and the HEAD stage on
    the gathered [T, J, C] tensors reproduces the un-sharded clip model bit for bit (frames are independent and
    every kernel is batch-size invariant), for the merge model and for SPNet."""
    from deephar_amd import parallel
    for builder in ('merge', 'spnet'):
        if builder == 'merge':
            m, _ = _merge(2, 8, 16, 2, num_actions=15)
        else:
            m, _, _, _ = _spnet(8, 'pa17j3d', 60, 2, [1, 2], 192)
        clips = np.random.default_rng(21).uniform(-1, 1, (2, 8, 256, 256, 3)).astype(np.float32)
        full = m.predict(clips, batch_size=2)
        fm, hm, info = parallel.split_frames(m, 2)
        from deephar_amd import weights
        tl = info['Tl']
        packed = np.concatenate([fm.predict(clips[:, r * tl:(r + 1) * tl], batch_size=2) for r in range(2)], axis=1)
        parts = [np.ascontiguousarray(packed[..., off:off + c]) for (_, off, c) in info['cut']]
        head = hm.predict(parts, batch_size=2)
        head = head if isinstance(head, list) else [head]
        for k, ci in info['passthrough'].items():
            assert np.array_equal(parts[ci], full[k]), (builder, 'passthrough', k)
        for k, o in zip(info['head_outputs'], head):
            assert np.array_equal(o, full[k]), (builder, 'head output', k)
        # and the runtime class with a 1-rank group (gloo) drives the same two stages
    import torch.distributed as dist
    import os
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        m, _ = _merge(2, 8, 16, 2, num_actions=15)
        clips = np.random.default_rng(22).uniform(-1, 1, (1, 8, 256, 256, 3)).astype(np.float32)
        outs = parallel.ShardedClipModel(m).predict(clips)
        ref = m.predict(clips, batch_size=1)
        assert all(np.array_equal(a, b) for a, b in zip(outs, ref))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('tag', ['rec2d', 'rec3d', 'merge2d', 'merge3d', 'spnet3d', 'spnet2d', 'spnet2dr'])
def test_hip_matches_reference_code_goldens(tag, hip_lib, cuda):
    """HIP engine vs the committed golden vectors that were computed by the reference's OWN model code
    (tests/golden/make_reference_golden.py): coordinates within a flat 1e-3 px of the reference code's fp64 run,
    identical arg-max action labels.  The three SPNet goldens here were made on per-pixel-noise inputs with
    un-calibrated heads (multi-modal maps, |logit| up to 100) and are kept as a STRESS case under
    paritylog.conditioned_tolerance; SPNet at the flat bar is tests/test_gpu_spnet_flat.py."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refgolden import build_case, golden
    m, x, run = build_case(tag)
    g32, g64 = golden(tag)
    hip = m.predict(x.astype(np.float32), batch_size=len(x))
    hip = hip if isinstance(hip, list) else [hip]
    assert [h.shape for h in hip] == [g.shape for g in g64]
    tols = None
    if tag.startswith('spnet'):
        # the golden weights are init_synthetic's un-calibrated ones (logit std up to ~18 on the coarse levels):
        # conditioning-aware tolerance from the fp64 oracle's logits (paritylog.conditioned_tolerance); the oracle
        # agrees with these goldens to 1e-9 (tests/test_reference_golden.py)
        t64 = {}
        run(torch.float64, taps=t64)
        tols = [paritylog.conditioned_tolerance(t64[k], t64.get(k[:-len('/logits')] + '/dlogits'))
                for k in t64 if k.endswith('/logits')]
    for k, (h, a, b) in enumerate(zip(hip, g32, g64)):
        is_scores = b.ndim == 2 and tag.startswith(('merge', 'spnet'))
        is_maps = b.ndim == 4 and tag == 'rec3d'
        if is_scores:
            _check('%s.act%d' % (tag, k), h, a, b, 1e-5)
            assert np.array_equal(h.argmax(-1), b.argmax(-1))
        elif is_maps:
            _check('%s.maps%d' % (tag, k), h, a, b, 2e-5, rel=True)
        elif tols is not None:
            flat = lambda v: v.reshape((-1,) + v.shape[-2:])
            dim = b.shape[-1] - 1
            txy, tz, tc = tols[k]
            paritylog.check_conditioned('%s.out%d.xy' % (tag, k), flat(h)[..., :2], flat(a)[..., :2], flat(b)[..., :2], txy)
            if dim == 3:
                paritylog.check_conditioned('%s.out%d.z' % (tag, k), flat(h)[..., 2], flat(a)[..., 2], flat(b)[..., 2], tz)
            paritylog.check_conditioned('%s.out%d.c' % (tag, k), flat(h)[..., dim], flat(a)[..., dim], flat(b)[..., dim], tc,
                                        px=False)
        else:
            _check('%s.out%d' % (tag, k), h, a, b, PX_TOL if b.shape[-1] != 1 else 1e-5, rel=(b.shape[-1] == 1))


@pytest.mark.parametrize('tag', ['rec2d_8', 'rec3d_8', 'merge2d_16'])
def test_hip_matches_reference_code_goldens_at_real_size(tag, hip_lib, cuda):
    """[r05] HIP engine vs golden vectors of the reference's OWN model code at the REAL size of BASELINE configs[1..3]
    (tests/golden/make_reference_golden.py --real): 8-block ReceptionNet 2-D / 3-D at 256 px within a flat 1e-3 px of the
    reference code's fp64 run; the merge model as exp/pennaction/eval_penn_ar_pe_merge.py:51-57 builds it (T = 16, 4 blocks):
    identical arg-max labels on all nine heads.  No oracle in the loop: the golden IS the reference code's output.
    (SPNet-NTU at T = 32: tests/test_gpu_spnet_flat.py, same file of goldens.)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refgolden import build_case, golden
    m, x, _ = build_case(tag)
    g32, g64 = golden(tag)
    hip = m.predict(x.astype(np.float32), batch_size=len(x))
    hip = hip if isinstance(hip, list) else [hip]
    assert [h.shape for h in hip] == [g.shape for g in g64]
    for k, (h, a, b) in enumerate(zip(hip, g32, g64)):
        if tag == 'merge2d_16':
            _check('%s.act%d' % (tag, k), h, a, b, 1e-5)
            assert np.array_equal(h.argmax(-1), b.argmax(-1)), 'action label differs on head %d' % k
        elif tag == 'rec3d_8':          # [N, 17, 4] = xyz + visibility (concat_pose_confidence=True)
            _check('%s.xyz%d' % (tag, k), h[..., :3], a[..., :3], b[..., :3], PX_TOL)
            _check('%s.vis%d' % (tag, k), h[..., 3:], a[..., 3:], b[..., 3:], 1e-6)
        else:                           # pose [N, 16, 2], visibility [N, 16, 1] per block
            _check('%s.out%d' % (tag, k), h, a, b, PX_TOL if b.shape[-1] != 1 else 1e-5, rel=(b.shape[-1] == 1))


@pytest.mark.parametrize('tag', ['rec2d', 'merge2d', 'spnet2d'])
def test_uint8_frames_equal_host_normalised_frames(tag, hip_lib, cuda):
    """Model.predict on raw uint8 frames (normalisation fused into the first convolution, 4x fewer input bytes)
    is bit-identical to predict on frames normalised on the host the way the reference's loaders do."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refgolden import build_case
    from deephar_amd.utils.transform import normalize_channels
    m, x, _ = build_case(tag)
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, x.shape, dtype=np.uint8)
    host = normalize_channels(frames.astype(np.float32))            # float32 in, float32 arithmetic (transform.py:122)
    assert host.dtype == np.float32
    a = m.predict(frames, batch_size=len(frames))
    b = m.predict(host, batch_size=len(frames))
    a, b = (a if isinstance(a, list) else [a]), (b if isinstance(b, list) else [b])
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    bp = m.executor.bound[(len(frames), repr(1))]
    assert bp.npre == 0 and any(getattr(c[1][0], '_obj', None) is not None and c[1][0]._obj.x_u8 for c in bp.calls
                                if c[2].kind == 'conv')               # really the fused path, not the fallback


def test_ragged_batches_and_chunking(hip_lib, cuda):
    """keras predict semantics: any number of frames, any batch_size (last chunk short), one frame; results do
    not depend on how the frames were chunked (bit-exact)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refgolden import build_case
    m, x, _ = build_case('rec2d')
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (5,) + x.shape[1:]).astype(np.float32)
    ref = m.predict(x, batch_size=5)
    for bs in (1, 2, 3, 4, 64):
        got = m.predict(x, batch_size=bs)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), bs
    one = m.predict(x[3:4])
    for a, b in zip(ref, one):
        assert np.array_equal(a[3:4], b)


@pytest.mark.parametrize('dtype', ['float32', 'uint8', 'float64'])
def test_ragged_tail_staged_in_parallel_slices(dtype, hip_lib, cuda):
    """a last chunk with m < batch_size rows, >= 8 MB of input (so the pinned-staging copy is cut
    into several row slices) and m % slice != 0 -- 27 frames at batch_size 16 leave m = 11 rows = 8.65 MB of float32,
    staged as rows [0:6] and [6:11].  The clamped destination slice used to be [6:12] against a source of [6:11]."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refgolden import build_case
    from deephar_amd.engine import executor as ex
    m, x, _ = build_case('rec2d')
    rng = np.random.default_rng(5)
    if dtype == 'uint8':
        x = rng.integers(0, 256, (27, 256, 256, 3), dtype=np.uint8)
    else:
        x = rng.uniform(-1, 1, (27, 256, 256, 3)).astype(dtype)
    tail = x[16:]
    nsl = max(1, min(ex._STAGING_THREADS, (tail.size * (1 if dtype == 'uint8' else 4)) >> 22, len(tail)))
    if dtype != 'uint8':
        assert nsl >= 2 and len(tail) % -(-len(tail) // nsl) != 0, 'the case no longer exercises a ragged slice'
    got = m.predict(x, batch_size=16)
    ref = m.predict(x[16:], batch_size=11)           # the tail on its own, one slice-free chunk of its own size
    one = m.predict(x[:16], batch_size=16)
    for g, r, o in zip(got, ref, one):
        assert g.shape[0] == 27 and np.array_equal(g[16:], r) and np.array_equal(g[:16], o)


def test_multi_stream_graphs_are_capped(hip_lib, cuda, monkeypatch):
    """hipGraphExecs of multi-stream plans can only be parked, never destroyed (ROCm 7.2), so their
    number per process is capped -- past the cap a multi-stream plan launches eagerly, bit-identically."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refgolden import build_case
    from deephar_amd.engine import executor as ex
    m, x, _ = build_case('rec2d')
    x = x.astype(np.float32)
    ref = m.predict(x, batch_size=2)
    m.num_streams = 2
    monkeypatch.setattr(ex, 'MAX_MULTISTREAM_GRAPHS', ex._MULTISTREAM_GRAPHS)       # the cap is reached: no new graph
    before = ex._MULTISTREAM_GRAPHS
    got = m.predict(x, batch_size=2)
    assert ex._MULTISTREAM_GRAPHS == before
    assert all(bp.graph is None for bp in m.executor.bound.values())
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    monkeypatch.setattr(ex, 'MAX_MULTISTREAM_GRAPHS', before + 1)
    got = m.predict(x[:1], batch_size=1)                                              # a new bound plan: captures
    assert ex._MULTISTREAM_GRAPHS == before + 1
    for a, b in zip(ref, got):
        assert np.array_equal(a[:1], b)


@pytest.mark.parametrize('tag', ['rec2d', 'rec3d', 'merge2d', 'merge3d', 'spnet3d', 'spnet2d', 'spnet2dr'])
def test_hip_matches_real_keras_outputs(tag, hip_lib, cuda):
    """The day tests/golden/keras_outputs.npz exists (produced on a machine with keras==2.1.4 + tensorflow==1.6 by
    the kit of tools/make_keras_parity_kit.py) this pins the remaining restated numerics -- TF-SAME padding, BN
    epsilon, pooling / up-sampling semantics -- against the REAL reference: 1e-3 px on coordinates, identical labels."""
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'keras_outputs.npz')
    if not os.path.exists(path):
        pytest.skip('no keras_outputs.npz yet (see tools/make_keras_parity_kit.py)')
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refgolden import build_case
    ref = np.load(path)
    m, x, _ = build_case(tag)
    hip = m.predict(x.astype(np.float32), batch_size=len(x))
    hip = hip if isinstance(hip, list) else [hip]
    for k, h in enumerate(hip):
        r = ref['%s/%d' % (tag, k)]
        assert h.shape == r.shape
        if r.ndim == 2:                                   # action scores
            assert np.array_equal(h.argmax(-1), r.argmax(-1))
            np.testing.assert_allclose(h, r, atol=1e-4)
        elif r.ndim == 4 and tag == 'rec3d':              # exported heat-maps
            np.testing.assert_allclose(h, r, rtol=1e-3, atol=1e-4)
        else:                                             # two fp32 implementations: a few times 1e-3 px is fp32 noise
            assert float(np.max(np.abs(h - r))) <= 4 * PX_TOL, (tag, k)


def test_pooled_epilogue_plan_is_bit_identical(hip_lib, cuda, monkeypatch):
    """Planner rule R7 (MaxPooling2D written by the producing convolution's epilogue) changes no result bit, in either
    GEMM mode."""
    from deephar_amd.models import reception
    for mode in ('f32', 'bf16x3'):
        outs = {}
        for fuse in ('1', '0'):
            monkeypatch.setenv('DEEPHAR_FUSE_POOL', fuse)
            m, _ = _build(2, 3, 16, num_context_per_joint=2)
            m.gemm_precision = mode
            x = np.random.default_rng(3).uniform(-1, 1, (3, 256, 256, 3)).astype(np.float32)
            outs[fuse] = m.predict(x, batch_size=3)
            n_pool = sum(1 for s in m.plan.steps if s.kind == 'pool' and s.ins['x'].shape[-2] in (16, 32))
            n_fused = sum(1 for s in m.plan.steps if s.kind == 'conv' and 'ypool' in s.outs)
            # three hourglasses x (32-column pool + [r06] the 16-column pool behind it)
            assert (n_fused, n_pool) == ((6, 0) if fuse == '1' else (0, 6)), (mode, fuse, n_fused, n_pool)
        monkeypatch.delenv('DEEPHAR_FUSE_POOL')
        for a, b in zip(outs['1'], outs['0']):
            assert np.array_equal(a, b), mode


def test_keras_h5_weight_files_drive_the_gpu_model(hip_lib, cuda, tmp_path):
    """SURVEY.md 8f rank 1 on the GPU: a Keras-layout .h5 written by save_weights is loaded BY ORDER into a fresh
    ReceptionNet (eval_mpii_singleperson.py:54) and BY NAME into a fresh SPNet (eval_penn_multitask.py:76) whose weights
    were different before; predict then reproduces the source model bit for bit and sits within tolerance of the
    oracle.  The same file re-written by the real libhdf5 (h5py under /opt/conda, every dataset chunked + gzip +
    shuffle) loads to the same bits."""
    import subprocess
    from deephar_amd import weights
    kw = dict(num_context_per_joint=2)
    src, wd = _build(2, 1, 16, **kw)
    x = np.random.default_rng(13).uniform(-1, 1, (2, 256, 256, 3)).astype(np.float32)
    ref = src.predict(x, batch_size=2)
    path = str(tmp_path / 'reception.h5')
    src.save_weights(path)
    dst, _ = _build(2, 1, 16, **kw)
    weights.init_synthetic(dst, seed=3)
    assert not np.array_equal(dst.predict(x, batch_size=2), ref)          # (also binds the plan with OTHER weights)
    dst.load_weights(path)
    got = dst.predict(x, batch_size=2)
    assert np.array_equal(got, ref)
    o64, _ = _oracle(wd, x, 2, 1, 16, torch.float64, **kw)
    o32, _ = _oracle(wd, x, 2, 1, 16, torch.float32, **kw)
    _check('h5.by_order.xy', got[..., :2], o32[0][..., :2], o64[0][..., :2], PX_TOL)

    conda = '/opt/conda/bin/python3.9'
    if os.path.exists(conda):
        lib_path = str(tmp_path / 'reception_libhdf5.h5')
        script = (
            "import h5py, sys\n"
            "src, dst = h5py.File(sys.argv[1], 'r'), h5py.File(sys.argv[2], 'w')\n"
            "for k, v in src.attrs.items(): dst.attrs[k] = v\n"
            "def walk(g, out):\n"
            "    for k, v in g.attrs.items(): out.attrs[k] = v\n"
            "    for name, item in g.items():\n"
            "        if isinstance(item, h5py.Group): walk(item, out.create_group(name))\n"
            "        elif item.shape == (): out.create_dataset(name, data=item[()])\n"
            "        else: out.create_dataset(name, data=item[...], chunks=tuple(max(1, s // 2) for s in item.shape),\n"
            "                                 compression='gzip', shuffle=True)\n"
            "walk(src, dst)\n")
        r = subprocess.run([conda, '-c', script, path, lib_path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        dst2, _ = _build(2, 1, 16, **kw)
        weights.init_synthetic(dst2, seed=4)
        dst2.load_weights(lib_path)
        assert np.array_equal(dst2.predict(x, batch_size=2), ref)

    # SPNet: every layer is named, files are loaded by name
    xs = np.random.default_rng(14).uniform(-1, 1, (1, 2, 256, 256, 3)).astype(np.float32)
    sp, _, _, _ = _spnet(2, 'pa16j2d', 15, 2, [2], 160, replica=True)
    ref_sp = sp.predict(xs, batch_size=1)
    sp_path = str(tmp_path / 'spnet.h5')
    sp.save_weights(sp_path)
    sp2, _, _, _ = _spnet(2, 'pa16j2d', 15, 2, [2], 160, replica=True)
    weights.init_synthetic(sp2, seed=5)
    sp2.load_weights(sp_path, by_name=True)
    for a, b in zip(sp2.predict(xs, batch_size=1), ref_sp):
        assert np.array_equal(a, b)
