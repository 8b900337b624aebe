"""CPU tests of the host logic: C-ABI surface, weight packing, graph builders, planner/fusion, memory plan,
weights, naming agreement between the product builders and the independent oracle.  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest


def test_capi_exports_every_declared_symbol(hip_lib):
    from deephar_amd import _lib
    hdr = open(os.path.join(os.path.dirname(_lib._HERE), 'include', 'deephar_hip.h')).read()
    declared = set(re.findall(r'\b(dh_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no prototypes parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(hip_lib, name), name
    assert hip_lib.dh_version() >= 100
    assert hip_lib.dh_error_string(-2).decode().startswith('configuration')


def test_ctypes_structs_match_header_layout(hip_lib, tmp_path):
    """Size and every field offset of the ctypes mirrors against what a C compiler makes of include/deephar_hip.h."""
    import re
    import subprocess
    from deephar_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'deephar_hip.h')).read()
    pairs = {'dh_conv_args': _lib.ConvArgs, 'dh_dw_args': _lib.DwArgs, 'dh_pool_args': _lib.PoolArgs,
             'dh_elt_args': _lib.EltArgs, 'dh_sam_args': _lib.SamArgs}
    structs = set(re.findall(r'typedef struct (dh_\w+_args)', header))
    assert structs == set(pairs), structs ^ set(pairs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "deephar_hip.h"', 'int main(void) {']
    for cname, ct in pairs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in ct._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append('return 0; }')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = str(tmp_path / 'layout')
    subprocess.run(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', exe], check=True)
    got = dict(l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, ct in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got['%s.%s' % (cname, fname)]) == getattr(ct, fname).offset, (cname, fname)


def test_weight_packing_roundtrip_and_layout(hip_lib):
    from deephar_amd.engine import packing
    rng = np.random.default_rng(0)
    for shape in [(1, 1, 576, 576), (3, 3, 3, 32), (1, 1, 48, 576), (5, 1, 64, 64), (1, 1, 576, 48), (3, 5, 2, 24)]:
        w = rng.standard_normal(shape).astype(np.float32)
        packed, kp, np_ = packing.pack_conv(w)
        k = shape[0] * shape[1] * shape[2]
        assert kp % 32 == 0 and np_ % 32 == 0 and kp >= k and np_ >= shape[3] and packed.size == kp * np_
        assert np.array_equal(packing.unpack_conv(packed, *shape), w)
        # [Kp/4][Np][4]: element (k, n) sits at ((k//4)*Np + n)*4 + k%4 ; padding is zero
        flat = w.reshape(k, shape[3])
        kk, nn = k - 1, shape[3] - 1
        assert packed[((kk // 4) * np_ + nn) * 4 + kk % 4] == flat[kk, nn]
        assert np.count_nonzero(packed) == np.count_nonzero(w)
    assert hip_lib.dh_conv2d_pack_weights_host(None, None, 1, 1, 1, 1) == -1


def test_tile_heuristic_prefers_full_tiles(hip_lib):
    pick = hip_lib.dh_conv2d_pick_tile_cfg
    assert pick(65536, 576) == 2          # 128x96 tiles, 4 waves along M (measured best)
    assert pick(65536, 288) == 2
    assert pick(65536, 48) == 3           # padded to 64
    assert pick(64, 32) == 8
    for m in (1, 100, 4096, 10 ** 6):
        for c in (1, 17, 48, 160, 272, 576):
            assert 0 <= pick(m, c) < hip_lib.dh_conv2d_num_tile_cfgs() // 2


def _mpii(blocks=2, **kw):
    from deephar_amd import graph
    from deephar_amd.models import reception
    graph.reset_naming()
    kw.setdefault('num_context_per_joint', 2)
    return reception.build((256, 256, 3), 16, dim=2, num_blocks=blocks, ksize=(5, 5), **kw)


def test_reception_graph_matches_survey_numbers():
    m = _mpii(8, concat_pose_confidence=False)
    assert m.count_params() == 14746560                      # 14.75 M
    assert abs(m.plan.total_flops() / 2e9 - 9.833) < 0.005   # 9.833 GMAC / frame
    assert len(m.outputs) == 16
    assert [t.shape for t in m.outputs[:2]] == [(16, 2), (16, 1)]
    assert m.input_shape == (None, 256, 256, 3) and m.get_input_shape_at(0) == (None, 256, 256, 3)
    names = [l.name for l in m.layers]
    for want in ['Stem', 'rBlock1', 'SepConv1', 'RegMap1', 'fReMap1', 'sSAM', 'cSAM', 'sjProb', 'cjProb', 'Agg',
                 'rBlock8', 'RegMap8']:
        assert want in names
    assert 'fReMap8' not in names
    assert m.get_layer('Stem').count_params() > 0


def test_h36m_graph():
    from deephar_amd import graph
    from deephar_amd.models import reception
    graph.reset_naming()
    m = reception.build((256, 256, 3), 17, dim=3, num_blocks=8, depth_maps=16, ksize=(5, 5))
    assert abs(m.count_params() / 1e6 - 16.68) < 0.01
    assert abs(m.plan.total_flops() / 2e9 - 11.814) < 0.005
    assert [t.shape for t in m.outputs] == [(17, 4)] * 8
    with pytest.raises(ValueError):
        reception.build((256, 256, 3), 16, dim=4)
    with pytest.raises(AssertionError):
        reception.build((256, 256, 3), 17, dim=3, num_context_per_joint=2)


def test_planner_fusion_rules():
    m = _mpii(2, concat_pose_confidence=True)
    plan = m.plan
    kinds = [s.kind for s in plan.steps]
    # nothing but kernels that exist; BN / ReLU / add / concat / upsample never survive as their own launch
    assert set(kinds) <= {'conv', 'dwconv', 'pool', 'sam_ctx'}
    # R7: the two poolings of every hourglass (reception.py:105-116) are second outputs of the convolutions in front of them
    # -- at 32 columns (pairs of waves pool) and, [r06], at 16 columns (a wave pools its two image rows alone)
    assert sorted(s.outs['y'].shape[-2] for s in plan.steps if s.kind == 'conv' and 'ypool' in s.outs) == [16, 16, 32, 32]
    convs = [s for s in plan.steps if s.kind == 'conv']
    # R3: both add([a, UpSampling2D(b)]) of every hourglass are second residuals, read at half resolution, of the
    # convolutions that produce `a`; those are emitted after the low-resolution branch that produces `b`
    down = [s for s in convs if s.attrs['res2_down']]
    assert len(down) == 4 and not any(s.attrs['up2'] for s in convs)
    for s in down:
        y, r2 = s.outs['y'], s.ins['res2']
        assert (r2.shape[-3], r2.shape[-2], r2.C) == (y.shape[-3] // 2, y.shape[-2] // 2, y.C) and 'res1' in s.ins
        writer = max(i for i, q in enumerate(plan.steps) if any(v is not None and v.buf is r2.buf for v in q.outs.values()))
        assert writer < plan.steps.index(s)
    assert any('res1' in s.ins and 'res2' in s.ins and not s.attrs['up2'] for s in convs)   # 3-way add
    first = plan.steps[0]
    assert first.attrs['post_relu'] == 1 and 'post_bn' in first.params and first.attrs['pt'] == 0  # TF-SAME s2
    # concat targets are written in place at channel offsets
    y = [s for s in plan.steps if s.kind == 'pool'][0].outs['y']
    assert (y.coff, y.ld) == (96, 160)
    # decoder (R5 + R5b): both soft-argmax read-outs and the context aggregation are ONE launch per block over the 48
    # heat-map channels; pose (x, y) and the joints' confidence land directly in the [J, 3] output
    dec = [s for s in plan.steps if s.kind == 'sam_ctx']
    assert len(dec) == 2
    assert (dec[0].ins['h'].coff, dec[0].ins['h'].ld, dec[0].ins['h'].C) == (0, 48, 48)
    assert (dec[0].attrs['J'], dec[0].attrs['nctx'], dec[0].attrs['alpha']) == (16, 2, 0.8)
    assert dec[0].outs['y'].ld == 3 and dec[0].outs['conf_raw'].ld == 3 and dec[0].outs['conf_raw'].coff == 2
    # ... unless something else reads the intermediate coordinates: exported heat-maps do not, exported poses would
    kept = _mpii(1, num_context_per_joint=2, export_heatmaps=True).plan
    assert sum(1 for s in kept.steps if s.kind == 'sam_ctx') == 1


def test_planner_wide_adds_become_conv_epilogues(monkeypatch):
    """R9: SPNet's four-operand re-injection sum (spnet.py:233) and the lateral add behind a residual unit (spnet.py:303)
    leave no element-wise launch: every partial sum is the epilogue of a convolution (two residual slots each); the model's
    own graph is not modified, the memory plan stays sound, FLOPs are conserved (DEEPHAR_SPLIT_ADDS=0: off)."""
    from deephar_amd import graph, utils
    from deephar_amd.config import ModelConfig
    from deephar_amd.models import spnet

    def build():
        graph.reset_naming()
        cfg = ModelConfig((4, 128, 128, 3), utils.pa17j3d, num_actions=[60], num_pyramids=2, action_pyramids=[1, 2],
                          num_levels=4, num_pose_features=64, num_visual_features=64)
        return spnet.build(cfg)
    monkeypatch.setenv('DEEPHAR_SPLIT_ADDS', '0')
    m0 = build()
    base = m0.plan
    monkeypatch.setenv('DEEPHAR_SPLIT_ADDS', '1')
    m1 = build()
    plan = m1.plan
    assert base.split_adds is False and plan.split_adds is True      # read once per plan, recorded in it
    adds = lambda p: [s for s in p.steps if s.kind == 'eltwise' and s.attrs.get('op', 0) == 0 and 'b' in s.ins]
    # (a sum that became a convolution's epilogue can also carry the pooling behind it: rule R7 -- those launches go too)
    npool = lambda p: sum(1 for s in p.steps if s.kind == 'conv' and 'ypool' in s.outs)
    assert len(adds(base)) == 15 and len(adds(plan)) == 0
    assert len(plan.steps) == len(base.steps) - 15 - (npool(plan) - npool(base))
    assert sum(1 for n in m1._nodes if n.op == 'add' and len(n.inputs) == 4) == 5          # the graph itself keeps its adds (the last block re-injects nothing)
    assert abs(sum(s.flops() for s in plan.steps) - sum(s.flops() for s in base.steps)) < 1.0
    assert sum(s.bytes() for s in plan.steps) < sum(s.bytes() for s in base.steps)
    two = [s for s in plan.steps if s.kind == 'conv' and 'res1' in s.ins and 'res2' in s.ins and not s.attrs.get('res2_down')]
    assert len(two) >= 9
    assert [v.shape for v in plan.outputs] == [v.shape for v in base.outputs]
    for i, s in enumerate(plan.steps):
        for v in list(s.ins.values()) + list(s.outs.values()):
            assert v.buf.start <= i <= v.buf.end
    for i, a in enumerate(plan.bufs):
        for b in plan.bufs[i + 1:]:
            live = not (a.end < b.start or b.end < a.start)
            space = not (a.offset + a.items <= b.offset or b.offset + b.items <= a.offset)
            assert not (live and space)


def test_planner_r3_spares_split_k_producers():
    """add([conv(x), UpSampling2D(b)]) must not become the half-resolution second residual (res2_down) of a
    convolution that dh_conv2d_f32 runs on the split-K kernel (per-frame output <= 256 pixels, K >= 64, Cout <= 256:
    conv_igemm.hip returns DH_EUNSUPPORTED for that pair) -- the plan falls back to an up-sampling kernel; a producer
    outside the rule still takes R3."""
    from deephar_amd import Model, graph
    from deephar_amd import layers as L
    from deephar_amd.engine.planner import split_k_rule

    def build(cin, cout, size):
        graph.reset_naming()
        inp = L.Input((size, size, cin))
        low = L.conv2d(L.MaxPooling2D(inp, (2, 2)), cout, (1, 1), name='low')
        a = L.conv2d(inp, cout, (1, 1), name='a')
        out = L.add([a, L.UpSampling2D(low, (2, 2))])
        return Model(inp, [out]).plan

    assert split_k_rule(16 * 16, 1024, 128, 1024)
    skinny = build(1024, 128, 16)
    convs = {s.name: s for s in skinny.steps if s.kind == 'conv'}
    assert not convs['a'].attrs['res2_down'] and 'res2' not in convs['a'].ins
    assert any(s.kind == 'upsample_add' for s in skinny.steps) or any(s.attrs.get('up2') for s in convs.values())
    assert not split_k_rule(16 * 16, 512, 288, 512)
    wide = build(512, 288, 16)
    convs = {s.name: s for s in wide.steps if s.kind == 'conv'}
    assert convs['a'].attrs['res2_down'] == 1 and not any(s.kind == 'upsample_add' for s in wide.steps)
    # sep-conv producer: the rule looks at its pointwise half (K = Cin)
    graph.reset_naming()
    inp = L.Input((16, 16, 1024))
    low = L.conv2d(L.MaxPooling2D(inp, (2, 2)), 128, (1, 1), name='low')
    a = L.sepconv2d(inp, 128, (3, 3), name='sep')
    plan = Model(inp, [L.add([a, L.UpSampling2D(low, (2, 2))])]).plan
    assert not any(s.attrs.get('res2_down') for s in plan.steps if s.kind == 'conv')


def test_tail_stream_policy_is_one_directional_and_sound():
    """Model.stream_policy = 'tail' (engine/schedule.py: assign_streams_tail; the latency regime of
    exp/pennaction/eval_speed2d.py): stream 1 runs SPNet's action stream, re-ordered by readiness -- the step order stays
    topological, every cross-stream dependency points from stream 0 to stream 1 and is waited for at most once per step,
    nothing on stream 0 reads stream 1, and no two buffers that may be live together share arena space."""
    from deephar_amd import graph, utils
    from deephar_amd.config import ModelConfig
    from deephar_amd.engine import schedule
    from deephar_amd.models import spnet

    def build(policy):
        graph.reset_naming()
        cfg = ModelConfig((8, 128, 128, 3), utils.pa16j2d, num_actions=[15], num_pyramids=2, action_pyramids=[1, 2],
                          num_levels=4, pose_replica=True, num_pose_features=160, num_visual_features=160)
        m = spnet.build(cfg)
        m.num_streams, m.stream_policy = 2, policy
        return m.plan
    one = build('list')
    plan = build('tail')
    assert plan.nstreams == 2 and sorted(s.name or s.kind for s in plan.steps) == sorted(s.name or s.kind for s in one.steps)
    deps = schedule.compute_deps(plan)
    stream = [s.stream for s in plan.steps]
    assert all(i < j for j, d in enumerate(deps) for i in d)                       # still a topological order
    assert all(not (stream[j] == 0 and stream[i] == 1) for j, d in enumerate(deps) for i in d)
    waits = [(j, w) for j, s in enumerate(plan.steps) for w in s.wait]
    assert waits and all(stream[j] == 1 and stream[w] == 0 and plan.steps[w].record for j, w in waits)
    assert len(waits) < sum(len(s.wait) for s in one.steps) / 4                    # (the list scheduler ping-pongs)
    # every cross-stream dependency is covered by a wait of this step or of an earlier step on the same stream
    covered = -1
    for j, d in enumerate(deps):
        if stream[j] == 1:
            covered = max([covered] + list(plan.steps[j].wait))
            assert all(i <= covered for i in d if stream[i] == 0), j
    # memory plan: buffers overlapping in space are ordered by happens-before
    reach = schedule.happens_before(len(plan.steps), deps, stream)
    acc = {}
    for j, s in enumerate(plan.steps):
        for v in list(s.ins.values()) + list(s.outs.values()):
            if v is not None:
                acc.setdefault(id(v.buf), (v.buf, set()))[1].add(j)
    bufs = list(acc.values())
    for x, (a, sa) in enumerate(bufs):
        for b, sb in bufs[x + 1:]:
            if a.pinned or b.pinned or a.kind == 'input' or b.kind == 'input':
                continue
            if not (a.offset + a.items <= b.offset or b.offset + b.items <= a.offset):
                ab = all(reach[i] >> j & 1 for i in sa for j in sb)
                ba = all(reach[j] >> i & 1 for i in sa for j in sb)
                assert ab or ba, (a.offset, b.offset)


def test_planner_pooled_output_rule(monkeypatch):
    """R7: the 32-column MaxPooling2D -- [r06] and the 16-column one behind it -- becomes a second output of the convolution
    that feeds it; same algorithmic FLOPs, a launch and a full-resolution read less each, and the memory plan stays sound
    (DEEPHAR_FUSE_POOL=0: off; DEEPHAR_FUSE_POOL_SMALL=0: 32 columns only)."""
    monkeypatch.setenv('DEEPHAR_FUSE_POOL', '0')
    base = _mpii(2).plan
    monkeypatch.setenv('DEEPHAR_FUSE_POOL', '1')
    plan = _mpii(2).plan
    fused = [s for s in plan.steps if s.kind == 'conv' and 'ypool' in s.outs]
    assert len(fused) == 4 and len(plan.steps) == len(base.steps) - 4
    for s in fused:
        y, yp = s.outs['y'], s.outs['ypool']
        assert y.shape[-2] in (16, 32) and yp.shape[-3:] == (y.shape[-3] // 2, y.shape[-2] // 2, y.C) and s.attrs['pool2'] == 1
    assert not any(s.kind == 'pool' and s.ins['x'].shape[-2] in (16, 32) for s in plan.steps)
    monkeypatch.setenv('DEEPHAR_FUSE_POOL_SMALL', '0')
    wide_only = _mpii(2).plan
    assert [s.outs['y'].shape[-2] for s in wide_only.steps if s.kind == 'conv' and 'ypool' in s.outs] == [32, 32]
    monkeypatch.delenv('DEEPHAR_FUSE_POOL_SMALL')
    assert sum(s.flops() for s in plan.steps) == sum(s.flops() for s in base.steps)
    assert sum(s.bytes() for s in plan.steps) < sum(s.bytes() for s in base.steps)
    for i, s in enumerate(plan.steps):
        for v in list(s.ins.values()) + list(s.outs.values()):
            assert v.buf.start <= i <= v.buf.end
    bufs = plan.bufs
    for i, a in enumerate(bufs):
        for b in bufs[i + 1:]:
            live = not (a.end < b.start or b.end < a.start)
            space = not (a.offset + a.items <= b.offset or b.offset + b.items <= a.offset)
            assert not (live and space)


def test_memory_plan_has_no_live_overlap():
    plan = _mpii(3).plan
    bufs = plan.bufs
    assert plan.arena_items >= max(b.offset + b.items for b in bufs)
    for i, a in enumerate(bufs):
        assert a.offset % 4 == 0
        for b in bufs[i + 1:]:
            live = not (a.end < b.start or b.end < a.start)
            space = not (a.offset + a.items <= b.offset or b.offset + b.items <= a.offset)
            assert not (live and space), 'two live buffers share arena space'
    # reuse actually happens: arena far smaller than the sum of all buffers
    assert plan.arena_items < 0.35 * sum(b.items for b in bufs)
    # every step's operands are alive at that step
    for i, s in enumerate(plan.steps):
        for v in list(s.ins.values()) + list(s.outs.values()):
            assert v.buf.start <= i <= v.buf.end


def test_weights_roundtrip_and_determinism(tmp_path):
    from deephar_amd import weights
    m = _mpii(1)
    with pytest.raises(RuntimeError):
        weights.save_weights(m, str(tmp_path / 'w.npz'))
    weights.init_synthetic(m, seed=3)
    a = weights.as_dict(m)
    path = str(tmp_path / 'w.npz')
    m.save_weights(path)
    m2 = _mpii(1)
    m2.load_weights(path)
    for k, v in weights.as_dict(m2).items():
        assert np.array_equal(v, a[k])
    m3 = _mpii(1)
    weights.init_synthetic(m3, seed=3)
    assert all(np.array_equal(v, a[k]) for k, v in weights.as_dict(m3).items())
    weights.init_synthetic(m3, seed=4)
    assert not np.array_equal(weights.as_dict(m3)['Stem/conv2d_1/kernel'], a['Stem/conv2d_1/kernel'])
    with pytest.raises(ValueError):
        m3.get_layer('Stem').layers[0].set_weights([np.zeros((1, 1, 1, 1))])


def test_oracle_and_builder_agree_on_names_and_shapes():
    """Independent restatements: the oracle walks the reference's source order and must ask for exactly the
    weights the product builder created (names + shapes), and use all of them."""
    import torch
    from deephar_amd import weights
    from oracle import reception as oref
    from oracle.naming import Weights
    m = _mpii(2, concat_pose_confidence=False)
    weights.init_synthetic(m)
    ow = Weights(weights.as_dict(m))
    x = np.random.default_rng(0).uniform(-1, 1, (1, 256, 256, 3)).astype(np.float32)
    outs = oref.forward(ow, x, 16, 2, num_context_per_joint=2, num_blocks=2, ksize=(5, 5),
                        concat_pose_confidence=False)
    assert ow.unused() == []
    assert [o.shape[1:] for o in outs] == [t.shape for t in m.outputs]
    assert all(np.all(np.isfinite(o)) for o in outs)


def test_synthetic_weights_keep_activations_sane():
    """heat-map logits must be neither flat nor one-hot, context confidences away from 0."""
    import torch
    from deephar_amd import weights
    from oracle import reception as oref
    m = _mpii(4, concat_pose_confidence=False)
    weights.init_synthetic(m)
    x = np.random.default_rng(1).uniform(-1, 1, (2, 256, 256, 3)).astype(np.float32)
    taps = {}
    oref.forward(weights.as_dict(m), x, 16, 2, num_context_per_joint=2, num_blocks=4, ksize=(5, 5),
                 concat_pose_confidence=False, taps=taps)
    for b in (1, 4):
        assert 1.0 < taps['heatmaps%d' % b].std() < 30.0
        assert 0.3 < taps['rblock%d' % b].std() < 30.0
        assert taps['vc%d' % b].reshape(2, 16, 2).sum(axis=2).min() > 1.0


def test_model_api_errors():
    from deephar_amd import Model, layers
    m = _mpii(1)
    with pytest.raises(ValueError):
        m.get_layer('nope')
    with pytest.raises(ValueError):
        Model(layers.Input((4, 4, 3)), m.outputs[0])      # disconnected graph
    with pytest.raises(TypeError):
        Model(m.input, [1.0])
    sub = m.get_layer('rBlock1')
    y = sub(layers.Input((7, 32, 32, 576)))               # TimeDistributed-style call: leading T dim
    assert y.shape == (7, 32, 32, 576)
    with pytest.raises(ValueError):
        sub(layers.Input((16, 16, 576)))


def test_no_cpu_execution_path():
    """Without a GPU predict() must fail loudly, never fall back to the oracle or torch-CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from deephar_amd import weights
    from deephar_amd._lib import DeepharHipError
    m = _mpii(1)
    weights.init_synthetic(m)
    with pytest.raises(DeepharHipError):
        m.predict(np.zeros((1, 256, 256, 3), np.float32))
    import deephar_amd
    src = ''.join(open(os.path.join(os.path.dirname(deephar_amd.__file__), f)).read()
                  for f in ('model.py', 'engine/executor.py', 'engine/planner.py', 'functional.py', 'layers.py'))
    assert 'import oracle' not in src and 'from oracle' not in src


def test_predict_validates_inputs_without_a_gpu():
    """Shape / count / dtype errors and the empty batch are host logic: no device is touched."""
    from deephar_amd.models import reception
    m = reception.build((64, 64, 3), 16, dim=2, num_context_per_joint=2, num_blocks=1, ksize=(3, 3))
    out = m.predict(np.zeros((0, 64, 64, 3), np.float32))
    assert out.shape == (0, 16, 3) and out.dtype == np.float32        # single output: a bare array, like Keras
    m2 = reception.build((64, 64, 3), 16, dim=2, num_context_per_joint=2, num_blocks=2, ksize=(3, 3),
                         concat_pose_confidence=False)
    assert [o.shape for o in m2.predict(np.zeros((0, 64, 64, 3), np.uint8))] == [(0, 16, 2), (0, 16, 1)] * 2
    with pytest.raises(ValueError):
        m.predict(np.zeros((2, 32, 32, 3), np.float32))
    with pytest.raises(ValueError):
        m.predict([np.zeros((2, 64, 64, 3), np.float32)] * 2)


@pytest.mark.parametrize('nstreams', [1, 2, 3, 4])
def test_multi_stream_memory_plan_is_race_free(nstreams):
    """With parallel graph branches two buffers may share arena space only if EVERY access of one happens-before
    every access of the other (data dependencies + same-stream order); cross-stream waits reference earlier steps
    on other streams only, and every waited-for step records an event."""
    from deephar_amd.engine import schedule
    from deephar_amd.engine.planner import build_plan
    m = _mpii(2)
    plan = build_plan(m.inputs, m.outputs, nstreams=nstreams)
    n = len(plan.steps)
    stream = [s.stream for s in plan.steps]
    assert max(stream) + 1 == plan.nstreams <= nstreams
    deps = schedule.compute_deps(plan)
    reach = schedule.happens_before(n, deps, stream)
    acc = {}
    for j, s in enumerate(plan.steps):
        for v in list(s.ins.values()) + list(s.outs.values()):
            if v is not None:
                acc.setdefault(id(v.buf), (v.buf, set()))[1].add(j)
    items = list(acc.values())
    shared = 0
    for i, (a, sa) in enumerate(items):
        for b, sb in items[i + 1:]:
            if a.offset + a.items <= b.offset or b.offset + b.items <= a.offset:
                continue
            shared += 1
            a_before_b = all(all((reach[x] >> y) & 1 for y in sb) for x in sa)
            b_before_a = all(all((reach[y] >> x) & 1 for x in sa) for y in sb)
            assert a_before_b or b_before_a, 'unordered buffers share arena space'
    assert shared > 50                                  # space is actually re-used
    for j, s in enumerate(plan.steps):
        for w in s.wait:
            assert w < j and stream[w] != stream[j] and plan.steps[w].record
        assert set(s.wait) <= set(deps[j])


def test_split_k_rule_matches_the_library(hip_lib):
    """engine.executor.split_k_rule (decides which weight packing a layer gets in bf16x3 mode) must agree with the
    library's own dispatch rule dh_conv2d_uses_split_k for every geometry."""
    import ctypes as C
    from deephar_amd import _lib
    from deephar_amd.engine.executor import split_k_rule
    a = _lib.ConvArgs()
    n = 0
    for oh, ow in ((4, 4), (16, 16), (16, 10), (8, 5), (17, 16), (32, 32), (1, 1)):
        for k in (1, 3, 5):
            for cin in (3, 30, 48, 112, 256, 288, 576, 1024):
                for cout in (15, 48, 256, 257, 576):
                    a.N, a.OH, a.OW, a.Cin, a.Cout, a.KH, a.KW, a.K = 3, oh, ow, cin, cout, k, k, k * k * cin
                    a.x_u8 = a.w_split = 0
                    assert bool(hip_lib.dh_conv2d_uses_split_k(C.byref(a))) == split_k_rule(oh * ow, k * k * cin, cout, cin, k, k), \
                        (oh, ow, k, cin, cout)
                    n += 1
    # the kernel-extent clauses: KH * KW < 256 (the tap decode), K * Cin < 2^31
    for kh, kw, cin in ((1, 255, 4), (2, 255, 4), (16, 16, 4), (15, 17, 4), (1, 3, 4096), (5, 5, 4096), (11, 11, 4096), (3, 3, 4097)):
        a.N, a.OH, a.OW, a.Cin, a.Cout, a.KH, a.KW, a.K = 1, 4, 4, cin, 64, kh, kw, kh * kw * cin
        assert bool(hip_lib.dh_conv2d_uses_split_k(C.byref(a))) == split_k_rule(16, kh * kw * cin, 64, cin, kh, kw), (kh, kw, cin)
        n += 1
    assert not split_k_rule(16, 2 * 255 * 4, 64, 4, 2, 255) and split_k_rule(16, 255 * 4, 64, 4, 1, 255)
    a.w_split = 1
    assert hip_lib.dh_conv2d_uses_split_k(C.byref(a)) == 0            # split-packed weights never take that kernel
    assert n > 500



def test_engine_options_replan_the_model():
    """Model.gemm_precision / num_streams set AFTER the plan exists take effect at the next predict: plan and executor
    are dropped (a silently ignored option once made a bf16x3 parity test run in fp32)."""
    m = _mpii(1)
    p1 = m.plan
    assert p1.gemm_precision == 'f32'
    m.gemm_precision = 'f32'
    assert m.plan is p1                                  # unchanged value: nothing is thrown away
    m.gemm_precision = 'bf16x3'
    assert m._plan is None and m._exec is None
    p2 = m.plan
    assert p2 is not p1 and p2.gemm_precision == 'bf16x3'
    m.num_streams = 2
    assert m._plan is None and m.plan.nstreams == 2


def test_split_eligibility_is_the_librarys_answer(hip_lib):
    """dh_conv2d_split_eligible (asked by the executor before it packs a layer's weights for bf16x3 mode):
    LDS-DMA GEMM shapes with an aligned float input and no BN prologue, not split-K layers, and -- what the Python
    mirror used to miss -- operands inside the kernel's 32-bit buffer offsets.  No launch is made."""
    import ctypes as C
    from deephar_amd import _lib

    def args(n=64, h=32, w=32, cin=576, cout=576, k=1, ldx=None, xptr=256):
        a = _lib.ConvArgs()
        a.x, a.y = xptr, 1 << 20
        a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, w, cin, ldx or cin, h, w, cout, cout
        a.KH = a.KW = k
        a.SH = a.SW = 1
        a.PT = a.PL = (k - 1) // 2
        a.K = k * k * cin
        kp, np_ = C.c_int(), C.c_int()
        assert hip_lib.dh_conv2d_packed_dims(k, k, cin, cout, C.byref(kp), C.byref(np_)) == 0
        a.Kp, a.Np = kp.value, np_.value
        return a
    ok = lambda a: hip_lib.dh_conv2d_split_eligible(C.byref(a))
    assert ok(args()) == 1                                    # the dominant pointwise GEMM
    assert ok(args(cin=64, cout=96, k=3, h=64, w=64)) == 1    # stem K x K, Cin % 32 == 0
    assert ok(args(cin=48, cout=96, k=3)) == 0                # K x K with Cin % 32 != 0
    assert ok(args(xptr=260)) == 0                            # input view not 16-byte aligned
    assert ok(args(ldx=578)) == 0                             # pixel pitch not a multiple of 4 floats
    a = args()
    a.pre_scale = a.pre_shift = 4096
    assert ok(a) == 0                                         # BN prologue
    a = args()
    a.x_u8 = 1
    assert ok(a) == 0
    assert ok(args(h=16, w=16, cin=256, cout=256, k=3)) == 0  # an action-head conv: the split-K kernel (fp32 packing)
    assert ok(args(n=1700)) == 1 and ok(args(n=1710)) == 0    # 1 710 x 32 x 32 x 576 x 4 B > 0xf0000000: fp32 fallback
    assert hip_lib.dh_conv2d_split_eligible(None) == 0


def test_integration_guide_declares_the_whole_conv_struct():
    """The ctypes example of INTEGRATION.md is what a maintainer copies: its ConvArgs must list every field of
    dh_conv_args, in order (a shorter struct makes the library read garbage for the tail)."""
    import re
    from deephar_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, 'INTEGRATION.md')).read()
    block = doc[doc.index('class ConvArgs(C.Structure)'):doc.index('lib.dh_conv2d_f32.argtypes')]
    names = re.findall(r"'([A-Za-z_0-9]+)'", block)
    assert names == [n for n, _ in _lib.ConvArgs._fields_]


def test_design_tables_are_generated_from_the_tracked_records():
    """DESIGN.md's tables of current-round numbers (parity worst cases, workloads, the speed protocol)
    are generated from profiles/*.json by tools/design_tables.py; a hand edit or a stale record fails here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'design_tables.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_heat_map_head_rules_on_the_host(monkeypatch):
    """[r05] Planner rules R4b / R10 without a GPU: on SPNet's heat-map head (spnet.py:24-48) the copy into
    concatenate([fw_maps, pred_maps]) disappears (pred_maps is written in place, the soft-argmax reads it there at pitch 2 J)
    and `_fw_maps` + `_conv1` become ONE convolution over both kernels side by side; the merged weight follows its parts."""
    from deephar_amd import graph, utils
    from deephar_amd.config import ModelConfig
    from deephar_amd.engine.planner import ConcatParam
    from deephar_amd.models import spnet

    def build():
        graph.reset_naming()
        cfg = ModelConfig((4, 128, 128, 3), utils.pa16j2d, num_actions=[15], num_pyramids=2, action_pyramids=[1, 2],
                          num_levels=4, pose_replica=True, num_pose_features=160, num_visual_features=160)
        return spnet.build(cfg)
    monkeypatch.setenv('DEEPHAR_MERGE_SIBLINGS', '0')     # (round-6 rules R10b / R10c have their own test below)
    monkeypatch.setenv('DEEPHAR_MERGE_KXK', '0')
    monkeypatch.setenv('DEEPHAR_MERGE_HEADS', '0')
    monkeypatch.setenv('DEEPHAR_CONCAT_SHARED', '0')
    base = build().plan
    monkeypatch.setenv('DEEPHAR_CONCAT_SHARED', '1')
    shared = build().plan
    monkeypatch.setenv('DEEPHAR_MERGE_HEADS', '1')
    m = build()
    plan = m.plan
    copies = lambda p: sum(1 for s in p.steps if s.kind == 'copy')
    assert copies(base) - copies(shared) == 5 and len(shared.steps) == len(base.steps) - 5       # five heads with forward maps
    merged = [s for s in plan.steps if s.kind == 'conv' and isinstance(s.params['w'], ConcatParam)]
    assert len(merged) == 5 and len(plan.steps) == len(shared.steps) - 5
    for s in merged:
        w, y = s.params['w'], s.outs['y']
        assert [p.key.split('/')[-2].split('_')[-2:] for p in w.parts] == [['fw', 'maps'], ['heatmaps', 'conv1']]
        assert s.attrs['Cout'] == 32 == y.C == y.ld and y.coff == 0 and w.shape[-1] == 32 and w.value is None
        # the soft-argmax of pred_maps reads the second slab of the same buffer
        sam = next(t for t in plan.steps if t.kind == 'sam' and t.ins['h'].buf is y.buf)
        assert (sam.ins['h'].coff, sam.ins['h'].C, sam.ins['h'].ld) == (16, 16, 32)
    from deephar_amd import weights
    weights.init_synthetic(m, seed=0)
    w = merged[0].params['w']
    v0, ver = w.value, w.version
    assert v0.shape == w.shape and np.array_equal(v0[..., :16], w.parts[0].value) and np.array_equal(v0[..., 16:], w.parts[1].value)
    w.parts[1].set(2.0 * w.parts[1].value)
    assert w.version == ver + 1 and np.array_equal(w.value[..., 16:], 2.0 * v0[..., 16:])
    assert sum(s.flops() for s in plan.steps) == pytest.approx(sum(s.flops() for s in base.steps))


def test_round6_planner_rules_on_the_host(monkeypatch):
    """[r06] Rules R10b (K x K siblings as one centred-kernel convolution), R10c (sibling 1x1 convolutions into a joint buffer,
    per-column affine, ReLU moved to the reader), R11 (UpSampling2D in front of an up-scaling unit: the shortcut runs at half
    resolution as the half-resolution residual, the depthwise conv up-samples on load) and R12 (pooling / up-sampling in front
    of a skinny-conv layer resampled on load) on the plan of the speed protocol's last model -- structure only, no GPU."""
    import bench
    from deephar_amd import Model, weights
    from deephar_amd.engine.planner import ConcatAffine, ConcatParam

    def plan(**env):
        for k in ('DEEPHAR_MERGE_KXK', 'DEEPHAR_MERGE_SIBLINGS', 'DEEPHAR_UP_COMMUTE', 'DEEPHAR_RESAMPLE_ON_LOAD', 'DEEPHAR_FOLD_POSE_MUL',
                  'DEEPHAR_MERGE_POOLS', 'DEEPHAR_FUSE_POOL_SMALL', 'DEEPHAR_POOL_SEGMENTS'):
            monkeypatch.setenv(k, env.get(k, '1'))
        full = bench.build_speed2d()
        m = Model(full.input, full.outputs[34:36])
        return m, m.plan
    _, off = plan(DEEPHAR_MERGE_KXK='0', DEEPHAR_MERGE_SIBLINGS='0', DEEPHAR_UP_COMMUTE='0', DEEPHAR_RESAMPLE_ON_LOAD='0',
                  DEEPHAR_FOLD_POSE_MUL='0', DEEPHAR_MERGE_POOLS='0', DEEPHAR_FUSE_POOL_SMALL='0', DEEPHAR_POOL_SEGMENTS='0')
    m, on = plan()
    assert len(off.steps) == 604 and len(on.steps) == 413
    _, r13 = plan(DEEPHAR_POOL_SEGMENTS='0')
    assert len(r13.steps) == 431
    # R10b: eighteen action heads, each opens with ONE 3x5 convolution of 70 columns whose parts sit centred in the window
    kxk = [s for s in on.steps if s.kind == 'conv' and (s.name or '').count('p_conv0') == 3]
    assert len(kxk) == 18 and all((s.attrs['kh'], s.attrs['kw'], s.attrs['pt'], s.attrs['pl'], s.attrs['Cout'], s.attrs['K']) ==
                                  (3, 5, 1, 2, 70, 30) for s in kxk)
    weights.init_synthetic(m, seed=0)
    w = kxk[0].params['w']
    assert isinstance(w, ConcatParam) and w.value.shape == (3, 5, 2, 70)
    assert np.array_equal(w.value[:, 2:3, :, :10], w.parts[0].value) and not w.value[:, [0, 1, 3, 4], :, :10].any()
    assert np.array_equal(w.value[:, 1:4, :, 10:30], w.parts[1].value) and not w.value[:, [0, 4], :, 10:30].any()
    assert np.array_equal(w.value[..., 30:], w.parts[2].value)
    # R10c: 36 residual units of the action heads (shortcut | conv1 -> 240 or 200 columns, affine on the second part, its
    # ReLU moved to conv2's prologue), 15 replica heads beside the forward / heat-map pair (48 columns)
    units = [s for s in on.steps if isinstance(s.params.get('post_affine'), ConcatAffine)]
    assert len(units) == 36 and sorted({s.attrs['Cout'] for s in units}) == [200, 240] and all(not s.attrs['post_relu'] for s in units)
    for s in units[:4]:
        (ca, la), (cb, lb) = s.params['post_affine'].parts
        assert la is None and lb is not None and ca == 160 and s.outs['y'].ld == ca + cb
        reader = next(q for q in on.steps if q.kind == 'conv' and q.ins['x'].buf is s.outs['y'].buf)
        assert reader.attrs['pre_relu'] == 1 and (reader.ins['x'].coff, reader.ins['x'].ld) == (160, ca + cb)
        assert reader.ins['res1'].buf is s.outs['y'].buf and reader.ins['res1'].coff == 0
    heads3 = [s for s in on.steps if s.kind == 'conv' and 'replica' in (s.name or '') and '+' in s.name]
    assert len(heads3) == 15 and all(s.attrs['Cout'] == 48 for s in heads3)
    # R11: nine up-scaling units -- no up-sampling launch on the pose side, shortcut at half resolution, depthwise up_in
    assert sum(1 for s in off.steps if s.kind == 'upsample_add') == 26 and not any(s.kind == 'upsample_add' for s in on.steps)
    up_dw = [s for s in on.steps if s.kind == 'dwconv' and s.attrs.get('up_in')]
    assert len(up_dw) == 9 and all(s.outs['y'].shape[-3] == 2 * s.ins['x'].shape[-3] for s in up_dw)
    down_res = [s for s in on.steps if s.kind == 'conv' and s.attrs.get('res2_down')]
    assert len(down_res) == 9 and all(s.ins['res2'].shape[-3] * 2 == s.outs['y'].shape[-3] for s in down_res)
    # R12: every action head's conv3 reads the class maps up-sampled on load, its conv2h reads x1 max+-min-pooled on load
    modes = sorted(s.attrs['x_resample'] for s in on.steps if s.kind == 'conv' and s.attrs.get('x_resample'))
    assert modes == [1] * 17 + [3] * 18
    for s in on.steps:
        if s.kind == 'conv' and s.attrs.get('x_resample') == 3:
            assert s.ins['x'].shape[-3] == 2 * s.outs['y'].shape[-3]
    # the replica read-outs write (x, y) * confidence themselves: no multiply launch is left
    assert sum(1 for s in off.steps if s.kind == 'eltwise') == 18 and not any(s.kind == 'eltwise' for s in on.steps)
    assert sum(1 for s in on.steps if s.kind == 'sam' and s.attrs.get('xy_times_conf')) == 18
    # R13: every action head pools its pose and appearance features in one launch out of a joint buffer
    pools = [s for s in r13.steps if s.kind == 'pool' and '+' in (s.name or '')]
    assert len(pools) == 18 and all(s.ins['x'].shape == (8, 16, 320) and s.outs['y'].shape == (8, 8, 320) for s in pools)
    for s in pools[:3]:
        writers = [q for q in r13.steps for v in q.outs.values() if v is not None and v.buf is s.ins['x'].buf]
        assert sorted((v.coff, v.C) for q in writers for v in q.outs.values() if v.buf is s.ins['x'].buf) == [(0, 160), (160, 160)]
        assert all(r13.steps.index(q) < r13.steps.index(s) for q in writers)
    # R14: that pooling is read through by the head's r2 unit -- concatenate([pool(U), xa]) as (x, x2) of one convolution
    segs = [s for s in on.steps if s.kind == 'conv' and s.attrs.get('seg')]
    assert len(segs) == 18 and not any(s.kind == 'pool' and '+' in (s.name or '') for s in on.steps)
    assert all(s.attrs['seg'] == dict(c_split=320, pool_sh=1) and s.ins['x'].shape == (8, 16, 320) and s.outs['y'].shape[:2] == (8, 8)
               for s in segs)
    assert sorted(s.attrs['Cin'] for s in segs) == [320] + [480] * 17 and sum(1 for s in segs if 'x2' in s.ins) == 17
    for s in segs:
        if 'x2' in s.ins:
            assert (s.ins['x2'].coff, s.ins['x2'].C, s.ins['x2'].ld) == (320, 160, 480)
    # R7 at 16 / 8 columns: the down-scaling units' MaxPooling2D is the second output of the prediction block's conv2
    pooled = [s for s in on.steps if s.kind == 'conv' and 'ypool' in s.outs]
    assert sorted(s.outs['y'].shape[-2] for s in pooled) == [8] * 3 + [16] * 3 + [32] * 2 and \
        all(s.outs['ypool'].shape[-2] * 2 == s.outs['y'].shape[-2] for s in pooled)
    # the arithmetic that is left: the up-scaling shortcuts run on a quarter of the pixels, nothing else changed
    assert sum(s.flops() for s in on.steps) < sum(s.flops() for s in off.steps)


def test_speed_protocol_configuration_is_the_reference_script():
    """bench.build_speed2d = exp/pennaction/eval_speed2d.py:31-36,50: six pyramids, actions on all six, pose_replica, 8-frame
    clips, 160 features; 18 pose + 18 action outputs, and the truncated model of block b is outputs[2b:2b+2]."""
    import bench
    from deephar_amd import Model
    from deephar_amd.models import spnet
    full = bench.build_speed2d()
    c = bench.SPEED2D_CFG
    assert (c['num_frames'], c['num_pyramids'], c['action_pyramids'], c['pose_replica']) == (8, 6, [1, 2, 3, 4, 5, 6], True)
    npred = spnet.get_num_predictions(6, 4)
    assert npred == 18 and len(full.outputs) == 2 * npred and full.inputs[0].shape == (8, 256, 256, 3)
    assert [o.shape for o in full.outputs[:npred]] == [(8, 16, 3)] * npred and [o.shape for o in full.outputs[npred:]] == [(15,)] * npred
    last = Model(full.input, full.outputs[2 * (npred - 1):2 * npred])
    first = Model(full.input, full.outputs[0:2])
    assert len(first.plan.steps) < 50 < 400 < len(last.plan.steps) < 480
    assert bench.WORKLOADS['speed2d']['per_gpu'] == 2 and bench.WORKLOADS['speed2d']['T'] == 8
