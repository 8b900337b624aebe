"""SURVEY.md 8b's plan / execute pair: Model.export_plan -> dh_plan_create / dh_forward / dh_forward_host /
dh_plan_destroy.  The C-level executor replays the bound, autotuned launch list on its own arena and weight image;
results must be the bits Model.predict returns.  Also a host with no Python in it: tools/c_host/predict_plan.c,
compiled with gcc against include/deephar_hip.h, run as a separate process."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(lib, blob):
    h = C.c_void_p()
    rc = lib.dh_plan_create(blob, len(blob), C.byref(h))
    assert rc == 0, rc
    return h


@pytest.mark.parametrize('kind', ['reception2d', 'reception3d', 'merge', 'spnet'])
def test_c_plan_reproduces_predict_bit_for_bit(kind, hip_lib, cuda, tmp_path):
    from test_gpu_models import _build, _merge, _spnet
    rng = np.random.default_rng(17)
    if kind == 'spnet':          # [r06] planner rule R11 (dh_dw_args.up_in) and the grouped launches (dh_conv2d_dw_group_f32) in a blob
        m, _, _, _ = _spnet(8, 'pa16j2d', 15, 2, [1, 2], 160, replica=True, res=128)
        x = rng.uniform(-1, 1, (3, 8, 128, 128, 3)).astype(np.float32)
    elif kind == 'reception2d':
        m, _ = _build(2, 2, 16, num_context_per_joint=2, concat_pose_confidence=False)
        x = rng.uniform(-1, 1, (5, 256, 256, 3)).astype(np.float32)
    elif kind == 'reception3d':
        m, _ = _build(3, 2, 17, depth_maps=16)
        x = rng.uniform(-1, 1, (5, 256, 256, 3)).astype(np.float32)
    else:
        m, _ = _merge(2, 4, 16, 2, num_actions=15)
        x = rng.uniform(-1, 1, (3, 4, 256, 256, 3)).astype(np.float32)
    n = len(x)
    ref = m.predict(x, batch_size=n)
    ref = ref if isinstance(ref, list) else [ref]
    if kind == 'spnet':
        bp = m.executor.bound[n]
        assert bp.grouped == 3 and sum(1 for s_ in m.plan.steps if s_.kind == 'dwconv' and s_.attrs.get('up_in')) == 3
    path = str(tmp_path / 'model.dhplan')
    nbytes = m.export_plan(path, n)
    blob = open(path, 'rb').read()
    assert len(blob) == nbytes and blob[:4] == b'DHPL'
    plan = _plan(hip_lib, blob)
    try:
        assert hip_lib.dh_plan_batch(plan) == n and hip_lib.dh_plan_num_inputs(plan) == 1
        assert hip_lib.dh_plan_num_outputs(plan) == len(ref)
        assert hip_lib.dh_plan_input_items(plan, 0) == x[0].size
        assert [hip_lib.dh_plan_output_items(plan, k) for k in range(len(ref))] == [r[0].size for r in ref]
        # device pointers, caller's stream
        xd = torch.from_numpy(x).to(cuda)
        outs = [torch.full(r.shape, float('nan'), device=cuda) for r in ref]
        ins_p = (C.c_void_p * 1)(xd.data_ptr())
        outs_p = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        st = torch.cuda.current_stream().cuda_stream
        assert hip_lib.dh_forward(plan, ins_p, n, outs_p, st) == 0
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            assert np.array_equal(o.cpu().numpy(), r)
        # fewer items than the plan was bound for; host pointers
        mrows = n - 2
        host = [np.full((mrows,) + r.shape[1:], np.nan, np.float32) for r in ref]
        xin = np.ascontiguousarray(x[:mrows])
        ins_h = (C.c_void_p * 1)(xin.ctypes.data)
        outs_h = (C.c_void_p * len(host))(*[h.ctypes.data for h in host])
        assert hip_lib.dh_forward_host(plan, ins_h, mrows, outs_h) == 0
        for h, r in zip(host, ref):
            assert np.array_equal(h, r[:mrows])
        assert hip_lib.dh_forward_host(plan, ins_h, n + 1, outs_h) != 0          # more items than the plan holds
    finally:
        assert hip_lib.dh_plan_destroy(plan) == 0
    # malformed blobs are refused, not executed
    h = C.c_void_p()
    assert hip_lib.dh_plan_create(blob[:100], 100, C.byref(h)) != 0
    assert hip_lib.dh_plan_create(b'XXXX' + blob[4:], len(blob), C.byref(h)) != 0
    assert hip_lib.dh_plan_create(blob[:-16], len(blob) - 16, C.byref(h)) != 0


@pytest.mark.parametrize('kind', ['reception2d', 'merge'])
def test_c_plan_with_uint8_frames(kind, hip_lib, cuda, tmp_path):
    """Round 4 (VERDICT r03 missing 5): Model.export_plan(..., uint8=True) -- the plan takes raw uint8 frames (region 3 of
    the blob: byte staging, normalised on the GPU inside the first convolution) and reproduces predict(uint8 array) bit
    for bit, through device and host pointers, also for fewer items than the plan holds."""
    from test_gpu_models import _build, _merge
    rng = np.random.default_rng(23)
    if kind == 'reception2d':
        m, _ = _build(2, 2, 16, num_context_per_joint=2, concat_pose_confidence=False)
        x = rng.integers(0, 256, (5, 256, 256, 3), dtype=np.uint8)
    else:
        m, _ = _merge(2, 4, 16, 2, num_actions=15)
        x = rng.integers(0, 256, (3, 4, 256, 256, 3), dtype=np.uint8)
    n = len(x)
    ref = m.predict(x, batch_size=n)
    path = str(tmp_path / 'model_u8.dhplan')
    m.export_plan(path, n, uint8=True)
    blob = open(path, 'rb').read()
    plan = _plan(hip_lib, blob)
    try:
        assert hip_lib.dh_plan_input_is_u8(plan, 0) == 1 and hip_lib.dh_plan_input_is_u8(plan, 1) == -1
        assert hip_lib.dh_plan_input_items(plan, 0) == x[0].size
        xd = torch.from_numpy(x).to(cuda)
        outs = [torch.full(r.shape, float('nan'), device=cuda) for r in ref]
        ins_p = (C.c_void_p * 1)(xd.data_ptr())
        outs_p = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        assert hip_lib.dh_forward(plan, ins_p, n, outs_p, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            assert np.array_equal(o.cpu().numpy(), r)
        mrows = n - 1
        host = [np.full((mrows,) + r.shape[1:], np.nan, np.float32) for r in ref]
        xin = np.ascontiguousarray(x[:mrows])
        ins_h = (C.c_void_p * 1)(xin.ctypes.data)
        outs_h = (C.c_void_p * len(host))(*[h.ctypes.data for h in host])
        assert hip_lib.dh_forward_host(plan, ins_h, mrows, outs_h) == 0
        for h, r in zip(host, ref):
            assert np.array_equal(h, r[:mrows])
    finally:
        assert hip_lib.dh_plan_destroy(plan) == 0
    # a float plan of the same model says so
    m.export_plan(path, n)
    fplan = _plan(hip_lib, open(path, 'rb').read())
    assert hip_lib.dh_plan_input_is_u8(fplan, 0) == 0
    hip_lib.dh_plan_destroy(fplan)


def test_frame_sharded_clip_model_as_two_c_plans(hip_lib, cuda, tmp_path):
    """VERDICT r03 missing 5: the frame-stage / head-stage pair of a frame-sharded clip model as C-level plans
    (ShardedClipModel.export_plans).  Two 'ranks' in one process: each runs the frame plan on its T/2 frames, the packed
    tensors are gathered rank-major by the host (what RCCL's all-gather writes), the head plan runs on the channel runs of
    the gathered buffer -- the model's outputs bit for bit (configs[3]: pose + action merge model)."""
    import json
    from deephar_amd import parallel
    from test_gpu_models import _merge
    m, _ = _merge(2, 8, 16, 2, num_actions=15)
    clips = np.random.default_rng(31).uniform(-1, 1, (2, 8, 256, 256, 3)).astype(np.float32)
    full = m.predict(clips, batch_size=2)
    world, n = 2, len(clips)
    scm = parallel.ShardedClipModel(m, rank=0, world=world, frame_fn=lambda x: x, head_fn=lambda t: t)
    info = scm.export_plans(str(tmp_path / 'merge'), n)
    assert json.load(open(tmp_path / 'merge.cut.json'))['packed_channels'] == info['packed_channels'] == 581
    fplan = _plan(hip_lib, open(tmp_path / 'merge.frames.dhplan', 'rb').read())
    hplan = _plan(hip_lib, open(tmp_path / 'merge.head.dhplan', 'rb').read())
    try:
        tl, cp = info['Tl'], info['packed_channels']
        assert hip_lib.dh_plan_num_outputs(fplan) == 1 and hip_lib.dh_plan_num_inputs(hplan) == len(info['cut'])
        gathered = np.empty((world, n, tl) + tuple(info['cut'][0]['shape'][:-1]) + (cp,), np.float32)     # rank-major
        for r in range(world):
            xin = np.ascontiguousarray(clips[:, r * tl:(r + 1) * tl])
            ins = (C.c_void_p * 1)(xin.ctypes.data)
            outs = (C.c_void_p * 1)(gathered[r].ctypes.data)
            assert hip_lib.dh_forward_host(fplan, ins, n, outs) == 0
        frames = np.moveaxis(gathered, 0, 1).reshape((n, world * tl) + gathered.shape[3:])                # [N, T, J, Cp]
        parts = [np.ascontiguousarray(frames[..., c['offset']:c['offset'] + c['channels']]) for c in info['cut']]
        for k, c in enumerate(info['cut']):
            assert hip_lib.dh_plan_input_items(hplan, k) == parts[k][0].size
        head = [np.empty((n, hip_lib.dh_plan_output_items(hplan, k)), np.float32)
                for k in range(hip_lib.dh_plan_num_outputs(hplan))]
        ins = (C.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
        outs = (C.c_void_p * len(head))(*[h.ctypes.data for h in head])
        assert hip_lib.dh_forward_host(hplan, ins, n, outs) == 0
        for k, ci in info['passthrough'].items():
            assert np.array_equal(parts[ci], full[int(k)]), ('passthrough', k)
        assert len(head) == len(info['head_outputs'])
        for k, o in zip(info['head_outputs'], head):
            assert np.array_equal(o.reshape(full[k].shape), full[k]), ('head output', k)
    finally:
        hip_lib.dh_plan_destroy(fplan)
        hip_lib.dh_plan_destroy(hplan)


def test_version_1_blobs_are_still_read(hip_lib, cuda, tmp_path):
    """A blob in the round-3 layout (version 1: no u8 region size in the header, inputs as bare arena offsets without a
    dtype) still loads and runs."""
    import struct
    from test_gpu_models import _build
    m, _ = _build(2, 1, 16, num_context_per_joint=2, concat_pose_confidence=True)
    x = np.random.default_rng(29).uniform(-1, 1, (2, 256, 256, 3)).astype(np.float32)
    ref = m.predict(x, batch_size=2)
    ref = ref if isinstance(ref, list) else [ref]
    path = str(tmp_path / 'm.dhplan')
    m.export_plan(path, 2)
    b = open(path, 'rb').read()
    ver, n, arena, wbytes, nin, nout, nsteps, u8b = struct.unpack_from('<IiQQIIIQ', b, 4)
    assert (ver, nin, u8b) == (2, 1, 0)
    pos = 4 + struct.calcsize('<IiQQIIIQ')
    tag, items, dtype, _ = struct.unpack_from('<QQII', b, pos)
    assert tag >> 60 == 1 and dtype == 0
    v1 = b[:4] + struct.pack('<IiQQIII', 1, n, arena, wbytes, nin, nout, nsteps) + \
        struct.pack('<QQ', tag & ((1 << 60) - 1), items) + b[pos + struct.calcsize('<QQII'):]
    plan = _plan(hip_lib, v1)
    try:
        host = [np.empty_like(r) for r in ref]
        ins_h = (C.c_void_p * 1)(x.ctypes.data)
        outs_h = (C.c_void_p * len(host))(*[h.ctypes.data for h in host])
        assert hip_lib.dh_forward_host(plan, ins_h, 2, outs_h) == 0
        for h, r in zip(host, ref):
            assert np.array_equal(h, r)
    finally:
        hip_lib.dh_plan_destroy(plan)


def test_pure_c_host_runs_an_exported_plan(hip_lib, cuda, tmp_path):
    """gcc-compiled host (no Python, no HIP headers): plan file + raw frames in, raw outputs out, equal to predict."""
    from test_gpu_models import _build
    m, _ = _build(2, 2, 16, num_context_per_joint=2, concat_pose_confidence=True)
    x = np.random.default_rng(18).uniform(-1, 1, (4, 256, 256, 3)).astype(np.float32)
    ref = m.predict(x, batch_size=4)
    m.export_plan(str(tmp_path / 'm.dhplan'), 4)
    x.tofile(str(tmp_path / 'x.f32'))
    exe = str(tmp_path / 'predict_plan')
    libdir = os.path.join(ROOT, 'deephar_amd', 'csrc')
    r = subprocess.run(['gcc', '-O2', '-I' + os.path.join(ROOT, 'include'),
                        os.path.join(ROOT, 'tools', 'c_host', 'predict_plan.c'), '-L' + libdir, '-ldeephar_hip',
                        '-Wl,-rpath,' + libdir, '-o', exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, str(tmp_path / 'm.dhplan'), str(tmp_path / 'x.f32'), str(tmp_path / 'out')],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())
    for k, want in enumerate(ref):
        got = np.fromfile(str(tmp_path / ('out.%d.f32' % k)), np.float32).reshape(want.shape)
        assert np.array_equal(got, want), k
