#!/usr/bin/env python
"""Generate tests/golden/host_golden.npz by RUNNING the reference's NumPy-only modules (this container only:
/root/reference does not exist on the GPU box).  The Keras/TensorFlow-backed modules cannot run, so only the
host-side helpers are pinned this way: soft-argmax grids (utils/math.py:6-19), affine / camera post-processing
(utils/transform.py, utils/camera.py), metrics (measures.py) and the pose-layout tables (utils/pose.py).

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/deephar'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_golden.npz')


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    # stubs so the numpy-only files import without Keras
    keras = types.ModuleType('keras')
    backend = types.ModuleType('keras.backend')
    backend.epsilon = lambda: 1e-7
    keras.backend = backend
    sys.modules.update({'keras': keras, 'keras.backend': backend})
    pkg = types.ModuleType('deephar')
    pkg.__path__ = [REF]
    utils = types.ModuleType('deephar.utils')
    utils.__path__ = [REF + '/utils']
    utils.__all__ = []
    sys.modules.update({'deephar': pkg, 'deephar.utils': utils})

    transform = _load('deephar.utils.transform', REF + '/utils/transform.py')
    camera = _load('deephar.utils.camera', REF + '/utils/camera.py')
    rmath = _load('deephar.utils.math', REF + '/utils/math.py')
    pose = _load('deephar.utils.pose', REF + '/utils/pose.py')
    measures = _load('deephar.measures', REF + '/measures.py')

    rng = np.random.default_rng(2024)
    g = {}
    for (r, c) in [(32, 32), (16, 16), (8, 8), (4, 4), (8, 16)]:
        g['grid_x_%dx%d' % (r, c)] = rmath.linspace_2d(r, c, dim=0)
        g['grid_y_%dx%d' % (r, c)] = rmath.linspace_2d(r, c, dim=1)

    n, j = 6, 16
    A = np.stack([np.array([[1 / s, 0, tx], [0, 1 / s, ty], [0, 0, 1]]) for s, tx, ty in
                  zip(rng.uniform(200, 400, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n))])
    poses = rng.uniform(0, 1, (n, j, 2))
    g['tps_A'], g['tps_poses'] = A.copy(), poses.copy()
    g['tps_out_batched'] = transform.transform_pose_sequence(A.copy(), poses.copy(), inverse=True)
    g['tps_out_single'] = transform.transform_pose_sequence(A[0].copy(), poses.copy(), inverse=False)
    g['t2d_out'] = transform.transform_2d_points(A[1], poses[0], transpose=True, inverse=True)
    frame = rng.uniform(0, 255, (5, 7, 3))
    g['norm_in'] = frame.copy()
    g['norm_out'] = transform.normalize_channels(frame.copy())
    g['norm_out_pow'] = transform.normalize_channels(frame.copy(), channel_power=(1, 2, 0.5))
    # every byte value through the loader path: T.asarray() hands normalize_channels a float32 array
    # (transform.py:122-124), so the whole normalisation runs in float32
    ramp = np.repeat(np.arange(256, dtype=np.float32)[:, None, None], 3, axis=2)
    g['norm_lut'] = transform.normalize_channels(ramp.copy())[:, 0, :].T.copy()
    g['norm_lut_pow'] = transform.normalize_channels(ramp.copy(), channel_power=(1, 2, 0.5))[:, 0, :].T.copy()

    th = 0.3
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    cam_args = (R, np.array([100., -50., 30.]), np.array([1145., 1144.]), np.array([512., 515.]),
                np.array([0.001, -0.002]))
    k = np.array([-0.2, 0.24, -0.002])
    pts = rng.uniform(-800, 800, (9, 3)) + np.array([0, 0, 4500.])
    g['cam_pts'] = pts
    for tag, kk in (('nok', None), ('k', k)):
        cam = camera.Camera(*cam_args, k=kk)
        uvd = cam.project(pts.copy())
        g['cam_uvd_' + tag] = uvd
        g['cam_back_' + tag] = cam.inverse_project(uvd.copy())
        g['cam_ser_' + tag] = cam.serialize()
    pred = rng.uniform(0, 1, (n, 17, 3))
    g['ppc_pred'], g['ppc_rootz'] = pred, rng.uniform(3000, 6000, n)
    g['ppc_out'] = camera.project_pred_to_camera(pred.copy(), A.copy(), 2000., g['ppc_rootz'].copy())

    yt = rng.uniform(0, 256, (12, 17, 3))
    yt[3, 5] = -1e9
    yt[7, 0, 1] = -1e9
    yp = yt + rng.normal(0, 60, yt.shape)
    hs = rng.uniform(40, 120, (12, 1))
    g['m_true'], g['m_pred'], g['m_head'] = yt, yp, hs
    g['m_mde'] = np.array(measures.mean_distance_error(yt, yp))
    g['m_pckh'] = np.array(measures.pckh(yt[:, :16, :2], yp[:, :16, :2], hs))
    g['m_pckh02'] = np.array(measures.pckh(yt[:, :16, :2], yp[:, :16, :2], hs, refp=0.2))
    g['m_pck3d'] = np.array(measures.pck3d(yt, yp))

    for name in ('pa16j2d', 'pa16j3d', 'pa17j2d', 'pa17j3d', 'pa20j3d', 'pa21j3d', 'coco17j'):
        lay = getattr(pose, name)
        g['pose_%s' % name] = np.array([lay.num_joints, lay.dim] + list(lay.map_hflip))
    g['pose_ntu25j3d'] = np.array([pose.ntu25j3d.num_joints, pose.ntu25j3d.dim])
    eval_tool_goldens(g, dict(transform=transform, camera=camera, pose=pose, measures=measures))
    np.savez_compressed(OUT, **g)
    print('wrote', OUT, len(g), 'arrays')


def eval_tool_goldens(g, mods):
    """Run the reference's exp/common/*_tools.py (bbox refinement loop, PCKh / mm-error drivers, single- and
    multi-clip action voting, box-from-pose) on the deterministic stand-ins of tests/evalstubs.py."""
    import contextlib
    import io as _io
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import evalstubs as S
    uio = _load('deephar.utils.io', REF + '/utils/io.py')
    fs = _load('deephar.utils.fs', REF + '/utils/fs.py')
    parser = _load('deephar.utils.parser', REF + '/utils/parser.py')
    bbox = _load('deephar.utils.bbox', REF + '/utils/bbox.py')
    utils = sys.modules['deephar.utils']
    for m in (uio, fs, bbox, mods['transform'], mods['camera'], mods['pose']):
        for k, v in vars(m).items():
            if not k.startswith('_'):
                setattr(utils, k, v)
    for k in ('TEST_MODE', 'TRAIN_MODE', 'VALID_MODE'):
        setattr(utils, k, getattr(parser, k))
    utils.__all__ = [k for k in vars(utils) if not k.startswith('_')]
    cb = types.ModuleType('keras.callbacks')
    cb.Callback = object
    data = types.ModuleType('deephar.data')
    data.BatchLoader = object
    h36 = types.ModuleType('deephar.data.human36m')
    h36.ACTION_LABELS = ['act%d' % i for i in range(16)]
    sys.modules.update({'keras.callbacks': cb, 'deephar.data': data, 'deephar.data.human36m': h36})
    np.float = float                      # removed from NumPy; the reference passes dtype=np.float ...
    _equal = np.equal                     # ... to np.equal, which NumPy 2 no longer resolves to a loop
    np.equal = lambda a, b, dtype=None: _equal(a, b) if dtype is None else _equal(a, b).astype(dtype)
    tools = {n: _load('refexp.' + n, '/root/reference/exp/common/%s.py' % n)
             for n in ('mpii_tools', 'h36m_tools', 'penn_tools', 'ntu_tools', 'generic')}
    rng = np.random.default_rng(77)
    quiet = contextlib.redirect_stdout(_io.StringIO())

    # bbox helpers
    pts = rng.uniform(0, 200, (5, 16, 2))
    pts[1, 3] = -1e9
    g['bb_pts'] = pts
    g['bb_valid'] = bbox.get_valid_bbox_array(pts)
    g['bb_valid_nosq'] = bbox.get_valid_bbox_array(pts, relsize=1.2, square=False)
    g['bb_grid'] = bbox.compute_grid_bboxes((640, 480))
    g['bb_grid_nosq'] = bbox.compute_grid_bboxes((640, 480), grid=(2, 3), square=False)
    op, ws = bbox.get_objpos_winsize(pts[0])
    g['bb_objpos'] = np.concatenate([op, ws])
    vis = (rng.uniform(0, 1, (5, 16)) > 0.3).astype(float)
    vis[2] = 0
    pts2 = pts.copy()
    pts2[2] = -1
    g['bb_vis'] = vis
    with quiet:
        g['bb_gt'] = bbox.get_gt_bbox(pts2, vis, (640, 480), scale=1.2, logkey='k')
    rootj = np.array([[320., 240., 4000.], [330., 250., 4100.]])
    o, w, z = bbox.get_crop_params(rootj, (1000, 1002), np.array([[1.1, 1.2]]), 1.3)
    g['bb_crop'] = np.concatenate([o, w, z])
    g['bb_posebbox'] = bbox.PoseBBox(pts)[1:4]      # (the reference's clip branch passes relsize as jprob and raises)

    # MPII: refinement loop and PCKh
    ds = S.FakeBoxDataset(6, seed=1)
    model = S.StubModel((8, 8, 3), [('pose', 16, 2)] * 3, ds=ds)
    with quiet:
        outs = tools['mpii_tools'].refine_pred(model, ds.frames(), ds.afmat(), ds.bbox(), ds, 2, 1, num_iter=3)
    g['mpii_refine'] = np.stack(outs)
    g['mpii_refine_log'] = np.array([len(ds.log), model.calls])
    fval = rng.uniform(-1, 1, (10, 8, 8, 3))
    pval = rng.uniform(0, 1, (10, 16, 2))
    A = np.stack([np.array([[1 / s, 0, tx], [0, 1 / s, ty], [0, 0, 1]]) for s, tx, ty in
                  zip(rng.uniform(200, 400, 10), rng.uniform(-0.5, 0.5, 10), rng.uniform(-0.5, 0.5, 10))])
    head = rng.uniform(60, 120, (10, 1))
    g['mpii_fval'], g['mpii_pval'], g['mpii_A'], g['mpii_head'] = fval, pval, A, head
    with quiet:
        g['mpii_pckh'] = np.array(tools['mpii_tools'].eval_singleperson_pckh(
            S.StubModel((8, 8, 3), [('pose', 16, 3)] * 4), fval, pval, A, head, refp=2.0, verbose=0))
        g['mpii_pckh_clip'] = np.array(tools['mpii_tools'].eval_singleperson_pckh(
            S.StubModel((4, 8, 8, 3), [('pose', 16, 3), ('action', 5)] * 2), fval, pval, A, head, refp=3.0,
            pred_per_block=2, verbose=0))
        g['mpii_abs'] = tools['mpii_tools'].absulute_pred(S.StubModel((8, 8, 3), [('pose', 16, 2)] * 2), fval, A, 1)

    # Human3.6M single-crop error
    cam = mods['camera'].Camera(np.eye(3), np.array([10., -20., 5.]), np.array([1145., 1144.]),
                                np.array([512., 515.]), np.array([0.001, -0.002]))
    n = 9
    g['h36_x'] = rng.uniform(-1, 1, (n, 8, 8, 3))
    g['h36_pw'] = rng.uniform(-800, 800, (n, 17, 3)) + np.array([0, 0, 4500.])
    g['h36_A'] = np.stack([np.array([[1 / s, 0, tx], [0, 1 / s, ty], [0, 0, 1]]) for s, tx, ty in
                           zip(rng.uniform(0.5, 2, n) * 1e-3, rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n))])
    g['h36_rootz'] = rng.uniform(3000, 6000, n)
    g['h36_scam'] = np.stack([cam.serialize()] * n)
    g['h36_action'] = rng.integers(0, 4, (n, 1))
    with quiet:
        g['h36_err'] = np.array(tools['h36m_tools'].eval_human36m_sc_error(
            S.StubModel((8, 8, 3), [('pose', 17, 4)] * 3), g['h36_x'], g['h36_pw'], g['h36_A'].copy(),
            g['h36_rootz'], g['h36_scam'], g['h36_action'], verbose=True))
        g['h36_err_clip'] = np.array(tools['h36m_tools'].eval_human36m_sc_error(
            S.StubModel((2, 8, 8, 3), [('pose', 17, 4)] * 2), g['h36_x'], g['h36_pw'], g['h36_A'].copy(),
            g['h36_rootz'], g['h36_scam'], g['h36_action'], verbose=False))

    # action: single clip, generator, multi-clip voting (Penn and NTU flavours)
    penn = S.FakeClipDataset(6, 4, 5, 'pennaction', seed=3)
    x_te = np.stack([penn.video[i, :4] for i in range(6)])
    a_te = np.eye(5)[penn.labels]
    am = lambda: S.StubModel((4, 8, 8, 3), [('action', 5)] * 3)
    with quiet, __import__('warnings').catch_warnings():
        __import__('warnings').simplefilter('ignore')
        g['act_single'] = np.array(tools['penn_tools'].eval_singleclip_gt_bbox(am(), x_te, a_te, verbose=0))
        g['act_gen_penn'] = np.array(tools['penn_tools'].eval_singleclip_gt_bbox_generator(am(), S.FakeSequence(penn), verbose=0))
        g['act_multi_penn'] = np.array(tools['penn_tools'].eval_multiclip_dataset(am(), penn, 2, verbose=0))
        ntu = S.FakeClipDataset(6, 4, 5, 'ntuaction', seed=4)
        g['act_gen_ntu'] = np.array(tools['ntu_tools'].eval_singleclip_gt_bbox_generator(am(), S.FakeSequence(ntu), verbose=0))
        g['act_multi_ntu'] = np.array(tools['ntu_tools'].eval_multiclip_dataset(am(), ntu, 2, verbose=0))
        import json
        import tempfile
        boxes = {'%04d.%d.%03d.%d' % (i, 2, f, h): [10, 20, 110 + 10 * i + f, 220] for i in range(6) for f in range(3)
                 for h in range(2) if not (i == 1 and f == 0)}
        with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as fh:
            json.dump(boxes, fh)
        ntu = S.FakeClipDataset(6, 4, 5, 'ntuaction', seed=4)
        g['act_multi_ntu_boxes'] = np.array(tools['ntu_tools'].eval_multiclip_dataset(am(), ntu, 2, bboxes_file=fh.name, verbose=0))
        g['act_multi_ntu_boxes_seen'] = np.array([len(ntu.bbox_seen), int(ntu.use_gt_bbox)])
        os.unlink(fh.name)

    # box from predicted poses
    poses = S.StubModel((8, 8, 3), [('pose', 16, 3)]).predict(fval[:3])
    g['gen_bbox'] = tools['generic'].get_bbox_from_poses(poses, A[0], scale=1.5)
    g['gen_bbox_clip'] = tools['generic'].get_bbox_from_poses(poses[None], A[1], scale=1.2)


if __name__ == '__main__':
    main()
