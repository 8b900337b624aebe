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
    np.savez_compressed(OUT, **g)
    print('wrote', OUT, len(g), 'arrays')


if __name__ == '__main__':
    main()
