"""Golden vectors produced by RUNNING the reference's NumPy-only modules (tests/golden/make_golden.py, this
container only) pin: the float32 soft-argmax grids used by the oracle AND by the HIP engine, the affine /
camera post-processing, the metrics and the pose-layout tables."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'host_golden.npz'))


@pytest.mark.parametrize('shape', [(32, 32), (16, 16), (8, 8), (4, 4), (8, 16)])
def test_softargmax_grids_bit_exact(shape):
    """utils/math.py:6-19 -> oracle.ops.linspace_2d (what the oracle multiplies with) and
    engine.executor.grid_x (what the kernel multiplies with) are bit-identical to the reference's arrays."""
    from oracle import ops
    from deephar_amd.engine.executor import grid_x
    r, c = shape
    gx, gy = G['grid_x_%dx%d' % shape], G['grid_y_%dx%d' % shape]
    assert gx.dtype == np.float32 and gx.shape == (r, c)
    assert np.array_equal(ops.linspace_2d(r, c, 0), gx) and np.array_equal(ops.linspace_2d(r, c, 1), gy)
    assert np.array_equal(np.broadcast_to(grid_x(c)[None, :], (r, c)), gx)
    assert np.array_equal(np.broadcast_to(grid_x(r)[:, None], (r, c)), gy)


def test_transform_pose_sequence_and_points():
    from deephar_amd.utils.transform import transform_pose_sequence, transform_2d_points, normalize_channels
    A, poses = G['tps_A'], G['tps_poses']
    np.testing.assert_allclose(transform_pose_sequence(A.copy(), poses.copy(), inverse=True), G['tps_out_batched'],
                               rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(transform_pose_sequence(A[0].copy(), poses.copy(), inverse=False),
                               G['tps_out_single'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(transform_2d_points(A[1], poses[0], transpose=True, inverse=True), G['t2d_out'],
                               rtol=1e-12, atol=1e-9)
    Ain = A.copy()
    transform_pose_sequence(Ain, poses.copy(), inverse=True)
    np.testing.assert_allclose(Ain, np.linalg.inv(A))          # the reference inverts a batched A in place
    np.testing.assert_allclose(normalize_channels(G['norm_in'].copy()), G['norm_out'], rtol=1e-14)
    np.testing.assert_allclose(normalize_channels(G['norm_in'].copy(), channel_power=(1, 2, 0.5)),
                               G['norm_out_pow'], rtol=1e-14)
    assert G['norm_out'].min() >= -1 and G['norm_out'].max() <= 1


def test_camera_roundtrip_and_projection():
    from deephar_amd.utils.camera import Camera, camera_deserialize, project_pred_to_camera
    for tag in ('nok', 'k'):
        cam = camera_deserialize(G['cam_ser_' + tag])
        assert (cam.k is None) == (tag == 'nok')
        np.testing.assert_allclose(cam.project(G['cam_pts'].copy()), G['cam_uvd_' + tag], rtol=1e-12)
        np.testing.assert_allclose(cam.inverse_project(G['cam_uvd_' + tag].copy()), G['cam_back_' + tag], rtol=1e-10)
        np.testing.assert_allclose(cam.serialize(), G['cam_ser_' + tag])
    np.testing.assert_allclose(G['cam_back_nok'], G['cam_pts'], rtol=1e-9)   # exact inverse without distortion
    out = project_pred_to_camera(G['ppc_pred'].copy(), G['tps_A'].copy(), 2000., G['ppc_rootz'].copy())
    np.testing.assert_allclose(out, G['ppc_out'], rtol=1e-12, atol=1e-9)


def test_measures():
    from deephar_amd import measures
    yt, yp, hs = G['m_true'], G['m_pred'], G['m_head']
    assert abs(measures.mean_distance_error(yt, yp) - float(G['m_mde'])) < 1e-9
    assert measures.pckh(yt[:, :16, :2], yp[:, :16, :2], hs) == float(G['m_pckh'])
    assert measures.pckh(yt[:, :16, :2], yp[:, :16, :2], hs, refp=0.2) == float(G['m_pckh02'])
    assert measures.pck3d(yt, yp) == float(G['m_pck3d'])
    assert 0 < float(G['m_pckh02']) < float(G['m_pckh']) < 1      # the fixture discriminates


def test_pose_layout_tables():
    from deephar_amd import utils
    for name in ('pa16j2d', 'pa16j3d', 'pa17j2d', 'pa17j3d', 'pa20j3d', 'pa21j3d', 'coco17j'):
        ref = G['pose_' + name]
        lay = getattr(utils, name)
        assert (lay.num_joints, lay.dim) == (int(ref[0]), int(ref[1]))
        assert list(lay.map_hflip) == [int(v) for v in ref[2:]]
    assert (utils.ntu25j3d.num_joints, utils.ntu25j3d.dim) == tuple(int(v) for v in G['pose_ntu25j3d'])


def test_uint8_normalisation_table_is_the_loaders_arithmetic():
    """engine.executor.normalization_lut (what the first convolution applies to raw uint8 frames) against the
    reference's normalize_channels run on float32 frames holding every byte value (transform.py:122-124,212-231)."""
    from deephar_amd.engine.executor import normalization_lut
    assert G['norm_lut'].dtype == np.float32 and G['norm_lut'].shape == (3, 256)
    assert np.array_equal(normalization_lut(3), G['norm_lut'])
    assert np.array_equal(normalization_lut(3, (1, 2, 0.5)), G['norm_lut_pow'])
    assert normalization_lut(3)[0, 0] == -1.0 and normalization_lut(3)[0, 255] == 1.0
