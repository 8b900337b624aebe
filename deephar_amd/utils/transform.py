"""Affine bookkeeping used after `predict` (reference deephar/utils/transform.py:134-231): mapping normalised crop
coordinates back to image pixels, and the input value range.  Image cropping/rotation (class T) belongs to the
data pipeline and is out of scope."""
import numpy as np


def transform_2d_points(A, x, transpose=False, inverse=False):
    """T(x) = A x for 2-D points; x is [2, N] (or [N, 2] with transpose=True, or a single [2] point)
    (transform.py:134-170)."""
    x = np.asarray(x, dtype=np.float64)
    single = x.ndim == 1
    if single:
        pts = x[:, None]
    else:
        pts = x.T if transpose else x
    assert pts.shape[0] == 2, 'transform_2d_points: Only 2D points are supported, get ' + str(pts.shape[0])
    A = np.linalg.inv(A) if inverse else np.asarray(A)
    y = (A @ np.vstack([pts, np.ones((1, pts.shape[1]))]))[0:2]
    if single:
        return np.squeeze(y)
    return y.T if transpose else y


def transform_pose_sequence(A, poses, inverse=True):
    """Apply one [3,3] (or one per sample, [N,3,3]) affine map to every pose of [N, J, 2]
    (transform.py:174-209).  NB: like the reference, a batched A is inverted IN PLACE when inverse=True."""
    assert poses.ndim == 3, 'transform_pose_sequence: expected 3D tensor, got ' + str(poses.shape)
    if A.ndim == 3:
        assert len(A) == len(poses), 'A is ' + str(A.shape) + ' and poses is ' + str(poses.shape)
        if inverse:
            A[:] = np.linalg.inv(A)
        mats = A
    else:
        mats = np.broadcast_to(np.linalg.inv(A) if inverse else A, (len(poses), 3, 3))
    homo = np.concatenate([poses[..., 0:2], np.ones(poses.shape[:2] + (1,))], axis=-1)     # [N, J, 3]
    return np.einsum('nij,nkj->nki', mats, homo)[..., 0:2]


def normalize_channels(frame, channel_power=1):
    """uint8-range frame -> [-1, 1] (transform.py:212-231); in place like the reference."""
    if type(channel_power) is not int:
        assert len(channel_power) == 3
    frame /= 255.
    if type(channel_power) is int:
        if channel_power != 1:
            frame = np.power(frame, channel_power)
    else:
        for c in range(3):
            if channel_power[c] != 1:
                frame[:, :, c] = np.power(frame[:, :, c], channel_power[c])
    frame -= .5
    frame *= 2.
    return frame
