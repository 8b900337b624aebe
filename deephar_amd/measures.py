"""Evaluation metrics on host arrays (reference deephar/measures.py:17-111), vectorised NumPy.
A joint is valid when every coordinate of the ground truth is > -1e6 (the reference's missing-joint marker)."""
import numpy as np

_MPII_PCKH_JOINTS = [2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15, 8, 9]   # measures.py:64 (pelvis/thorax ignored)
_PCK3D_JOINTS = list(range(1, 17))                                      # measures.py:94


def _valid(y, min_valid=-1e6):
    return np.all(y > min_valid, axis=-1).astype(np.float64)


def _dist(a, b):
    return np.sqrt(np.sum(np.square(a - b), axis=-1))


def mean_distance_error(y_true, y_pred):
    """MPJPE-style mean Euclidean error over valid joints (measures.py:17-43)."""
    assert y_true.shape == y_pred.shape
    valid = _valid(y_true)
    return float((_dist(y_true, y_pred) * valid).sum() / valid.sum())


def pckh(y_true, y_pred, head_size, refp=0.5):
    """PCKh@refp on the 14 MPII evaluation joints (measures.py:45-75)."""
    assert y_true.shape == y_pred.shape
    assert len(y_true) == len(head_size)
    yt, yp = y_true[:, _MPII_PCKH_JOINTS, :], y_pred[:, _MPII_PCKH_JOINTS, :]
    valid = _valid(yt)
    dist = _dist(yt, yp) / np.reshape(head_size, (len(yt), 1))
    return float(((dist <= refp) * valid).sum() / valid.sum())


def pck3d(y_true, y_pred, refp=150):
    """PCK3D with an absolute threshold in mm on joints 1..16 (measures.py:78-105)."""
    assert y_true.shape == y_pred.shape
    yt, yp = y_true[:, _PCK3D_JOINTS, :], y_pred[:, _PCK3D_JOINTS, :]
    valid = _valid(yt)
    return float(((_dist(yt, yp) <= refp) * valid).sum() / valid.sum())


def pckh_per_joint(y_true, y_pred, head_size, pose_layout, refp=0.5, verbose=0):
    """Per-joint PCKh (measures.py:108-149); returns the array instead of only printing it."""
    assert y_true.shape == y_pred.shape and len(y_true) == len(head_size)
    valid = _valid(y_true)
    dist = _dist(y_true, y_pred) / np.reshape(head_size, (len(y_true), 1))
    scores = ((dist <= refp) * valid).sum(axis=0) / valid.sum(axis=0)
    if verbose:
        print(' | '.join('%.2f' % (100 * s) for s in scores))
    return scores
