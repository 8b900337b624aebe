"""Pin-hole camera (reference deephar/utils/camera.py:6-121): pixel+depth <-> world millimetres, used to turn the
3-D network output into MPJPE numbers (exp/common/h36m_tools.py:72-91)."""
import numpy as np

from .transform import transform_pose_sequence


def get_r2_radial_tan(x, k, p):
    """camera.py:83-94"""
    assert x.ndim == 2 and x.shape[1] == 2
    assert k.shape == (3,) and p.shape == (1, 2)
    r2 = np.square(x[:, 0]) + np.square(x[:, 1])
    radial = 1. + r2 * k[0] + r2 ** 2 * k[1] + r2 ** 3 * k[2]
    return r2, radial, np.sum(x * p, axis=-1)


class Camera(object):
    """R (3,3), t (3,), f (2,), c (2,), p (2,) skew, optional k (3,) radial distortion (camera.py:6-30)."""

    def __init__(self, R, t, f, c, p, k=None):
        self.R = R
        self.R_inv = np.linalg.inv(self.R)
        self.t = np.reshape(t, (3, 1))
        self.f = np.reshape(f, (1, 2))
        self.c = np.reshape(c, (1, 2))
        self.p = np.reshape(p, (1, 2))
        self.k = None if k is None else np.reshape(k, (3,))

    def project(self, points_w):
        """world mm [n,3] -> (u, v, depth) (camera.py:32-50)"""
        assert points_w.ndim == 2 and points_w.shape[1] == 3
        x = (self.R @ (points_w.T - self.t)).T
        x[:, 0:2] /= x[:, 2:3]
        if self.k is not None:
            r2, radial, tan = get_r2_radial_tan(x[:, 0:2], self.k, self.p)
            x[:, 0:2] *= (radial + tan)[:, None]
            x[:, 0:2] += r2[:, None] @ self.p
        x[:, 0:2] = x[:, 0:2] * self.f + self.c
        return x

    def inverse_project(self, points_uvd):
        """(u, v, depth) [n,3] -> world mm (camera.py:52-71)"""
        assert points_uvd.ndim == 2 and points_uvd.shape[1] == 3
        x = points_uvd.copy()
        x[:, 0:2] = (x[:, 0:2] - self.c) / self.f
        if self.k is not None:
            r2, radial, tan = get_r2_radial_tan(x[:, 0:2], self.k, self.p)
            x[:, 0:2] -= r2[:, None] @ self.p
            x[:, 0:2] /= (radial + tan)[:, None]
        x[:, 0:2] *= x[:, 2:3]
        return (self.R_inv @ x.T + self.t).T

    def serialize(self):
        parts = [np.array(self.R).reshape(9), self.t.reshape(3), self.f.reshape(2), self.c.reshape(2),
                 self.p.reshape(2)] + ([self.k] if self.k is not None else [])
        return np.concatenate(parts)


def camera_deserialize(s):
    """camera.py:97-109"""
    s = np.asarray(s)
    k = s[18:21] if len(s) > 18 else None
    return Camera(s[0:9].reshape(3, 3), s[9:12], s[12:14], s[14:16], s[16:18], k)


def project_pred_to_camera(pred, afmat, resol_z, root_z):
    """Network output [N,J,3] in crop units -> (u, v, absolute depth) (camera.py:112-121)."""
    proj = np.zeros(pred.shape)
    proj[:, :, 0:2] = transform_pose_sequence(afmat, pred[:, :, 0:2], inverse=True)
    proj[:, :, 2] = resol_z * (pred[:, :, 2] - 0.5) + np.expand_dims(root_z, axis=-1)
    return proj
