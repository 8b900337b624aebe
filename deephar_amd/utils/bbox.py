"""Bounding-box helpers used by the callers of the pose path (reference deephar/utils/bbox.py): boxes are
[x1, y1, x2, y2] float arrays; "objpos / winsize" is the (centre, (w, h)) form the croppers take.
A joint is valid when all its coordinates are > -1e6 (utils/pose.py:162-163)."""
import warnings

import numpy as np

RELSIZE = 1.5          # bbox.py:9-10
SQUARE = True


def get_valid_joints(x):
    return np.all(np.asarray(x) > -1e6, axis=-1).astype(np.int64)


def get_visible_joints(x, margin=0.0):
    x = np.asarray(x)
    return (np.all(x > margin, axis=-1) & np.all(x < 1 - margin, axis=-1)).astype(np.int64)


def _span(x, y, relsize, square):
    cx, cy = (x.min() + x.max()) / 2., (y.min() + y.max()) / 2.
    w, h = relsize * (x.max() - x.min()), relsize * (y.max() - y.min())
    if square:
        w = h = max(w, h)
    return cx, cy, w, h


def get_valid_bbox(points, jprob=None, relsize=RELSIZE, square=SQUARE):
    """Box around the valid joints of one pose, enlarged by `relsize` (bbox.py:53-76).  `jprob` (per-joint
    confidence or boolean) replaces the validity test with jprob > 0.5."""
    points = np.asarray(points)
    keep = (np.squeeze(np.asarray(jprob) > 0.5) if jprob is not None else get_valid_joints(points)).astype(bool)
    if not keep.any():
        raise ValueError('get_valid_bbox: all points are invalid!')
    cx, cy, w, h = _span(points[keep, 0], points[keep, 1], relsize, square)
    return np.array([cx - w / 2., cy - h / 2., cx + w / 2., cy + h / 2.])


def get_valid_bbox_array(pointarray, jprob=None, relsize=RELSIZE, square=SQUARE):
    return np.stack([get_valid_bbox(p, jprob=None if jprob is None else jprob[i], relsize=relsize, square=square)
                     for i, p in enumerate(pointarray)]).reshape(len(pointarray), 4)


def get_objpos_winsize(points, relsize=RELSIZE, square=SQUARE):
    """bbox.py:91-102 (no validity filtering here, like the reference)."""
    points = np.asarray(points)
    cx, cy, w, h = _span(points[:, 0], points[:, 1], relsize, square)
    return np.array([cx, cy]), (w, h)


def compute_grid_bboxes(frame_size, grid=(3, 2), relsize=RELSIZE, square=SQUARE):
    """Full frame, full frame x relsize, then a grid of overlapping windows (bbox.py:104-140)."""
    def half(a, b):
        return (max(a, b), max(a, b)) if square else (a, b)

    def box(cx, cy, rw, rh):
        return [cx - rw, cy - rh, cx + rw, cy + rh]

    cx, cy = frame_size[0] / 2, frame_size[1] / 2
    rw, rh = half(cx, cy)
    out = [box(cx, cy, rw, rh), box(cx, cy, rw * relsize, rh * relsize)]
    sx, sy = frame_size[0] / (grid[0] + 1), frame_size[1] / (grid[1] + 1)
    rw, rh = half(sx, sy)
    for j in range(1, grid[1] + 1):
        for i in range(1, grid[0] + 1):
            out.append(box(i * sx, j * sy, rw, rh))
    return np.array(out, dtype=np.float64)


def bbox_to_objposwin(bbox):
    return np.array([(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2]), (bbox[2] - bbox[0], bbox[3] - bbox[1])


def objposwin_to_bbox(objpos, winsize):
    hw, hh = winsize[0] / 2, winsize[1] / 2
    return np.array([objpos[0] - hw, objpos[1] - hh, objpos[0] + hw, objpos[1] + hh])


def _key_frames(n):
    return [0] if n == 1 else [0, int(n / 2 + 0.5), n - 1]


_warned = set()


def get_gt_bbox(pose, visible, image_size, scale=1.0, logkey=None):
    """Union of the boxes of the first / middle / last frame of a clip (bbox.py:160-199); a frame without any
    usable joint contributes the full image."""
    pose, visible = np.asarray(pose), np.asarray(visible)
    assert pose.ndim == 3 and pose.shape[-1] >= 2, \
        'Invalid pose shape ({}), expected (num_frames, num_joints, dim) vector'.format(pose.shape)
    assert len(pose) == len(visible), 'pose and visible should have the same langth'
    lo, hi = np.array([np.inf, np.inf]), np.array([-np.inf, -np.inf])
    for i in _key_frames(len(pose)):
        pts = pose[i, visible[i] >= 0.5]
        if len(pts) == 0:
            pts = pose[i, pose[i] > 0]
        if len(pts) > 0:
            b = get_valid_bbox(pts, relsize=1.5 * scale)
        else:
            if logkey not in _warned:
                warnings.warn('No ground-truth bounding box, using full image (key {})!'.format(logkey))
            _warned.add(logkey)
            b = np.array([0., 0., image_size[0], image_size[1]])
        lo, hi = np.minimum(lo, b[:2]), np.maximum(hi, b[2:])
    return np.concatenate([lo, hi])


def get_crop_params(rootj, imgsize, f, scale):
    """Crop window and depth range from the root joint (bbox.py:202-229).  Like the reference, every key frame
    reads rootj[0] / f[0] -- the loop index is not used there."""
    rootj = np.asarray(rootj)
    assert rootj.ndim == 2 and rootj.shape[-1] == 3, \
        'Invalid rootj shape ({}), expected (n, 3) vector'.format(rootj.shape)
    d = rootj[0, 2]
    win = (2.25 * scale) * max(imgsize[0] * f[0, 0] / d, imgsize[1] * f[0, 1] / d)
    box = objposwin_to_bbox(np.array([rootj[0, 0], rootj[0, 1] + scale]), (win, win))
    objpos, winsize = bbox_to_objposwin(box)
    return objpos, winsize, np.array([d - scale * 1000., d + scale * 1000.])


class PoseBBox:
    """Lazy per-sample boxes over an array of poses [N, J, D] or clips [N, T, J, D] (bbox.py:12-51)."""

    def __init__(self, poses, relsize=RELSIZE, square=SQUARE):
        self.poses, self.relsize, self.square = poses, relsize, square
        self.num_frames = poses.shape[1] if poses.ndim == 4 else None

    def __len__(self):
        return len(self.poses)

    @property
    def shape(self):
        return (len(self), 4) if self.num_frames is None else (len(self), self.num_frames, 4)

    def _one(self, p):
        if self.num_frames is None:
            return get_valid_bbox(p, relsize=self.relsize, square=self.square)
        return np.stack([get_valid_bbox(p[f], None, self.relsize, self.square) for f in range(self.num_frames)])

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self._one(self.poses[key])
        idx = range(*key.indices(len(self))) if isinstance(key, slice) else list(key)
        out = np.zeros((len(idx),) + self.shape[1:])
        for k, i in enumerate(idx):
            out[k] = self._one(self.poses[i])
        return out
