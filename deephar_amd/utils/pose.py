"""Pose layouts: joint counts / dimensionality consumed by ModelConfig (reference deephar/utils/pose.py:127-143,
config.py:171-172).  Only what the hot path reads (num_joints, dim) plus the horizontal-flip maps used by the
multi-clip evaluation (penn_tools.py:117-127) is kept; plotting tables are out of scope."""


class _Layout:
    num_joints = 0
    dim = 0
    map_hflip = None


def _layout(name, joints, dim, hflip=None):
    return type(name, (_Layout,), dict(num_joints=joints, dim=dim, map_hflip=hflip))


_HFLIP16 = [0, 1, 2, 3, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14]
pa16j2d = _layout('pa16j2d', 16, 2, _HFLIP16)
pa16j3d = _layout('pa16j3d', 16, 3, _HFLIP16)
pa17j2d = _layout('pa17j2d', 17, 2, _HFLIP16 + [16])
pa17j3d = _layout('pa17j3d', 17, 3, _HFLIP16 + [16])
pa20j3d = _layout('pa20j3d', 20, 3, _HFLIP16 + [17, 16, 19, 18])
pa21j3d = _layout('pa21j3d', 21, 3, _HFLIP16 + [17, 16, 19, 18, 20])
coco17j = _layout('coco17j', 17, 2, [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15])
ntu25j3d = _layout('ntu25j3d', 25, 3)
