"""Small helpers the builders and callers share with the reference (deephar/utils/__init__.py)."""
from .pose import *  # noqa: F401,F403
from .bbox import (PoseBBox, get_valid_bbox, get_valid_bbox_array, get_objpos_winsize, compute_grid_bboxes,  # noqa: F401
                   bbox_to_objposwin, objposwin_to_bbox, get_gt_bbox, get_crop_params, get_valid_joints,
                   get_visible_joints)
from .io import (HEADER, OKBLUE, OKGREEN, WARNING, FAIL, ENDC, printc, printcn, printnl, warning, sprintcn,  # noqa: F401
                 mkdir)
from .camera import Camera, camera_deserialize, project_pred_to_camera  # noqa: F401
from .transform import transform_2d_points, transform_pose_sequence, normalize_channels  # noqa: F401

TEST_MODE, TRAIN_MODE, VALID_MODE = 0, 1, 2     # deephar/utils/parser.py:12-14


def appstr(s, a):
    """Safe string append: None stays None (deephar/utils/parser.py:254-259)."""
    try:
        return s + a
    except Exception:
        return None
