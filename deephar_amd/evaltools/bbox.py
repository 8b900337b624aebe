"""Callers that wrap `predict` in a pose -> box -> re-crop loop: the iterative box refinement of the MPII
evaluation (exp/common/mpii_tools.py:13-44), boxes from predicted poses (exp/common/generic.py:7-27) and the
per-frame box prediction pass of exp/pennaction/predict_bboxes.py:53-68 / exp/ntu/predict_bboxes.py."""
import numpy as np

from ..utils import (bbox_to_objposwin, get_valid_bbox_array, objposwin_to_bbox, transform_2d_points,
                     transform_pose_sequence)


def refine_pred(model, frames, afmat, bbox, ds, mode, outidx, num_iter=2, winsize_scale=1.50, momentum=0.8,
                batch_size=8):
    """Predict, turn every predicted pose into a tighter box (centre blended with the current box by
    `momentum`, side = winsize_scale x the larger pose extent), hand the boxes to the dataset
    (`ds.set_custom_bboxes`) so that `frames` / `afmat` / `bbox` -- live views of the dataset -- re-crop, and
    predict again.  Returns the list of per-iteration poses in image coordinates."""
    out, refined = [], []
    for t in range(num_iter):
        ds.set_custom_bboxes(mode, refined)
        pred = model.predict(frames, batch_size=batch_size, verbose=1)[outidx]
        A, current = afmat[:], bbox[:]
        if len(refined) == 0:
            refined = current.copy()
        pred = transform_pose_sequence(A.copy(), pred, inverse=True)
        out.append(pred)
        if t == num_iter - 1:
            break
        lo, hi = pred[:, :, 0:2].min(axis=1), pred[:, :, 0:2].max(axis=1)
        for i in range(len(pred)):
            centre_p = np.array([(lo[i, 0] + hi[i, 0]) / 2, (lo[i, 1] + hi[i, 1]) / 2])
            side = winsize_scale * max(hi[i, 0] - lo[i, 0], hi[i, 1] - lo[i, 1])
            centre_t, _ = bbox_to_objposwin(current[i])
            refined[i, :] = objposwin_to_bbox(momentum * centre_t + (1 - momentum) * centre_p, (side, side))
    ds.clear_custom_bboxes(mode)
    return out


def get_bbox_from_poses(poses, afmat, scale=1.5):
    """One box (image coordinates) around all confident joints (> 0.25) of a batch of predicted poses
    [N, J, >=3] or of the first clip of [1, T, J, >=3]; the confidence is read from channel -2 like the
    reference does (generic.py:9-13)."""
    if poses.ndim == 3:
        sel = poses
    elif poses.ndim == 4:
        sel = poses[0]
    else:
        raise ValueError('Invalid poses shape {}'.format(poses.shape))
    boxes = get_valid_bbox_array(sel[:, :, 0:2], jprob=sel[:, :, -2:-1] > 0.25, relsize=scale)
    corners = np.array([[boxes[:, 0].min(), boxes[:, 1].min()], [boxes[:, 2].max(), boxes[:, 3].max()]])
    c = np.reshape(transform_2d_points(afmat, corners, transpose=True, inverse=True), (4,))
    return np.array([min(c[0], c[2]), min(c[1], c[3]), max(c[0], c[2]), max(c[1], c[3])])


def predict_frame_bboxes(model, ds, mode, scale=1.5, key=None):
    """The loop of exp/pennaction/predict_bboxes.py:53-68: one forward per sample, box from the predicted pose,
    integer box keyed '<seq_idx>.<frame>' (or key(data, i))."""
    boxes = {}
    for i in range(ds.get_length(mode)):
        data = ds.get_data(i, mode)
        poses = model.predict(np.expand_dims(data['frame'], axis=0))
        k = key(data, i) if key is not None else '%d.%d' % (data['seq_idx'], data['frame_list'][0])
        boxes[k] = get_bbox_from_poses(poses, data['afmat'], scale=scale).astype(int).tolist()
    return boxes
