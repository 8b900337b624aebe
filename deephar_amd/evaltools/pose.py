"""Pose-accuracy drivers: PCKh on MPII (exp/common/mpii_tools.py:55-128) and the single-crop mm error on
Human3.6M (exp/common/h36m_tools.py:12-137)."""
import numpy as np

from ..measures import mean_distance_error, pckh, pckh_per_joint
from ..utils import camera_deserialize, pa16j2d, transform_pose_sequence


def _say(verbose, *a):
    if verbose:
        print(*a)


def _fold_clips(num_frames, *arrays):
    """Video models take [N/T, T, ...]: drop the tail that does not fill a clip (mpii_tools.py:62-72)."""
    n = (len(arrays[0]) // num_frames) * num_frames
    return [a[:n] for a in arrays]


def _block_poses(y, is_clip, dim):
    """[N, (T,) J, >=dim] -> [N(*T), J, dim]"""
    y = y[..., 0:dim]
    return np.reshape(y, (-1,) + y.shape[-2:]) if is_clip else y


def absulute_pred(model, frames, afmat, outidx, batch_size=8):
    """Predicted poses of output `outidx` mapped back to image coordinates (mpii_tools.py:46-52; the name keeps
    the reference's spelling)."""
    pred = model.predict(frames, batch_size=batch_size, verbose=1)[outidx]
    return transform_pose_sequence(afmat[:].copy(), pred, inverse=True)


def eval_singleperson_pckh(model, fval, pval, afmat_val, headsize_val, win=None, batch_size=8, refp=0.5,
                           map_to_pa16j=None, pred_per_block=1, verbose=1):
    """PCKh of every prediction block; returns the list of scores (fractions)."""
    in_shape = model.get_input_shape_at(0)
    is_clip = len(in_shape) == 5
    if is_clip:
        fval, pval, afmat_val, headsize_val = _fold_clips(in_shape[1], fval, pval, afmat_val, headsize_val)
        fval = np.reshape(fval, (-1, in_shape[1]) + fval.shape[1:])
    num_blocks = int(len(model.outputs) / pred_per_block)
    inputs = [fval]
    if win is not None:
        num_blocks -= 1
        inputs.append(win)
    pred = model.predict(inputs, batch_size=batch_size, verbose=1)
    if win is not None:
        del pred[0]

    A = afmat_val[:]
    y_true = transform_pose_sequence(A.copy(), pval[:], inverse=True)
    if map_to_pa16j is not None:
        y_true = y_true[:, map_to_pa16j, :]
    _say(verbose, 'PCKh on validation:')
    scores = []
    for b in range(num_blocks):
        y_pred = _block_poses(pred[pred_per_block * b] if num_blocks > 1 else pred, is_clip, 2)
        if map_to_pa16j is not None:
            y_pred = y_pred[:, map_to_pa16j, :]
        y_pred = transform_pose_sequence(A.copy(), y_pred, inverse=True)
        scores.append(pckh(y_true, y_pred, headsize_val, refp=refp))
        _say(verbose, ' %.1f' % (100 * scores[-1]))
        if b == num_blocks - 1:
            pckh_per_joint(y_true, y_pred, headsize_val, pa16j2d, verbose=verbose)
    return scores


def eval_human36m_sc_error(model, x, pose_w, afmat, rootz, scam, action, resol_z=2000., batch_size=8,
                           map_to_pa17j=None, logdir=None, verbose=True, action_labels=None):
    """Root-relative mean joint error in mm for every block: predictions go crop -> image plane
    (inverse affine) -> absolute depth (resol_z, root z) -> world (camera un-projection).  Returns the list of
    errors; per-action errors of the best block are printed.  `action_labels` replaces the reference's
    dataset-global ACTION_LABELS (deephar/data/human36m.py:10,58-59)."""
    assert len(x) == len(pose_w) == len(afmat) == len(scam) == len(action)
    in_shape = model.input_shape
    is_clip = len(in_shape) == 5
    if is_clip:
        x, pose_w, afmat, rootz, scam, action = _fold_clips(in_shape[1], x, pose_w, afmat, rootz, scam, action)
        x = np.reshape(x, (-1, in_shape[1]) + x.shape[1:])
    num_blocks = len(model.outputs)
    y_true_w = pose_w.copy()
    if map_to_pa17j is not None:
        y_true_w = y_true_w[:, map_to_pa17j, :]
    y_pred_w = np.zeros((num_blocks,) + y_true_w.shape)
    if rootz.ndim == 1:
        rootz = np.expand_dims(rootz, axis=-1)
    pred = model.predict(x, batch_size=batch_size, verbose=1)
    y_true_w -= y_true_w[:, 0:1, :]
    _say(verbose, 'Avg. mm. error:')
    scores = []
    for b in range(num_blocks):
        # the reference edits the float32 prediction array in place, so image-plane / depth values are rounded
        # to float32 before the camera un-projection; keep that dtype (on a copy)
        y = np.array(_block_poses(pred[b] if num_blocks > 1 else pred, is_clip, 3))
        y[:, :, 0:2] = transform_pose_sequence(afmat.copy(), y[:, :, 0:2], inverse=True)
        y[:, :, 2] = (resol_z * (y[:, :, 2] - 0.5)) + rootz
        uvd = y[:, map_to_pa17j, 0:3] if map_to_pa17j is not None else y
        for j in range(len(uvd)):
            y_pred_w[b, j] = camera_deserialize(scam[j]).inverse_project(uvd[j])
        y_pred_w[b] -= y_pred_w[b, :, 0:1, :]
        scores.append(mean_distance_error(y_true_w, y_pred_w[b]))
        _say(verbose, ' %.1f' % scores[-1])
    if logdir is not None:
        np.save('%s/y_pred_w.npy' % logdir, y_pred_w)
        np.save('%s/y_true_w.npy' % logdir, y_true_w)
    best = int(np.argmin(scores))          # first minimum, like the reference's strict '<' scan
    if verbose:
        act = np.asarray(action)[:, 0]
        for a in sorted(set(int(v) for v in act)):
            sel = act == a
            label = action_labels[a] if action_labels is not None else 'action %d' % a
            print('%s: %.1f' % (label, mean_distance_error(y_true_w[sel], y_pred_w[best][sel])))
        print('Final averaged error (mm): %.3f' % scores[best])
    return scores
