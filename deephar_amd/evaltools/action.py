"""Action-accuracy drivers (exp/common/penn_tools.py:14-163, exp/common/ntu_tools.py:13-152): single-clip accuracy
per prediction block, and multi-clip voting -- every test sequence is cut into several clips, each clip is also
evaluated horizontally flipped, and the per-clip class probabilities are MULTIPLIED before the arg-max."""
import json
import os
import time
import warnings

import numpy as np

from ..utils import TEST_MODE


def _say(verbose, *a):
    if verbose:
        print(*a)


def _accuracy(y_true, y_pred):
    hit = np.argmax(y_true, axis=-1) == np.argmax(y_pred, axis=-1)
    return hit.astype(np.float64)


def eval_singleclip_gt_bbox(model, x_te, action_te, batch_size=1, verbose=1):
    """Fraction of clips whose arg-max label is right, per output block (penn_tools.py:14-41)."""
    t0 = time.time()
    pred = model.predict(x_te, batch_size=batch_size, verbose=verbose)
    dt = time.time() - t0
    scores = []
    for b in range(len(model.outputs)):
        hit = _accuracy(action_te, pred[b])
        scores.append(sum(hit) / len(hit))
    _say(verbose, 'single-clip, action acc.%:', ' '.join('%.1f' % (100 * s) for s in scores))
    _say(verbose, '%d samples in %.1f sec: %.1f clips per sec' % (len(x_te), dt, len(x_te) / max(dt, 1e-9)))
    return scores


def eval_singleclip_gt_bbox_generator(model, datagen, verbose=1, logdir=None):
    """Same, pulling one `[x], [y]` batch per index from a Sequence-like loader (penn_tools.py:44-84,
    ntu_tools.py:13-50)."""
    nb, n = len(model.outputs), len(datagen)
    y_true = y_pred = None
    t0 = time.time()
    for i in range(n):
        [x], [y] = datagen[i]
        if y_true is None:
            y_true = np.zeros((n,) + y.shape[1:])
            y_pred = np.zeros((n, nb) + y.shape[1:])
        y_true[i, :] = y
        pred = model.predict(x)
        for b in range(nb):
            y_pred[i, b, :] = pred[b]
    dt = time.time() - t0
    scores = []
    for b in range(nb):
        hit = _accuracy(y_true, y_pred[:, b, :])
        scores.append(sum(hit) / len(hit))
        if logdir is not None:
            os.makedirs(os.path.join(logdir, 'single-clip'), exist_ok=True)
            np.save(os.path.join(logdir, 'single-clip', '%02d.npy' % b), hit)
    _say(verbose, 'single-clip, GT bbox, action acc.%:', ' '.join('%.1f' % (100 * s) for s in scores))
    _say(verbose, '%d samples in %.1f sec: %.1f clips per sec' % (n, dt, n / max(dt, 1e-9)))
    return scores


def _multiclip(model, ds, label_key, subsampling, bboxes_data, logdir, verbose, pred_file, pass_bbox=False):
    n, nb = ds.get_length(TEST_MODE), len(model.outputs)
    shape = (n,) + tuple(ds.get_shape(label_key))
    a_true = np.zeros(shape)
    a_pred = np.ones((nb,) + shape)          # running PRODUCT of per-clip probabilities
    missing, right = {}, 0
    keep_hflip = ds.dataconf.fixed_hflip
    allpred = None
    try:
        for i in range(n):
            clips = ds.get_clip_index(i, TEST_MODE, subsamples=[subsampling])
            allpred = np.ones((nb, 2 * len(clips)) + shape[1:])
            for f, frames in enumerate(clips):
                for hflip in (0, 1):
                    try:
                        ds.dataconf.fixed_hflip = hflip
                        kw = {'bbox': None} if pass_bbox else {}
                        if bboxes_data is not None:
                            key = '%04d.%d.%03d.%d' % (i, subsampling, f, hflip)
                            if key in bboxes_data:
                                kw['bbox'] = np.array(bboxes_data[key])
                            else:
                                warnings.warn('Missing bounding box key ' + key)
                        data = ds.get_data(i, TEST_MODE, frame_list=frames, **kw)
                        a_true[i, :] = data[label_key]
                        pred = model.predict(np.expand_dims(data['frame'], axis=0))
                        for b in range(nb):
                            allpred[b, 2 * f + hflip, :] = pred[b][0]
                            a_pred[b, i, :] *= pred[b][0]
                        if np.argmax(a_true[i]) != np.argmax(a_pred[-1, i]):
                            missing['%04d.%03d.%d' % (i, f, hflip)] = [int(np.argmax(a_true[i])),
                                                                       int(np.argmax(a_pred[-1, i]))]
                    except Exception as e:       # the reference keeps going on unreadable clips
                        warnings.warn('eval_multiclip, exception on sample %d frame %d: %s' % (i, f, e))
            right += int(np.argmax(a_true[i]) == np.argmax(a_pred[-1, i]))
            _say(verbose, '%04d/%04d\t%.1f' % (i, n, 100 * right / (i + 1)))
    finally:
        ds.dataconf.fixed_hflip = keep_hflip
    if logdir is not None:
        np.save(os.path.join(logdir, pred_file), allpred if pred_file == 'allpred.npy' else a_pred)
        np.save(os.path.join(logdir, 'a_true.npy'), a_true)
        with open(os.path.join(logdir, 'missing-clips.json'), 'w') as fid:
            json.dump(missing, fid)
    hit = np.argmax(a_true, axis=-1)[None, :] == np.argmax(a_pred, axis=-1)
    scores = 100 * np.sum(hit, axis=-1) / n
    _say(verbose, 'multi-clip:', np.array2string(np.array(scores), precision=2), 'best: %.2f' % max(scores))
    return scores


def penn_eval_multiclip_dataset(model, penn, subsampling, bboxes_file=None, logdir=None, verbose=1):
    """penn_tools.eval_multiclip_dataset (penn_tools.py:87-163); `bboxes_file` is accepted and unused there."""
    return _multiclip(model, penn, 'pennaction', subsampling, None, logdir, verbose, 'allpred.npy')


def ntu_eval_multiclip_dataset(model, ntu, subsampling, bboxes_file=None, logdir=None, verbose=1):
    """ntu_tools.eval_multiclip_dataset (ntu_tools.py:53-152): optional predicted boxes from a JSON file keyed
    '<sample>.<subsampling>.<clip>.<hflip>'; the dataset's use_gt_bbox flag is switched accordingly and restored."""
    boxes = None
    if bboxes_file is not None:
        with open(bboxes_file, 'r') as fid:
            boxes = json.load(fid)
    keep = ntu.use_gt_bbox
    ntu.use_gt_bbox = boxes is None
    try:
        return _multiclip(model, ntu, 'ntuaction', subsampling, boxes, logdir, verbose, 'a_pred.npy', pass_bbox=True)
    finally:
        ntu.use_gt_bbox = keep
