"""SURVEY.md 8f rank 4 / BASELINE.json configs[0] ON THE ENGINE: the reference's caller loops
(exp/common/mpii_tools.py:13-105, exp/mpii/eval_mpii_singleperson.py:41-74, exp/common/h36m_tools.py:12-106) drive a
HIP ReceptionNet and, beside it, a stand-in whose `predict` is the fp64 CPU oracle on the SAME weights:

  * `refine_pred` (3 iterations: predict -> box from the predicted pose -> re-crop -> predict) on 16 synthetic
    scenes -- the sequence of boxes the dataset was handed is identical, poses within 1e-3 crop-px;
  * `eval_singleperson_pckh` through the script's re-wrapping idiom `Model(model.input, [concatenate([pose_b, vis_b])])`
    -- every block's PCKh score equal;
  * `eval_human36m_sc_error` with the synthetic camera of SURVEY.md 8d on a 3-D ReceptionNet -- mm errors equal
    to 1e-3 mm.

tests/test_evaltools.py pins these drivers bit-exactly to the reference's own exp/common/*_tools.py on stub models
(CPU); this file is where they meet the HIP `Model`."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wellcond                                    # noqa: E402

pytestmark = pytest.mark.gpu

RES = 256


class SceneDataset:
    """MPII-shaped validation split over synthetic scenes: `frames` / `afmat` / `bbox` are live views that re-crop when
    `set_custom_bboxes` hands over new boxes (deephar/data/mpii.py:62-122 behaviour).  Boxes are snapped to whole
    pixels, as a dataset that stores integer annotations does."""

    def __init__(self, n, seed, size=384):
        self.scenes = wellcond.lowpass_fields(n, size, seed, sigma=5.0).astype(np.float32)
        rng = np.random.default_rng(seed)
        x1 = rng.integers(10, 60, (n, 2))
        side = rng.integers(240, 300, (n, 1))
        self.box0 = np.concatenate([x1, x1 + side], axis=1).astype(np.float64)
        self.custom = []
        self.history = []

    def set_custom_bboxes(self, mode, boxes):
        self.custom = [] if len(boxes) == 0 else np.rint(np.asarray(boxes, dtype=np.float64))
        self.history.append(np.array(self.boxes()))

    def clear_custom_bboxes(self, mode):
        self.custom = []

    def boxes(self):
        return self.box0 if len(self.custom) == 0 else self.custom

    def crop(self):
        """Bilinear re-sampling of every scene's box to RES x RES (zero outside the scene), float32 in [-1, 1]."""
        from scipy.ndimage import map_coordinates
        out = np.empty((len(self.scenes), RES, RES, 3), np.float32)
        t = (np.arange(RES) + 0.5) / RES
        for i, b in enumerate(self.boxes()):
            xs, ys = b[0] + t * (b[2] - b[0]) - 0.5, b[1] + t * (b[3] - b[1]) - 0.5
            yy, xx = np.meshgrid(ys, xs, indexing='ij')
            for c in range(3):
                out[i, :, :, c] = map_coordinates(self.scenes[i, :, :, c], [yy, xx], order=1, mode='constant')
        return out

    def affine(self):
        b = self.boxes()
        a = np.zeros((len(b), 3, 3))
        a[:, 0, 0], a[:, 1, 1] = 1 / (b[:, 2] - b[:, 0]), 1 / (b[:, 3] - b[:, 1])
        a[:, 0, 2], a[:, 1, 2] = -b[:, 0] * a[:, 0, 0], -b[:, 1] * a[:, 1, 1]
        a[:, 2, 2] = 1
        return a

    class _View:
        def __init__(self, fn):
            self.fn = fn

        def __getitem__(self, k):
            return self.fn()[k]

        def __len__(self):
            return len(self.fn())

    frames = property(lambda self: self._View(self.crop))
    afmat = property(lambda self: self._View(self.affine))
    bbox = property(lambda self: self._View(lambda: np.array(self.boxes())))


class OracleModel:
    """The Keras-Model surface the drivers touch, with `predict` = the fp64 CPU oracle (float32 results, like Keras)."""

    def __init__(self, wd, joints, dim, concat_blocks, **kw):
        self.wd, self.joints, self.dim, self.kw, self.concat = wd, joints, dim, kw, concat_blocks
        self.input_shape = (None, RES, RES, 3)
        nb = kw['num_blocks']
        self.outputs = [None] * nb

    def get_input_shape_at(self, i):
        return self.input_shape

    def predict(self, x, batch_size=None, verbose=0):
        from oracle import reception as oref
        x = x[0] if isinstance(x, (list, tuple)) else x
        x = np.asarray(x[:], dtype=np.float32)
        outs = oref.forward(self.wd, x, self.joints, self.dim, dtype=torch.float64, **self.kw)
        if self.concat:       # eval_mpii_singleperson.py:56-61
            outs = [np.concatenate([outs[2 * b], outs[2 * b + 1]], axis=-1) for b in range(len(outs) // 2)]
        outs = [o.astype(np.float32) for o in outs]
        return outs if len(outs) > 1 else outs[0]


def test_mpii_refine_and_pckh_loop_on_the_hip_model(hip_lib, cuda):
    from deephar_amd import Model, concatenate, graph, weights
    from deephar_amd.evaltools import mpii_tools
    from deephar_amd.models import reception
    blocks, joints, n = 8, 16, 16
    kw = dict(num_context_per_joint=2, num_blocks=blocks, ksize=(5, 5), concat_pose_confidence=False)
    graph.reset_naming()
    m = reception.build((RES, RES, 3), joints, dim=2, **kw)
    weights.init_synthetic(m, seed=0)
    outs = [concatenate([m.outputs[2 * b], m.outputs[2 * b + 1]], name='blk%d' % (b + 1)) for b in range(blocks)]
    hip = Model(m.input, outs, name=m.name)                                   # eval_mpii_singleperson.py:56-61
    ora = OracleModel(weights.as_dict(m), joints, 2, True, **kw)
    assert len(hip.outputs) == len(ora.outputs) == blocks and hip.get_input_shape_at(0)[1:] == (RES, RES, 3)

    res = {}
    for tag, model in (('hip', hip), ('oracle', ora)):
        ds = SceneDataset(n, seed=3)
        poses = mpii_tools.refine_pred(model, ds.frames, ds.afmat, ds.bbox, ds, 2, blocks - 1, num_iter=3, batch_size=8)
        assert len(ds.custom) == 0 and len(ds.history) == 3
        res[tag] = (poses, ds.history)
    for t in range(3):
        assert np.array_equal(res['hip'][1][t], res['oracle'][1][t]), 'box sequence differs at refinement step %d' % t
        side = (res['oracle'][1][t][:, 2] - res['oracle'][1][t][:, 0])[:, None, None]
        d_crop_px = np.abs(res['hip'][0][t] - res['oracle'][0][t]) * RES / side
        print('refine step %d: max |d| = %.2e crop-px, boxes moved %.1f px on average' % (
            t, d_crop_px.max(), np.abs(res['oracle'][1][t] - res['oracle'][1][0]).mean()))
        assert d_crop_px.max() <= 1e-3
    assert not np.array_equal(res['oracle'][1][0], res['oracle'][1][1])       # the loop really re-cropped

    # PCKh of every block, the way the script calls it (mpii_tools.py:55-128): ground truth = the oracle's last-block
    # pose + noise of about one threshold, so scores are neither 0 nor 1
    ds = SceneDataset(n, seed=4)
    x_val, a_val = ds.crop(), ds.affine()
    head = np.random.default_rng(5).uniform(40, 120, n)
    truth = ora.predict(x_val)[-1][..., :2].astype(np.float64)
    truth += np.random.default_rng(6).normal(0, 0.08, truth.shape)
    s_hip = mpii_tools.eval_singleperson_pckh(hip, x_val, truth, a_val.copy(), head, batch_size=8, verbose=0)
    s_ora = mpii_tools.eval_singleperson_pckh(ora, x_val, truth, a_val.copy(), head, batch_size=8, verbose=0)
    print('PCKh per block: hip %s | oracle %s' % (np.round(s_hip, 4), np.round(s_ora, 4)))
    assert len(s_hip) == blocks and s_hip == s_ora
    assert 0.05 < s_ora[-1] < 0.95


def test_h36m_error_loop_on_the_hip_model(hip_lib, cuda):
    from deephar_amd import graph, weights
    from deephar_amd.evaltools import h36m_tools
    from deephar_amd.models import reception
    from deephar_amd.utils import Camera
    blocks, joints, n = 4, 17, 16
    kw = dict(num_blocks=blocks, depth_maps=16, ksize=(5, 5))
    graph.reset_naming()
    hip = reception.build((RES, RES, 3), joints, dim=3, **kw)
    weights.init_synthetic(hip, seed=0)
    ora = OracleModel(weights.as_dict(hip), joints, 3, False, **kw)
    ora.outputs = [None] * blocks
    assert hip.input_shape[1:] == (RES, RES, 3) and len(hip.outputs) == blocks
    ds = SceneDataset(n, seed=8)
    x, afmat = ds.crop(), ds.affine()
    rng = np.random.default_rng(9)
    rootz = rng.uniform(3000, 6000, n)
    cam = Camera(np.eye(3), np.zeros(3), np.array([1145., 1144.]), np.array([512., 515.]), np.zeros(2))
    scam = np.stack([cam.serialize()] * n)
    action = rng.integers(0, 15, (n, 1))
    pose_w = rng.normal(0, 400, (n, joints, 3)) + np.array([0, 0, 4500.])
    e_hip = h36m_tools.eval_human36m_sc_error(hip, x, pose_w, afmat.copy(), rootz, scam, action, batch_size=8,
                                              verbose=False)
    e_ora = h36m_tools.eval_human36m_sc_error(ora, x, pose_w, afmat.copy(), rootz, scam, action, batch_size=8,
                                              verbose=False)
    print('mm error per block: hip %s | oracle %s' % (np.round(e_hip, 4), np.round(e_ora, 4)))
    assert len(e_hip) == blocks
    assert np.abs(np.array(e_hip) - np.array(e_ora)).max() <= 1e-3
