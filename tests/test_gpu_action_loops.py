"""The ACTION-side caller loops of the reference on the HIP engine:

  * `eval_multiclip_dataset` (exp/common/penn_tools.py:85-163, exp/common/ntu_tools.py:53-151): every test sequence is cut
    into hop windows (deephar/data/pennaction.py:207-221), each clip is predicted as it is and horizontally flipped, the
    per-clip soft-max scores are MULTIPLIED and the arg-max of the product is the label -- driven once by the HIP model
    and once by a stand-in whose `predict` is the fp64 CPU oracle on the same weights: identical labels per output
    block, identical `missing-clips.json`, score products within 1e-4 relative;
      - the PennAction merge model as exp/pennaction/eval_penn_ar_pe_merge.py:42-57 builds it (16-frame clips, 4 blocks,
        15 actions, output_poses=False: p1..p4, v1..v4, m),
      - SPNet's action half from `split_model` (exp/ntu/eval_ntu_multitask.py:35-60), with predicted boxes from a JSON
        file on the NTU path;
  * the per-frame box pass of exp/pennaction/predict_bboxes.py:50-68 / exp/ntu/predict_bboxes.py:35-60: a pose-only SPNet
    re-wrapped as `Model(full.input, full.outputs[-1])`, one forward per frame, box from the predicted pose.

tests/test_evaltools.py pins these drivers bit-exactly to the reference's own exp/common/*_tools.py on stub models (CPU);
this file is where they meet the HIP `Model`."""
import json
import os
import sys
import warnings

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wellcond                                    # noqa: E402

pytestmark = pytest.mark.gpu
RES = 256


class _Conf:
    fixed_hflip = 0


class VideoDataset:
    """Penn / NTU-shaped test split over synthetic videos: `get_clip_index` is the reference's hop-window rule
    (pennaction.py:207-221), `get_data` honours dataconf.fixed_hflip (mirrors the frames) and an optional box (NTU:
    the clip is re-cropped to the box, nearest neighbour)."""

    def __init__(self, n, length, clip_size, num_actions, label_key, seed):
        self.n, self.clip_size, self.C, self.key = n, clip_size, num_actions, label_key
        self.video = [wellcond.video_clips(1, length, RES, seed * 10 + i, phase=2.0)[0] for i in range(n)]
        self.labels = np.random.default_rng(seed).integers(0, num_actions, n)
        self.dataconf = _Conf()
        self.use_gt_bbox = True
        self.bbox_seen = []

    def get_length(self, mode):
        return self.n

    def get_shape(self, key):
        assert key == self.key
        return (self.C,)

    def get_clip_index(self, i, mode, subsamples=(2,)):
        out = []
        for sub in subsamples:
            start = 0
            while start + self.clip_size * sub <= len(self.video[i]):
                out.append(range(start, start + self.clip_size * sub, sub))
                start += int(self.clip_size / 2) + (sub - 1)
        return out

    def get_data(self, i, mode, frame_list=None, bbox=None):
        fr = self.video[i][list(frame_list)]
        if bbox is not None:
            self.bbox_seen.append(tuple(int(v) for v in bbox))
            x0, y0, x1, y1 = [int(v) for v in bbox]
            xs = np.clip(np.rint(np.linspace(x0, x1 - 1, RES)).astype(int), 0, RES - 1)
            ys = np.clip(np.rint(np.linspace(y0, y1 - 1, RES)).astype(int), 0, RES - 1)
            fr = fr[:, ys][:, :, xs]
        if self.dataconf.fixed_hflip:
            fr = fr[:, :, ::-1]
        onehot = np.zeros(self.C)
        onehot[self.labels[i]] = 1
        return {'frame': np.ascontiguousarray(fr), self.key: onehot}


class OracleActionModel:
    """The Keras-Model surface the drivers touch; `predict` = the fp64 CPU oracle (float32 results, like Keras)."""

    def __init__(self, nout, run):
        self.outputs, self.run = [None] * nout, run

    def predict(self, x, batch_size=None, verbose=0):
        outs = [o.astype(np.float32) for o in self.run(np.asarray(x, dtype=np.float32))]
        return outs if len(outs) > 1 else outs[0]


def _vote(tool, model, make_ds, tmp, **kw):
    ds = make_ds()
    os.makedirs(tmp, exist_ok=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        scores = tool(model, ds, 2, logdir=tmp, verbose=0, **kw)
    a_true = np.load(os.path.join(tmp, 'a_true.npy'))
    missing = json.load(open(os.path.join(tmp, 'missing-clips.json')))
    return np.asarray(scores), a_true, missing, ds


def test_penn_merge_multiclip_voting_on_the_hip_model(hip_lib, cuda, tmp_path):
    from deephar_amd import graph, weights
    from deephar_amd.evaltools import penn_tools
    from deephar_amd.models import action, reception
    from oracle import action as oact
    T, blocks, joints, nact = 16, 4, 16, 15
    graph.reset_naming()
    pe = reception.build((RES, RES, 3), joints, dim=2, num_blocks=blocks, num_context_per_joint=2, ksize=(5, 5),
                         concat_pose_confidence=False)
    hip = action.build_merge_model(pe, nact, (RES, RES, 3), T, joints, blocks, pose_dim=2, pose_net_version='v1',
                                   full_trainable=False)                        # eval_penn_ar_pe_merge.py:50-57
    weights.init_synthetic(hip, seed=0)
    wd = weights.as_dict(hip)
    assert len(hip.outputs) == 9
    ora = OracleActionModel(9, lambda x: oact.forward_merge(wd, x, nact, joints, blocks, dtype=torch.float64, pose_dim=2,
                                                             pose_net_version='v1', output_poses=False))
    make = lambda: VideoDataset(2, 52, T, nact, 'pennaction', seed=5)          # 52 frames, hop 9 at subsampling 2: 3 clips
    assert len(make().get_clip_index(0, 0, subsamples=[2])) == 3
    s_h, t_h, miss_h, ds_h = _vote(penn_tools.eval_multiclip_dataset, hip, make, str(tmp_path / 'hip'))
    s_o, t_o, miss_o, ds_o = _vote(penn_tools.eval_multiclip_dataset, ora, make, str(tmp_path / 'ora'))
    ap_h, ap_o = np.load(tmp_path / 'hip' / 'allpred.npy'), np.load(tmp_path / 'ora' / 'allpred.npy')
    assert ap_h.shape == ap_o.shape == (9, 6, nact)                            # (blocks, 3 clips x 2 flips, classes)
    print('multi-clip scores per block: hip %s | oracle %s; per-clip score max |d| = %.2e' % (
        s_h, s_o, np.abs(ap_h - ap_o).max()))
    assert np.array_equal(t_h, t_o) and np.array_equal(s_h, s_o)
    assert miss_h == miss_o, 'missing-clips.json differs'
    assert np.array_equal(ap_h.argmax(-1), ap_o.argmax(-1)), 'a per-clip arg-max label differs'
    np.testing.assert_allclose(ap_h, ap_o, rtol=1e-4, atol=1e-6)
    assert ds_h.dataconf.fixed_hflip == 0


def test_spnet_action_multiclip_voting_with_boxes_on_the_hip_model(hip_lib, cuda, tmp_path):
    from deephar_amd.evaltools import ntu_tools
    from deephar_amd.models import spnet, split_model
    from oracle import spnet as osp
    from test_gpu_models import _spnet
    T, nact, pyr, apyr = 8, 60, 2, [1, 2]
    full, cfg, wd, ocfg = _spnet(T, 'pa17j3d', nact, pyr, apyr, 192)
    hip = split_model(full, cfg)[1]                                             # eval_ntu_multitask.py:57-60
    npose = spnet.get_num_predictions(pyr, 4)
    nout = spnet.get_num_predictions(len(apyr), 4)
    assert len(hip.outputs) == nout
    ora = OracleActionModel(nout, lambda x: osp.forward(wd, x, ocfg, dtype=torch.float64)[npose:])
    make = lambda: VideoDataset(2, 28, T, nact, 'ntuaction', seed=6)           # 28 frames, hop 5: 3 clips
    nclips = len(make().get_clip_index(0, 0, subsamples=[2]))
    assert nclips == 3
    boxes = {'%04d.%d.%03d.%d' % (i, 2, f, h): [8 + 4 * i, 12 + 2 * f, 240 - 6 * f, 250 - 3 * i]
             for i in range(2) for f in range(nclips) for h in range(2) if not (i == 1 and f == 2 and h == 1)}
    bf = tmp_path / 'boxes.json'
    json.dump(boxes, open(bf, 'w'))
    s_h, t_h, miss_h, ds_h = _vote(ntu_tools.eval_multiclip_dataset, hip, make, str(tmp_path / 'hip'), bboxes_file=str(bf))
    s_o, t_o, miss_o, ds_o = _vote(ntu_tools.eval_multiclip_dataset, ora, make, str(tmp_path / 'ora'), bboxes_file=str(bf))
    ap_h, ap_o = np.load(tmp_path / 'hip' / 'a_pred.npy'), np.load(tmp_path / 'ora' / 'a_pred.npy')
    assert ap_h.shape == ap_o.shape == (nout, 2, nact)                          # running product over clips and flips
    print('multi-clip scores per block: hip %s | oracle %s; product max rel |d| = %.2e' % (
        s_h, s_o, (np.abs(ap_h - ap_o) / np.maximum(ap_o, 1e-30)).max()))
    assert np.array_equal(t_h, t_o) and np.array_equal(s_h, s_o)
    assert miss_h == miss_o, 'missing-clips.json differs'
    assert np.array_equal(ap_h.argmax(-1), ap_o.argmax(-1)), 'a voted label differs'
    np.testing.assert_allclose(ap_h, ap_o, rtol=2e-4)
    assert ds_h.bbox_seen == ds_o.bbox_seen and len(ds_h.bbox_seen) == 2 * nclips * 2 - 1   # one key is missing
    assert ds_h.use_gt_bbox is True                                             # restored


class FrameDataset:
    """'frames' topology for predict_bboxes.py: one frame per sample with its crop affine, sequence index and frame number."""

    def __init__(self, n, seed):
        from test_gpu_caller_loop import SceneDataset
        self.sd = SceneDataset(n, seed)
        self.x, self.a = self.sd.crop(), self.sd.affine()

    def get_length(self, mode):
        return len(self.x)

    def get_data(self, i, mode):
        return {'frame': self.x[i], 'afmat': self.a[i], 'seq_idx': 3 + i // 4, 'frame_list': [10 * (i % 4)]}


def test_frame_bbox_pass_with_rewrapped_spnet_on_the_hip_model(hip_lib, cuda):
    from deephar_amd import Model, graph, utils, weights
    from deephar_amd.config import ModelConfig
    from deephar_amd.evaltools.bbox import get_bbox_from_poses, predict_frame_bboxes
    from deephar_amd.models import spnet
    from oracle import spnet as osp
    graph.reset_naming()
    cfg = ModelConfig((RES, RES, 3), utils.pa16j2d, num_pyramids=2, action_pyramids=[], num_levels=4)   # frames topology,
    full = spnet.build(cfg)                                                     # pose only (predict_bboxes.py:35-41)
    weights.init_synthetic(full, seed=0)
    hip = Model(full.input, full.outputs[-1])                                   # "squeeze the model for only one output"
    wd = weights.as_dict(full)
    ocfg = dict(num_joints=16, dim=2, num_actions=[], num_pyramids=2, action_pyramids=[], num_levels=4, kernel_size=(5, 5),
                growth=96, image_div=8, num_pose_features=0, num_visual_features=0, sam_alpha=1)

    class Ora:
        outputs = [None]

        def predict(self, x, batch_size=None, verbose=0):
            return osp.forward(wd, np.asarray(x, np.float32), ocfg, dtype=torch.float64)[-1].astype(np.float32)
    ds = FrameDataset(8, seed=12)
    b_h = predict_frame_bboxes(hip, ds, 0)
    b_o = predict_frame_bboxes(Ora(), ds, 0)
    assert sorted(b_h) == sorted(b_o) == sorted('%d.%d' % (3 + i // 4, 10 * (i % 4)) for i in range(8))
    # the boxes are floor()ed floats: a coordinate within the pose tolerance of an integer may land on either side
    fh = np.array([get_bbox_from_poses(hip.predict(ds.x[i][None]), ds.a[i]) for i in range(8)])
    fo = np.array([get_bbox_from_poses(Ora().predict(ds.x[i][None]), ds.a[i]) for i in range(8)])
    side = (ds.sd.box0[:, 2] - ds.sd.box0[:, 0])[:, None]
    print('boxes: max |d| = %.2e image px (= %.2e crop px); identical integer boxes: %d of 8' % (
        np.abs(fh - fo).max(), (np.abs(fh - fo) * RES / side).max(), sum(b_h[k] == b_o[k] for k in b_h)))
    assert (np.abs(fh - fo) * RES / side).max() <= 5e-3                         # 1.5 x the pose extent of poses within 3e-3 px
    for k in b_h:
        assert np.abs(np.array(b_h[k]) - np.array(b_o[k])).max() <= 1
    assert sum(b_h[k] == b_o[k] for k in b_h) >= 6
    assert (fo[:, 2] - fo[:, 0]).min() > 20                                     # real boxes, not collapsed ones
