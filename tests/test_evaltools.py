"""Callers of the pose/action path: deephar_amd.evaltools and utils.bbox against
outputs of the REFERENCE's exp/common/*_tools.py and deephar/utils/bbox.py, both driven by the same deterministic
stand-in model / datasets (tests/evalstubs.py; goldens from tests/golden/make_golden.py).  CPU only."""
import json
import os
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import evalstubs as S                                             # noqa: E402
from deephar_amd import utils as U                                # noqa: E402
from deephar_amd.evaltools import generic, h36m_tools, mpii_tools, ntu_tools, penn_tools   # noqa: E402
from deephar_amd.evaltools.bbox import predict_frame_bboxes       # noqa: E402

G = np.load(os.path.join(HERE, 'golden', 'host_golden.npz'))
EXACT = dict(rtol=0, atol=0)


def test_bbox_helpers_match_reference():
    pts, vis = G['bb_pts'], G['bb_vis']
    np.testing.assert_allclose(U.get_valid_bbox_array(pts), G['bb_valid'], **EXACT)
    np.testing.assert_allclose(U.get_valid_bbox_array(pts, relsize=1.2, square=False), G['bb_valid_nosq'], **EXACT)
    np.testing.assert_allclose(U.compute_grid_bboxes((640, 480)), G['bb_grid'], **EXACT)
    np.testing.assert_allclose(U.compute_grid_bboxes((640, 480), grid=(2, 3), square=False), G['bb_grid_nosq'], **EXACT)
    op, ws = U.get_objpos_winsize(pts[0])
    np.testing.assert_allclose(np.concatenate([op, ws]), G['bb_objpos'], **EXACT)
    pts2 = pts.copy()
    pts2[2] = -1
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        np.testing.assert_allclose(U.get_gt_bbox(pts2, vis, (640, 480), scale=1.2, logkey='k'), G['bb_gt'], **EXACT)
    rootj = np.array([[320., 240., 4000.], [330., 250., 4100.]])
    o, w, z = U.get_crop_params(rootj, (1000, 1002), np.array([[1.1, 1.2]]), 1.3)
    np.testing.assert_allclose(np.concatenate([o, w, z]), G['bb_crop'], rtol=1e-15)
    np.testing.assert_allclose(U.PoseBBox(pts)[1:4], G['bb_posebbox'], **EXACT)
    assert U.PoseBBox(pts).shape == (5, 4) and U.PoseBBox(pts.reshape(1, 5, 16, 2))[0].shape == (5, 4)
    b = np.array([10., 20., 110., 60.])
    c, wh = U.bbox_to_objposwin(b)
    np.testing.assert_array_equal(U.objposwin_to_bbox(c, wh), b)
    with pytest.raises(ValueError):
        U.get_valid_bbox(np.full((4, 2), -1e9))


def test_bbox_refinement_loop():
    ds = S.FakeBoxDataset(6, seed=1)
    model = S.StubModel((8, 8, 3), [('pose', 16, 2)] * 3, ds=ds)
    outs = mpii_tools.refine_pred(model, ds.frames(), ds.afmat(), ds.bbox(), ds, 2, 1, num_iter=3)
    np.testing.assert_allclose(np.stack(outs), G['mpii_refine'], rtol=1e-13, atol=1e-10)
    assert [len(ds.log), model.calls] == list(G['mpii_refine_log'])
    assert ds.log[0] == ('set', 2, 0) and ds.log[-1] == ('clear', 2) and len(ds.custom) == 0
    assert not np.allclose(outs[0], outs[1])          # the second pass really saw re-cropped inputs


def test_pckh_driver_frames_and_clips():
    fval, pval, A, head = G['mpii_fval'], G['mpii_pval'], G['mpii_A'], G['mpii_head']
    s = mpii_tools.eval_singleperson_pckh(S.StubModel((8, 8, 3), [('pose', 16, 3)] * 4), fval, pval, A, head,
                                          refp=2.0, verbose=0)
    np.testing.assert_allclose(s, G['mpii_pckh'], **EXACT)
    s = mpii_tools.eval_singleperson_pckh(S.StubModel((4, 8, 8, 3), [('pose', 16, 3), ('action', 5)] * 2), fval, pval,
                                          A, head, refp=3.0, pred_per_block=2, verbose=0)      # 10 frames -> 2 clips
    np.testing.assert_allclose(s, G['mpii_pckh_clip'], **EXACT)
    assert 0 < min(G['mpii_pckh']) and max(G['mpii_pckh']) < 1          # the golden is not degenerate
    out = mpii_tools.absulute_pred(S.StubModel((8, 8, 3), [('pose', 16, 2)] * 2), fval, A, 1)
    np.testing.assert_allclose(out, G['mpii_abs'], rtol=1e-13, atol=1e-10)


def test_h36m_error_driver(capsys):
    args = (G['h36_x'], G['h36_pw'], G['h36_A'].copy(), G['h36_rootz'], G['h36_scam'], G['h36_action'])
    e = h36m_tools.eval_human36m_sc_error(S.StubModel((8, 8, 3), [('pose', 17, 4)] * 3), *args, verbose=True,
                                          action_labels=['act%d' % i for i in range(16)])
    np.testing.assert_allclose(e, G['h36_err'], rtol=1e-12)
    assert 'act0' in capsys.readouterr().out
    args = (G['h36_x'], G['h36_pw'], G['h36_A'].copy(), G['h36_rootz'], G['h36_scam'], G['h36_action'])
    e = h36m_tools.eval_human36m_sc_error(S.StubModel((2, 8, 8, 3), [('pose', 17, 4)] * 2), *args, verbose=False)
    np.testing.assert_allclose(e, G['h36_err_clip'], rtol=1e-12)


def _action_model():
    return S.StubModel((4, 8, 8, 3), [('action', 5)] * 3)


def test_single_clip_accuracy():
    penn = S.FakeClipDataset(6, 4, 5, 'pennaction', seed=3)
    x_te = np.stack([penn.video[i, :4] for i in range(6)])
    s = penn_tools.eval_singleclip_gt_bbox(_action_model(), x_te, np.eye(5)[penn.labels], verbose=0)
    np.testing.assert_allclose(s, G['act_single'], **EXACT)
    s = penn_tools.eval_singleclip_gt_bbox_generator(_action_model(), S.FakeSequence(penn), verbose=0)
    np.testing.assert_allclose(s, G['act_gen_penn'], **EXACT)
    ntu = S.FakeClipDataset(6, 4, 5, 'ntuaction', seed=4)
    s = ntu_tools.eval_singleclip_gt_bbox_generator(_action_model(), S.FakeSequence(ntu), verbose=0)
    np.testing.assert_allclose(s, G['act_gen_ntu'], **EXACT)


def test_multi_clip_product_voting(tmp_path):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        penn = S.FakeClipDataset(6, 4, 5, 'pennaction', seed=3)
        penn.dataconf.fixed_hflip = 7
        s = penn_tools.eval_multiclip_dataset(_action_model(), penn, 2, verbose=0, logdir=str(tmp_path))
        np.testing.assert_allclose(s, G['act_multi_penn'], **EXACT)
        assert penn.dataconf.fixed_hflip == 7                                  # restored
        assert os.path.exists(tmp_path / 'allpred.npy') and json.load(open(tmp_path / 'missing-clips.json')) is not None
        ntu = S.FakeClipDataset(6, 4, 5, 'ntuaction', seed=4)
        s = ntu_tools.eval_multiclip_dataset(_action_model(), ntu, 2, verbose=0)
        np.testing.assert_allclose(s, G['act_multi_ntu'], **EXACT)
        boxes = {'%04d.%d.%03d.%d' % (i, 2, f, h): [10, 20, 110 + 10 * i + f, 220] for i in range(6)
                 for f in range(3) for h in range(2) if not (i == 1 and f == 0)}
        bf = tmp_path / 'boxes.json'
        json.dump(boxes, open(bf, 'w'))
        ntu = S.FakeClipDataset(6, 4, 5, 'ntuaction', seed=4)
        s = ntu_tools.eval_multiclip_dataset(_action_model(), ntu, 2, bboxes_file=str(bf), verbose=0)
        np.testing.assert_allclose(s, G['act_multi_ntu_boxes'], **EXACT)
        assert [len(ntu.bbox_seen), int(ntu.use_gt_bbox)] == list(G['act_multi_ntu_boxes_seen'])


def test_bbox_from_predicted_poses():
    fval, A = G['mpii_fval'], G['mpii_A']
    model = S.StubModel((8, 8, 3), [('pose', 16, 3)])
    poses = model.predict(fval[:3])
    np.testing.assert_allclose(generic.get_bbox_from_poses(poses, A[0], scale=1.5), G['gen_bbox'], rtol=1e-14)
    np.testing.assert_allclose(generic.get_bbox_from_poses(poses[None], A[1], scale=1.2), G['gen_bbox_clip'], rtol=1e-14)
    with pytest.raises(ValueError):
        generic.get_bbox_from_poses(poses[0], A[0])

    class DS:
        def get_length(self, mode):
            return 3

        def get_data(self, i, mode):
            return {'frame': fval[i], 'afmat': A[i], 'seq_idx': 4, 'frame_list': [10 + i]}
    out = predict_frame_bboxes(model, DS(), 0)
    assert sorted(out) == ['4.10', '4.11', '4.12'] and all(len(v) == 4 and isinstance(v[0], int) for v in out.values())


@pytest.mark.skipif(not os.path.isdir('/root/reference/exp/common'), reason='needs the reference checkout')
def test_reference_harness_runs_unmodified_on_the_compat_modules():
    """deephar_amd.compat.install(): the reference's OWN exp/common/mpii_tools.py and h36m_tools.py, loaded as they
    are, resolve `deephar.*` / `keras.*` to this package and reproduce their golden numbers -- i.e. measures,
    transform, camera, bbox and the printing helpers are drop-ins for what the harness star-imports."""
    import contextlib
    import importlib.util
    import io
    import deephar_amd.compat as compat
    assert 'keras' not in sys.modules and 'deephar' not in sys.modules
    names = compat.install()
    try:
        assert 'deephar.models.reception' in names and 'keras.models' in names
        from deephar.models import reception, split_model          # noqa: F401  (what the eval scripts import)
        from deephar.config import ModelConfig, mpii_sp_dataconf    # noqa: F401
        from keras.models import Model
        from keras.layers import concatenate
        import deephar_amd
        assert Model is deephar_amd.Model and concatenate is deephar_amd.concatenate
        assert mpii_sp_dataconf.input_shape == (256, 256, 3)
        import deephar.data
        with pytest.raises(NotImplementedError):
            deephar.data.MpiiSinglePerson('datasets/MPII')

        def load(name):
            spec = importlib.util.spec_from_file_location('refexp_' + name, '/root/reference/exp/common/%s.py' % name)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
        mpii, h36 = load('mpii_tools'), load('h36m_tools')
        sys.modules['deephar.data.human36m'] = type(sys)('deephar.data.human36m')
        sys.modules['deephar.data.human36m'].ACTION_LABELS = ['act%d' % i for i in range(16)]
        fval, pval, A, head = G['mpii_fval'], G['mpii_pval'], G['mpii_A'], G['mpii_head']
        with contextlib.redirect_stdout(io.StringIO()):
            s = mpii.eval_singleperson_pckh(S.StubModel((8, 8, 3), [('pose', 16, 3)] * 4), fval, pval, A, head,
                                            refp=2.0, verbose=0)
            e = h36.eval_human36m_sc_error(S.StubModel((8, 8, 3), [('pose', 17, 4)] * 3), G['h36_x'], G['h36_pw'],
                                           G['h36_A'].copy(), G['h36_rootz'], G['h36_scam'], G['h36_action'])
        np.testing.assert_allclose(s, G['mpii_pckh'], **EXACT)
        np.testing.assert_allclose(e, G['h36_err'], rtol=1e-12)
        ds = S.FakeBoxDataset(6, seed=1)
        model = S.StubModel((8, 8, 3), [('pose', 16, 2)] * 3, ds=ds)
        with contextlib.redirect_stdout(io.StringIO()):
            outs = mpii.refine_pred(model, ds.frames(), ds.afmat(), ds.bbox(), ds, 2, 1, num_iter=3)
        np.testing.assert_allclose(np.stack(outs), G['mpii_refine'], rtol=1e-13, atol=1e-10)
    finally:
        compat.uninstall()
        sys.modules.pop('deephar.data.human36m', None)
    assert 'keras' not in sys.modules and 'deephar' not in sys.modules
