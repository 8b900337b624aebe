"""SPNet at the FLAT 1e-3 px bar (reference deephar/models/spnet.py:151-248).

Round 4: three input seeds x two clips per configuration, predicted as ONE batch, and a margin
sweep over the head sensitivity (test_spnet_margin_sweep).

Vectors: tests/wellcond.py -- video clips of low-pass noise and heat-map heads fitted on the oracle so that every
prediction block has one peak per joint (read-out sensitivity S = sum p |g - x| <= 0.05, asserted from the fp64
oracle's logits; maps not one-hot).  Checks: plain `paritylog.check(..., PX_TOL)` on x/y (and z) of every prediction
block -- no conditioned tolerance, no relative clause -- in the default fp32-MFMA mode and in the opt-in bf16x3
mode, `hip - o32` recorded beside `hip - o64`; confidences to 1e-5 absolute, action scores to 1e-5, identical labels.

Configurations at the real 256x256 resolution:
  ntu3d_T8        exp/ntu/eval_ntu_multitask.py:35-38 (pa17j3d, 60 actions, 2 pyramids, T = 8)
  penn2d_T16      2-D, T = 16 (time_stride 2 branch, spnet.py:100), action on pyramid 2
  penn_shipped    exp/pennaction/eval_penn_multitask.py:36-40: 6 pyramids, actions on 5 and 6, pose_replica=True
  cfg5_ntu_T32    BASELINE.json configs[4]: the NTU model at T = 32
  speed2d         exp/pennaction/eval_speed2d.py:31-43: 6 pyramids, actions on ALL six, pose_replica=True, T = 8 [r06]
and the three reference-code goldens 'spnet3d_s', 'spnet2d_s', 'spnet2dr_s' (the reference's own spnet.py run on the
same kind of vectors, tests/golden/make_reference_golden.py --smooth).
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import paritylog                                   # noqa: E402
import wellcond                                    # noqa: E402
from paritylog import PX_TOL, check                # noqa: E402

pytestmark = pytest.mark.gpu

CONFIGS = {
    'ntu3d_T8': (8, 'pa17j3d', 60, 2, [1, 2], 192, False),
    'penn2d_T16': (16, 'pa16j2d', 15, 2, [2], 160, False),
    'penn_shipped': (8, 'pa16j2d', 15, 6, [5, 6], 160, True),
    'cfg5_ntu_T32': (32, 'pa17j3d', 60, 2, [1, 2], 192, False),
    # [r06] the model of the reference's speed protocol (exp/pennaction/eval_speed2d.py:31-43): actions on all six pyramids
    'speed2d': (8, 'pa16j2d', 15, 6, [1, 2, 3, 4, 5, 6], 160, True),
}
SEEDS = (0, 1, 2)        # input seeds {0, 1, 2}; the synthetic weights keep their fixed seed
NCLIPS = 2               # clips per configuration, predicted in ONE batch (configs[4] is 8 clips per GPU: the other six
                         # are bridged by the bit-exact batch-size invariance of tests/test_gpu_full_configs.py)
_CACHE = {}


def _prepare(name, seed, nclips=NCLIPS, s_target=None, conditioned=True):
    """Model with fitted heads + clips + fp32 / fp64 oracle outputs; shared by the two GEMM modes of a configuration."""
    key = (name, seed, nclips, s_target)
    if key in _CACHE:
        return _CACHE[key]
    from test_gpu_models import _spnet
    from deephar_amd import weights
    from oracle import spnet as osp
    T, layout, nact, pyr, apyr, feats, replica = CONFIGS[name]
    vseed = 40 + 10 * seed + sorted(CONFIGS).index(name)      # (seed 0 = the round-3 vectors)
    x = wellcond.video_cuts(nclips, T, 256, vseed)   # one scene per seed, cut into the clips of the batch
    m, cfg, _, ocfg = _spnet(T, layout, nact, pyr, apyr, feats, replica=replica)
    wellcond.fit_spnet_heads(m, ocfg, x, wellcond.scene_positions(nclips, T, ocfg['num_joints'], vseed),
                             per_joint=True, **({} if s_target is None else dict(s_target=s_target)))
    wd = weights.as_dict(m)
    t64 = {}
    o32 = osp.forward(wd, x, ocfg, dtype=torch.float32)
    o64 = osp.forward(wd, x, ocfg, dtype=torch.float64, taps=t64)
    stats = wellcond.assert_well_conditioned(t64, name) if conditioned else wellcond.conditioning_stats(t64)
    _CACHE.clear()                                  # one configuration at a time (clips are small; models are not)
    _CACHE[key] = (m, x, ocfg, o32, o64, stats, (pyr, apyr))
    return _CACHE[key]


def _flat_checks(tag, hip, o32, o64, dim, npose, case):
    flat = lambda a: a.reshape((-1,) + a.shape[-2:])
    for k in range(npose):
        h, a, b = flat(hip[k]), flat(o32[k]), flat(o64[k])
        check('%s.out%d.xy' % (tag, k), h[..., :2], a[..., :2], b[..., :2], PX_TOL, case=case)
        if dim == 3:
            check('%s.out%d.z' % (tag, k), h[..., 2], a[..., 2], b[..., 2], PX_TOL, case=case)
        check('%s.out%d.conf' % (tag, k), h[..., dim], a[..., dim], b[..., dim], 1e-5, case=case)
    for k in range(npose, len(hip)):
        check('%s.act%d' % (tag, k - npose), hip[k], o32[k], o64[k], 1e-5, case=case)
        assert np.array_equal(hip[k].argmax(-1), o64[k].argmax(-1)), '%s: action label differs on head %d' % (tag, k - npose)


@pytest.mark.parametrize('name,seed,mode', [(n, s, md) for n in CONFIGS for s in SEEDS for md in ('f32', 'bf16x3')])
def test_spnet_flat_1e3_px(name, seed, mode, hip_lib, cuda):
    """Three input seeds x two clips per configuration (one batch), both GEMM modes: flat 1e-3 px on every prediction."""
    from deephar_amd.models import spnet
    m, x, ocfg, o32, o64, stats, (pyr, apyr) = _prepare(name, seed)
    m.gemm_precision = mode                          # (an engine option: changing it re-plans the model)
    hip = m.predict(x, batch_size=len(x))
    nsplit = sum(1 for s in m.plan.steps if s.kind == 'conv' and s.attrs.get('w_split') == 1)
    assert (nsplit > 20) == (mode == 'bf16x3'), (mode, nsplit)          # the mode under test is the mode that ran
    npose = spnet.get_num_predictions(pyr, 4)
    assert len(hip) == npose + spnet.get_num_predictions(len(apyr), 4)
    assert [h.shape for h in hip] == [o.shape for o in o64]
    assert hip[0].shape[0] == NCLIPS
    print('%s seed %d: S_max per block %s' % (name, seed, ' '.join('%.3f' % s['S_max'] for s in stats.values())))
    _flat_checks('%s.s%d.%s' % (name, seed, mode), hip, o32, o64, ocfg['dim'], npose,
                 case='spnet_flat/%s/seed%d/%s' % (name, seed, mode))


SWEEP_S = (0.02, 0.04, 0.08, 0.15)


@pytest.mark.parametrize('name', ['ntu3d_T8', 'penn2d_T16'])
def test_spnet_margin_sweep(name, hip_lib, cuda):
    """How far the engine is from the 1e-3 px bar as the read-out sensitivity S = sum p |g - x| of the fitted heads grows
    (VERDICT r03 item 1b): the same fit at S_TARGET in {0.02, 0.04, 0.08, 0.15}, one clip, fp32 mode.  RECORDED in
    gpurun_out/parity_r06.json (`margin_sweep`: S asked / measured, worst |hip - o64|, |o32 - o64|, |hip - o32| over every
    prediction block); ASSERTED only where the fit reached its target and S <= wellcond.S_MAX = 0.05, the conditioning the
    flat test itself requires (S = 0.02 is not reachable on these maps: a peak between two cells keeps S at half a cell)."""
    from deephar_amd.models import spnet
    for s_target in SWEEP_S:
        m, x, ocfg, o32, o64, stats, (pyr, apyr) = _prepare(name, 0, nclips=1, s_target=s_target, conditioned=False)
        m.gemm_precision = 'f32'
        hip = m.predict(x, batch_size=1)
        npose, dim = spnet.get_num_predictions(pyr, 4), ocfg['dim']
        s_max = max(s['S_max'] for s in stats.values())
        flat = lambda a: a.reshape((-1,) + a.shape[-2:])[..., :dim]
        cat = lambda outs: np.concatenate([flat(o).reshape(-1) for o in outs[:npose]])
        # the fit reaches its target unless a peak sits between two cells: such a map keeps S ~ half a cell however sharp it
        # is made, the bisection runs into its cap (x 50) and the logits into the hundreds -- another regime (fp32 rounding
        # of |logit| ~ 500 is 3e-5 per operation; the CPU fp32 oracle is then several 1e-3 px from fp64 itself)
        reached = s_max <= 1.05 * s_target
        r = paritylog.record('%s.S%.2f.pose' % (name, s_target), cat(hip), cat(o32), cat(o64),
                             case='spnet_margin_sweep/%s' % name, S_target=s_target, S_max_measured=s_max,
                             fit_reached_target=bool(reached),
                             logit_absmax=max(s['logit_absmax'] for s in stats.values()), tol=1e-3,
                             asserted_here=bool(reached and s_max <= wellcond.S_MAX))
        print('%s S_target %.2f (measured %.3f, |logit| <= %.0f): hip-o64 %.2e  o32-o64 %.2e  hip-o32 %.2e px' % (
            name, s_target, s_max, r['logit_absmax'], r['hip_vs_o64'], r['o32_vs_o64'], r['hip_vs_o32']))
        if reached and s_max <= wellcond.S_MAX:
            assert r['hip_vs_o64'] <= 1e-3, 'S = %.3f: %.3e px' % (s_max, r['hip_vs_o64'])
        same = all(np.array_equal(hip[k].argmax(-1), o64[k].argmax(-1)) for k in range(npose, len(hip)))
        r['action_labels_identical'] = bool(same)
        assert same or not reached, 'action label differs at S = %.2f' % s_target


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
@pytest.mark.parametrize('tag', ['spnet3d_s', 'spnet2d_s', 'spnet2dr_s', 'spnet3d_32_s'])
def test_hip_matches_smooth_reference_code_goldens(tag, mode, hip_lib, cuda):
    """HIP engine vs golden vectors computed by the reference's OWN spnet.py / common.py / layers.py (on mini-Keras) for
    the well-conditioned vectors: flat 1e-3 px on every pose output, identical arg-max action labels."""
    from refgolden import build_case, golden
    m, x, run = build_case(tag)
    t64 = {}
    run(torch.float64, taps=t64)
    wellcond.assert_well_conditioned(t64, tag)
    g32, g64 = golden(tag)
    m.gemm_precision = mode
    hip = m.predict(x.astype(np.float32), batch_size=len(x))
    assert [h.shape for h in hip] == [g.shape for g in g64]
    npose = sum(1 for g in g64 if g.ndim == 4)
    _flat_checks('%s.%s' % (tag, mode), hip, g32, g64, g64[0].shape[-1] - 1, npose,
                 case='spnet_flat_golden/%s/%s' % (tag, mode))
