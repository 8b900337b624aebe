"""Well-conditioned SPNet test vectors.  TEST INFRASTRUCTURE.

The 1e-3 px criterion of BASELINE.json north_star can only be resolved by a test whose soft-argmax read-outs are
conditioned like a trained network's: d x / d logit_i = p_i (g_i - x), so an fp32 rounding error e on the logits moves
a coordinate by up to e * S with S = sum_i p_i |g_i - x|.  A trained SPNet produces one peak per joint (S ~ 0.03);
`weights.init_synthetic` on per-pixel uniform noise produces multi-modal maps (S = 0.3 .. 0.5) with |logit| up to 100
on the coarse pyramid levels, where ANY fp32 implementation -- the PyTorch-CPU oracle included -- is 1 .. 3e-3 px from
fp64 (profiles/r02_spnet_noise.json).  These vectors make the synthetic network behave like a trained one:

  * clips are short *videos*: spatially low-pass noise fields (Gaussian, sigma 4 px, wrap-around) that rotate slowly
    in the plane spanned by two base fields, so the frames of a clip are strongly correlated like real frames
    (`video_clips`); every frame still has variance 1/3 like the U(-1, 1) inputs init_synthetic's BatchNorm
    statistics assume;
  * every heat-map head '<block>_heatmaps_conv1' is FITTED, in prediction order, by ridge regression of the tensor
    the head reads (oracle fp32 pass) onto one Gaussian peak per (clip, joint) at a random position, then scaled by
    bisection until max S = S_TARGET on the calibration clips (`fit_spnet_heads`): a closed-form "training" of the
    1x1 heads alone; all other weights stay init_synthetic's.  Coarse levels (<= 8 cells) get their peak on a cell
    centre (a peak between two cells of a 4x4 map has S = 0.17 whatever the network does).

The tests then assert S <= S_MAX on every prediction block from the fp64 oracle's logits, that the maps are not
one-hot (the coordinate still depends on several pixels), and a plain `paritylog.check(..., PX_TOL)` -- no
conditioned tolerance.  The per-pixel-noise cases of tests/test_gpu_models.py stay as a stress test.
"""
import numpy as np

S_TARGET = 0.04      # calibration target for max S over the calibration clips
S_MAX = 0.05         # asserted bound (S = sum p |g - x| <= 0.05 on every prediction block)
PEAK = 12.0          # logit height of the fitted peaks before the bisection scaling
RIDGE = 1e-3         # ridge, relative to the mean feature energy


def lowpass_fields(n, res, seed, sigma=4.0):
    """n fields [res, res, 3] of Gaussian low-pass noise (wrap-around), std 1/sqrt(3) like U(-1, 1), clipped to [-1, 1]."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    x = gaussian_filter(rng.standard_normal((n, res, res, 3)), sigma=(0, sigma, sigma, 0), mode='wrap')
    x /= x.std()
    return np.clip(x / np.sqrt(3.0), -1.0, 1.0)


def video_clips(n_clips, frames, res, seed, sigma=4.0, phase=1.0):
    """[n_clips, frames, res, res, 3] float32: frame t of a clip = cos(phi_t) B0 + sin(phi_t) B1, phi_t = phase t / T."""
    out = np.empty((n_clips, frames, res, res, 3), np.float32)
    phi = phase * np.arange(frames) / float(frames)
    for c in range(n_clips):
        b = lowpass_fields(2, res, seed * 1000 + c, sigma)
        out[c] = np.cos(phi)[:, None, None, None] * b[0] + np.sin(phi)[:, None, None, None] * b[1]
    return out


def video_cuts(n_clips, frames, res, seed, sigma=4.0, phase=1.0):
    """[n_clips, frames, res, res, 3] float32: ONE video of n_clips * frames frames (the rotation of `video_clips`, same
    total angle `phase`), cut into n_clips consecutive clips -- the clips of a batch show the same scene at different
    times, so their frames differ while the joints stay where they are (`scene_positions`).  Why not independent scenes
    per clip: the heads are FITTED, and a 1x1 head over C channels can place J peaks for about as many independent scenes
    as C / (map cells) allows -- one, at 32 x 32 maps (measured: two independent clips leave the ridge fit of the 32 x 32
    block at S ~ 0.25 before scaling and the bisection at its cap).  Different seeds are different scenes."""
    b = lowpass_fields(2, res, seed * 1000, sigma)
    phi = phase * np.arange(n_clips * frames) / float(n_clips * frames)
    v = np.cos(phi)[:, None, None, None] * b[0] + np.sin(phi)[:, None, None, None] * b[1]
    return v.astype(np.float32).reshape(n_clips, frames, res, res, 3)


def scene_positions(n_clips, frames, joints, seed):
    """Target positions for `video_cuts`: one per joint for the whole video, [n_clips * frames, J, 2]."""
    p = np.random.default_rng(seed + 77).uniform(0.1, 0.9, (1, joints, 2))
    return np.repeat(p, n_clips * frames, axis=0)


def joint_positions(n_clips, frames, joints, seed):
    """One target position per (clip, joint) in normalised [0.1, 0.9]^2, constant over the clip: [n_clips*frames, J, 2]."""
    p = np.random.default_rng(seed + 77).uniform(0.1, 0.9, (n_clips, 1, joints, 2))
    return np.repeat(p, frames, axis=1).reshape(n_clips * frames, joints, 2)


def sensitivity(logits):
    """S = max(sum p |gx - x|, sum p |gy - y|) per (frame, joint) and the per-map maximum probability, in float64,
    with the reference's own grid (utils/math.py:6-19 via oracle.ops.linspace_2d)."""
    import torch
    from oracle import ops
    l = torch.from_numpy(np.asarray(logits, dtype=np.float64))
    p = ops.channel_softmax_2d(l, 1.0)
    xy = ops.softargmax2d_from_prob(p).numpy()
    p = p.numpy()
    h, w = p.shape[1], p.shape[2]
    gx = ops.linspace_2d(h, w, 0).astype(np.float64)[None, :, :, None]
    gy = ops.linspace_2d(h, w, 1).astype(np.float64)[None, :, :, None]
    sx = (p * np.abs(gx - xy[:, None, None, :, 0])).sum(axis=(1, 2))
    sy = (p * np.abs(gy - xy[:, None, None, :, 1])).sum(axis=(1, 2))
    return np.maximum(sx, sy), p.max(axis=(1, 2))


def peak_targets(pos, h, w, peak=PEAK):
    """Target logits [F, h, w, J]: one Gaussian peak per (frame, joint); on coarse maps the peak sits on a cell."""
    gx = np.linspace(0, 1, w)[None, None, :, None]
    gy = np.linspace(0, 1, h)[None, :, None, None]
    px, py = pos[:, None, None, :, 0], pos[:, None, None, :, 1]
    if w <= 8:
        px, py = np.round(px * (w - 1)) / (w - 1), np.round(py * (h - 1)) / (h - 1)
    width = 0.9 if w > 8 else 0.6                       # cells
    r2 = ((gx - px) * (w - 1)) ** 2 + ((gy - py) * (h - 1)) ** 2
    return peak * np.exp(-r2 / (2.0 * width * width))


def _scale_for(logits, target):
    lo, hi = 0.05, 50.0
    for _ in range(30):
        mid = (lo * hi) ** 0.5
        if sensitivity(logits * mid)[0].max() > target:
            lo = mid
        else:
            hi = mid
    return hi


def _scale_per_joint(logits, target):
    """One factor per joint (output channel of the head), [J]: every joint's maps are brought to max S = target on their
    own, so that one badly placed peak (a tie between cells keeps S high until the map is nearly one-hot) does not drive
    the OTHER joints' maps one-hot as the common factor of `_scale_for` does."""
    j = logits.shape[-1]
    lo, hi = np.full(j, 0.05), np.full(j, 50.0)
    for _ in range(30):
        mid = np.sqrt(lo * hi)
        over = sensitivity(logits * mid)[0].max(axis=0) > target
        lo, hi = np.where(over, mid, lo), np.where(over, hi, mid)
    return hi


def fit_spnet_heads(model, ocfg, clips, pos, s_target=S_TARGET, per_joint=False):
    """Fit every '<block>_heatmaps_conv1' of a synthetic SPNet (deephar/models/spnet.py:24-48) so that its maps have
    one peak per joint at `pos`; heads feed the re-injection convs of later blocks, so they are fitted in prediction
    order, one fp32 oracle pass each.  A replica head ('_conv1_replica', spnet.py:36-38) gets 0.9 x the fitted kernel
    (distinct numbers, same peaks).  per_joint: the bisection factor is chosen per joint instead of per head (round 4
    vectors; the committed round-3 goldens were fitted with the common factor).  Returns {layer name: float32 kernel} of
    everything it changed."""
    import torch
    from deephar_amd import weights
    from oracle import spnet as osp
    layers = {l.name: l for n in model._nodes for l in n.layers.values()}
    taps = {}
    osp.forward(weights.as_dict(model), clips, ocfg, dtype=torch.float32, taps=taps)
    blocks = [k[:-len('/logits')] for k in taps if k.endswith('/logits')]
    changed = {}
    for b in blocks:
        taps = {'want_head_inputs': True, 'stop_at': b + '_heatmaps/in'}
        try:
            osp.forward(weights.as_dict(model), clips, ocfg, dtype=torch.float32, taps=taps)
        except osp.StopForward:
            pass
        f = taps[b + '_heatmaps/in'].astype(np.float64)
        n, h, w, c = f.shape
        x = f.reshape(-1, c)
        y = peak_targets(pos, h, w).reshape(n * h * w, -1)
        g = x.T @ x
        k = np.linalg.solve(g + RIDGE * np.trace(g) / c * np.eye(c), x.T @ y)          # [C, J]
        k *= (_scale_per_joint if per_joint else _scale_for)((x @ k).reshape(n, h, w, -1), s_target)
        for name, factor in ((b + '_heatmaps_conv1', 1.0), (b + '_heatmaps_conv1_replica', 0.9)):
            if name in layers:
                p = layers[name].params[0]
                p.set((factor * k).reshape(p.shape).astype(np.float32))
                changed[name] = p.value.copy()
    return changed


def apply_heads(model, heads):
    layers = {l.name: l for n in model._nodes for l in n.layers.values()}
    for name, k in heads.items():
        p = layers[name].params[0]
        assert tuple(p.shape) == tuple(k.shape), (name, p.shape, k.shape)
        p.set(np.asarray(k, np.float32))


def conditioning_stats(t64):
    """Per prediction block, from the fp64 oracle's logits: S_max, S_median, min over maps of the top probability,
    max |logit|.  No assertion (margin sweep)."""
    stats = {}
    for key in [k for k in t64 if k.endswith('/logits')]:
        s, pmax = sensitivity(t64[key])
        stats[key[:-len('/logits')]] = dict(S_max=float(s.max()), S_median=float(np.median(s)),
                                            pmax_min=float(pmax.min()), logit_absmax=float(np.abs(t64[key]).max()))
    return stats


def assert_well_conditioned(t64, label=''):
    """t64: taps of the fp64 oracle pass.  Every prediction block: S <= S_MAX, not one-hot, finite.  Returns stats."""
    stats = {}
    for key in [k for k in t64 if k.endswith('/logits')]:
        s, pmax = sensitivity(t64[key])
        b = key[:-len('/logits')]
        stats[b] = dict(S_max=float(s.max()), S_median=float(np.median(s)), pmax_min=float(pmax.min()),
                        logit_absmax=float(np.abs(t64[key]).max()))
        assert s.max() <= S_MAX, '%s %s: read-out sensitivity S = %.3f > %.2f' % (label, b, s.max(), S_MAX)
        assert np.median(s) >= 1e-4 and pmax.min() < 0.9999, \
            '%s %s: heat-maps are one-hot -- the coordinate would not depend on the logits (vacuous test)' % (label, b)
    return stats
