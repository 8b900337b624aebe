"""Pin the CPU oracle with analytic known answers and a second, independent NumPy-fp64 statement of the
decoder ops (the reference has no golden vectors: parity is otherwise unpinned, see oracle/__init__.py)."""
import numpy as np
import pytest
import torch

from oracle import ops


def test_same_padding_is_tf_asymmetric():
    # 3x3 s2 on even sizes pads (0 top/left, 1 bottom/right); 7x7 s2 -> (2,3); 5x5 s1 -> (2,2)
    assert ops.same_pad(256, 3, 2) == (0, 1, 128)
    assert ops.same_pad(256, 7, 2) == (2, 3, 128)
    assert ops.same_pad(32, 5, 1) == (2, 2, 32)
    assert ops.same_pad(64, 1, 1) == (0, 0, 64)
    assert ops.same_pad(17, 2, 2) == (0, 1, 9)


def test_conv_same_stride2_uses_bottom_right_padding():
    x = torch.zeros(1, 4, 4, 1)
    x[0, 3, 3, 0] = 1.0          # bottom-right pixel
    k = torch.zeros(3, 3, 1, 1)
    k[0, 0, 0, 0] = 1.0          # top-left tap
    y = ops.conv2d(x, k, (2, 2), 'same')
    # out(1,1) reads in(2..4, 2..4) (pad after): top-left tap hits in(2,2)=0 ; out(1,1) with tap (1,1) would hit (3,3)
    assert y.shape == (1, 2, 2, 1) and float(y.abs().sum()) == 0.0
    k = torch.zeros(3, 3, 1, 1)
    k[1, 1, 0, 0] = 1.0
    y = ops.conv2d(x, k, (2, 2), 'same')
    assert float(y[0, 1, 1, 0]) == 1.0   # symmetric (torch-style) padding would put it elsewhere


def test_identity_pointwise_and_depthwise():
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((2, 6, 5, 4)).astype(np.float32))
    eye = torch.eye(4).reshape(1, 1, 4, 4)
    assert torch.equal(ops.conv2d(x, eye), x)
    dw = torch.zeros(5, 5, 4, 1)
    dw[2, 2, :, 0] = 1.0
    assert torch.equal(ops.depthwise_conv2d(x, dw), x)
    assert torch.equal(ops.sepconv2d(x, dw, eye), x)


def test_batchnorm_formula():
    x = torch.tensor([[[[1.0, 2.0]]]])
    y = ops.batchnorm(x, beta=torch.tensor([0.5, -0.5]), mean=torch.tensor([1.0, 0.0]),
                      var=torch.tensor([3.0, 0.999]))
    exp = np.array([0.5, -0.5 + 2.0 / np.sqrt(0.999 + 1e-3)], dtype=np.float32)
    np.testing.assert_allclose(y.numpy().ravel(), exp, rtol=1e-6)
    y2 = ops.batchnorm(x, torch.zeros(2), torch.zeros(2), torch.ones(2) - 1e-3, gamma=torch.tensor([2.0, 3.0]))
    np.testing.assert_allclose(y2.numpy().ravel(), [2.0, 6.0], rtol=1e-6)


def test_maxpool_same_ignores_padding():
    x = -torch.ones(1, 3, 3, 1)
    y = ops.maxpool2d(x, (3, 3), (2, 2), 'same')
    assert y.shape == (1, 2, 2, 1) and torch.all(y == -1)   # zero padding would give 0
    y = ops.maxpool2d(x, (2, 2))
    assert y.shape == (1, 1, 1, 1)


def test_upsample_nearest():
    x = torch.arange(4.0).reshape(1, 2, 2, 1)
    y = ops.upsample2d(x)
    assert y[0, :, :, 0].tolist() == [[0, 0, 1, 1], [0, 0, 1, 1], [2, 2, 3, 3], [2, 2, 3, 3]]


@pytest.mark.parametrize('hw', [(32, 32), (8, 16)])
def test_softargmax_of_peaked_map_is_grid_coordinate(hw):
    H, W = hw
    h = torch.full((1, H, W, 3), -1e4)
    pts = [(0, 0), (H - 1, W - 1), (H // 2, 3)]
    for c, (r, q) in enumerate(pts):
        h[0, r, q, c] = 50.0
    xy = ops.softargmax2d(h).numpy()[0]
    for c, (r, q) in enumerate(pts):
        # endpoints 0 and 1 inclusive (np.linspace(0,1,W)), not pixel centres
        np.testing.assert_allclose(xy[c], [q / (W - 1), r / (H - 1)], atol=1e-6)


def test_softargmax_of_constant_map_is_centre_and_probs_sum_to_one():
    h = torch.zeros(2, 16, 16, 5)
    p = ops.channel_softmax_2d(h)
    np.testing.assert_allclose(p.sum(dim=(1, 2)).numpy(), 1.0, rtol=1e-6)
    np.testing.assert_allclose(ops.softargmax2d(h).numpy(), 0.5, atol=1e-6)


def test_softmax_alpha_temperature():
    rng = np.random.default_rng(1)
    h = torch.from_numpy(rng.standard_normal((1, 8, 8, 2)).astype(np.float32))
    a = ops.channel_softmax_2d(h, alpha=3.0)
    b = ops.channel_softmax_2d(3.0 * h)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-6)


def test_joint_probability_is_max_of_2x2_window_sums():
    x = torch.zeros(1, 4, 4, 2)
    x[0, 1, 1, 0] = 1.0
    x[0, 1, 2, 0] = 2.0
    x[0, 2, 1, 0] = 3.0
    x[0, 2, 2, 0] = 4.0
    x[0, 3, 3, 1] = 7.0     # bottom-right corner pixel: only one window contains it
    v = ops.joints_probability(x).numpy()[0, :, 0]
    np.testing.assert_allclose(v, [10.0, 7.0])


def test_context_aggregation_weights_and_grouping():
    # 2 joints x 2 contexts; contexts of joint j are channels 2j, 2j+1 (blocks.py:229-231)
    ys = torch.tensor([[[0.2, 0.4], [0.6, 0.8]]])
    yc = torch.tensor([[[0.0, 0.0], [1.0, 1.0], [0.5, 0.25], [0.5, 0.75]]])
    pc = torch.tensor([[[1.0], [3.0], [2.0], [2.0]]])
    y = ops.context_aggregation(ys, yc, pc, 2, 2, alpha=0.8).numpy()[0]
    np.testing.assert_allclose(y[0], 0.8 * np.array([0.2, 0.4]) + 0.2 * np.array([0.75, 0.75]), rtol=1e-6)
    np.testing.assert_allclose(y[1], 0.8 * np.array([0.6, 0.8]) + 0.2 * np.array([0.5, 0.5]), rtol=1e-6)


def test_softargmax1d_grid_is_pixel_centred():
    D, J = 16, 3
    hz = torch.full((1, D, J), -1e4)
    hz[0, 0, 0] = 0.0
    hz[0, D - 1, 1] = 0.0
    hz[0, 5, 2] = 0.0
    z = ops.softargmax1d(hz).numpy()[0, :, 0]
    np.testing.assert_allclose(z, [1 / (2 * D), 1 - 1 / (2 * D), (2 * 5 + 1) / (2 * D)], atol=1e-6)


def test_kronecker_and_maxmin_pooling():
    rng = np.random.default_rng(2)
    hm = torch.from_numpy(rng.random((2, 3, 4, 4, 5)).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((2, 3, 4, 4, 7)).astype(np.float32))
    f = ops.kronecker_prod(hm, x).numpy()
    ref = (hm.numpy()[..., :, None] * x.numpy()[..., None, :]).sum(axis=(2, 3))
    np.testing.assert_allclose(f, ref, rtol=1e-5, atol=1e-5)
    t = torch.from_numpy(rng.standard_normal((1, 5, 6, 2)).astype(np.float32))
    mm = ops.max_min_pooling(t).numpy()
    assert mm.shape == (1, 3, 3, 2)
    np.testing.assert_allclose(mm[0, 0, 0], t[0, :2, :2].amax(dim=(0, 1)).numpy() + t[0, :2, :2].amin(dim=(0, 1)).numpy(),
                               rtol=1e-6)
    g = ops.global_max_min_pooling(t).numpy()
    np.testing.assert_allclose(g[0], t[0].amax(dim=(0, 1)).numpy() + t[0].amin(dim=(0, 1)).numpy(), rtol=1e-6)


# ---- second, independent statement (NumPy fp64 loops) of the decoder --------------------------------------

def _np_decoder(h, alpha):
    F, H, W, C = h.shape
    h = h.astype(np.float64)
    xy = np.zeros((F, C, 2))
    conf = np.zeros((F, C))
    gx = np.linspace(0, 1, W).astype(np.float32).astype(np.float64)
    gy = np.linspace(0, 1, H).astype(np.float32).astype(np.float64)
    for f in range(F):
        for c in range(C):
            m = alpha * h[f, :, :, c]
            e = np.exp(m - m.max())
            p = e / max(e.sum(), 1e-7)
            xy[f, c, 0] = (p * gx[None, :]).sum()
            xy[f, c, 1] = (p * gy[:, None]).sum()
            raw = h[f, :, :, c]
            win = raw[:-1, :-1] + raw[:-1, 1:] + raw[1:, :-1] + raw[1:, 1:]
            conf[f, c] = win.max()
    return xy, conf


def test_torch_oracle_agrees_with_numpy_fp64_decoder():
    rng = np.random.default_rng(3)
    h = (rng.standard_normal((2, 32, 32, 6)) * 4).astype(np.float32)
    xy_np, conf_np = _np_decoder(h, alpha=1.0)
    t = torch.from_numpy(h)
    np.testing.assert_allclose(ops.softargmax2d(t).numpy(), xy_np, atol=2e-6)
    np.testing.assert_allclose(ops.joints_probability(t).numpy()[..., 0], conf_np, rtol=1e-5, atol=1e-5)
    t64 = t.double()
    np.testing.assert_allclose(ops.softargmax2d(t64).numpy(), xy_np, atol=1e-12)


O = ops


# ---- an implementation of the Keras layer semantics that shares NOTHING with oracle/ops.py or mini-keras ----------------
# oracle/ops.py and oracle/refrun/minikeras.py are both written on torch.nn.functional by the same author, so agreement
# between them does not pin what Keras 2.1.4 / TF 1.6 decide inside a layer.  These are plain NumPy float64
# loops written from the TensorFlow documentation of each op: 'SAME' padding = ceil(in / stride) outputs,
# total padding (out - 1) * stride + k - in, the ODD cell goes to the bottom / right; max-pool ignores padding cells;
# UpSampling2D repeats rows and columns; BatchNormalization inference = gamma * (x - mean) / sqrt(var + 1e-3) + beta;
# SeparableConv2D = depthwise (per-channel) then 1x1, no activation between.
def _np_same(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, out


def _np_conv2d_same(x, w, s):
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    pt, oh = _np_same(h, kh, s)
    pl, ow = _np_same(wd, kw, s)
    y = np.zeros((n, oh, ow, cout))
    for b in range(n):
        for i in range(oh):
            for j in range(ow):
                for a in range(kh):
                    for c in range(kw):
                        ii, jj = i * s - pt + a, j * s - pl + c
                        if 0 <= ii < h and 0 <= jj < wd:
                            y[b, i, j] += x[b, ii, jj] @ w[a, c]
    return y


def _np_depthwise_same(x, w):
    n, h, wd, ch = x.shape
    kh, kw = w.shape[:2]
    pt, _ = _np_same(h, kh, 1)
    pl, _ = _np_same(wd, kw, 1)
    y = np.zeros_like(x)
    for i in range(h):
        for j in range(wd):
            for a in range(kh):
                for c in range(kw):
                    ii, jj = i - pt + a, j - pl + c
                    if 0 <= ii < h and 0 <= jj < wd:
                        y[:, i, j] += x[:, ii, jj] * w[a, c, :, 0]
    return y


def _np_maxpool_same(x, k, s):
    n, h, wd, ch = x.shape
    pt, oh = _np_same(h, k, s)
    pl, ow = _np_same(wd, k, s)
    y = np.full((n, oh, ow, ch), -np.inf)
    for i in range(oh):
        for j in range(ow):
            for a in range(k):
                for c in range(k):
                    ii, jj = i * s - pt + a, j * s - pl + c
                    if 0 <= ii < h and 0 <= jj < wd:
                        y[:, i, j] = np.maximum(y[:, i, j], x[:, ii, jj])
    return y


@pytest.mark.parametrize('h,w,k,s', [(9, 8, 3, 2), (8, 8, 5, 1), (7, 10, 7, 2), (6, 6, 1, 1), (5, 9, 3, 1), (8, 6, 2, 2)])
def test_layer_semantics_against_independent_numpy_loops(h, w, k, s):
    rng = np.random.default_rng(h * 100 + w * 10 + k + s)
    x = rng.standard_normal((2, h, w, 3))
    wt = rng.standard_normal((k, k, 3, 4))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    np.testing.assert_allclose(O.conv2d(t(x), t(wt), (s, s), 'same').numpy(), _np_conv2d_same(x, wt, s), atol=1e-12)
    if s == 1:
        dw = rng.standard_normal((k, k, 3, 1))
        pw = rng.standard_normal((1, 1, 3, 5))
        want = _np_conv2d_same(_np_depthwise_same(x, dw), pw, 1)
        np.testing.assert_allclose(O.sepconv2d(t(x), t(dw), t(pw)).numpy(), want, atol=1e-12)
    if k in (2, 3):
        np.testing.assert_array_equal(O.maxpool2d(t(x), (k, k), (s, s), 'same').numpy(), _np_maxpool_same(x, k, s))
        mm = _np_maxpool_same(x, 2, 2) - _np_maxpool_same(-x, 2, 2)      # layers.max_min_pooling: max(x) - max(-x)
        np.testing.assert_allclose(O.max_min_pooling(t(x)).numpy(), mm, atol=0)
    up = np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)
    np.testing.assert_array_equal(O.upsample2d(t(x)).numpy(), up)
    beta, mean, var, gamma = (rng.standard_normal(3), rng.standard_normal(3), rng.uniform(0.5, 2, 3), rng.uniform(0.5, 2, 3))
    bn = gamma * (x - mean) / np.sqrt(var + 1e-3) + beta
    np.testing.assert_allclose(O.batchnorm(t(x), t(beta), t(mean), t(var), gamma=t(gamma)).numpy(), bn, atol=1e-12)
    bn0 = (x - mean) / np.sqrt(var + 1e-3) + beta                         # layers.py BatchNormalization(scale=False)
    np.testing.assert_allclose(O.batchnorm(t(x), t(beta), t(mean), t(var)).numpy(), bn0, atol=1e-12)


@pytest.mark.parametrize('h,w', [(256, 256), (128, 128), (33, 31), (17, 20), (8, 16)])
@pytest.mark.parametrize('k,s', [(1, 1), (3, 1), (3, 2), (5, 1), (5, 2), (7, 2), (2, 2)])
def test_same_padding_against_a_third_party_statement(h, w, k, s):
    """[r05] One more witness for the Keras / TF semantics this oracle restates ("Keras-layer numerics restated
    by the same author in every witness"): `apply_tf_padding` of HuggingFace transformers' MobileNetV2 port -- written by
    others to reproduce TensorFlow checkpoints bit for bit, so TF's 'SAME' rule (pad_total from in % stride, the extra pixel
    on the bottom / right) is stated there independently.  The oracle's Conv2D / depthwise convolution on every geometry
    the models use (kernels 1-7, strides 1-2, even and odd maps) equals torch's convolution behind that padding, and the
    product's own `layers.same_pad` (which fills dh_conv_args.PT / PL) names the same pixels."""
    tf_pad = pytest.importorskip('transformers.models.mobilenet_v2.modeling_mobilenet_v2').apply_tf_padding
    from deephar_amd.layers import same_pad
    rng = np.random.default_rng(h * 131 + w * 7 + k * 3 + s)
    cin, cout = 3, 4
    x = torch.from_numpy(rng.standard_normal((2, h, w, cin)))
    kern = torch.from_numpy(rng.standard_normal((k, k, cin, cout)))
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, bias=False).double()
    with torch.no_grad():
        conv.weight.copy_(kern.permute(3, 2, 0, 1))
        want = conv(tf_pad(x.permute(0, 3, 1, 2), conv)).permute(0, 2, 3, 1)
    got = ops.conv2d(x, kern, (s, s), 'same')
    assert got.shape == want.shape == (2, -(-h // s), -(-w // s), cout)
    assert torch.allclose(got, want, rtol=0, atol=1e-12)
    # the depthwise half of SeparableConv2D goes through the same rule
    dwk = torch.from_numpy(rng.standard_normal((k, k, cin, 1)))
    dconv = torch.nn.Conv2d(cin, cin, k, stride=s, groups=cin, bias=False).double()
    with torch.no_grad():
        dconv.weight.copy_(dwk.permute(2, 3, 0, 1))
        dwant = dconv(tf_pad(x.permute(0, 3, 1, 2), dconv)).permute(0, 2, 3, 1)
    assert torch.allclose(ops.depthwise_conv2d(x, dwk, (s, s), 'same'), dwant, rtol=0, atol=1e-12)
    # the host-side rule of the product: pixels of padding before each axis = what the third-party statement pads
    padded = tf_pad(torch.zeros(1, 1, h, w), conv)
    pt, pb, oh = same_pad(h, k, s)
    pl, pr, ow = same_pad(w, k, s)
    assert (padded.shape[-2], padded.shape[-1]) == (h + pt + pb, w + pl + pr) and (oh, ow) == tuple(want.shape[1:3])
    probe = tf_pad(torch.ones(1, 1, h, w), conv)[0, 0]
    assert bool((probe[:pt] == 0).all()) and bool((probe[:, :pl] == 0).all()) and probe[pt, pl] == 1


@pytest.mark.parametrize('h,w', [(128, 128), (33, 31), (17, 20), (8, 16), (5, 5)])
@pytest.mark.parametrize('k,s', [(3, 2), (2, 2), (3, 1)])
def test_same_max_pooling_geometry_against_a_third_party_statement(h, w, k, s):
    """[r05] `layers.maxpooling2d` = MaxPooling2D((3, 3), strides 2, 'same') (layers.py:92-97) and `max_min_pooling`
    (layers.py:411-425): the window geometry of TF 'SAME' pooling as HuggingFace transformers' BiT port states it
    (`BitMaxPool2d` + `DynamicPad2d`: a ceil-based formula written to reproduce TF checkpoints), with the padding value this
    oracle states itself (TF ignores padded cells: -inf).  All-negative inputs, so a zero-padded pool would differ."""
    bit = pytest.importorskip('transformers.models.bit.modeling_bit')
    rng = np.random.default_rng(h * 17 + w + k + s)
    x = -torch.from_numpy(rng.random((2, h, w, 5))) - 0.5
    pool = bit.BitMaxPool2d(k, stride=s, padding_value=float('-inf'))
    want = pool(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    got = ops.maxpool2d(x, (k, k), (s, s), 'same')
    assert got.shape == want.shape == (2, -(-h // s), -(-w // s), 5) and torch.equal(got, want)
    assert float(got.max()) < 0 and not torch.equal(got, bit.BitMaxPool2d(k, stride=s)(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)) \
        or (h % s == 0 and w % s == 0 and k <= s)


def test_upsampling_and_batchnorm_against_torch_kernels():
    """[r05] UpSampling2D (nearest) and inference BatchNormalization (epsilon inside the square root, `scale=False` layers
    have no gamma) against torch's own kernels -- third-party code for the formula; Keras' default epsilon 1e-3
    (`deephar/layers.py:51-58` passes none) remains this oracle's statement."""
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 5, 7, 6)))
    up = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode='nearest').permute(0, 2, 3, 1)
    assert torch.equal(ops.upsample2d(x, (2, 2)), up)
    mean, var = torch.from_numpy(rng.standard_normal(6)), torch.from_numpy(rng.random(6) + 0.5)
    beta, gamma = torch.from_numpy(rng.standard_normal(6)), torch.from_numpy(rng.random(6) + 0.5)
    for g in (None, gamma):
        want = F.batch_norm(x.permute(0, 3, 1, 2), mean, var, weight=g, bias=beta, training=False, eps=1e-3).permute(0, 2, 3, 1)
        got = ops.batchnorm(x, beta, mean, var, g)
        assert torch.allclose(got, want, rtol=0, atol=1e-12)
