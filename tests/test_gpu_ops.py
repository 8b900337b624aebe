"""Op-level parity: every kernel entry point of the C-ABI vs the CPU oracle on seeded inputs.

fp32 tolerance: the kernels and the oracle both compute in IEEE fp32 but sum in different orders
(MFMA k-ordered fmaf chain vs oneDNN blocking), so outputs are compared with
|hip - oracle| <= ATOL + RTOL * |oracle|, RTOL = 2e-5 (a few ulp times sqrt(K)), scaled ATOL.
"""
import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def _rand(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _close(got, ref, atol, rtol=RTOL, what=''):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    assert np.all(err <= tol), '%s: max err %.3e (tol %.3e) at %s' % (
        what, err.max(), tol.flat[err.argmax()], np.unravel_index(err.argmax(), err.shape))


CONV_CASES = [
    # (N, H, W, Cin, Cout, kh, kw, stride, padding)
    (2, 32, 32, 576, 576, 1, 1, 1, 'same'),     # the dominant pointwise GEMM
    (3, 16, 16, 288, 288, 1, 1, 1, 'same'),
    (2, 32, 32, 576, 48, 1, 1, 1, 'same'),      # RegMap: Cout not a multiple of 32
    (2, 32, 32, 48, 576, 1, 1, 1, 'same'),      # fReMap: K not a multiple of 32
    (2, 64, 64, 3, 32, 3, 3, 2, 'same'),        # first stem conv: Cin=3 (scalar gather), TF-SAME stride 2
    (2, 33, 31, 32, 64, 3, 3, 1, 'same'),       # odd sizes, ragged M
    (1, 32, 32, 64, 96, 3, 3, 2, 'same'),
    (2, 20, 24, 64, 64, 5, 1, 1, 'same'),
    (2, 20, 24, 64, 64, 1, 5, 1, 'same'),
    (1, 40, 40, 8, 64, 7, 7, 2, 'same'),        # SPNet entry-flow shape class (7x7 s2)
    (2, 16, 16, 2, 24, 3, 5, 1, 'same'),        # action-head conv over the (T,J) plane, Cin=2
    (1, 9, 9, 160, 64, 1, 1, 1, 'valid'),
    (1, 12, 12, 32, 32, 3, 3, 1, 'valid'),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_plain(case, hip_lib, cuda):
    from deephar_amd import functional as F
    n, h, w, cin, cout, kh, kw, s, pad = case
    rng = np.random.default_rng(sum(v for v in case if isinstance(v, int)))
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (kh, kw, cin, cout), np.sqrt(1.0 / (kh * kw * cin)))
    ref = O.conv2d(torch.from_numpy(x), torch.from_numpy(k), (s, s), pad)
    ref64 = O.conv2d(torch.from_numpy(x).double(), torch.from_numpy(k).double(), (s, s), pad)
    got = F.conv2d(torch.from_numpy(x).to(cuda), k, (s, s), pad)
    torch.cuda.synchronize()
    _close(got, ref, atol=2e-5, what='conv %s' % (case,))
    # and not further from the fp64 truth than a few times the fp32 CPU result is
    e_hip = (got.cpu().double() - ref64).abs().max().item()
    e_cpu = (ref.double() - ref64).abs().max().item()
    assert e_hip <= 4 * e_cpu + 1e-6, (e_hip, e_cpu)


@pytest.mark.parametrize('cfg', range(18))
def test_conv2d_every_tile_config(cfg, hip_lib, cuda):
    """cfg 0..8: general implicit-GEMM kernel (3x3 here); cfg 9..17: the same nine tile shapes on the LDS-DMA kernel (1x1).
    All tilings must agree bit-for-bit (same K summation order), which is what lets the autotuner pick freely."""
    from deephar_amd import functional as F
    assert hip_lib.dh_conv2d_num_tile_cfgs() == 18
    rng = np.random.default_rng(cfg % 9)
    ks = 3 if cfg < 9 else 1
    x = _rand(rng, (2, 19, 23, 96))           # M = 874: ragged in every BM
    k = _rand(rng, (ks, ks, 96, 200), 0.05)   # Cout = 200: ragged in every BN
    r1 = _rand(rng, (2, 19, 23, 200))
    ref = O.conv2d(O.relu(torch.from_numpy(x)), torch.from_numpy(k)) + torch.from_numpy(r1)
    d = lambda a: torch.from_numpy(a).to(cuda)
    got = F.conv2d(d(x), k, pre_relu=True, res1=d(r1), tile_cfg=cfg)
    base = F.conv2d(d(x), k, pre_relu=True, res1=d(r1), tile_cfg=8 if cfg < 9 else 17)
    torch.cuda.synchronize()
    _close(got, ref, atol=3e-5, what='cfg %d' % cfg)
    assert torch.equal(got, base), 'tilings disagree bitwise'


@pytest.mark.parametrize('cin,cout', [(576, 576), (48, 576), (576, 48), (100, 36), (288, 272)])
def test_pointwise_dma_kernel_matches_general_kernel(cin, cout, hip_lib, cuda):
    """K tails (k >= K clamped, zero weights), Cout tails and M tails of the LDS-DMA kernel."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(cin + cout)
    x = _rand(rng, (3, 13, 11, cin))
    k = _rand(rng, (1, 1, cin, cout), np.sqrt(1.0 / cin))
    d = lambda a: torch.from_numpy(a).to(cuda)
    a = F.conv2d(d(x), k, pre_relu=True, tile_cfg=3)
    b = F.conv2d(d(x), k, pre_relu=True, tile_cfg=12)
    c = F.conv2d(d(x), k, pre_relu=True, tile_cfg=-1)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)
    _close(a, O.conv2d(O.relu(torch.from_numpy(x)), torch.from_numpy(k)), atol=2e-5, what='pointwise')


@pytest.mark.parametrize('cin,cout,relu', [(96, 200, True), (64, 96, True), (100, 36, False), (288, 272, True), (576, 48, False)])
def test_pointwise_bn_prologue_on_the_dma_kernel(cin, cout, relu, hip_lib, cuda):
    """Round 4: BatchNormalization (+ ReLU) prologue of a 1x1 convolution on the LDS-DMA GEMM (scale / shift tables in
    LDS, applied to the A fragments) -- every tiling bit-identical to the general kernel's prologue (same fused
    multiply-add, same K order), K tails (k >= K: zero table entries against zero weights), Cout and M tails, BN + residual
    epilogue on top; vs the oracle chain relu?(x * s + b) -> conv."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(cin * 7 + cout)
    x = _rand(rng, (3, 13, 11, cin))
    k = _rand(rng, (1, 1, cin, cout), np.sqrt(1.0 / cin))
    ps, pb = rng.uniform(0.5, 1.5, cin).astype(np.float32), _rand(rng, (cin,), 0.5)
    qs, qb = rng.uniform(0.5, 1.5, cout).astype(np.float32), _rand(rng, (cout,), 0.3)
    r1 = _rand(rng, (3, 13, 11, cout))
    d = lambda a: torch.from_numpy(a).to(cuda)
    t = lambda a: torch.from_numpy(a)
    pro = t(x) * t(ps) + t(pb)
    ref = O.conv2d(O.relu(pro) if relu else pro, t(k)) * t(qs) + t(qb) + t(r1)
    kw = dict(pre_scale=d(ps), pre_shift=d(pb), pre_relu=relu, post_scale=d(qs), post_shift=d(qb), res1=d(r1))
    base = F.conv2d(d(x), k, tile_cfg=3, **kw)                       # general implicit-GEMM kernel
    _close(base, ref, atol=3e-5, what='general kernel')
    for cfg in range(9, 18):
        got = F.conv2d(d(x), k, tile_cfg=cfg, **kw)
        torch.cuda.synchronize()
        assert torch.equal(got, base), 'DMA tiling %d differs from the general kernel' % cfg
    assert torch.equal(F.conv2d(d(x), k, tile_cfg=-1, **kw), base)


@pytest.mark.parametrize('ks,cout,bn', [(3, 32, True), (7, 64, False), (7, 32, True), (3, 64, False)])
def test_first_layer_kernel(ks, cout, bn, hip_lib, cuda):
    """Round 4: conv_stem.hip -- 3 input channels, stride 2, TF-SAME asymmetric padding, 256 x 256 frames -> 128 x 128 maps
    (reception.py:61-66: 3x3 3->32 + BN + ReLU; spnet.py:317-322: 7x7 3->64).  The library takes such a layer by a rule on
    its geometry (dh_conv2d_uses_first_layer_kernel), whatever tiling is asked for; checked vs the oracle (fp32 and fp64),
    raw uint8 frames = the loader's float32 values (bitwise), batch-size invariance (bitwise)."""
    from deephar_amd import functional as F, _lib
    from deephar_amd.engine.executor import normalization_lut
    import ctypes as C
    rng = np.random.default_rng(ks * 100 + cout)
    n = 3
    k = _rand(rng, (ks, ks, 3, cout), np.sqrt(1.0 / (ks * ks * 3)))
    qs, qb = rng.uniform(0.5, 1.5, cout).astype(np.float32), _rand(rng, (cout,), 0.3)
    d = lambda a: torch.from_numpy(a).to(cuda)
    kw = dict(post_scale=d(qs), post_shift=d(qb), post_relu=True) if bn else {}
    xb = rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    lut = normalization_lut(3, 1)
    x = lut[np.arange(3)[None, None, None, :], xb]                             # the loader's float32 values
    t = lambda a: torch.from_numpy(a)
    ref, ref64 = O.conv2d(t(x), t(k), (2, 2), 'same'), O.conv2d(t(x).double(), t(k).double(), (2, 2), 'same')
    if bn:
        ref, ref64 = O.relu(ref * t(qs) + t(qb)), O.relu(ref64 * t(qs).double() + t(qb).double())
    got = F.conv2d(d(x), k, (2, 2), 'same', **kw)
    got8 = F.conv2d(d(xb), k, (2, 2), 'same', in_lut=d(lut), **kw)
    forced = F.conv2d(d(x), k, (2, 2), 'same', tile_cfg=4, **kw)                # the rule overrides the tiling
    one = F.conv2d(d(x[1:2]), k, (2, 2), 'same', **kw)
    torch.cuda.synchronize()
    assert got.shape == (n, 128, 128, cout)
    _close(got, ref, atol=3e-5, what='first layer %dx%d -> %d' % (ks, ks, cout))
    e_hip = (got.cpu().double() - ref64).abs().max().item()
    e_cpu = (ref.double() - ref64).abs().max().item()
    assert e_hip <= 4 * e_cpu + 1e-6, (e_hip, e_cpu)
    assert torch.equal(got, got8), 'uint8 frames differ from the normalised float frames'
    assert torch.equal(got, forced) and torch.equal(got[1:2], one)
    # the rule itself
    a = _lib.ConvArgs()
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, 256, 256, 3, 3, 128, 128, cout, cout
    a.KH, a.KW, a.SH, a.SW, a.PT, a.PL, a.K = ks, ks, 2, 2, (ks - 2) // 2, (ks - 2) // 2, ks * ks * 3
    a.x = a.w = a.y = 16
    assert hip_lib.dh_conv2d_uses_first_layer_kernel(C.byref(a)) == 1
    for field, val in (('OW', 64), ('Cin', 4), ('SH', 1), ('Cout', 48), ('w_split', 1), ('pre_relu', 1)):
        b_ = _lib.ConvArgs.from_buffer_copy(a)
        setattr(b_, field, val)
        assert hip_lib.dh_conv2d_uses_first_layer_kernel(C.byref(b_)) == 0, field


def test_conv2d_fused_prologue_epilogue(hip_lib, cuda):
    """BN -> ReLU -> conv -> BN -> (+res1 +res2) -> ReLU in one launch vs the unfused oracle chain."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(7)
    n, h, w, cin, cout = 2, 16, 16, 64, 96
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (3, 3, cin, cout), 0.06)
    ps, pb = rng.uniform(0.5, 1.5, cin).astype(np.float32), _rand(rng, (cin,), 0.3)
    qs, qb = rng.uniform(0.5, 1.5, cout).astype(np.float32), _rand(rng, (cout,), 0.3)
    r1, r2 = _rand(rng, (n, h, w, cout)), _rand(rng, (n, h, w, cout))
    t = lambda a: torch.from_numpy(a)
    ref = O.relu(t(x) * t(ps) + t(pb))               # zero padding happens AFTER the activation
    ref = O.conv2d(ref, t(k)) * t(qs) + t(qb) + t(r1) + t(r2)
    ref_relu = O.relu(ref)
    d = lambda a: torch.from_numpy(a).to(cuda)
    got = F.conv2d(d(x), k, pre_scale=d(ps), pre_shift=d(pb), pre_relu=True, post_scale=d(qs), post_shift=d(qb),
                   res1=d(r1), res2=d(r2))
    got_relu = F.conv2d(d(x), k, pre_scale=d(ps), pre_shift=d(pb), pre_relu=True, post_scale=d(qs),
                        post_shift=d(qb), res1=d(r1), res2=d(r2), post_relu=True)
    torch.cuda.synchronize()
    _close(got, ref, atol=3e-5, what='fused')
    _close(got_relu, ref_relu, atol=3e-5, what='fused+relu')


def test_conv2d_prologue_keeps_padding_zero(hip_lib, cuda):
    """A pre-affine with a positive shift must not leak into the zero padding (TF pads after BN+ReLU)."""
    from deephar_amd import functional as F
    x = np.zeros((1, 6, 6, 4), np.float32)
    k = np.ones((3, 3, 4, 32), np.float32)
    ps, pb = np.ones(4, np.float32), np.full(4, 2.0, np.float32)
    d = lambda a: torch.from_numpy(a).to(cuda)
    got = F.conv2d(d(x), k, pre_scale=d(ps), pre_shift=d(pb), pre_relu=True).cpu().numpy()
    # interior: 9 taps * 4 ch * 2.0 = 72 ; corner: 4 taps * 4 * 2 = 32
    assert got[0, 3, 3, 0] == 72.0 and got[0, 0, 0, 0] == 32.0 and got[0, 0, 3, 5] == 48.0


@pytest.mark.parametrize('cout,cfg', [(288, -1), (576, -1), (96, 5), (64, 1), (576, 11), (288, 13), (96, 9)])
def test_conv2d_fused_upsample_add(cout, cfg, hip_lib, cuda):
    """conv -> BN -> +res1 -> UpSampling2D -> +res2 (reception.py:122-127) in the conv epilogue."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(cout)
    n, h, w, cin = 2, 8, 8, 288
    x, k = _rand(rng, (n, h, w, cin)), _rand(rng, (1, 1, cin, cout), 0.06)
    qs, qb = rng.uniform(0.5, 1.5, cout).astype(np.float32), _rand(rng, (cout,), 0.3)
    r1, r2 = _rand(rng, (n, h, w, cout)), _rand(rng, (n, 2 * h, 2 * w, cout))
    t = lambda a: torch.from_numpy(a)
    ref = O.upsample2d(O.conv2d(t(x), t(k)) * t(qs) + t(qb) + t(r1)) + t(r2)
    d = lambda a: torch.from_numpy(a).to(cuda)
    got = F.conv2d(d(x), k, post_scale=d(qs), post_shift=d(qb), res1=d(r1), res2=d(r2), up2=True, tile_cfg=cfg)
    torch.cuda.synchronize()
    _close(got, ref, atol=3e-5, what='up2')


@pytest.mark.parametrize('shape,k', [((2, 32, 32, 576), 5), ((3, 16, 16, 288), 5), ((2, 8, 8, 288), 5),
                                     ((2, 32, 32, 384), 3), ((1, 7, 9, 20), 5), ((2, 5, 6, 6), 3),
                                     ((2, 16, 16, 16), 1),
                                     # the row ring of the LDS kernel: many bands, two column tiles, a last band of 4 rows,
                                     # a map lower than one band, a column tile that is half outside the frame
                                     ((2, 64, 64, 64), 5), ((1, 64, 64, 32), 3), ((1, 44, 32, 32), 5), ((2, 4, 32, 32), 5),
                                     ((1, 24, 48, 64), 5), ((1, 20, 128, 32), 3), ((3, 128, 128, 32), 5)])
def test_dwconv(shape, k, hip_lib, cuda):
    """Depthwise conv vs the oracle, three prologues (none, ReLU, BatchNorm + ReLU with the zero padding applied after it)."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(shape[3] + k)
    x = _rand(rng, shape)
    dw = _rand(rng, (k, k, shape[3], 1), 1.0 / k)
    ps, pb = rng.uniform(0.5, 1.5, shape[3]).astype(np.float32), _rand(rng, (shape[3],), 0.3)
    t = lambda a: torch.from_numpy(a)
    d = lambda a: torch.from_numpy(a).to(cuda)
    _close(F.dwconv2d(d(x), dw), O.depthwise_conv2d(t(x), t(dw)), atol=1e-5, what='dw plain')
    _close(F.dwconv2d(d(x), dw, pre_relu=True), O.depthwise_conv2d(O.relu(t(x)), t(dw)), atol=1e-5, what='dw relu')
    _close(F.dwconv2d(d(x), dw, pre_scale=d(ps), pre_shift=d(pb), pre_relu=True),
           O.depthwise_conv2d(O.relu(t(x) * t(ps) + t(pb)), t(dw)), atol=1e-5, what='dw bn relu')


@pytest.mark.parametrize('shape,k', [((2, 16, 16, 384), 5), ((3, 8, 8, 480), 5), ((2, 4, 4, 576), 5), ((2, 32, 32, 96), 5),
                                     ((1, 22, 16, 32), 5), ((2, 8, 8, 64), 3), ((2, 2, 2, 32), 5), ((1, 3, 5, 20), 3),
                                     ((2, 4, 6, 6), 5)])
def test_dwconv_reads_an_upsampled_input(shape, k, hip_lib, cuda):
    """[r06] dh_dw_args.up_in: the depthwise convolution of UpSampling2D((2, 2))(x) read straight from the half-resolution x
    (planner rule R11: SPNet's up-scaling unit, common.py:89-108) -- bit for bit the convolution of the explicitly
    up-sampled tensor, on all three kernels (LDS ring incl. several bands, register strips, generic) and all three
    prologues; zero padding and the BatchNormalization prologue act on the up-sampled pixels."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(sum(shape) + k)
    x = _rand(rng, shape)
    dw = _rand(rng, (k, k, shape[3], 1), 1.0 / k)
    ps, pb = rng.uniform(0.5, 1.5, shape[3]).astype(np.float32), _rand(rng, (shape[3],), 0.3)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    up = np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)
    for kw in (dict(), dict(pre_relu=True), dict(pre_scale=d(ps), pre_shift=d(pb), pre_relu=True),
               dict(pre_scale=d(ps), pre_shift=d(pb))):
        got = F.dwconv2d(d(x), dw, up_in=True, **kw)
        want = F.dwconv2d(d(up), dw, **kw)
        assert got.shape == want.shape and torch.equal(got, want), sorted(kw)
    t = lambda a: torch.from_numpy(a)
    _close(F.dwconv2d(d(x), dw, pre_scale=d(ps), pre_shift=d(pb), pre_relu=True, up_in=True),
           O.depthwise_conv2d(O.relu(t(up) * t(ps) + t(pb)), t(dw)), atol=1e-5, what='dw of an up-sampled input')


@pytest.mark.parametrize('case', [
    # (frames, H, W, C, Cout, epilogue of the conv, depthwise reads the half-resolution x)
    (16, 8, 8, 384, 480, 'plain', False),          # SPNet down-scaling unit on the 8 x 8 level (64-thread depthwise kernel)
    (16, 16, 16, 288, 384, 'plain', False),        # ... 16 x 16 (256-thread depthwise kernel)
    (16, 4, 4, 576, 480, 'plain', True),           # up-scaling unit: shortcut at 4 x 4, depthwise at 8 x 8 reading it up-sampled
    (16, 16, 16, 384, 288, 'plain', True),         # ... depthwise at 32 x 32: several bands
    (3, 8, 8, 96, 288, 'bn+res1', False),          # an epilogue on the conv side, ragged GEMM tiles (M = 192)
    (2, 16, 16, 64, 272, 'res1+res2', False),
])
def test_conv_dw_grouped_launch(case, hip_lib, cuda):
    """[r06] dh_conv2d_dw_group_f32: the 1x1 shortcut convolution and the depthwise convolution of a pre-activation residual
    unit (common.py:25-67) as ONE launch -- bit for bit the two stand-alone launches (every work-group runs its kernel's own
    code), incl. the up-scaling unit of planner rule R11 (conv at half resolution, depthwise up-samples on load)."""
    import ctypes as C
    from deephar_amd import functional as F, _lib
    from deephar_amd.layers import same_pad
    n, h, w, c, cout, epi, up = case
    rng = np.random.default_rng(sum(int(v) for v in case[:5]) + len(epi))
    x = _rand(rng, (n, h, w, c))
    k = _rand(rng, (1, 1, c, cout), np.sqrt(1.0 / c))
    dwk = _rand(rng, (5, 5, c, 1), 0.2)
    ps, pb = rng.uniform(0.5, 1.5, c).astype(np.float32), _rand(rng, (c,), 0.3)
    qs, qb = (rng.uniform(0.5, 1.5, cout).astype(np.float32), _rand(rng, (cout,), 0.3)) if 'bn' in epi else (None, None)
    r1 = _rand(rng, (n, h, w, cout)) if 'res1' in epi else None
    r2 = _rand(rng, (n, h, w, cout)) if 'res2' in epi else None
    d = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    xd, psd, pbd = d(x), d(ps), d(pb)
    kw = dict(pre_scale=psd, pre_shift=pbd, pre_relu=True, post_scale=d(qs), post_shift=d(qb), res1=d(r1), res2=d(r2))
    want_conv = F.conv2d(xd, k, **kw)
    want_dw = F.dwconv2d(xd, dwk, pre_scale=psd, pre_shift=pbd, pre_relu=True, up_in=up)
    torch.cuda.synchronize()
    # the same two argument structs through the grouped entry point
    wt, kp, np_ = F.pack_conv_weight(k, cuda)
    dwt = torch.from_numpy(np.ascontiguousarray(dwk.reshape(25, c))).to(cuda)
    yc = torch.full_like(want_conv, float('nan'))
    yd = torch.full_like(want_dw, float('nan'))
    keep = [d(qs), d(qb), d(r1), d(r2)]
    ptr = lambda t_: t_.data_ptr() if t_ is not None else None
    a = _lib.ConvArgs()
    a.x, a.w, a.y, a.pre_scale, a.pre_shift = xd.data_ptr(), wt.data_ptr(), yc.data_ptr(), psd.data_ptr(), pbd.data_ptr()
    a.post_scale, a.post_shift, a.res1, a.res2 = ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3])
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, w, c, c, h, w, cout, cout
    a.KH = a.KW = a.SH = a.SW = 1
    a.K, a.Kp, a.Np, a.ldr1, a.ldr2, a.pre_relu = c, kp, np_, cout, cout, 1
    g = _lib.DwArgs()
    dh, dw_ = (2 * h, 2 * w) if up else (h, w)
    g.x, g.w, g.y, g.pre_scale, g.pre_shift = xd.data_ptr(), dwt.data_ptr(), yd.data_ptr(), psd.data_ptr(), pbd.data_ptr()
    g.N, g.H, g.W, g.C, g.ldx, g.ldy = n, dh, dw_, c, c, c
    g.KH = g.KW = 5
    g.PT, g.PL, g.pre_relu, g.up_in = same_pad(dh, 5, 1)[0], same_pad(dw_, 5, 1)[0], 1, int(up)
    _lib.check(hip_lib.dh_conv2d_dw_group_f32(C.byref(a), C.byref(g), torch.cuda.current_stream().cuda_stream), 'group')
    torch.cuda.synchronize()
    assert torch.equal(yc, want_conv), 'conv half of the grouped launch differs'
    assert torch.equal(yd, want_dw), 'depthwise half of the grouped launch differs'
    # a pair the group does not cover answers DH_EUNSUPPORTED and touches nothing
    a.pre_relu = 0
    yc.fill_(7.0)
    assert hip_lib.dh_conv2d_dw_group_f32(C.byref(a), C.byref(g), torch.cuda.current_stream().cuda_stream) == -2
    torch.cuda.synchronize()
    assert bool((yc == 7.0).all())


def _skinny_args(rng, cuda, n, h, w, cin, cout, ks, bn, relu, res, keep):
    """(filled dh_conv_args on NaN-initialised output, expected output from dh_conv2d_f32 on the same operands)"""
    from deephar_amd import functional as F, _lib
    from deephar_amd.layers import same_pad
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (ks, ks, cin, cout), np.sqrt(1.0 / (ks * ks * cin)))
    d = lambda a_: None if a_ is None else torch.from_numpy(a_).to(cuda)
    ps, pb = (d(rng.uniform(0.5, 1.5, cin).astype(np.float32)), d(_rand(rng, (cin,), 0.3))) if bn else (None, None)
    qs, qb = d(rng.uniform(0.5, 1.5, cout).astype(np.float32)), d(_rand(rng, (cout,), 0.3))
    r1 = d(_rand(rng, (n, h, w, cout))) if res else None
    xd = d(x)
    want = F.conv2d(xd, k, pre_scale=ps, pre_shift=pb, pre_relu=relu, post_scale=qs, post_shift=qb, res1=r1, post_relu=not res)
    wt, kp, np_ = F.pack_conv_weight(k, cuda)
    y = torch.full_like(want, float('nan'))
    keep += [xd, wt, ps, pb, qs, qb, r1, y]
    ptr = lambda t_: t_.data_ptr() if t_ is not None else None
    a = _lib.ConvArgs()
    a.x, a.w, a.y, a.pre_scale, a.pre_shift = xd.data_ptr(), wt.data_ptr(), y.data_ptr(), ptr(ps), ptr(pb)
    a.post_scale, a.post_shift, a.res1 = qs.data_ptr(), qb.data_ptr(), ptr(r1)
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, w, cin, cin, h, w, cout, cout
    a.KH = a.KW = ks
    a.SH = a.SW = 1
    a.PT, a.PL = same_pad(h, ks, 1)[0], same_pad(w, ks, 1)[0]
    a.K, a.Kp, a.Np, a.ldr1, a.pre_relu, a.post_relu = ks * ks * cin, kp, np_, cout, int(relu), int(not res)
    return a, y, want


@pytest.mark.parametrize('case', [
    # (n, h, w, cin, cout, k, bn prologue, relu, residual) x 2: the action head's pairs and the corners of the kernel family
    ((2, 8, 16, 70, 240, 1, True, True, False), (2, 8, 16, 576, 160, 1, False, False, False)),     # 4 waves beside 16 (K = 70 | 576)
    ((2, 8, 16, 240, 160, 3, False, True, True), (2, 8, 8, 160, 160, 1, False, False, False)),     # 3x3 K = 2160 beside a 1x1 on another map
    ((3, 4, 4, 66, 15, 3, True, True, False), (1, 16, 16, 128, 200, 1, True, False, True)),        # scalar loads (Cin % 4 != 0), ragged tiles
    ((2, 8, 16, 384, 160, 1, False, False, False), (2, 8, 16, 384, 48, 1, True, True, False)),     # 8 waves | 8 waves
])
def test_conv_pair_launch(case, hip_lib, cuda):
    """[r06] dh_conv2d_pair_f32: two independent skinny-conv layers as ONE launch -- each half bit for bit its own
    dh_conv2d_f32 launch (its own wave count, its own load form), whatever the other half is; pairs outside the rule are
    refused and touch nothing."""
    import ctypes as C
    from deephar_amd import _lib
    rng = np.random.default_rng(sum(int(v) for c_ in case for v in c_))
    keep = []
    a, ya, want_a = _skinny_args(rng, cuda, *case[0], keep)
    b, yb, want_b = _skinny_args(rng, cuda, *case[1], keep)
    assert hip_lib.dh_conv2d_uses_split_k(C.byref(a)) == 1 and hip_lib.dh_conv2d_uses_split_k(C.byref(b)) == 1
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    for first, second in ((a, b), (b, a)):
        ya.fill_(float('nan'))
        yb.fill_(float('nan'))
        _lib.check(hip_lib.dh_conv2d_pair_f32(C.byref(first), C.byref(second), st), 'pair')
        torch.cuda.synchronize()
        assert torch.equal(ya, want_a) and torch.equal(yb, want_b)
    # a layer of another kernel family, or one that resamples its input on load: DH_EUNSUPPORTED, nothing written
    ya.fill_(7.0)
    yb.fill_(7.0)
    b.x_resample = 1
    assert hip_lib.dh_conv2d_pair_f32(C.byref(a), C.byref(b), st) == -2
    b.x_resample = 0
    cout = b.Cout
    b.Cout = 512                                             # more than 256 output channels: not a skinny layer
    assert hip_lib.dh_conv2d_pair_f32(C.byref(a), C.byref(b), st) in (-1, -2)
    b.Cout = cout
    assert hip_lib.dh_conv2d_pair_f32(C.byref(a), None, st) == -1
    torch.cuda.synchronize()
    assert bool((ya == 7.0).all()) and bool((yb == 7.0).all())


@pytest.mark.parametrize('case', [
    # n, h (pooled), w (pooled), c pooled, c direct, cout, k, pool_sh, bn prologue, residual
    (2, 8, 8, 320, 160, 200, 1, 1, True, False),      # the action head's r2 unit at T = 8: [x1 | x2 | xa] -> shortcut | conv1
    (2, 8, 8, 320, 0, 240, 1, 1, True, False),        # the first head: no features handed on
    (3, 8, 8, 320, 160, 200, 1, 2, True, True),       # T = 16: disjoint windows
    (2, 4, 8, 64, 32, 48, 3, 1, False, True),         # 3 x 3: zero padding acts on the concatenated pixels
    (1, 8, 4, 66, 30, 15, 3, 2, True, False),         # scalar loads (c_split % 4 != 0), ragged tiles
])
def test_skinny_conv_reads_pooled_and_direct_segments(case, hip_lib, cuda):
    """[r06] dh_conv2d_seg_f32: the input of a skinny-conv layer is concatenate([MaxPooling2D((2, 2), strides=(pool_sh, 2),
    'same')(x), x2]) read in place (spnet.py:126-141) -- bit for bit dh_conv2d_f32 on the tensor the pooling kernel and a
    concatenation write; layers outside the skinny-conv rule and malformed segments are refused."""
    import ctypes as C
    from deephar_amd import functional as F, _lib
    n, h, w, cp, cd, cout, ks, sh, bn, res = case
    rng = np.random.default_rng(sum(int(v) for v in case))
    x = _rand(rng, (n, h * sh, 2 * w, cp))
    x2 = _rand(rng, (n, h, w, cd)) if cd else None
    cin = cp + cd
    k = _rand(rng, (ks, ks, cin, cout), np.sqrt(1.0 / (ks * ks * cin)))
    d = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    ps, pb = (d(rng.uniform(0.5, 1.5, cin).astype(np.float32)), d(_rand(rng, (cin,), 0.3))) if bn else (None, None)
    qs, qb = d(rng.uniform(0.5, 1.5, cout).astype(np.float32)), d(_rand(rng, (cout,), 0.3))
    r1 = d(_rand(rng, (n, h, w, cout))) if res else None
    kw = dict(pre_scale=ps, pre_shift=pb, pre_relu=True, post_scale=qs, post_shift=qb, res1=r1, post_relu=not res)
    xd, x2d = d(x), d(x2)
    pooled = F.pool2d(xd, (2, 2), (sh, 2), 'same')
    assert pooled.shape == (n, h, w, cp)
    cat = torch.cat([pooled, x2d], dim=-1).contiguous() if cd else pooled
    want = F.conv2d(cat, k, **kw)
    got = F.conv2d(xd, k, seg=(x2d, sh), **kw)
    assert torch.equal(got, want)
    # refusals: an up-sampling epilogue, a layer of another family (Cout > 256), a split point outside the channels
    with pytest.raises(Exception):
        F.conv2d(xd, _rand(rng, (ks, ks, cin, 512), 0.1), seg=(x2d, sh), pre_relu=True)
    with pytest.raises(Exception):
        F.conv2d(xd, k, seg=(x2d, 3), **kw)


@pytest.mark.parametrize('h,w,c,k', [(32, 32, 64, 5), (16, 16, 32, 5), (8, 8, 32, 3), (40, 64, 32, 3)])
def test_dwconv_on_channel_slabs(h, w, c, k, hip_lib, cuda):
    """The planner hands the depthwise kernel views into wider tensors (concat slabs): ldx, ldy > C and a channel offset.
    The buffer descriptors of the LDS kernel are sized from ld, not from C; nothing outside the slab may be touched."""
    import ctypes as C
    from deephar_amd import _lib
    from deephar_amd.layers import same_pad
    rng = np.random.default_rng(h + w + c + k)
    n, cx, cy, ox, oy = 3, 3 * c, 3 * c, c, 2 * c              # input slab [c, 2c), output slab [2c, 3c) of 3c-channel tensors
    xf = torch.from_numpy(_rand(rng, (n, h, w, cx))).to(cuda)
    yf = torch.full((n, h, w, cy), 7.0, device=cuda)
    dw = _rand(rng, (k, k, c, 1), 1.0 / k)
    wt = torch.from_numpy(np.ascontiguousarray(dw.reshape(k * k, c))).to(cuda)
    a = _lib.DwArgs()
    a.x, a.w, a.y = xf.data_ptr() + 4 * ox, wt.data_ptr(), yf.data_ptr() + 4 * oy
    a.N, a.H, a.W, a.C, a.ldx, a.ldy = n, h, w, c, cx, cy
    a.KH = a.KW = k
    a.PT, a.PL, a.pre_relu = same_pad(h, k, 1)[0], same_pad(w, k, 1)[0], 1
    _lib.check(hip_lib.dh_dwconv2d_f32(C.byref(a), torch.cuda.current_stream().cuda_stream), 'dw slab')
    torch.cuda.synchronize()
    ref = O.depthwise_conv2d(O.relu(xf[..., ox:ox + c].cpu()), torch.from_numpy(dw))
    _close(yf[..., oy:oy + c], ref, atol=1e-5, what='dw slab')
    assert torch.all(yf[..., :oy] == 7.0) and torch.all(yf[..., oy + c:] == 7.0), 'wrote outside its channel slab'


@pytest.mark.parametrize('shape,pool,strides,pad', [
    ((2, 128, 128, 64), (3, 3), (2, 2), 'same'), ((2, 64, 64, 192), (2, 2), (2, 2), 'valid'),
    ((2, 32, 32, 576), (2, 2), None, 'valid'), ((2, 16, 17, 30), (2, 2), (2, 2), 'same'),
    ((1, 9, 7, 5), (3, 3), (2, 2), 'same'), ((2, 32, 20, 8), (2, 2), (2, 2), 'same')])
def test_pool_bit_exact(shape, pool, strides, pad, hip_lib, cuda):
    from deephar_amd import functional as F
    rng = np.random.default_rng(sum(shape))
    x = _rand(rng, shape)
    ref = O.maxpool2d(torch.from_numpy(x), pool, strides, pad)
    got = F.pool2d(torch.from_numpy(x).to(cuda), pool, strides, pad).cpu()
    assert torch.equal(got, ref)
    if pool == (2, 2) and pad == 'same':
        ref = O.max_min_pooling(torch.from_numpy(x), pool, pad)
        got = F.pool2d(torch.from_numpy(x).to(cuda), pool, strides, pad, mode=1).cpu()
        assert torch.equal(got, ref)


def test_upsample_add_bit_exact(hip_lib, cuda):
    from deephar_amd import functional as F
    rng = np.random.default_rng(5)
    a, b = _rand(rng, (2, 16, 16, 288)), _rand(rng, (2, 8, 8, 288))
    ref = torch.from_numpy(a) + O.upsample2d(torch.from_numpy(b))
    got = F.upsample2x_add(torch.from_numpy(b).to(cuda), torch.from_numpy(a).to(cuda)).cpu()
    assert torch.equal(got, ref)
    assert torch.equal(F.upsample2x_add(torch.from_numpy(b).to(cuda)).cpu(), O.upsample2d(torch.from_numpy(b)))


@pytest.mark.parametrize('shape,alpha', [((4, 32, 32, 16), 1.0), ((3, 32, 32, 32), 1.0), ((2, 32, 32, 17), 1.0),
                                         ((2, 16, 16, 17), 2.5), ((3, 8, 8, 20), 1.0), ((2, 4, 4, 17), 0.7),
                                         ((1, 32, 32, 272), 1.0)])
def test_softargmax2d(shape, alpha, hip_lib, cuda):
    """Coordinates within 1e-3 px of a 256-px crop (|d| <= 3.9e-6 in normalised units)."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(shape[3])
    h = _rand(rng, shape, 4.0)
    t = torch.from_numpy(h)
    out = F.softargmax2d(t.to(cuda), alpha=alpha, conf_scale=4.0, want_prob=True)
    p = O.channel_softmax_2d(t, alpha)
    p64 = O.channel_softmax_2d(t.double(), alpha)
    xy64 = O.softargmax2d_from_prob(p64)
    xy = out['xy'].cpu()
    assert (xy.double() - xy64).abs().max().item() <= 3.9e-6 / 2
    assert (xy - O.softargmax2d_from_prob(p)).abs().max().item() <= 3.9e-6
    _close(out['prob'], p, atol=1e-9, rtol=1e-5, what='prob')
    _close(out['conf_raw'], O.joints_probability(4.0 * t), atol=1e-5, what='conf_raw')
    _close(out['conf_prob'], O.joints_probability(p), atol=1e-9, rtol=1e-5, what='conf_prob')
    assert torch.equal(out['gmax'].cpu(), torch.amax(t, dim=(1, 2)))


def test_softargmax2d_known_answers(hip_lib, cuda):
    from deephar_amd import functional as F
    H = W = 32
    h = np.full((1, H, W, 3), -1e4, np.float32)
    pts = [(0, 0), (H - 1, W - 1), (7, 20)]
    for c, (r, q) in enumerate(pts):
        h[0, r, q, c] = 30.0
    xy = F.softargmax2d(torch.from_numpy(h).to(cuda))['xy'].cpu().numpy()[0]
    for c, (r, q) in enumerate(pts):
        np.testing.assert_allclose(xy[c], [np.float32(q / (W - 1)), np.float32(r / (H - 1))], atol=1e-7)
    flat = F.softargmax2d(torch.zeros(2, 16, 16, 5, device=cuda))['xy'].cpu().numpy()
    np.testing.assert_allclose(flat, 0.5, atol=1e-6)


@pytest.mark.parametrize('shape,joints,nctx', [((3, 32, 32, 48), 16, 2), ((2, 16, 16, 52), 12, 3), ((2, 32, 32, 8), 4, 1),
                                               ((1, 8, 8, 64), 16, 2)])
def test_softargmax2d_with_context_in_one_launch(shape, joints, nctx, hip_lib, cuda):
    """dh_softargmax2d_context_f32 = the 2-D decoder with context of reception.pose_regression_2d_context in one launch:
    against the oracle's soft-argmax / keypoint confidence / context aggregation (1e-3 px on the pose) and against the
    three-launch path it replaces (same maps: coordinates agree to fp32 rounding)."""
    from deephar_amd import functional as F
    f, hh, ww, ld = shape
    rng = np.random.default_rng(sum(shape) + nctx)
    h = _rand(rng, shape, 4.0) + 1.0                     # (positive offset: context confidences away from 0)
    t = torch.from_numpy(h)
    c = joints * (1 + nctx)
    pose, conf = F.softargmax2d_context(t.to(cuda), joints, nctx, 0.8, conf_scale=1.0)
    hs, hc = t[..., :joints].double(), t[..., joints:c].double()
    ys = O.softargmax2d_from_prob(O.channel_softmax_2d(hs, 1.0))
    yc = O.softargmax2d_from_prob(O.channel_softmax_2d(hc, 1.0))
    pc = O.joints_probability(hc)
    assert float(pc.reshape(f, joints, nctx).sum(-1).min()) > 1.0
    ref = O.context_aggregation(ys, yc, pc, joints, nctx, 0.8)
    assert (pose.cpu().double() - ref).abs().max().item() <= 3.9e-6
    _close(conf, O.joints_probability(hs), atol=1e-5, what='joint confidence')
    d = t.to(cuda)
    a = F.softargmax2d(d[..., :joints].contiguous())
    b = F.softargmax2d(d[..., joints:c].contiguous())
    three = F.context_aggregation(a['xy'], b['xy'], b['conf_raw'], nctx, 0.8)
    assert float((three - pose).abs().max()) <= 2e-6 and torch.equal(a['conf_raw'], conf)
    with pytest.raises(Exception):
        F.softargmax2d_context(torch.zeros(1, 8, 8, 15, device=cuda), 5, 2, 0.8)      # J % 4 != 0: not this kernel


def test_context_aggregation(hip_lib, cuda):
    from deephar_amd import functional as F
    rng = np.random.default_rng(11)
    ys, yc = rng.random((5, 16, 2)).astype(np.float32), rng.random((5, 32, 2)).astype(np.float32)
    pc = rng.uniform(0.5, 3.0, (5, 32, 1)).astype(np.float32)
    ref = O.context_aggregation(torch.from_numpy(ys), torch.from_numpy(yc), torch.from_numpy(pc), 16, 2, 0.8)
    d = lambda a: torch.from_numpy(a).to(cuda)
    _close(F.context_aggregation(d(ys), d(yc), d(pc), 2, 0.8), ref, atol=2e-7, rtol=1e-6, what='agg')


def test_pose3d_pieces(hip_lib, cuda):
    from deephar_amd import functional as F
    rng = np.random.default_rng(12)
    D, J = 16, 17
    h = _rand(rng, (3, 32, 32, D * J), 3.0)
    t = torch.from_numpy(h)
    h5 = t.reshape(3, 32, 32, D, J)
    hxy, hz = F.depth_means(t.to(cuda), D, J)
    _close(hxy, h5.mean(dim=3), atol=2e-6, what='hxy')
    _close(hz, h5.mean(dim=(1, 2)), atol=2e-6, what='hz')
    z, vz = F.softargmax1d(hz)
    ref_z = O.softargmax1d(h5.double().mean(dim=(1, 2)))
    assert (z.cpu().double() - ref_z).abs().max().item() <= 3.9e-6
    _close(vz, torch.amax(h5.mean(dim=(1, 2)), dim=1), atol=2e-6, what='vz')


@pytest.mark.parametrize('hw,D,J', [((32, 32), 16, 17), ((32, 32), 8, 20), ((30, 30), 16, 17), ((7, 9), 4, 5)])
def test_depth_means_paths_share_one_summation_order(hip_lib, cuda, hw, D, J):
    """reception.py:193-222: both volume means.  The one-pass kernel (>= 96 frames) and the two small-batch kernels sum
    in the same stated order: a frame's result does not depend on the batch it is in (bitwise), and both sit within a few
    ulp of the fp64 mean (the round-4 one-pass kernel ran 256 serial additions per accumulator)."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(120 + D + J)
    h = _rand(rng, (4,) + hw + (D * J,), 30.0) + 100.0           # large common offset: rounding of long sums shows
    t = torch.from_numpy(h)
    small = [a.cpu() for a in F.depth_means(t.to(cuda), D, J)]
    big = [a.cpu() for a in F.depth_means(t.repeat(32, 1, 1, 1).to(cuda), D, J)]     # 128 frames: one-pass kernel
    for a, b, what in zip(small, big, ('hxy', 'hz')):
        for r in range(32):
            assert torch.equal(a, b[4 * r:4 * r + 4]), what
    h5 = t.double().reshape(4, hw[0], hw[1], D, J)
    assert (small[0].double() - h5.mean(dim=3)).abs().max().item() <= 4 * 130 * 2.0 ** -24
    assert (small[1].double() - h5.mean(dim=(1, 2))).abs().max().item() <= 3 * 130 * 2.0 ** -24    # numpy emulation: 1.4 (new order), 5.5 (round-4 order)


def test_kronecker_and_action_top(hip_lib, cuda):
    from deephar_amd import functional as F
    rng = np.random.default_rng(13)
    hm = rng.random((6, 32, 32, 16)).astype(np.float32)
    hm /= hm.sum(axis=(1, 2), keepdims=True)
    x = _rand(rng, (6, 32, 32, 576))
    ref = O.kronecker_prod(torch.from_numpy(hm).double(), torch.from_numpy(x).double())
    got = F.kronecker(torch.from_numpy(hm).to(cuda), torch.from_numpy(x).to(cuda))
    _close(got.cpu().double(), ref, atol=2e-6, rtol=1e-5, what='kron')
    a = _rand(rng, (4, 8, 10, 60), 2.0)
    ref = torch.softmax(O.global_max_min_pooling(torch.from_numpy(a)), dim=-1)
    _close(F.global_maxmin_softmax(torch.from_numpy(a).to(cuda)), ref, atol=1e-8, rtol=1e-5, what='action_top')
    ref = O.global_max_min_pooling(torch.from_numpy(a))
    assert torch.equal(F.global_maxmin_softmax(torch.from_numpy(a).to(cuda), softmax=False).cpu(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('b,h,w,j,c', [(3, 32, 32, 17, 288), (2, 16, 16, 20, 384), (2, 8, 8, 25, 60), (2, 5, 3, 16, 64),
                                       (2, 9, 7, 41, 132), (2, 8, 8, 16, 30)])
def test_kronecker_shapes(b, h, w, j, c, hip_lib, cuda):
    """layers.kronecker_prod (layers.py:478-508) on the tiled kernel: joints that are no multiple of 4 (NTU: 17 + ...),
    more joints than one pass holds, channel slabs with a tail, fewer pixels than pixel groups; C % 4 != 0 takes the
    one-channel-per-thread kernel."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(b + h + w + j + c)
    hm = rng.random((b, h, w, j)).astype(np.float32)
    hm /= hm.sum(axis=(1, 2), keepdims=True)
    x = _rand(rng, (b, h, w, c))
    ref = O.kronecker_prod(torch.from_numpy(hm).double(), torch.from_numpy(x).double())
    got = F.kronecker(torch.from_numpy(hm).to(cuda), torch.from_numpy(x).to(cuda))
    _close(got.cpu().double(), ref, atol=2e-6, rtol=1e-5, what='kron %s' % ((b, h, w, j, c),))
    packed = F.kronecker(torch.from_numpy(hm).to(cuda), torch.from_numpy(x).to(cuda), out_pitch=c + 3)   # odd pitch
    assert torch.equal(packed.contiguous(), got)


@pytest.mark.gpu
@pytest.mark.parametrize('k,s,cout,power', [(3, 2, 32, 1), (7, 2, 64, 1), (3, 1, 16, (1, 2, 0.5))])
def test_conv2d_uint8_frames_normalised_on_load(k, s, cout, power, hip_lib, cuda):
    """dh_conv_args.x_u8: the first convolution reads raw uint8 frames and applies the loader's normalisation
    (transform.py:212-231) per byte -- bit-identical to normalising first and convolving the fp32 frames, and
    to the stand-alone dh_normalize_u8_f32."""
    from deephar_amd import functional as F
    from deephar_amd.engine.executor import normalization_lut
    rng = np.random.default_rng(k * 10 + s)
    frames = rng.integers(0, 256, (3, 37, 41, 3), dtype=np.uint8)
    frames[0, :2] = 0
    frames[1, -2:] = 255
    w = _rand(rng, (k, k, 3, cout), 0.3)
    lut_h = normalization_lut(3, power)
    lut = torch.from_numpy(lut_h).cuda()
    xb = torch.from_numpy(frames).cuda()
    xf = F.normalize_u8(xb, lut)
    ref = lut_h[np.arange(3)[None, None, None, :], frames]
    assert np.array_equal(xf.cpu().numpy(), ref)
    post = torch.from_numpy(_rand(rng, (cout,), 1.0)).cuda()
    for cfg in (-1, 3, 6, 8):
        y8 = F.conv2d(xb, w, strides=(s, s), in_lut=lut, post_scale=post, post_shift=post, post_relu=True, tile_cfg=cfg)
        yf = F.conv2d(xf, w, strides=(s, s), post_scale=post, post_shift=post, post_relu=True, tile_cfg=cfg)
        assert torch.equal(y8, yf), cfg
    want = O.conv2d(torch.from_numpy(ref.astype(np.float64)), torch.from_numpy(w.astype(np.float64)), (s, s), 'same')
    got = F.conv2d(xb, w, strides=(s, s), in_lut=lut)
    _close(got, want, 2e-5, what='u8 conv vs fp64 oracle')
    with pytest.raises(Exception):
        F.conv2d(xb, w, strides=(s, s), in_lut=lut, tile_cfg=17)   # DMA GEMM: no u8


@pytest.mark.gpu
@pytest.mark.parametrize('case', [
    # (every output map has more than 256 positions: smaller ones belong to the skinny-conv kernel by rule [r05])
    (2, 33, 31, 32, 64, 3, 3, 1, 'same'), (1, 40, 40, 64, 96, 3, 3, 2, 'same'), (2, 20, 24, 64, 64, 5, 1, 1, 'same'),
    (2, 20, 24, 64, 64, 1, 5, 1, 'same'), (1, 20, 20, 32, 32, 3, 3, 1, 'valid'), (2, 35, 37, 96, 72, 5, 5, 2, 'same'),
    (1, 34, 34, 192, 192, 3, 3, 2, 'same'), (2, 33, 33, 160, 64, 1, 1, 2, 'same')])
@pytest.mark.parametrize('cfg', [9, 11, 12, 13, 15, 17])
def test_kxk_conv_on_the_dma_kernel(case, cfg, hip_lib, cuda):
    """K x K / strided / TF-SAME convolutions with Cin % 32 == 0 on the LDS-DMA kernel (taps in the zero padding
    are fed from a page of zeros): bit-identical to the register-staged general kernel, with the fused ReLU
    prologue + BN + residual epilogue, on every tiling."""
    from deephar_amd import functional as F
    n, h, w, cin, cout, kh, kw, s, pad = case
    rng = np.random.default_rng(sum(v for v in case if isinstance(v, int)))
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (kh, kw, cin, cout), np.sqrt(1.0 / (kh * kw * cin)))
    d = lambda a: torch.from_numpy(a).to(cuda)
    base = F.conv2d(d(x), k, (s, s), pad, pre_relu=True, tile_cfg=8)
    r1 = _rand(rng, tuple(base.shape))
    qs, qb = _rand(rng, (cout,)), _rand(rng, (cout,))
    kw_ = dict(pre_relu=True, post_scale=d(qs), post_shift=d(qb), res1=d(r1), post_relu=True)
    a = F.conv2d(d(x), k, (s, s), pad, tile_cfg=cfg, **kw_)
    b = F.conv2d(d(x), k, (s, s), pad, tile_cfg=cfg - 9, **kw_)
    assert torch.equal(a, b)
    ref = O.conv2d(O.relu(torch.from_numpy(x).double()), torch.from_numpy(k).double(), (s, s), pad)
    ref = torch.relu(ref * torch.from_numpy(qs).double() + torch.from_numpy(qb).double() + torch.from_numpy(r1).double())
    _close(a, ref, atol=5e-5, what='kxk dma conv')


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 32, 32, 576, 576, False), (3, 16, 16, 288, 288, False), (2, 32, 32, 64, 100, True),
                                  (1, 8, 8, 48, 40, False),
                                  # the direct epilogue's index split: rows wider than a wave's 32 pixels, a wide low map,
                                  # and maps that are not 2^a x 2^b (those stay on the staged epilogue)
                                  (1, 64, 64, 32, 64, False), (2, 8, 128, 32, 96, True), (1, 12, 24, 32, 64, False),
                                  (2, 4, 8, 64, 32, False)])
def test_conv2d_second_residual_at_half_resolution(case, hip_lib, cuda):
    """dh_conv_args.res2_down: out = BN(conv(x)) + res1 + UpSampling2D(res2) with res2 at half resolution -- the
    hourglass' add([a, UpSampling2D(b)]) (reception.py:122-127) folded into the convolution that produces a.  Equal, bit
    for bit, to adding the explicitly up-sampled tensor as a full-resolution residual, on every tiling and in bf16x3."""
    from deephar_amd import functional as F
    n, h, w, cin, cout, relu = case
    rng = np.random.default_rng(sum(int(v) for v in case))
    x, k = _rand(rng, (n, h, w, cin)), _rand(rng, (1, 1, cin, cout), np.sqrt(1.0 / cin))
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), _rand(rng, (cout,), 0.1)
    r1, r2 = _rand(rng, (n, h, w, cout)), _rand(rng, (n, h // 2, w // 2, cout))
    up = np.repeat(np.repeat(r2, 2, axis=1), 2, axis=2)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    kw = dict(pre_relu=relu, post_scale=d(sc), post_shift=d(sh), res1=d(r1))
    from deephar_amd.engine.planner import split_k_rule
    skinny = split_k_rule(h * w, cin, cout, cin)      # fp32 layers on the skinny-conv kernel have no half-resolution residual
    for split in (False, True):
        if split and cout % 4:
            continue
        ncfg = hip_lib.dh_conv2d_num_split_tile_cfgs() if split else hip_lib.dh_conv2d_num_tile_cfgs()
        done = 0
        for cfg in range(-1, ncfg):
            try:
                got = F.conv2d(d(x), k, res2=d(r2), res2_down=True, tile_cfg=cfg, split=split, **kw)
                want = F.conv2d(d(x), k, res2=d(up), tile_cfg=cfg, split=split, **kw)
            except Exception as e:
                assert 'rc=-2' in str(e), e
                continue
            assert torch.equal(got, want), (split, cfg)
            done += 1
        # (the planner never asks a skinny layer for it: test_planner_r3_spares_split_k_producers; the library says
        #  DH_EUNSUPPORTED on every tiling)
        assert done == 0 if skinny else done >= 4      # (a skinny layer stays on the fp32 skinny kernel in bf16x3 mode too)
    xin = torch.from_numpy(x).double()
    ref = O.conv2d(O.relu(xin) if relu else xin, torch.from_numpy(k).double(), (1, 1), 'same')
    ref = ref * torch.from_numpy(sc).double() + torch.from_numpy(sh).double() + torch.from_numpy(r1).double() + \
        torch.from_numpy(up).double()
    if not skinny:
        _close(F.conv2d(d(x), k, res2=d(r2), res2_down=True, **kw), ref, atol=5e-5, what='res2_down')
    with pytest.raises(Exception):
        F.conv2d(d(x), k, res2=d(r2), res2_down=True, up2=True, **kw)


@pytest.mark.parametrize('case', [
    # (frames, logical H, W of the conv's input, Cin, Cout, k, mode, BN prologue, ReLU prologue)
    (16, 8, 8, 15, 160, 3, 1, False, True),       # action head conv3 on relu(UpSampling2D(class maps)): Cin = 15 (dword loads)
    (16, 4, 4, 160, 15, 3, 3, True, True),        # action head conv2h on relu(BN(max_min_pooling(x1)))
    (3, 8, 8, 64, 48, 3, 2, True, False),         # plain max-pooling on load
    (2, 16, 16, 96, 200, 1, 1, False, False),     # 1x1, up-sampled input, no prologue
    (5, 6, 10, 20, 33, 5, 3, False, True),        # 5x5, ragged everything
    (2, 16, 16, 24, 64, 3, 2, False, True),
])
def test_skinny_conv_resamples_on_load(case, hip_lib, cuda):
    """[r06] dh_conv_args.x_resample (planner rule R12): the skinny-conv kernel reads a half-resolution tensor as if
    up-sampled (1) or a double-resolution tensor through a 2x2 max (2) / max+-min (3) pooling -- bit for bit the convolution
    of the tensor the stand-alone up-sampling / pooling launch writes; BN / ReLU prologue and zero padding act on the
    resampled pixels; other kernel families refuse the flag."""
    from deephar_amd import functional as F
    n, h, w, cin, cout, k, mode, bn, relu = case
    rng = np.random.default_rng(sum(int(v) for v in case))
    phys = (n, h // 2, w // 2, cin) if mode == 1 else (n, 2 * h, 2 * w, cin)
    x = _rand(rng, phys)
    kern = _rand(rng, (k, k, cin, cout), np.sqrt(1.0 / (k * k * cin)))
    ps, pb = (rng.uniform(0.5, 1.5, cin).astype(np.float32), _rand(rng, (cin,), 0.4)) if bn else (None, None)
    r1 = _rand(rng, (n, h, w, cout))
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    if mode == 1:
        explicit = d(np.repeat(np.repeat(x, 2, axis=1), 2, axis=2))
    else:
        explicit = F.pool2d(d(x), (2, 2), (2, 2), 'valid', mode=1 if mode == 3 else 0)
    kw = dict(pre_scale=d(ps), pre_shift=d(pb), pre_relu=relu, res1=d(r1))
    want = F.conv2d(explicit, kern, **kw)
    got = F.conv2d(d(x), kern, x_resample=mode, **kw)
    torch.cuda.synchronize()
    assert got.shape == want.shape and torch.equal(got, want)
    t = lambda a: torch.from_numpy(a).double()
    xin = explicit.cpu().double()
    if bn:
        xin = xin * t(ps) + t(pb)
    if relu:
        xin = O.relu(xin)
    _close(got, O.conv2d(xin, t(kern), (1, 1), 'same') + t(r1), atol=3e-5, what='resample on load')
    half = n // 2
    if half:
        assert torch.equal(F.conv2d(d(x[:half]), kern, x_resample=mode, **dict(kw, res1=d(r1[:half]))), got[:half])


def test_resampling_on_load_is_the_skinny_kernels_only(hip_lib, cuda):
    from deephar_amd import functional as F
    x = torch.randn(1, 32, 32, 64, device=cuda)                    # a 64 x 64 map: not a skinny layer
    with pytest.raises(Exception, match='rc=-2'):
        F.conv2d(x, np.zeros((3, 3, 64, 64), np.float32), x_resample=1)


# ---- halo-resident K x K kernel (dh_conv_args.w_split = 2, conv_halo.hip) ------------------------------------------------
HALO_CASES = [
    # n, h, w, cin, cout, kh, kw, relu prologue, residual
    (2, 128, 128, 32, 64, 3, 3, False, False),      # ReceptionNet stem: conv_bn_act(x, 64, (3, 3)) at 128 x 128
    (1, 128, 128, 32, 32, 3, 3, False, True),       # ... and its 32-channel sibling; one tile = one image row
    (2, 64, 64, 64, 96, 3, 3, False, False),        # stem branch a / b: conv_bn(., 96, (3, 3)); tile = 2 rows
    (2, 64, 64, 64, 64, 5, 1, False, True),         # stem: (5, 1) and (1, 5)
    (2, 64, 64, 64, 64, 1, 5, True, False),
    (1, 32, 32, 144, 288, 3, 3, True, True),        # SPNet res3 / res4 conv2: Cin = 144 = 9 chunks, tile = 4 rows
    (1, 128, 128, 48, 96, 3, 3, True, False),       # SPNet res0 conv2: Cin = 48
    (3, 32, 32, 32, 40, 3, 3, False, True),         # ragged Cout
    (1, 16, 256, 32, 64, 3, 3, True, False),        # rows wider than a tile: two 128-column runs per row
    (2, 32, 32, 64, 96, 5, 5, True, True),          # 5 x 5
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', HALO_CASES)
def test_conv2d_halo_kernel(case, hip_lib, cuda):
    """Dense K x K convolutions with the input halo tile resident in LDS (chunk-major K order, w_split = 2): against the
    fp64 oracle with the fused ReLU prologue / BN / residual / ReLU epilogue; its three tilings bit-identical; a frame's
    result does not depend on the batch it sits in; the library's eligibility rule takes exactly these layers."""
    import ctypes as C
    from deephar_amd import _lib, functional as F
    n, h, w, cin, cout, kh, kw, relu, res = case
    rng = np.random.default_rng(sum(int(v) for v in case))
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (kh, kw, cin, cout), np.sqrt(1.0 / (kh * kw * cin)))
    qs, qb = rng.uniform(0.5, 1.5, cout).astype(np.float32), _rand(rng, (cout,), 0.1)
    r1 = _rand(rng, (n, h, w, cout)) if res else None
    d = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    kw_ = dict(pre_relu=relu, post_scale=d(qs), post_shift=d(qb), res1=d(r1), post_relu=True)
    outs = []
    for cfg in range(-1, hip_lib.dh_conv2d_num_halo_tile_cfgs()):
        try:
            outs.append(F.conv2d(d(x), k, halo=True, tile_cfg=cfg, **kw_))
        except Exception as e:                   # a tiling whose stages do not fit the LDS budget (5 x 5 with 96 columns)
            assert cfg >= 0 and 'rc=-2' in str(e), e
    torch.cuda.synchronize()
    assert len(outs) >= 3
    for y in outs[1:]:
        assert torch.equal(y, outs[0])
    xin = torch.from_numpy(x).double()
    ref = O.conv2d(O.relu(xin) if relu else xin, torch.from_numpy(k).double(), (1, 1), 'same')
    ref = ref * torch.from_numpy(qs).double() + torch.from_numpy(qb).double()
    if res:
        ref = ref + torch.from_numpy(r1).double()
    _close(outs[0], torch.relu(ref), atol=5e-5, what='halo conv %s' % (case,))
    # frames are independent: the last frame alone gives the same bits
    one = F.conv2d(d(x[-1:]), k, halo=True, **dict(kw_, res1=d(None if r1 is None else r1[-1:])))
    assert torch.equal(one, outs[0][-1:])
    # the tap-major kernels agree to fp32 rounding (different K order: not the same bits)
    tap = F.conv2d(d(x), k, **kw_)
    assert float((tap - outs[0]).abs().max()) < 5e-5
    a = _lib.ConvArgs()
    a.x = 4096
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, w, cin, cin, h, w, cout, cout
    a.KH, a.KW, a.SH, a.SW, a.PT, a.PL, a.K = kh, kw, 1, 1, (kh - 1) // 2, (kw - 1) // 2, kh * kw * cin
    want = int(cin % 32 != 0)                    # the binding rule: only what the tap-major DMA kernel cannot take
    assert hip_lib.dh_conv2d_halo_eligible(C.byref(a)) == want
    a.N = 1                                      # the rule never looks at the batch size
    assert hip_lib.dh_conv2d_halo_eligible(C.byref(a)) == want
    a.SH = a.SW = 2
    assert hip_lib.dh_conv2d_halo_eligible(C.byref(a)) == 0


@pytest.mark.gpu
def test_conv2d_halo_rejects_other_layers(hip_lib, cuda):
    """w_split = 2 on a layer outside the rule is refused (DH_EUNSUPPORTED), never run on another kernel."""
    from deephar_amd import functional as F
    rng = np.random.default_rng(3)
    d = lambda a: torch.from_numpy(a).to(cuda)
    for shape, k, strides in (((1, 16, 16, 32), (3, 3, 32, 32), (1, 1)),       # small map
                              ((1, 64, 64, 32), (3, 3, 32, 32), (2, 2)),       # strided
                              ((1, 64, 64, 32), (1, 1, 32, 32), (1, 1)),       # pointwise
                              ((1, 64, 48, 32), (3, 3, 32, 32), (1, 1))):      # rows that do not tile into 128 pixels
        with pytest.raises(Exception) as e:
            F.conv2d(d(_rand(rng, shape)), _rand(rng, k), strides, halo=True)
        assert 'rc=-2' in str(e.value), e.value
    with pytest.raises(ValueError):
        F.conv2d(d(_rand(rng, (1, 64, 64, 24))), _rand(rng, (3, 3, 24, 32)), halo=True)     # Cin % 16 != 0: no such packing


# ---- split-bf16 GEMM (dh_conv_args.w_split, gemm1x1s.hip) --------------------------------------------------------------
SPLIT_CASES = [
    # (N, H, W, Cin, Cout, k, stride, relu, residual, up2)
    (2, 32, 32, 576, 576, 1, 1, True, True, False),      # the dominant pointwise GEMM
    (3, 16, 16, 288, 288, 1, 1, False, True, False),
    (2, 32, 32, 576, 48, 1, 1, True, False, False),      # RegMap: ragged Cout
    (2, 32, 32, 48, 576, 1, 1, True, False, False),      # fReMap: K = 48 (padded to 64)
    (2, 16, 16, 288, 576, 1, 1, False, True, True),      # fused up-sampling epilogue
    (2, 33, 31, 64, 96, 3, 1, True, False, False),       # K x K through the zero-page DMA, ragged M
    (1, 32, 32, 64, 288, 3, 2, False, False, False),     # strided (288 output channels: a 16 x 16 map with <= 256 is skinny)
    (2, 19, 23, 96, 200, 1, 1, True, True, False),       # ragged everywhere
]


@pytest.mark.parametrize('case', [(3, 32, 32, 96, 200, 1, True, True, False), (2, 32, 32, 48, 576, 1, True, False, True),
                                  (2, 64, 64, 64, 96, 3, False, True, False), (1, 6, 32, 288, 288, 1, False, False, False),
                                  (2, 32, 32, 64, 100, 3, True, True, False)])
def test_conv2d_pooled_second_output(case, hip_lib, cuda):
    """dh_conv_args.y_pool: the epilogue also writes MaxPooling2D((2, 2)) of the final output.  Equal to pooling the
    first output with the stand-alone kernel, for every tiling that takes it (fp32 general + DMA GEMM, split-bf16 incl.
    its wide tiling, which runs the epilogue twice through one slab), with BN / residual / ReLU epilogues, two residuals, strided K x K producers and ragged channel tiles;
    tilings without a wave pair, other widths and the up-sampling epilogue refuse it."""
    from deephar_amd import functional as F
    n, h, w, cin, cout, ks, relu, res, res2 = case
    st = 2 if h == 64 else 1                                        # 64 x 64 input, stride 2 -> 32 x 32 output
    rng = np.random.default_rng(sum(int(v) for v in case))
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (ks, ks, cin, cout), np.sqrt(1.0 / (ks * ks * cin)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = _rand(rng, (cout,), 0.1)
    oh, ow = h // st, w // st
    d = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    r1 = _rand(rng, (n, oh, ow, cout)) if res else None
    kw = dict(strides=(st, st), padding='same', pre_relu=relu, post_scale=d(sc), post_shift=d(sh), res1=d(r1), post_relu=res2)
    base = F.conv2d(d(x), k, **kw)
    ref_pool = F.pool2d(base, (2, 2))
    took = {}
    for split in (False, True):
        ncfg = hip_lib.dh_conv2d_num_split_tile_cfgs() if split else hip_lib.dh_conv2d_num_tile_cfgs()
        ref_y = F.conv2d(d(x), k, split=split, **kw)
        for cfg in range(-1, ncfg):
            try:
                y, yp = F.conv2d(d(x), k, split=split, tile_cfg=cfg, pool2=True, **kw)
            except Exception as e:
                assert 'rc=-2' in str(e), e
                continue
            took[(split, cfg)] = True
            assert torch.equal(y, ref_y), (split, cfg)
            assert torch.equal(yp, F.pool2d(y, (2, 2))), (split, cfg)
            if not split:
                assert torch.equal(yp, ref_pool)
    assert (False, -1) in took and (False, 11) in took and (False, 13) in took and (False, 2) in took
    assert (False, 8) not in took and (False, 17) not in took and (False, 0) not in took      # no wave pair / 64-row waves
    if cin % 32 == 0 or ks == 1:
        assert (True, 14) in took and (True, 2) in took
    with pytest.raises(Exception):                                                          # 12 columns: not built
        F.conv2d(d(np.ascontiguousarray(x[:, :, :12 * st])), k, pool2=True, **dict(kw, res1=None))


@pytest.mark.parametrize('case', [(3, 16, 16, 32, 384, 1, True, True, True), (2, 8, 8, 480, 480, 1, False, False, False),
                                  (2, 16, 16, 96, 320, 3, True, True, False), (5, 8, 8, 64, 300, 1, True, True, False),
                                  (1, 4, 8, 288, 288, 1, False, True, False), (2, 32, 16, 64, 96, 3, False, True, True)])
def test_conv2d_pooled_second_output_small_maps(case, hip_lib, cuda):
    """[r06] dh_conv_args.y_pool at OW = 16 / OW = 8 (OH * OW a multiple of 32): a wave's 32 output rows are two / four whole
    image rows and the wave pools its own slab -- every tiling with 32-row waves takes it, incl. the one-wave 32 x 32 tiling
    the latency regime runs (SPNet's down path below 32 x 32: common.py:70-86 after the prediction block's conv2 with its two
    residuals).  Equal to the stand-alone pooling of the first output, which is unchanged; fp32 and split-bf16."""
    from deephar_amd import functional as F
    n, h, w, cin, cout, ks, relu, res, res2 = case
    rng = np.random.default_rng(7 + sum(int(v) for v in case))
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (ks, ks, cin, cout), np.sqrt(1.0 / (ks * ks * cin)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = _rand(rng, (cout,), 0.1)
    d = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    r1 = _rand(rng, (n, h, w, cout)) if res else None
    r2 = _rand(rng, (n, h, w, cout)) if res2 else None
    kw = dict(padding='same', pre_relu=relu, post_scale=d(sc), post_shift=d(sh), res1=d(r1), res2=d(r2), post_relu=not res2)
    base = F.conv2d(d(x), k, **kw)
    ref_pool = F.pool2d(base, (2, 2))
    took = {}
    for split in (False, True):
        ncfg = hip_lib.dh_conv2d_num_split_tile_cfgs() if split else hip_lib.dh_conv2d_num_tile_cfgs()
        ref_y = F.conv2d(d(x), k, split=split, **kw)
        for cfg in range(-1, ncfg):
            try:
                y, yp = F.conv2d(d(x), k, split=split, tile_cfg=cfg, pool2=True, **kw)
            except Exception as e:
                assert 'rc=-2' in str(e), e
                continue
            took[(split, cfg)] = True
            assert yp.shape == (n, h // 2, w // 2, cout)
            assert torch.equal(y, ref_y), (split, cfg)
            assert torch.equal(yp, F.pool2d(y, (2, 2))), (split, cfg)
            if not split:
                assert torch.equal(yp, ref_pool)
    # the library's own pick, the one-wave tiling (no partner needed here) and the paired ones; 64-row waves still refuse
    assert (False, -1) in took and (False, 8) in took and (False, 2) in took and (False, 0) not in took
    if ks == 1:
        assert (False, 17) in took and (False, 13) in took
    if cin % 32 == 0 or ks == 1:
        assert (True, -1) in took


SKINNY_CASES = [
    # n, h, w, cin, cout, k, stride, bn prologue, relu, residual
    (4, 16, 16, 256, 256, 3, 1, False, True, False),     # merge action head: 3x3 over [T x J], K = 2304
    (4, 16, 16, 112, 15, 3, 1, True, True, False),       # K = 1008, Cout = 15 (ragged tile), BN + ReLU prologue
    (3, 32, 20, 192, 192, 3, 2, False, True, True),      # strided (out 16 x 10), residual
    (2, 4, 4, 256, 15, 3, 1, False, False, False),       # 16 output pixels per clip: half-empty row tile
    (5, 8, 8, 1024, 60, 1, 1, False, True, False),       # pointwise with a long K
    # [r05] the rule reaches down to K = 64 and takes any Cin: the action heads of SPNet at T = 8 (spnet.py:51-148)
    (2, 8, 16, 70, 160, 1, 1, True, True, False),        # K = 70: Cin % 4 != 0 -> dword loads, 4 waves, BN + ReLU prologue
    (2, 8, 8, 15, 160, 3, 1, False, True, True),         # K = 135 = 3 x 3 x 15: k-groups straddle taps
    (2, 8, 16, 80, 160, 3, 1, False, False, True),       # K = 720: 16 waves, one chunk each
    (2, 8, 8, 480, 40, 1, 1, True, True, False),         # K = 480: 8 waves
    (3, 16, 16, 384, 16, 1, 1, True, True, False),       # a heat-map head of the 16 x 16 level: Cout = 16
    (2, 8, 8, 160, 160, 3, 1, True, True, False),        # K = 1440: two chunks per wave, prologue + zero padding
    (1, 4, 4, 576, 16, 1, 1, True, True, False),         # 16 positions in all: half a row tile
]


@pytest.mark.parametrize('case', SKINNY_CASES)
def test_conv2d_split_k_kernel(case, hip_lib, cuda):
    """Tiny per-frame output + long reduction -> conv_splitk.hip, chosen by a rule on the layer's geometry: against the
    fp64 truth, identical for every tile_cfg, batch-size invariant, identical through an unaligned input view."""
    import ctypes as C
    from deephar_amd import functional as F, _lib
    n, h, w, cin, cout, ks, st, bn, relu, res = case
    rng = np.random.default_rng(sum(int(v) for v in case))
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (ks, ks, cin, cout), np.sqrt(1.0 / (ks * ks * cin)))
    ps = rng.uniform(0.5, 1.5, cin).astype(np.float32) if bn else None
    pb = _rand(rng, (cin,), 0.1) if bn else None
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = _rand(rng, (cout,), 0.1)
    oh, ow = -(-h // st), -(-w // st)
    r1 = _rand(rng, (n, oh, ow, cout)) if res else None
    t = lambda a: torch.from_numpy(a).double()
    xin = t(x) * t(ps) + t(pb) if bn else t(x)
    xin = O.relu(xin) if relu else xin
    ref = O.conv2d(xin, t(k), (st, st), 'same') * t(sc) + t(sh)
    if res:
        ref = ref + t(r1)
    xin32 = torch.from_numpy(x) * torch.from_numpy(ps) + torch.from_numpy(pb) if bn else torch.from_numpy(x)
    xin32 = O.relu(xin32) if relu else xin32
    cpu32 = O.conv2d(xin32, torch.from_numpy(k), (st, st), 'same') * torch.from_numpy(sc) + torch.from_numpy(sh)
    if res:
        cpu32 = cpu32 + torch.from_numpy(r1)
    d = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    kw = dict(strides=(st, st), padding='same', pre_scale=d(ps), pre_shift=d(pb), pre_relu=relu, post_scale=d(sc),
              post_shift=d(sh))
    got = F.conv2d(d(x), k, res1=d(r1), **kw)
    torch.cuda.synchronize()
    e_hip = (got.cpu().double() - ref).abs().max().item()
    e_cpu = (cpu32.double() - ref).abs().max().item()
    assert e_hip <= 4 * e_cpu + 1e-6, (e_hip, e_cpu)
    for cfg in (0, 8, 12, hip_lib.dh_conv2d_num_tile_cfgs() - 1):           # the rule overrides the tiling
        assert torch.equal(F.conv2d(d(x), k, res1=d(r1), tile_cfg=cfg, **kw), got)
    half = F.conv2d(d(x[:2]), k, res1=d(None if r1 is None else r1[:2]), **kw)      # batch-size invariant
    assert torch.equal(half, got[:2])
    # the rule itself, and an input view that is not 16-byte aligned (pitch cin + 1, offset 1): same bits
    a = _lib.ConvArgs()
    wt, kp, np_ = F.pack_conv_weight(k, cuda)
    wide = torch.zeros(n, h, w, cin + 1, device=cuda)
    wide[..., 1:] = d(x)
    y = torch.empty_like(got)
    keep = [d(ps), d(pb), d(sc), d(sh), d(r1)]
    a.x, a.w, a.y = wide.data_ptr() + 4, wt.data_ptr(), y.data_ptr()
    a.pre_scale, a.pre_shift = (keep[0].data_ptr(), keep[1].data_ptr()) if bn else (None, None)
    a.post_scale, a.post_shift = keep[2].data_ptr(), keep[3].data_ptr()
    a.res1 = keep[4].data_ptr() if res else None
    a.N, a.H, a.W, a.Cin, a.ldx, a.OH, a.OW, a.Cout, a.ldy = n, h, w, cin, cin + 1, oh, ow, cout, cout
    a.KH = a.KW = ks; a.SH = a.SW = st
    a.PT = max((oh - 1) * st + ks - h, 0) // 2; a.PL = max((ow - 1) * st + ks - w, 0) // 2
    a.K, a.Kp, a.Np, a.ldr1, a.pre_relu = ks * ks * cin, kp, np_, cout, int(relu)
    assert hip_lib.dh_conv2d_uses_split_k(C.byref(a)) == 1
    assert hip_lib.dh_conv2d_f32(C.byref(a), -1, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, got)
    a.K = 288; a.Cin = 288; a.Cout = 288; a.KH = a.KW = 1                     # an MPII 16 x 16 pointwise conv: not skinny
    assert hip_lib.dh_conv2d_uses_split_k(C.byref(a)) == 0                   # (more than 256 output channels)


@pytest.mark.parametrize('case', SPLIT_CASES)
def test_conv2d_dma_gemm_tilings_bitwise(case, hip_lib, cuda):
    """The fp32 LDS-DMA GEMM family (cfg 9..17) on the same shapes: pointwise, K x K, strided, fused up-sampling, ragged
    tails -- all tilings bit-identical."""
    from deephar_amd import functional as F
    n, h, w, cin, cout, ks, st, relu, res, up2 = case
    rng = np.random.default_rng(sum(int(v) for v in case) + 1)
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (ks, ks, cin, cout), np.sqrt(1.0 / (ks * ks * cin)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = _rand(rng, (cout,), 0.1)
    oh, ow = -(-h // st), -(-w // st)
    r1 = _rand(rng, (n, oh, ow, cout)) if res else None
    r2 = _rand(rng, (n, 2 * oh, 2 * ow, cout)) if up2 else None
    d = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    kw = dict(strides=(st, st), padding='same', pre_relu=relu, post_scale=d(sc), post_shift=d(sh), res1=d(r1), res2=d(r2),
              up2=up2)
    outs = {}
    for cfg in range(9, hip_lib.dh_conv2d_num_tile_cfgs()):
        try:
            outs[cfg] = F.conv2d(d(x), k, tile_cfg=cfg, **kw)
        except Exception as e:
            assert 'rc=-2' in str(e), e
    torch.cuda.synchronize()
    assert len(outs) >= 4
    first = next(iter(outs.values()))
    for cfg, y in outs.items():
        assert torch.equal(y, first), 'DMA GEMM tiling %d differs' % cfg
    general = F.conv2d(d(x), k, tile_cfg=4, **kw)                  # conv_igemm_kernel, same K order
    assert torch.equal(general, first)


@pytest.mark.parametrize('case', [(64, 32, 32, 48, 576, True), (3, 32, 32, 64, 96, False), (2, 16, 16, 288, 288, True),
                                  (1, 30, 30, 64, 100, False)])
def test_three_way_add_every_tiling(case, hip_lib, cuda):
    """[r05] out = BN(conv(relu?(x))) + res1 + res2 with BOTH residuals at full resolution -- fReMap's re-injection,
    add([ident_map, x, h]) of reception.py:312 (an HBM-bound launch: K = 48, 97 us at 4.8 TB/s).  Same additions in the same
    order on every tiling of both fp32 families: bit-identical, and the fp64 truth near.  (Storing it straight from the
    accumulators like the one-residual launches was built and measured this round: 97.8 / 97.1 -> 98.7 / 97.2 us per launch,
    12.86 / 12.83 -> 12.89 / 12.87 ms per MPII step, same box -- the launch is bound by its bytes, not by its epilogue's
    form; not kept.)"""
    from deephar_amd import functional as F
    n, h, w, cin, cout, relu = case
    rng = np.random.default_rng(sum(int(v) for v in case) + 5)
    x, k = _rand(rng, (n, h, w, cin)), _rand(rng, (1, 1, cin, cout), np.sqrt(1.0 / cin))
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), _rand(rng, (cout,), 0.1)
    r1, r2 = _rand(rng, (n, h, w, cout)), _rand(rng, (n, h, w, cout))
    d = lambda a: torch.from_numpy(a).to(cuda)
    kw = dict(pre_relu=relu, post_scale=d(sc), post_shift=d(sh), res1=d(r1), res2=d(r2))
    outs = {}
    for cfg in range(0, hip_lib.dh_conv2d_num_tile_cfgs()):
        try:
            outs[cfg] = F.conv2d(d(x), k, tile_cfg=cfg, **kw)
        except Exception as e:
            assert 'rc=-2' in str(e), e
    assert len(outs) >= 12
    first = outs[4]                                                   # the general kernel (staged epilogue)
    for cfg, y in outs.items():
        assert torch.equal(y, first), 'tiling %d differs' % cfg
    xin = torch.from_numpy(x).double()
    ref = O.conv2d(O.relu(xin) if relu else xin, torch.from_numpy(k).double(), (1, 1), 'same')
    ref = ref * torch.from_numpy(sc).double() + torch.from_numpy(sh).double() + torch.from_numpy(r1).double() + \
        torch.from_numpy(r2).double()
    _close(first, ref, atol=5e-5, what='three-way add')


@pytest.mark.parametrize('case', SPLIT_CASES)
def test_conv2d_split_bf16(case, hip_lib, cuda):
    """Split-bf16 convolution: (a) as close to the fp64 truth as the fp32-MFMA path (the six-product split loses less
    than the fp32 accumulation does), (b) every tiling bit-identical to every other (same K order)."""
    from deephar_amd import functional as F
    n, h, w, cin, cout, ks, st, relu, res, up2 = case
    rng = np.random.default_rng(sum(int(v) for v in case))
    x = _rand(rng, (n, h, w, cin))
    k = _rand(rng, (ks, ks, cin, cout), np.sqrt(1.0 / (ks * ks * cin)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = _rand(rng, (cout,), 0.1)
    oh, ow = -(-h // st), -(-w // st)
    r1 = _rand(rng, (n, oh, ow, cout)) if res else None
    r2 = _rand(rng, (n, 2 * oh, 2 * ow, cout)) if up2 else None
    t = lambda a: torch.from_numpy(a).double()
    xin = O.relu(t(x)) if relu else t(x)
    ref = O.conv2d(xin, t(k), (st, st), 'same') * t(sc) + t(sh)
    if res:
        ref = ref + t(r1)
    if up2:
        ref = O.upsample2d(ref) + t(r2)
    d = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    kw = dict(strides=(st, st), padding='same', pre_relu=relu, post_scale=d(sc), post_shift=d(sh), res1=d(r1), res2=d(r2),
              up2=up2)
    f32 = F.conv2d(d(x), k, **kw)
    outs = {}
    for cfg in range(-1, hip_lib.dh_conv2d_num_split_tile_cfgs()):
        try:
            outs[cfg] = F.conv2d(d(x), k, split=True, tile_cfg=cfg, **kw)
        except Exception as e:
            assert 'rc=-2' in str(e), e
    torch.cuda.synchronize()
    assert len(outs) >= 3
    first = next(iter(outs.values()))
    for cfg, y in outs.items():
        assert torch.equal(y, first), 'split tiling %d differs' % cfg
    e_split = (first.cpu().double() - ref).abs().max().item()
    e_f32 = (f32.cpu().double() - ref).abs().max().item()
    print('case %s: |split - fp64| = %.3e   |fp32 mfma - fp64| = %.3e' % (case, e_split, e_f32))
    assert e_split <= 2.0 * e_f32 + 1e-6, (e_split, e_f32)


def test_conv2d_split_rejects_what_the_gemm_family_cannot_run(hip_lib, cuda):
    from deephar_amd import functional as F
    from deephar_amd._lib import DeepharHipError
    x = torch.randn(1, 16, 16, 3, device=cuda)                      # Cin = 3: general implicit-GEMM kernel only
    with pytest.raises(DeepharHipError):
        F.conv2d(x, np.zeros((3, 3, 3, 32), np.float32), split=True)
