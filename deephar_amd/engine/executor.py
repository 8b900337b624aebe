"""Bind a Plan to device memory for a batch size and run it through the C-ABI on one HIP stream.

PyTorch-ROCm is used for exactly three things: the activation arena / weight tensors (device memory),
the stream handle, and host<->device copies.  Every kernel is a libdeephar_hip.so entry point; there is no
torch compute and no CPU fallback.  The launch sequence of a bound plan is captured once into a hipGraph
(dh_graph_*), so steady-state `predict` costs one graph launch per batch instead of ~300 kernel launches.
"""
import ctypes as C
import os

import numpy as np

from .. import _lib
from . import packing

BN_EPS = 1e-3  # keras BatchNormalization default epsilon


def _torch():
    import torch
    return torch


class WeightStore:
    """Device copies of the model weights in kernel-ready layout; refreshed in place when Params change."""

    def __init__(self, device):
        self.device = device
        self.conv = {}    # id(param) -> (tensor, Kp, Np, version)
        self.dw = {}      # id(param) -> (tensor, version)
        self.bn = {}      # id(layer) -> (scale, shift, versions)
        self.const = {}   # key -> tensor

    def _dev(self, arr):
        torch = _torch()
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)

    @staticmethod
    def _require(p):
        if p.value is None:
            raise RuntimeError('weight %s is not set: call model.load_weights(...) or '
                               'deephar_amd.weights.init_synthetic(model) before predict' % p.key)
        return p.value

    def conv_weight(self, p, split=0):
        """split = dh_conv_args.w_split: 0 fp32 tap-major, 1 the split-bf16 packing, 2 fp32 chunk-major (halo-resident
        K x K kernel); different packings of one Param are kept side by side when several are in use."""
        key = (id(p), int(split))
        ent = self.conv.get(key)
        if ent is None or ent[3] != p.version:
            packer = {0: packing.pack_conv, 1: packing.pack_conv_split, 2: packing.pack_conv_halo}[int(split)]
            packed, kp, np_ = packer(self._require(p))
            if ent is None:
                ent = (self._dev(packed), kp, np_, p.version)
            else:
                ent[0].copy_(self._dev(packed))
                ent = (ent[0], kp, np_, p.version)
            self.conv[key] = ent
        return ent[0], ent[1], ent[2]

    def dw_weight(self, p):
        ent = self.dw.get(id(p))
        if ent is None or ent[1] != p.version:
            w = self._require(p)
            flat = np.ascontiguousarray(w.reshape(w.shape[0] * w.shape[1], w.shape[2]))
            if ent is None:
                ent = (self._dev(flat), p.version)
            else:
                ent[0].copy_(self._dev(flat))
                ent = (ent[0], p.version)
            self.dw[id(p)] = ent
        return ent[0]

    def _bn_host(self, layer):
        byrole = {p.role: self._require(p) for p in layer.params}
        inv = (np.float32(1.0) / np.sqrt(byrole['var'].astype(np.float32) + np.float32(BN_EPS))).astype(np.float32)
        if 'gamma' in byrole:
            inv = (inv * byrole['gamma']).astype(np.float32)
        return inv, (byrole['beta'] - byrole['mean'] * inv).astype(np.float32)

    def bn_affine(self, layer):
        """BatchNormalization inference as y = x*scale + shift (fp32, like tf.nn.batch_normalization).  A
        planner.ConcatAffine (the affine behind a horizontally merged convolution, rule R10c) gives the per-column scale /
        shift of its parts side by side, identity (1, 0) for a part without a BatchNormalization."""
        vers = tuple(p.version for p in layer.params)
        ent = self.bn.get(id(layer))
        if ent is None or ent[2] != vers:
            if hasattr(layer, 'parts'):
                pieces = [self._bn_host(l) if l is not None else (np.ones(c, np.float32), np.zeros(c, np.float32))
                          for c, l in layer.parts]
                inv = np.concatenate([a for a, _ in pieces]).astype(np.float32)
                shift = np.concatenate([b for _, b in pieces]).astype(np.float32)
            else:
                inv, shift = self._bn_host(layer)
            if ent is None:
                ent = (self._dev(inv), self._dev(shift), vers)
            else:
                ent[0].copy_(self._dev(inv))
                ent[1].copy_(self._dev(shift))
                ent = (ent[0], ent[1], vers)
            self.bn[id(layer)] = ent
        return ent[0], ent[1]

    def constant(self, key, make):
        t = self.const.get(key)
        if t is None:
            t = self._dev(make())
            self.const[key] = t
        return t


def grid_x(w):
    """float32 x grid of utils/math.py:6-19 (np.linspace(0,1,W) stored into a float32 array)."""
    return np.linspace(0.0, 1.0, num=w).astype(np.float32)


def normalization_lut(channels, channel_power=1):
    """utils/transform.normalize_channels (transform.py:212-231) tabulated for every byte value, per channel:
    the reference runs it in float32 in place (T.asarray() yields float32, transform.py:122-124), so the same
    NumPy float32 operations on 0..255 give bit-identical values.  Returns float32 [channels, 256]."""
    powers = [channel_power] * channels if isinstance(channel_power, int) else list(channel_power)
    if len(powers) != channels:
        raise ValueError('channel_power expected to be int or a list with one entry per channel')
    lut = np.empty((channels, 256), np.float32)
    for c, pw in enumerate(powers):
        v = np.arange(256, dtype=np.float32)
        v /= 255.
        if pw != 1:
            v = np.power(v, pw)
        v -= .5
        v *= 2.
        lut[c] = v
    return lut


def grid_depth(d):
    """layers.py:141-143: linspace(1/(2D), 1-1/(2D), D) stored in float32 Conv1D weights."""
    s = 1 / (2 * d)
    return np.linspace(s, 1 - s, num=d).astype(np.float32)


class _InputStep:
    """Pseudo-step for the stand-alone uint8 normalisation launch (first on stream 0, nothing to wait for)."""
    kind, name, stream, wait, record, deps, params = 'normalize_u8', 'input', 0, (), False, (), {}

    def __init__(self, v):
        self.outs = {'y': v}
        self.ins = {}
        self.attrs = {}

    def flops(self, n):
        return 0

    def bytes(self, n):
        v = self.outs['y']
        return 5 * n * int(np.prod(v.shape))


from .planner import split_k_rule      # noqa: E402  (the rule is stated once, next to the planner rule that uses it)


class BoundPlan:
    """A Plan with concrete device pointers for batch size n."""

    def __init__(self, plan, n, store, device, u8_norm=None):
        torch = _torch()
        self.plan, self.n, self.store, self.device = plan, n, store, device
        self.lib = _lib.load()
        self.arena = torch.empty(max(plan.arena_items * n, 4), dtype=torch.float32, device=device)
        self.base = self.arena.data_ptr()
        self.calls = []       # (fn, args tuple without stream, step)
        self._keep = []       # ctypes structs kept alive
        self.graph = None
        # test hook (schedule stress tests): called as perturb(call index, step, stream of the step, [all streams]) right
        # before every launch of launch_all -- e.g. to put dh_stream_spin_us delays on one stream
        self.perturb = None
        # uint8 frames: every model input gets a byte staging buffer; convolutions that read an input directly
        # normalise on load (dh_conv_args.x_u8), anything else is fed by a stand-alone normalisation launch
        self.u8 = None
        self.npre = 0         # launches in front of plan.steps[0] (step indices in `wait` lists are offset by it)
        if u8_norm is not None:
            self.u8 = {}
            for v in plan.inputs:
                lut = store.constant(('u8lut', v.C, repr(u8_norm)), lambda v=v: normalization_lut(v.C, u8_norm))
                buf = torch.zeros((n,) + v.shape, dtype=torch.uint8, device=device)
                users = [s for s in plan.steps for w in s.ins.values() if w is not None and w.buf is v.buf]
                fused = all(s.kind == 'conv' and s.ins['x'].buf is v.buf and s.ins['x'].C == v.C and
                            s.ins['x'].coff == v.coff and v.C % 4 != 0 and 'pre_bn' not in s.params and
                            s.ins.get('res1') is None and s.ins.get('res2') is None for s in users)
                self.u8[id(v.buf)] = (buf, lut, fused)
                if not fused:
                    self.calls.append((self.lib.dh_normalize_u8_f32,
                                       (buf.data_ptr(), lut.data_ptr(), self.ptr(v), n * int(np.prod(v.shape[:-1])),
                                        v.C), _InputStep(v)))
                    self.npre += 1
        for step in plan.steps:
            self._bind(step)
        self.grouped = 0      # pairs of launches merged into one (group_launches: latency regime only)
        self.noop_calls = set()   # indices of `calls` whose work moved into an earlier, grouped launch
        self.paired = []          # (index of the paired launch, index of the launch it absorbed): _pair_skinny_convs
        self.absorbed = {}        # index of a grouped / paired launch -> the Step whose work it also does

    def weight_layout(self, args):
        """dh_conv_args.w_split of a conv step: 1 = split-bf16 (plan.gemm_precision == 'bf16x3' and the library takes the
        layer: dh_conv2d_split_eligible), 2 = fp32 chunk-major for the halo-resident K x K kernel
        (dh_conv2d_halo_eligible: a rule on the per-frame geometry), else 0.  The LIBRARY decides, asked with the launch's
        own argument struct before the weights are packed -- a layer is never bound with a packing its launch rejects."""
        if getattr(self.plan, 'gemm_precision', 'f32') == 'bf16x3' and self.lib.dh_conv2d_split_eligible(C.byref(args)):
            return 1
        if os.environ.get('DEEPHAR_HALO_CONV', '1') != '0' and self.lib.dh_conv2d_halo_eligible(C.byref(args)):
            return 2
        return 0

    # ---- views -------------------------------------------------------------------------------------------
    def ptr(self, v):
        return self.base + 4 * (v.buf.offset * self.n + v.coff)

    def tensor(self, v):
        """torch view [n, *shape] of a Value (strided when it lives inside a wider buffer)."""
        torch = _torch()
        shape = (self.n,) + v.shape
        strides = [1] * len(shape)
        strides[-1] = 1
        if len(shape) >= 2:
            strides[-2] = v.ld
            for i in range(len(shape) - 3, -1, -1):
                strides[i] = strides[i + 1] * shape[i + 1]
        return torch.as_strided(self.arena, shape, strides, v.buf.offset * self.n + v.coff)

    # ---- binding -----------------------------------------------------------------------------------------
    def _bind(self, s):
        lib, n, k = self.lib, self.n, s.kind
        a = s.attrs
        P = self.ptr

        def opt(role, d=s.ins):
            v = d.get(role)
            return P(v) if v is not None else None

        if k == 'conv':
            x, y = s.ins['x'], s.outs['y']
            args = _lib.ConvArgs()
            args.x, args.y = P(x), P(y)
            if 'pre_bn' in s.params:
                sc, sh = self.store.bn_affine(s.params['pre_bn'])
                args.pre_scale, args.pre_shift = sc.data_ptr(), sh.data_ptr()
            if 'post_bn' in s.params or 'post_affine' in s.params:
                sc, sh = self.store.bn_affine(s.params.get('post_bn') or s.params['post_affine'])
                args.post_scale, args.post_shift = sc.data_ptr(), sh.data_ptr()
            up = 2 if a['up2'] else 1
            args.N = n * x.lead(3)
            args.H, args.W, args.Cin, args.ldx = x.shape[-3], x.shape[-2], x.C, x.ld
            sg = a.get('seg')                          # planner rule R14: x = [MaxPooling2D(x) | x2], read in place
            if sg is not None:                         # (H, W: the extent the convolution sees; Cin: both segments)
                args.H, args.W, args.Cin = x.shape[-3] // sg['pool_sh'], x.shape[-2] // 2, a['Cin']
            rs = a.get('x_resample', 0)                # planner rule R12: x is stored at another resolution than the conv sees
            if rs:
                args.x_resample = rs
                args.H, args.W = (2 * x.shape[-3], 2 * x.shape[-2]) if rs == 1 else (x.shape[-3] // 2, x.shape[-2] // 2)
            args.OH, args.OW, args.Cout, args.ldy = y.shape[-3] // up, y.shape[-2] // up, a['Cout'], y.ld
            args.KH, args.KW, args.SH, args.SW, args.PT, args.PL = a['kh'], a['kw'], a['sh'], a['sw'], a['pt'], a['pl']
            kp_np = (C.c_int(), C.c_int())
            _lib.check(lib.dh_conv2d_packed_dims(a['kh'], a['kw'], args.Cin, a['Cout'], C.byref(kp_np[0]), C.byref(kp_np[1])))
            args.K, args.Kp, args.Np = a['K'], kp_np[0].value, kp_np[1].value
            r1, r2 = s.ins.get('res1'), s.ins.get('res2')
            if r1 is not None:
                args.res1, args.ldr1 = P(r1), r1.ld
            if r2 is not None:
                args.res2, args.ldr2 = P(r2), r2.ld
            args.pre_relu, args.post_relu, args.up2 = a['pre_relu'], a['post_relu'], a['up2']
            args.res2_down = a.get('res2_down', 0)
            yp = s.outs.get('ypool')
            if yp is not None:                         # planner rule R7: the 2x2 max-pooled second output
                args.y_pool, args.ldyp = P(yp), yp.ld
            if self.u8 is not None and id(x.buf) in self.u8 and self.u8[id(x.buf)][2]:
                buf, lut, _ = self.u8[id(x.buf)]
                args.x, args.in_lut, args.x_u8 = buf.data_ptr(), lut.data_ptr(), 1
            a['x_u8'] = int(args.x_u8)                 # (read by bench.py to name the kernel)
            split = self.weight_layout(args)           # every field but w / w_split is final here
            wt, kp, np_ = self.store.conv_weight(s.params['w'], split=split)
            assert (kp, np_) == (args.Kp, args.Np)
            args.w, args.w_split = wt.data_ptr(), int(split)
            a['w_split'] = int(split)                  # (read by bench.py to name the kernel)
            self._keep.append(args)
            if sg is not None:
                seg = _lib.ConvSeg()
                x2 = s.ins.get('x2')
                if x2 is not None:
                    seg.x2, seg.ldx2 = P(x2), x2.ld
                seg.c_split, seg.pool_sh = sg['c_split'], sg['pool_sh']
                self._keep.append(seg)
                self.calls.append((lib.dh_conv2d_seg_f32, (C.byref(args), C.byref(seg)), s))
                return
            self.calls.append((lib.dh_conv2d_f32, (C.byref(args), a.get('tile_cfg', -1)), s))
        elif k == 'dwconv':
            x, y = s.ins['x'], s.outs['y']
            args = _lib.DwArgs()
            args.x, args.w, args.y = P(x), self.store.dw_weight(s.params['w']).data_ptr(), P(y)
            if 'pre_bn' in s.params:
                sc, sh = self.store.bn_affine(s.params['pre_bn'])
                args.pre_scale, args.pre_shift = sc.data_ptr(), sh.data_ptr()
            args.N, args.H, args.W, args.C = n * y.lead(3), y.shape[-3], y.shape[-2], x.C      # (up_in: x is at half resolution)
            args.ldx, args.ldy = x.ld, y.ld
            args.KH, args.KW, args.PT, args.PL, args.pre_relu = a['kh'], a['kw'], a['pt'], a['pl'], a['pre_relu']
            args.up_in = a.get('up_in', 0)
            self._keep.append(args)
            self.calls.append((lib.dh_dwconv2d_f32, (C.byref(args),), s))
        elif k == 'pool':
            x, y = s.ins['x'], s.outs['y']
            args = _lib.PoolArgs()
            args.x, args.y = P(x), P(y)
            args.N, args.H, args.W, args.C, args.ldx = n * x.lead(3), x.shape[-3], x.shape[-2], x.C, x.ld
            args.OH, args.OW, args.ldy = y.shape[-3], y.shape[-2], y.ld
            args.KH, args.KW, args.SH, args.SW, args.PT, args.PL = a['kh'], a['kw'], a['sh'], a['sw'], a['pt'], a['pl']
            args.mode = a.get('mode', 0)
            self._keep.append(args)
            self.calls.append((lib.dh_pool2d_f32, (C.byref(args),), s))
        elif k == 'upsample_add':
            b, y = s.ins['b'], s.outs['y']
            av = s.ins.get('a')
            self.calls.append((lib.dh_upsample2x_add_f32,
                               (P(av) if av is not None else None, av.ld if av is not None else 0, P(b), b.ld,
                                P(y), y.ld, n * y.lead(3), y.shape[-3], y.shape[-2], y.C), s))
        elif k == 'eltwise':
            av, y = s.ins['a'], s.outs['y']
            args = _lib.EltArgs()
            args.a, args.y, args.lda, args.ldy = P(av), P(y), av.ld, y.ld
            bv, cv = s.ins.get('b'), s.ins.get('c')
            if bv is not None:
                args.b, args.ldb = P(bv), bv.ld
            if cv is not None:
                args.c, args.ldc = P(cv), cv.ld
            if 'bn' in s.params:
                sc, sh = self.store.bn_affine(s.params['bn'])
                args.scale, args.shift = sc.data_ptr(), sh.data_ptr()
            elif 'scale_const' in a:
                kc, cc = a['scale_const'], av.C
                args.scale = self.store.constant(('const', kc, cc), lambda: np.full(cc, kc, np.float32)).data_ptr()
                args.shift = self.store.constant(('const', 0.0, cc), lambda: np.zeros(cc, np.float32)).data_ptr()
            args.npix, args.C = n * av.npix, av.C
            args.relu, args.op, args.bcast_b = a.get('relu', 0), a.get('op', 0), a.get('bcast_b', 0)
            self._keep.append(args)
            self.calls.append((lib.dh_eltwise_f32, (C.byref(args),), s))
        elif k == 'sam':
            h = s.ins['h']
            H, W = h.shape[-3], h.shape[-2]
            args = _lib.SamArgs()
            args.h = P(h)
            args.gx = self.store.constant(('gx', W), lambda: grid_x(W)).data_ptr()
            args.gy = self.store.constant(('gx', H), lambda: grid_x(H)).data_ptr()
            o = s.outs
            if o.get('xy') is not None:
                args.xy, args.ldxy = P(o['xy']), o['xy'].ld
            if o.get('conf_raw') is not None:
                args.conf_raw, args.ldcr = P(o['conf_raw']), o['conf_raw'].ld
            if o.get('conf_prob') is not None:
                args.conf_prob, args.ldcp = P(o['conf_prob']), o['conf_prob'].ld
            if o.get('prob') is not None:
                args.prob, args.ldp = P(o['prob']), o['prob'].ld
            if o.get('gmax') is not None:
                assert o['gmax'].dense
                args.gmax = P(o['gmax'])
            args.F, args.H, args.W, args.C, args.ldh = n * h.lead(3), H, W, h.C, h.ld
            args.alpha, args.conf_scale = a['alpha'], a['conf_scale']
            args.xy_times_conf = a.get('xy_times_conf', 0)
            self._keep.append(args)
            self.calls.append((lib.dh_softargmax2d_f32, (C.byref(args),), s))
        elif k == 'sam_ctx':
            h, y = s.ins['h'], s.outs['y']
            H, W = h.shape[-3], h.shape[-2]
            args = _lib.SamArgs()
            args.h = P(h)
            args.gx = self.store.constant(('gx', W), lambda: grid_x(W)).data_ptr()
            args.gy = self.store.constant(('gx', H), lambda: grid_x(H)).data_ptr()
            cr = s.outs.get('conf_raw')
            if cr is not None:
                args.conf_raw, args.ldcr = P(cr), cr.ld
            args.F, args.H, args.W, args.C, args.ldh = n * h.lead(3), H, W, h.C, h.ld
            args.alpha, args.conf_scale = a['sam_alpha'], a['conf_scale']
            self._keep.append(args)
            self.calls.append((lib.dh_softargmax2d_context_f32,
                               (C.byref(args), a['J'], a['nctx'], a['alpha'], P(y), y.ld), s))
        elif k == 'context_agg':
            ys, y = s.ins['ys'], s.outs['y']
            self.calls.append((lib.dh_context_aggregation_f32,
                               (P(ys), P(s.ins['yc']), P(s.ins['pc']), P(y), n * ys.lead(2), ys.shape[-2],
                                a['nctx'], a['alpha'], y.ld), s))
        elif k == 'depth_means':
            h = s.ins['h']
            self.calls.append((lib.dh_depth_means_f32,
                               (P(h), h.ld, opt('hxy', s.outs), opt('hz', s.outs), n * h.lead(3),
                                h.shape[-3] * h.shape[-2], a['D'], a['J']), s))
        elif k == 'softargmax1d':
            hz = s.ins['hz']
            D, J = hz.shape[-2], hz.shape[-1]
            grid = self.store.constant(('gd', D), lambda: grid_depth(D))
            z = s.outs.get('z')
            self.calls.append((lib.dh_softargmax1d_f32,
                               (P(hz), grid.data_ptr(), P(z) if z is not None else None,
                                z.ld if z is not None else 1, opt('vz', s.outs), n * hz.lead(2), D, J), s))
        elif k == 'kronecker':
            hm, x, y = s.ins['hm'], s.ins['x'], s.outs['y']
            self.calls.append((lib.dh_kronecker_f32,
                               (P(hm), hm.ld, P(x), x.ld, P(y), y.ld, n * hm.lead(3),
                                hm.shape[-3] * hm.shape[-2], hm.C, x.C), s))
        elif k == 'globalmaxmin':
            x, y = s.ins['x'], s.outs['y']
            self.calls.append((lib.dh_global_maxmin_softmax_f32,
                               (P(x), x.ld, P(y), n * x.lead(3), x.shape[-3] * x.shape[-2], x.C, a['softmax']), s))
        elif k == 'copy':
            x, y = s.ins['x'], s.outs['y']
            self.calls.append((lib.dh_copy_channels_f32, (P(x), x.ld, P(y), y.ld, n * x.npix, x.C), s))
        elif k == 'zeropad':
            x, y = s.ins['x'], s.outs['y']
            self.calls.append((lib.dh_zeropad2d_f32,
                               (P(x), P(y), n * x.lead(3), x.shape[-3], x.shape[-2], x.C, y.shape[-3],
                                y.shape[-2], a.get('pt', 0), a.get('pl', 0)), s))
        elif k == 'depthsum':
            d, h, z = s.ins['d'], s.ins['h'], s.outs['z']
            self.calls.append((lib.dh_depth_from_maps_f32,
                               (P(d), d.ld, P(h), h.ld, P(z), z.ld, n * d.lead(3), d.shape[-3] * d.shape[-2], d.C), s))
        else:
            raise NotImplementedError('no binding for step kind %r' % k)

    # ---- execution -----------------------------------------------------------------------------------------
    def _side_streams(self, main_ptr=None):
        """The extra HIP streams + sync events for the parallel branches of the plan.  [r06] The streams are the engine's
        role streams other than the one this plan is launched on (shared_stream below: every stream the package creates costs one
        of ROCm's four hardware queues, and a side stream that shares a queue with its plan's main stream is slow); only a plan
        with more branches than those creates streams of its own."""
        if getattr(self, '_streams', None) is None:
            lib = self.lib
            self._streams, self._events, self._join, self._fork, self._relay = [], {}, [], C.c_void_p(), {}
            roles = [shared_stream(self.device, r) for r in ('head', 'comm', 'copy')]
            spare = [st for st in roles if main_ptr is None or st.cuda_stream != main_ptr]
            for k in range(self.plan.nstreams - 1):
                st, ev = C.c_void_p(), C.c_void_p()
                if k < len(spare):
                    st = C.c_void_p(spare[k].cuda_stream)
                else:
                    _lib.check(lib.dh_stream_create(C.byref(st)), 'stream create')
                _lib.check(lib.dh_event_create_sync(C.byref(ev)), 'event create')
                self._streams.append(st)
                self._join.append(ev)
            _lib.check(lib.dh_event_create_sync(C.byref(self._fork)), 'event create')
            for i, (_, _, step) in enumerate(self.calls):
                if step.record:
                    ev = C.c_void_p()
                    _lib.check(lib.dh_event_create_sync(C.byref(ev)), 'event create')
                    self._events[i] = ev
        return self._streams

    def launch_all(self, stream_ptr):
        """Enqueue the whole plan.  With one stream: plain in-order launches.  With several: steps go to
        their scheduled stream, cross-stream dependencies become event waits, side streams fork from / join
        back into `stream_ptr` (so the sequence is capturable into one hipGraph)."""
        lib = self.lib
        if self.plan.nstreams <= 1:
            for i, (fn, args, step) in enumerate(self.calls):
                if self.perturb is not None:
                    self.perturb(i, step, stream_ptr, [stream_ptr])
                rc = fn(*args, stream_ptr)
                if rc != 0:
                    _lib.check(rc, 'step %s (%s)' % (step.kind, step.name))
            return
        side = self._side_streams(stream_ptr)
        ptrs = [stream_ptr] + [st for st in side]
        for fn, args, step in self.calls[:self.npre]:          # input staging runs before the fork
            _lib.check(fn(*args, stream_ptr), step.kind)
        _lib.check(lib.dh_event_record(self._fork, stream_ptr), 'fork')
        for st in side:
            _lib.check(lib.dh_stream_wait_event(st, self._fork), 'fork wait')
        for i, (fn, args, step) in enumerate(self.calls):
            if i < self.npre:
                continue
            sp = ptrs[step.stream]
            for w in step.wait:
                ev = self._events[w + self.npre]
                src = self.calls[w + self.npre][2].stream
                if step.stream != 0 and src > step.stream:
                    # ROCm 7.2 stream capture segfaults (hipStreamEndCapture) once two SIDE streams have waited on
                    # each other's events in both directions (tools/micro/capture_pattern.hip).  A lower-numbered
                    # side stream therefore never waits on a higher-numbered one directly: the origin stream waits
                    # and re-publishes the dependency.
                    key = (i, w)
                    if key not in self._relay:
                        r = C.c_void_p()
                        _lib.check(lib.dh_event_create_sync(C.byref(r)), 'event create')
                        self._relay[key] = r
                    _lib.check(lib.dh_stream_wait_event(stream_ptr, ev), 'relay wait')
                    _lib.check(lib.dh_event_record(self._relay[key], stream_ptr), 'relay record')
                    ev = self._relay[key]
                _lib.check(lib.dh_stream_wait_event(sp, ev), 'dependency wait')
            if self.perturb is not None:
                self.perturb(i, step, sp, ptrs)
            rc = fn(*args, sp)
            if rc != 0:
                _lib.check(rc, 'step %s (%s)' % (step.kind, step.name))
            if step.record:
                _lib.check(lib.dh_event_record(self._events[i], sp), 'event record')
        for st, ev in zip(side, self._join):
            _lib.check(lib.dh_event_record(ev, st), 'join record')
            _lib.check(lib.dh_stream_wait_event(stream_ptr, ev), 'join wait')

    # ---- grouped launches (latency regime) ------------------------------------------------------------------------
    GROUP_MAX_ROWS = 8192           # conv rows (pixels of the bound batch) up to which a pair is merged
    GROUP_MAX_DW_ELEMS = 1 << 23    # ... and depthwise elements

    def group_launches(self, stream_ptr):
        """[r06] Merge every (1x1 shortcut convolution, depthwise convolution) pair of a pre-activation residual unit that is
        launched back to back on one stream, reads the same tensor and is small enough to be bound by the cost of a node
        rather than by its work into ONE launch (dh_conv2d_dw_group_f32: bit-identical, the work-groups of one grid run
        either kernel's code).  The depthwise entry of `calls` stays as a no-op so that step indices (event waits, per-step
        profiles) keep their meaning.  DEEPHAR_GROUP_LAUNCHES=0 switches it off.  Returns the number of pairs merged."""
        if os.environ.get('DEEPHAR_GROUP_LAUNCHES', '1') == '0' or self.grouped or self.paired:
            return self.grouped
        lib = self.lib
        for i in range(self.npre, len(self.calls) - 1):
            fc, ac, sc = self.calls[i]
            # the next launch ON THE SAME STREAM (a multi-stream plan interleaves the streams in launch order): the pair is
            # consecutive there, so running the depthwise conv at the shortcut's place moves it past nothing it is ordered
            # with -- it waits for no event of its own (checked), and what it writes was planned to be free from here on
            j = next((k for k in range(i + 1, len(self.calls)) if self.calls[k][2].stream == sc.stream), None)
            if j is None:
                continue
            fd, ad, sd = self.calls[j]
            if sc.kind != 'conv' or sd.kind != 'dwconv' or sd.wait or i in self.noop_calls or fc is lib.dh_conv2d_dw_group_f32:
                continue
            xc, xd = sc.ins['x'], sd.ins['x']
            if xc.buf is not xd.buf or any(v is not None and v.buf is sc.outs['y'].buf for v in sd.ins.values()):
                continue                               # not the same input, or the depthwise conv reads the conv's result
            ca, da = ac[0]._obj, ad[0]._obj
            if ca.N * ca.OH * ca.OW > self.GROUP_MAX_ROWS or da.N * da.H * da.W * da.C > self.GROUP_MAX_DW_ELEMS:
                continue
            if lib.dh_conv2d_dw_group_f32(ac[0], ad[0], stream_ptr) != 0:      # (a real launch: the pair's own outputs)
                continue
            sc.attrs['grouped'] = True                 # (for bench.py's kernel names: the plan's steps are shared by every
            self.calls[i] = (lib.dh_conv2d_dw_group_f32, (ac[0], ad[0]), sc)      # batch size it is bound to -- what counts
            self.calls[j] = (_noop_launch, (), sd)                                 # for execution is `calls` / `noop_calls`)
            self.noop_calls.add(j)
            self.absorbed[i] = sd
            self.grouped += 1
        self._pair_skinny_convs(stream_ptr)
        if (self.grouped or self.paired) and self.graph is not None:
            if self.plan.nstreams > 1:
                _GRAPH_GRAVEYARD.append(self.graph)
            else:
                lib.dh_graph_destroy(self.graph)
            self.graph = None
        return self.grouped

    PAIR_LOOKAHEAD = 3              # launches of the same stream a partner may be pulled forward past

    @staticmethod
    def _values_overlap(v, w):
        """May the two views touch the same floats?  Same buffer: unless they are disjoint channel runs of one pitch;
        different buffers: if their arena ranges meet (the memory plan re-uses space)."""
        if v is None or w is None:
            return False
        if v.buf is w.buf:
            return not (v.ld == w.ld and (v.coff + v.C <= w.coff or w.coff + w.C <= v.coff))
        a, b = v.buf, w.buf
        return not (a.offset + a.items <= b.offset or b.offset + b.items <= a.offset)

    @classmethod
    def _independent(cls, s, t):
        """neither step reads or overwrites what the other writes"""
        for o in s.outs.values():
            if any(cls._values_overlap(o, v) for v in list(t.ins.values()) + list(t.outs.values())):
                return False
        for o in t.outs.values():
            if any(cls._values_overlap(o, v) for v in s.ins.values()):
                return False
        return True

    def _pair_skinny_convs(self, stream_ptr):
        """[r06] Two independent convolutions of the skinny-conv kernel that follow each other on one stream -- the residual unit
        on the action head's pose features beside `v_conv0` on its appearance features (spnet.py:113-133), the pointwise half of
        the separable unit on the previous head's features beside this head's conv2 -- become ONE launch (dh_conv2d_pair_f32:
        work-groups of one grid run either convolution's own code: bit-identical).  The partner is the next launch of the same
        stream, or one up to PAIR_LOOKAHEAD launches further down that is independent of everything it is pulled past; it must
        wait for no event itself and pass no launch that does (an event a launch waits for may be the one that orders the
        partner's inputs too).  DEEPHAR_PAIR_CONVS=0 switches it off."""
        if os.environ.get('DEEPHAR_PAIR_CONVS', '1') == '0':
            return
        lib = self.lib

        def skinny(k):
            fn, args, st = self.calls[k]
            if fn is not lib.dh_conv2d_f32 or k in self.noop_calls or st.kind != 'conv':
                return False
            ca = args[0]._obj                          # (the latency regime only, like the conv + depthwise groups)
            return not ca.x_resample and ca.N * ca.OH * ca.OW <= self.GROUP_MAX_ROWS and lib.dh_conv2d_uses_split_k(args[0]) == 1

        i = self.npre
        while i < len(self.calls) - 1:
            if not skinny(i):
                i += 1
                continue
            fa, aa, sa = self.calls[i]
            passed, npassed = [], 0
            for k in range(i + 1, len(self.calls)):
                fk, ak, sk = self.calls[k]
                if sk.stream != sa.stream or k in self.noop_calls:
                    continue                           # (a no-op's work runs where its group / pair is: see `absorbed` below)
                if sk.wait or npassed > self.PAIR_LOOKAHEAD:
                    break
                if skinny(k) and self._independent(sa, sk) and all(self._independent(q, sk) for q in passed) and \
                        lib.dh_conv2d_pair_f32(aa[0], ak[0], stream_ptr) == 0:
                    self.calls[i] = (lib.dh_conv2d_pair_f32, (aa[0], ak[0]), sa)
                    self.calls[k] = (_noop_launch, (), sk)
                    self.noop_calls.add(k)
                    self.absorbed[i] = sk
                    self.paired.append((i, k))
                    break
                passed.append(sk)
                npassed += 1
                if k in self.absorbed:                 # a grouped / paired launch does a second step's work at this place
                    passed.append(self.absorbed[k])
            i += 1

    def capture(self, stream_ptr):
        """Capture the launch sequence into a hipGraph.  -> False (nothing captured; the caller launches eagerly) when
        this is a multi-stream plan and the process already holds MAX_MULTISTREAM_GRAPHS such graph execs: they can only
        be parked, never destroyed (__del__), so their number is what bounds the leak of a long-lived multi-model process."""
        global _MULTISTREAM_GRAPHS
        lib = self.lib
        if self.plan.nstreams > 1:
            if _MULTISTREAM_GRAPHS >= MAX_MULTISTREAM_GRAPHS:
                return False
        _lib.check(lib.dh_graph_begin_capture(stream_ptr), 'graph capture begin')
        try:
            self.launch_all(stream_ptr)
        finally:
            g = C.c_void_p()
            rc = lib.dh_graph_end_capture(stream_ptr, C.byref(g))
        _lib.check(rc, 'graph capture end')
        self.graph = g
        if self.plan.nstreams > 1:
            _MULTISTREAM_GRAPHS += 1      # counted only once a graph exec exists (a failed capture holds no slot: ADVICE r04)
        return True

    def replay(self, stream_ptr):
        _lib.check(self.lib.dh_graph_launch(self.graph, stream_ptr), 'graph launch')

    # ---- autotuning of the conv tiling -------------------------------------------------------------------
    @staticmethod
    def _conv_signature(step):
        a, x, y = step.attrs, step.ins['x'], step.outs['y']
        r1, r2 = step.ins.get('res1'), step.ins.get('res2')
        # the 16-byte alignment of every view decides which kernels / epilogues are eligible (gemm1x1_eligible, the
        # vector epilogue), so it is part of the identity of a tuned shape -- as are padding and the output size
        align = (x.coff % 4, y.coff % 4, r1.coff % 4 if r1 is not None else 0, r2.coff % 4 if r2 is not None else 0,
                 r1.ld % 4 if r1 is not None else 0, r2.ld % 4 if r2 is not None else 0)
        return (x.lead(3), x.shape[-3], x.shape[-2], x.C, x.ld, y.ld, a['Cout'], a['kh'], a['kw'], a['sh'],
                a['sw'], a['pre_relu'], a['post_relu'], a['up2'], 'res1' in step.ins, 'res2' in step.ins,
                'pre_bn' in step.params, 'post_bn' in step.params or 'post_affine' in step.params, a['pt'], a['pl'], y.shape[-3], y.shape[-2], a.get('res2_down', 0), 'ypool' in step.outs) + align

    def autotune(self, stream_ptr, table=None, reps=3):
        """Time every tile configuration of dh_conv2d_f32 for each distinct conv shape of this bound plan
        (HIP events on the launch stream) and keep the fastest.  All configurations sum K in the same order,
        so the choice never changes a result bit.  `table` (signature -> cfg) is filled / reused."""
        lib = self.lib
        table = {} if table is None else table
        ncfgs = {'conv': lib.dh_conv2d_num_tile_cfgs()}
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.check(lib.dh_event_create(C.byref(e0)))
        _lib.check(lib.dh_event_create(C.byref(e1)))
        for i, (fn, args, step) in enumerate(self.calls):
            if step.kind not in ncfgs:
                continue
            if fn is lib.dh_conv2d_seg_f32:                      # (skinny-conv kernel by its own entry point: nothing to choose)
                step.attrs['tile_cfg'] = -1
                step.attrs['split_k'] = True
                continue
            ncfg = ncfgs[step.kind]
            cargs = args[0]._obj
            sig = (self.n, step.kind) + self._conv_signature(step) + ((('u8',) if cargs.x_u8 else ())) + \
                ((({1: 'bf16x3', 2: 'halo'}[cargs.w_split],) if cargs.w_split else ()))
            if cargs.w_split:
                ncfg = lib.dh_conv2d_num_halo_tile_cfgs() if cargs.w_split == 2 else lib.dh_conv2d_num_split_tile_cfgs()
            if step.kind == 'conv' and lib.dh_conv2d_uses_split_k(args[0]):
                step.attrs['tile_cfg'] = -1                      # shape rule: split-K kernel, no tilings to choose from
                step.attrs['split_k'] = True
                self.calls[i] = (fn, (args[0], -1), step)
                continue
            if step.kind == 'conv' and lib.dh_conv2d_uses_first_layer_kernel(args[0]):
                step.attrs['tile_cfg'] = -1                      # shape rule: first-layer kernel (conv_stem.hip)
                step.attrs['first_layer'] = True
                self.calls[i] = (fn, (args[0], -1), step)
                continue
            shape_key = ('shape', int(cargs.w_split), self.n, cargs.N * cargs.OH * cargs.OW, cargs.K, cargs.Cout, cargs.KH,
                         cargs.KW, cargs.up2)
            if sig not in table:
                def time_cfg(cfg, nrep, trials):
                    t_cfg = float('inf')
                    for _trial in range(trials):         # best of several trials: one noisy sample must not pick the tiling
                        _lib.check(lib.dh_event_record(e0, stream_ptr))
                        for _ in range(nrep):
                            fn(args[0], cfg, stream_ptr)
                        _lib.check(lib.dh_event_record(e1, stream_ptr))
                        _lib.check(lib.dh_event_synchronize(e1))
                        ms = C.c_float()
                        _lib.check(lib.dh_event_elapsed_ms(e0, e1, C.byref(ms)))
                        t_cfg = min(t_cfg, ms.value / nrep)
                    return t_cfg
                timed = {}
                for cfg in range(ncfg):
                    if fn(args[0], cfg, stream_ptr) != 0:     # unsupported combination (warm-up launch)
                        continue
                    timed[cfg] = time_cfg(cfg, reps, 2)
                best = -1
                if timed:
                    lo = min(timed.values())
                    # candidates within 8 % of the fastest are timed again, longer: a 3-4 % difference between two
                    # tilings of the dominant GEMM is worth more than the first pass can resolve
                    close = [c for c, t in timed.items() if t <= 1.08 * lo]
                    if len(close) > 1:
                        for c in close:
                            timed[c] = time_cfg(c, 4 * reps, 3)
                        best = min(close, key=lambda c: timed[c])
                        # tilings within 1 % of each other are the same speed as far as this timing can tell: the same
                        # GEMM (M x K x N, taps) under another epilogue signature keeps the tiling chosen first, so one
                        # layer family runs ONE instantiation (per-kernel profiles stay readable, PMC look-ups hit)
                        pref = table.get(shape_key)
                        if pref in close and timed[pref] <= 1.01 * timed[best]:
                            best = pref
                    else:
                        best = close[0]
                table[sig] = best
                table.setdefault(shape_key, best)
            step.attrs['tile_cfg'] = table[sig]
            self.calls[i] = (fn, (args[0], table[sig]), step)
        lib.dh_event_destroy(e0)
        lib.dh_event_destroy(e1)
        if self.graph is not None:
            if self.plan.nstreams > 1:
                _GRAPH_GRAVEYARD.append(self.graph)      # see __del__
            else:
                lib.dh_graph_destroy(self.graph)
            self.graph = None
        return table

    def profile(self, stream_ptr, reps=1):
        """Per-step device time (ms, mean over reps) with HIP events recorded on the launch stream."""
        lib = self.lib
        evs = []
        for _ in range(len(self.calls) + 1):
            e = C.c_void_p()
            _lib.check(lib.dh_event_create(C.byref(e)))
            evs.append(e)
        times = np.zeros(len(self.calls))
        for _ in range(reps):
            _lib.check(lib.dh_event_record(evs[0], stream_ptr))
            for i, (fn, args, step) in enumerate(self.calls):
                _lib.check(fn(*args, stream_ptr), step.kind)
                _lib.check(lib.dh_event_record(evs[i + 1], stream_ptr))
            _lib.check(lib.dh_event_synchronize(evs[-1]))
            for i in range(len(self.calls)):
                ms = C.c_float()
                _lib.check(lib.dh_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms)))
                times[i] += ms.value
        for e in evs:
            lib.dh_event_destroy(e)
        return times / reps

    def __del__(self):
        # ROCm 7.2: hipGraphExecDestroy of a graph captured across several streams (event fork/join nodes) leaves
        # the runtime in a state where a LATER capture + launch of another multi-stream graph segfaults
        # (tools/repro_seg.py: two bound plans of one model destroyed, next model's first replay crashes; the
        # same sequence with one stream, or without the destroy, is fine).  Multi-stream graph execs are therefore
        # parked until process exit instead of destroyed -- a few KB of kernel-node parameters each.
        try:
            if self.graph is not None:
                if self.plan.nstreams > 1:
                    _GRAPH_GRAVEYARD.append(self.graph)
                else:
                    self.lib.dh_graph_destroy(self.graph)
                self.graph = None
        except Exception:
            pass


def _noop_launch(*_args):
    """Placeholder of a launch that was merged into the one before it (BoundPlan.group_launches)."""
    return 0


_GRAPH_GRAVEYARD = []
# multi-stream graph execs ever created by this process (live + parked): capped, see BoundPlan.capture.  One-stream
# plans (the default, Model.num_streams = 1) destroy their graphs and are not counted.
_MULTISTREAM_GRAPHS = 0
MAX_MULTISTREAM_GRAPHS = max(0, int(os.environ.get('DEEPHAR_MAX_MULTISTREAM_GRAPHS', '64')))
# ---- the package's HIP streams, one per ROLE and device ----------------------------------------------------------------
# ROCm maps every HIP stream onto one of GPU_MAX_HW_QUEUES (default 4) hardware queues when it is created.  Until round 6 every
# Model owned a compute stream, every predict() a copy stream, every frame-sharded clip model a collective stream: the fourth
# or fifth stream of a process landed on a hardware queue another one already used, and two streams that order each other with
# events through ONE hardware queue cost the frame-sharded SPNet / merge models 4.7 % of their step (22.0 -> 23.0 ms, 7.85 ->
# 8.24 ms) -- in any process that had called Model.predict on another model before (profiles/r06_hw_queue_collision.md).
# Now: ONE stream per role -- 'compute' (every Model's launches; models of a process run one after the other anyway), 'copy'
# (predict's host-device staging), 'head' (the head stage of a frame-sharded clip model), 'comm' (its collective) -- four
# streams, four hardware queues, whatever the process did before.
_ROLE_STREAMS = {}
_ROLE_LOCKS = {}        # (device index, role) -> RLock: the models of a process share a stream, so ONE thread at a time enqueues on it
                        # (a hipGraph capture on the stream would otherwise swallow another thread's launches)


def stream_lock(device, role):
    import threading
    torch = _torch()
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    lk = _ROLE_LOCKS.get(key)
    if lk is None:
        lk = _ROLE_LOCKS.setdefault(key, threading.RLock())
    return lk


def shared_stream(device, role):
    torch = _torch()
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    st = _ROLE_STREAMS.get(key)
    if st is None:
        with torch.cuda.device(device):
            for r in ('compute', 'copy', 'head', 'comm'):           # a fixed creation order: the mapping does not depend on
                k = (key[0], r)                                      # which role a process happens to need first
                if k not in _ROLE_STREAMS:
                    _ROLE_STREAMS[k] = torch.cuda.Stream(device=device)
        st = _ROLE_STREAMS.get(key)
        if st is None:
            with torch.cuda.device(device):
                st = _ROLE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


_STAGING_THREADS = max(1, min(8, (os.cpu_count() or 2) // 2, int(os.environ.get('DEEPHAR_STAGING_THREADS', '6'))))
_POOL = None


def _staging_pool():
    """Threads that copy / cast host arrays into pinned staging for Model.predict (run_pipelined); orchestration only."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=_STAGING_THREADS, thread_name_prefix='deephar-staging')
    return _POOL


class Executor:
    def __init__(self, plan, device=None, use_graph=True, autotune=True, stream_role='compute'):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.DeepharHipError('no HIP device visible: deephar_amd runs only on an AMD GPU (gfx950); '
                                       'there is no CPU execution path')
        _lib.load()
        self.device = torch.device(device or 'cuda:%d' % torch.cuda.current_device())
        self.plan = plan
        self.use_graph = use_graph
        self.autotune = autotune
        self.tune_table = {}
        self.store = WeightStore(self.device)
        self.bound = {}            # key -> BoundPlan, most recently used last; at most `max_bound` are kept
        self.max_bound = max(1, int(os.environ.get('DEEPHAR_MAX_BOUND_PLANS', '4')))
        self._wstamp = None        # sum of Param.version over plan.params at the last refresh / first bind
        self.stream = shared_stream(self.device, stream_role)
        self._lock = stream_lock(self.device, stream_role)

    @property
    def stream_ptr(self):
        return self.stream.cuda_stream

    def bind(self, n, u8_norm=None):
        with self._lock:          # one thread at a time on the shared stream (engine/executor.py: stream_lock)
            return self._bind_locked(n, u8_norm)

    def _bind_locked(self, n, u8_norm=None):
        """u8_norm: None for float inputs; a channel_power (1 or a per-channel list) when the inputs are raw uint8
        frames to be normalised on the GPU like utils/transform.normalize_channels."""
        key = n if u8_norm is None else (n, repr(u8_norm))
        bp = self.bound.get(key)
        if bp is None:
            torch = _torch()
            with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
                bp = BoundPlan(self.plan, n, self.store, self.device, u8_norm=u8_norm)
                if self.autotune:
                    bp.autotune(self.stream_ptr, self.tune_table)
                bp.group_launches(self.stream_ptr)
            self.bound[key] = bp
            while len(self.bound) > self.max_bound:         # dataset tails / box-refinement loops bind many sizes:
                old = next(iter(self.bound))                # drop the least recently used arena + graph
                if old == key:
                    break
                del self.bound[old]
        else:
            self.bound[key] = self.bound.pop(key)           # mark as most recently used
        return bp

    def _weight_stamp(self):
        return sum(p.version for p in self.plan.params)

    def sync_weights(self):
        """Device weights follow the host Params: any Param.set since the last forward (Model.set_weights,
        Layer.set_weights, load_weights through ANOTHER model that shares these layers, init_synthetic ...) bumps
        its version; the sum over the plan's params is compared before every forward and changed tensors are
        re-packed and copied in place (pointers stay valid, captured graphs need no re-capture)."""
        stamp = self._weight_stamp()
        if self._wstamp is None:
            self._wstamp = stamp
        elif stamp != self._wstamp:
            self.refresh_weights()

    def refresh_weights(self):
        """Push changed Params to their (already bound) device tensors in place."""
        torch = _torch()
        self._wstamp = self._weight_stamp()
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            for s in self.plan.steps:
                for role, p in s.params.items():
                    if role == 'w' and s.kind != 'dwconv':
                        for (pid, split) in [k_ for k_ in self.store.conv if k_[0] == id(p)]:
                            self.store.conv_weight(p, split=split)
                    elif role == 'w':
                        self.store.dw_weight(p)
                    elif role == 'dw':
                        self.store.dw_weight(p)
                    else:
                        self.store.bn_affine(p)
        self.stream.synchronize()

    def set_inputs(self, bp, arrays):
        """Copy host (or device) arrays [m<=n, ...] into the input buffers (fp32, or the byte staging buffers of
        a uint8-bound plan)."""
        torch = _torch()
        for v, arr in zip(self.plan.inputs, arrays):
            if bp.u8 is not None:
                dst = bp.u8[id(v.buf)][0]
                src = torch.from_numpy(np.ascontiguousarray(arr)) if isinstance(arr, np.ndarray) else arr
                if src.dtype != torch.uint8 or tuple(src.shape[1:]) != tuple(dst.shape[1:]) or src.shape[0] > bp.n:
                    raise ValueError('uint8-bound plan expects uint8 input [<=%d, %s], got %s %s' %
                                     (bp.n, tuple(dst.shape[1:]), src.dtype, tuple(src.shape)))
                dst[:src.shape[0]].copy_(src.to(self.device, non_blocking=True))
                continue
            dst = bp.tensor(v)
            if isinstance(arr, np.ndarray):
                src = torch.from_numpy(np.ascontiguousarray(arr))
            else:
                src = arr
            m = src.shape[0]
            if tuple(src.shape[1:]) != tuple(dst.shape[1:]) or m > bp.n:
                raise ValueError('input has shape %s, model expects [<=%d, %s]' % (tuple(src.shape), bp.n,
                                                                                  tuple(dst.shape[1:])))
            dst[:m].copy_(src.to(self.device, non_blocking=True).to(torch.float32))

    def forward(self, bp):
        with self._lock:          # one thread at a time on the shared stream (engine/executor.py: stream_lock)
            return self._forward_locked(bp)

    def _forward_locked(self, bp):
        """Enqueue one forward pass of the bound plan on the executor stream."""
        if self.use_graph and (bp.graph is not None or bp.capture(self.stream_ptr)):
            bp.replay(self.stream_ptr)
        else:
            bp.launch_all(self.stream_ptr)

    def run(self, arrays, n=None, u8_norm=None):
        """arrays: list of host arrays with equal leading dim m <= n.  Returns list of np.float32 outputs."""
        return self.run_pipelined(arrays, n or arrays[0].shape[0], u8_norm=u8_norm)

    # ---- the Model.predict boundary: host arrays in, host arrays out ---------------------------------------------
    def _io(self, bp, depth):
        """Per bound plan: a ring of `depth` slots of pinned host staging (inputs in the dtype that crosses PCIe: uint8
        frames or float32 -- float64 loader arrays are cast on the HOST while being copied into the pinned buffer),
        their device twins, one pinned buffer per slot for the packed outputs, and the events ordering the streams."""
        io = getattr(bp, '_io', None)
        if io is None:
            io = bp._io = dict(host_in=[], dev_in=[], host_out=[], h2d=[], done=[], pending=[])
        torch = _torch()
        dt = torch.uint8 if bp.u8 is not None else torch.float32
        while len(io['host_in']) < depth:
            io['host_in'].append([torch.empty((bp.n,) + v.shape, dtype=dt, pin_memory=True) for v in self.plan.inputs])
            io['dev_in'].append([torch.empty((bp.n,) + v.shape, dtype=dt, device=self.device) for v in self.plan.inputs])
            io['host_out'].append(torch.empty(max(self.plan.out_items * bp.n, 4), dtype=torch.float32, pin_memory=True))
            io['h2d'].append(torch.cuda.Event())
            io['done'].append(torch.cuda.Event())
            io['pending'].append(None)
        return io

    def _host_outputs(self, bp, flat, m):
        """Slice the packed output region (host copy of arena[0 : out_items * n]) into one fresh array per output."""
        torch = _torch()
        outs = []
        for v in self.plan.outputs:
            shape = (bp.n,) + v.shape
            strides = [1] * len(shape)
            if len(shape) >= 2:
                strides[-2] = v.ld
                for i in range(len(shape) - 3, -1, -1):
                    strides[i] = strides[i + 1] * shape[i + 1]
            view = torch.as_strided(flat, shape, strides, v.buf.offset * bp.n + v.coff)
            outs.append(np.array(view[:m].numpy(), dtype=np.float32, copy=True, order='C'))
        return outs

    def run_pipelined(self, arrays, bs, u8_norm=None, verbose=0):
        with self._lock:          # one thread at a time on the shared stream (engine/executor.py: stream_lock)
            return self._run_pipelined_locked(arrays, bs, u8_norm, verbose)

    def _run_pipelined_locked(self, arrays, bs, u8_norm=None, verbose=0):
        """Forward over all rows of `arrays` (host arrays, equal leading dim) in chunks of `bs`, as a 3-stage pipeline:
             host   : cast / copy chunk i+2 into pinned staging on a small THREAD POOL (row slices in parallel; the
                      copy releases the GIL) while the main thread enqueues chunk i -- a float32 batch of 64 frames is
                      50 MB, and one core writing pinned (fine-grained) host memory was the serial stage of the float32
                      boundary (66 % of the device-resident rate over 2 048 frames before the pool)
             copy   : pinned -> device staging on a copy stream      (overlaps the graph replay of chunk i)
             compute: staging -> plan input (D2D), hipGraph replay, ONE D2H of the packed outputs of the whole model
                      (arena[0 : out_items * n]) into pinned memory.
           The ring of staging slots is DEEP (up to DEEPHAR_PREDICT_DEPTH = 8 chunks in flight, capped at 512 MB of
           pinned staging per bound plan): on this stack every host-side wait that actually has to block is followed by
           a 40-70 ms hole in the GPU's timeline, even though later chunks are already queued
           (profiles/r02_predict_boundary.md) -- so the host enqueues ahead and waits as late as it can.
           Returns one np.float32 array per model output, rows in input order (keras Model.predict semantics:
           exp/common/*_tools.py; timing method of exp/pennaction/eval_speed2d.py:70-77)."""
        torch = _torch()
        total = arrays[0].shape[0]
        bs = int(min(bs, total))
        nchunks = (total + bs - 1) // bs
        with torch.cuda.device(self.device):
            if self.bound:
                self.sync_weights()
            with torch.cuda.stream(self.stream):
                bp = self.bind(bs, u8_norm=u8_norm)
            if self._wstamp is None:
                self._wstamp = self._weight_stamp()
            if getattr(self, 'copy_stream', None) is None:
                self.copy_stream = shared_stream(self.device, 'copy')
            want = torch.uint8 if bp.u8 is not None else torch.float32
            slot_bytes = sum(bs * int(np.prod(v.shape)) for v in self.plan.inputs) * (1 if want == torch.uint8 else 4)
            depth = max(2, min(nchunks, int(os.environ.get('DEEPHAR_PREDICT_DEPTH', '8')),
                               max(2, (512 << 20) // max(slot_bytes, 1))))
            io = self._io(bp, depth)
            io['pending'] = [None] * len(io['pending'])        # a call that died mid-loop must not leak rows into this one
            results = []
            ahead = min(2, depth - 1)                           # chunks staged ahead of the one being enqueued
            pool = _staging_pool()
            staged = {}

            def collect(slot):
                m = io['pending'][slot]
                if m is not None:
                    io['done'][slot].synchronize()
                    results.append(self._host_outputs(bp, io['host_out'][slot], m))
                    io['pending'][slot] = None

            def stage(ci):
                """Start the host-side copies of chunk ci into its slot; returns (m, on_dev, device sources, futures)."""
                slot, start = ci % depth, ci * bs
                collect(slot)                                   # chunk ci - depth used this slot (its H2D is long done)
                m = min(bs, total - start)
                on_dev, dev_src, futs = [], [], []
                for k, (v, arr) in enumerate(zip(self.plan.inputs, arrays)):
                    part = arr[start:start + m]
                    if tuple(part.shape[1:]) != tuple(v.shape):
                        raise ValueError('input has shape %s, model expects [N, %s]' % (tuple(arr.shape), tuple(v.shape)))
                    src = torch.from_numpy(np.ascontiguousarray(part)) if isinstance(part, np.ndarray) else part
                    if (src.dtype == torch.uint8) != (want == torch.uint8):
                        raise ValueError('plan bound for %s inputs, got %s' % (want, src.dtype))
                    on_dev.append(bool(src.is_cuda))
                    dev_src.append(src if src.is_cuda else None)
                    if not src.is_cuda:                         # host-side cast (f64 -> f32) + pin, row slices in parallel
                        dst = io['host_in'][slot][k]
                        nsl = max(1, min(_STAGING_THREADS, src.numel() * src.element_size() >> 22, m))
                        step = -(-m // nsl)
                        for a0 in range(0, m, step):
                            a1 = min(a0 + step, m)              # a ragged last chunk: both slices stop at m
                            futs.append(pool.submit(dst[a0:a1].copy_, src[a0:a1]))
                return m, on_dev, dev_src, futs

            try:
                for ci in range(min(ahead, nchunks)):
                    staged[ci] = stage(ci)
                for ci in range(nchunks):
                    if ci + ahead < nchunks:
                        staged[ci + ahead] = stage(ci + ahead)
                    slot = ci % depth
                    m, on_dev, dev_src, futs = staged.pop(ci)
                    for fu in futs:
                        fu.result()
                    with torch.cuda.stream(self.copy_stream):
                        if any(on_dev):                         # the caller's device data may still be in flight on ITS stream
                            self.copy_stream.wait_stream(torch.cuda.current_stream())
                        for k in range(len(arrays)):
                            src = dev_src[k] if on_dev[k] else io['host_in'][slot][k][:m]
                            io['dev_in'][slot][k][:m].copy_(src, non_blocking=True)
                        io['h2d'][slot].record(self.copy_stream)
                    with torch.cuda.stream(self.stream):
                        self.stream.wait_event(io['h2d'][slot])                       # GPU-side wait
                        for k, v in enumerate(self.plan.inputs):
                            dst = bp.u8[id(v.buf)][0] if bp.u8 is not None else bp.tensor(v)
                            dst[:m].copy_(io['dev_in'][slot][k][:m], non_blocking=True)
                        self.forward(bp)
                        io['host_out'][slot].copy_(bp.arena[:io['host_out'][slot].numel()], non_blocking=True)
                        io['done'][slot].record(self.stream)
                    io['pending'][slot] = m
                    if verbose:
                        print('%d/%d' % (min((ci + 1) * bs, total), total))
                for k in range(depth):                                  # oldest pending chunk first
                    collect((nchunks + k) % depth)
            finally:
                for _, _, _, futs in staged.values():                   # (an exception mid-loop: let the copies finish)
                    for fu in futs:
                        fu.cancel() or fu.exception()
                if any(p is not None for p in io['pending']):
                    self.stream.synchronize()
                    io['pending'] = [None] * len(io['pending'])
        nres = len(self.plan.outputs)
        return [np.concatenate([r[k] for r in results], axis=0) if len(results) > 1 else results[0][k]
                for k in range(nres)]

    def run_device(self, tensors, n=None, inputs_copied=None):
        with self._lock:          # one thread at a time on the shared stream (engine/executor.py: stream_lock)
            return self._run_device_locked(tensors, n, inputs_copied)

    def _run_device_locked(self, tensors, n=None, inputs_copied=None):
        """Device tensors in ([m <= n, ...] float32 on this device), device VIEWS of the outputs out (valid until the
        next forward of the same bound plan); everything is enqueued on `self.stream`, nothing touches the host.
        An input may be any strided view with the input's element count whose leading dim is m -- e.g. the
        [m, G, T/G, J, c] view of a rank-major all-gather result: the copy into the plan's contiguous [m, T, J, c]
        input does the re-ordering.  `inputs_copied`: optional torch event recorded on `self.stream` right after the
        input copies (the producer may then overwrite its buffers).
        Used by the frame-sharded clip runtime (parallel.py) around the RCCL all-gather."""
        torch = _torch()
        m = tensors[0].shape[0]
        n = n or m
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            if self.bound:
                self.sync_weights()
            bp = self.bind(n)
            if self._wstamp is None:
                self._wstamp = self._weight_stamp()
            for v, t in zip(self.plan.inputs, tensors):
                dst = bp.tensor(v)[:m] if m <= bp.n else None
                if dst is None or t.shape[0] != m or t.numel() != dst.numel() or \
                        tuple(t.shape[-1:]) != tuple(v.shape[-1:]):
                    raise ValueError('input has shape %s, model expects [<=%d, %s]' % (tuple(t.shape), bp.n, v.shape))
                if tuple(t.shape) != tuple(dst.shape):
                    dst = dst.view(tuple(t.shape))
                dst.copy_(t, non_blocking=True)
            if inputs_copied is not None:
                inputs_copied.record(self.stream)
            self.forward(bp)
            return [bp.tensor(v)[:m] for v in self.plan.outputs]
