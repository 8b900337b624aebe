"""Graph -> kernel plan: fusion + lowering + activation memory planning (host logic, no GPU needed).

Input: the Keras-granularity node graph of a Model (deephar_amd/graph.py).
Output: a `Plan` = ordered list of `Step`s, one per gfx950 kernel launch through the C-ABI
(include/deephar_hip.h), over `Value`s placed in one activation arena.

Fusion rules (what the reference leaves to TensorFlow as separate kernels):
  R1 prologue : BatchNormalization -> ReLU chains feeding a conv / depthwise conv are applied while loading
                the conv input (layers.py:258-301 "act_conv_bn", common.py:40-54); never materialised.
  R2 epilogue : conv -> BN -> ReLU -> add chains with a single consumer are applied on the accumulator
                (layers.py:202-241, reception.py:57,312).
  R3 upsample : add([a, UpSampling2D(b)]) (reception.py:122-127) is folded into the convolution that produces `a`, which
                reads `b` at half resolution as its second residual (dh_conv_args.res2_down); that convolution is
                emitted once `b` exists (the low-resolution branch goes first).  Saves writing `a` and reading it
                back.  Where `a` has no free residual slot: conv -> UpSampling2D -> add is written by the conv
                epilogue at 2x resolution; a free-standing UpSampling2D -> add becomes one upsample_add kernel.
  R4 concat   : producers write straight into the concatenation buffer at their channel offset
                (reception.py:75,83,87); Lambda channel slices are pointer/ld views (reception.py:171-172).
  R7 pool     : MaxPooling2D((2, 2)) of a 32-column convolution output -- [r06] also of a 16- or 8-column one -- is a second
                output of that convolution's epilogue (dh_conv_args.y_pool; reception.py:105-116, common.py:70-86).
  R9 wide add : add([...]) with four or more operands is re-associated into one add per convolution that produces an operand
                (SPNet's re-injection sum, spnet.py:233), a second two-operand add behind a residual add takes the free
                residual slot (spnet.py:303) -- no element-wise launches are left for either.
  R5 decoder  : channel soft-max + both lin_interpolation_2d + joint confidence (+ global max) on the same
                maps are one soft-argmax kernel (blocks.py:306-343).
  R12 on load : [r06] a 2x2 / stride-2 (max or max+-min) pooling, or a nearest up-sampling, whose only reader is a convolution on
                the skinny-conv kernel (the action heads: spnet.py:77-91) is not written out: that kernel resamples while it
                loads its input (dh_conv_args.x_resample).  Bit-identical.
  R11 up-unit : [r06] UpSampling2D in FRONT of a residual unit (SPNet's up-scaling unit, common.py:89-108:
                residual_unit(UpSampling2D(x)) -- BN, then a 1x1 shortcut convolution and a separable convolution, both of
                relu(BN(.))) is never written out: nearest up-sampling commutes with everything element-wise and with a
                1x1 convolution, so the shortcut runs at HALF resolution (a quarter of the work; its result is the
                half-resolution second residual of the separable convolution's pointwise half, dh_conv_args.res2_down) and
                the depthwise half reads the half-resolution tensor as if up-sampled (dh_dw_args.up_in).  Same products, same
                sums; only the order of the two residual additions changes.
  R13 pools   : [r06] two poolings with the same window whose results are concatenated right away (the action head's pose and
                appearance features, spnet.py:126-139) read ONE joint buffer their producers fill and are one launch.
  R14 segments: [r06] a pooling that fills the first channels of a concatenation whose only reader is a convolution on the
                skinny-conv kernel (the head's second residual unit, spnet.py:139-141) is not written out: that kernel reads
                concatenate([pool(x), x2]) in place (dh_conv2d_seg_f32).  Bit-identical.
  (R10 / R10b / R10c: sibling convolutions of one tensor merged into one launch -- see the passes below.)
All tensors are fp32; sizes are tracked per batch item so one plan serves any batch size.
"""
import os

import numpy as np

from .. import graph as G


class Buf:
    def __init__(self, items, kind='act'):
        self.items = int(items)       # floats per batch item
        self.kind = kind              # 'act' | 'input'
        self.start = None             # first step writing it
        self.end = None               # last step reading/writing it
        self.offset = None            # floats per batch item inside the arena
        self.pinned = False           # model output: lives to the end


class Value:
    """A view: logical shape (batch excluded), last dim = channels with pixel stride `ld`."""

    def __init__(self, shape, buf, coff=0, ld=None):
        self.shape = tuple(shape)
        self.buf = buf
        self.coff = coff
        self.ld = self.shape[-1] if ld is None else ld

    @property
    def C(self):
        return self.shape[-1]

    @property
    def npix(self):
        return int(np.prod(self.shape[:-1])) if len(self.shape) > 1 else 1

    @property
    def dense(self):
        return self.coff == 0 and self.ld == self.C

    def lead(self, nd):
        """product of the dims in front of the last `nd` dims (folded into the batch)."""
        return int(np.prod(self.shape[:-nd])) if len(self.shape) > nd else 1


def split_k_rule(out_pixels, K, cout, cin, kh=1, kw=1):
    """Mirror of conv_is_skinny (csrc/conv_splitk.hip; dh_conv2d_uses_split_k) for float inputs and fp32-packed weights:
    the layers dh_conv2d_f32 runs on its in-work-group split-K kernel -- per-frame geometry only, so a layer's bits depend
    on neither batch size nor tiling.  Every clause of the C++ rule is stated here (the two had drifted apart
    on the kernel-extent clauses); tests/test_host_logic.py sweeps both over the same shapes."""
    return out_pixels <= 256 and K >= _SKINNY_MIN_K and cout <= 256 and 2 <= cin <= 4096 and kh * kw < 256 and K * cin < (1 << 31)


_SKINNY_MIN_K = int(os.environ.get('DEEPHAR_SKINNY_MIN_K', '64'))      # (A/B aid, read once like the library's copy)


class ConcatParam:
    """Several convolution kernels over the same Cin side by side along Cout: the weight a horizontally merged convolution
    (rule R10) reads.  Looks like a graph.Param to the weight store: `value` is assembled on demand, `version` moves whenever
    a part is set.  [r06] The parts may have different (odd) extents: each is centred inside the largest [kh, kw] window,
    the taps around it are zero -- a 3x1 and a 3x3 kernel beside a 3x5 one (rule R10b)."""
    role = 'conv'

    def __init__(self, parts):
        self.parts = list(parts)
        self.key = ' + '.join(p.key for p in self.parts)
        self.name = 'kernel'
        self.kh = max(p.shape[0] for p in self.parts)
        self.kw = max(p.shape[1] for p in self.parts)
        assert all((self.kh - p.shape[0]) % 2 == 0 and (self.kw - p.shape[1]) % 2 == 0 and p.shape[2] == self.parts[0].shape[2]
                   for p in self.parts)
        self.shape = (self.kh, self.kw, self.parts[0].shape[2], sum(p.shape[-1] for p in self.parts))

    @property
    def value(self):
        if any(p.value is None for p in self.parts):
            return None
        out = np.zeros(self.shape, np.float32)
        off = 0
        for p in self.parts:
            v = np.asarray(p.value, np.float32)
            t, l = (self.kh - v.shape[0]) // 2, (self.kw - v.shape[1]) // 2
            out[t:t + v.shape[0], l:l + v.shape[1], :, off:off + v.shape[3]] = v
            off += v.shape[3]
        return out

    @property
    def version(self):
        return sum(p.version for p in self.parts)


class ConcatAffine:
    """The BatchNormalization behind a horizontally merged convolution whose parts disagree about it (rule R10c): per part
    its BatchNormalization layer, or None for a part without one (scale 1, shift 0: `t * 1 + 0` is t).  The executor's
    weight store assembles the per-column scale / shift (WeightStore.affine_concat)."""

    def __init__(self, parts):
        self.parts = list(parts)          # [(channels, graph.Layer or None)]

    @property
    def params(self):
        return [p for _, layer in self.parts if layer is not None for p in layer.params]


class Step:
    def __init__(self, kind, ins, outs, attrs=None, params=None, name=None):
        self.stream, self.deps, self.wait, self.record = 0, [], [], False   # filled by schedule.finalize
        self.kind = kind
        self.ins = ins          # dict role -> Value
        self.outs = outs        # dict role -> Value
        self.attrs = attrs or {}
        self.params = params or {}  # role -> graph.Layer / Param
        self.name = name

    def flops(self, n=1):
        """Algorithmic FLOPs (2*MAC) for n batch items (conv/GEMM-shaped steps only)."""
        a = self.attrs
        if self.kind == 'conv':
            y = self.outs['y']
            m = y.npix // (4 if a.get('up2') else 1)
            return 2.0 * n * m * a['K'] * a['Cout']
        if self.kind == 'dwconv':
            y = self.outs['y']
            return 2.0 * n * y.npix * y.C * a['kh'] * a['kw']
        if self.kind == 'kronecker':
            return 2.0 * n * self.ins['x'].npix * self.ins['hm'].C * self.ins['x'].C
        return 0.0

    def bytes(self, n=1):
        """Algorithmic HBM bytes (each operand read once, result written once)."""
        tot = 0
        for v in list(self.ins.values()) + list(self.outs.values()):
            if v is not None:
                tot += v.npix * v.C
        return 4.0 * n * tot


class Plan:
    def __init__(self):
        self.steps = []
        self.inputs = []       # Values of the model inputs
        self.outputs = []      # Values of the model outputs
        self.bufs = []
        self.arena_items = 0   # floats per batch item
        self.out_items = 0     # the model outputs occupy [0, out_items) of it (schedule.allocate)
        self.params = []       # ordered unique graph.Param list
        self.nstreams = 1
        self.gemm_precision = 'f32'   # 'bf16x3': eligible convs run split-bf16 on the bf16 matrix cores (executor)
        self.split_adds = True        # rule R9 / second-add rule applied (Planner: DEEPHAR_SPLIT_ADDS, read once)

    def total_flops(self, n=1):
        return sum(s.flops(n) for s in self.steps)


# ---------------------------------------------------------------------------------------------------------

class _Lazy:
    """A not-yet-materialised element-wise chain on top of a Value (R1)."""

    def __init__(self, base, bn=None, relu=False):
        self.base = base      # Value
        self.bn = bn          # graph.Layer or None
        self.relu = relu


class _UpView:
    """A nearest-up-sampled tensor that has not been written out (R11): `low` = the Value at half resolution, `shape` = the
    logical (up-sampled) shape, `full` = the materialised Value once some consumer needed one."""

    def __init__(self, low, shape):
        self.low, self.shape, self.full = low, tuple(shape), None


class _PoolView:
    """A MaxPooling2D((2, 2)) / max_min_pooling((2, 2)) result that has not been written out (R12): `base` = the Value at full
    resolution, `shape` = the pooled shape, `attrs` = the pooling node's attributes, `full` = the materialised Value once
    some consumer needed one."""

    def __init__(self, base, shape, attrs):
        self.base, self.shape, self.attrs, self.full = base, tuple(shape), dict(attrs), None


class Planner:
    def __init__(self, inputs, outputs, nstreams=1, stream_policy='list'):
        self.nstreams = nstreams
        self.stream_policy = stream_policy
        self.producer = {}            # id(Value) -> the Step that writes it
        self.g_inputs = inputs
        self.g_outputs = outputs
        self.nodes = G.topo_nodes(outputs)
        # R9 and the second-add rule re-associate fp32 sums (conv + a + b for the reference's a + b + conv), so the switch
        # -- it exists for A/B measurements -- is read ONCE per plan and recorded in it: a plan is built
        # entirely under one setting, and Plan.split_adds says which.
        self.split_adds = os.environ.get('DEEPHAR_SPLIT_ADDS', '1') != '0'
        if self.split_adds:
            self.nodes = self._split_wide_adds(self.nodes, outputs)
        self.plan = Plan()
        self.plan.split_adds = self.split_adds
        self.val = {}          # tensor uid -> Value | _Lazy
        self.absorbed = set()  # node uids folded into another step
        self.consumers = {}
        for n in self.nodes:
            for i, t in enumerate(n.inputs):
                self.consumers.setdefault(t.uid, []).append((n, i))
        self.out_uids = {}
        for t in outputs:
            self.out_uids[t.uid] = self.out_uids.get(t.uid, 0) + 1
        self.concat_val = {}   # concat node uid -> Value
        self.processed = set()
        self.deferred = {}     # tensor uid a deferred conv node waits for -> [node]   (R3)

    # ---- R9: n-ary adds are spread over the convolutions that produce their operands ------------------------
    @staticmethod
    def _split_wide_adds(nodes, outputs):
        """add([a, b, c, d, ...]) with four or more operands (SPNet's re-injection sum, spnet.py:233: block input + sep-conv
        output + one 1x1 convolution per prediction branch) would run as a chain of element-wise launches: a convolution's
        epilogue takes two residuals.  Re-associate it (on a copy of the node list; the model's graph is untouched) into
        one add per convolution whose ONLY reader is this sum: the first takes up to two of the other operands, each further
        one the running partial sum (+ one more operand) -- every partial add is then the epilogue of its convolution (R2)
        and no element-wise launch is left.  fp32 addition is re-ordered (conv + a + b instead of a + b + conv): within the
        rounding the 2-residual epilogue of R2 already has."""
        index = {n.uid: i for i, n in enumerate(nodes)}
        readers = {}
        for n in nodes:
            for t in n.inputs:
                readers[t.uid] = readers.get(t.uid, 0) + 1
        for t in outputs:
            readers[t.uid] = readers.get(t.uid, 0) + 1
        insert_after, replace = {}, {}
        for n in nodes:
            if n.op != 'add' or len(n.inputs) < 4 or len({t.uid for t in n.inputs}) != len(n.inputs):
                continue
            prods = [t for t in n.inputs if t.node is not None and t.node.op in ('conv', 'sepconv') and
                     t.node.uid in index and readers.get(t.uid, 0) == 1]
            if not prods:
                continue
            prods.sort(key=lambda t: index[t.node.uid])
            others = [t for t in n.inputs if all(t.uid != q.uid for q in prods)]
            pos = lambda t: index[t.node.uid] if t.node is not None and t.node.uid in index else -1
            partial = None
            for k, p in enumerate(prods):
                last = k == len(prods) - 1
                take = [partial] if partial is not None else []
                ready = [t for t in others if pos(t) < index[p.node.uid]]
                while len(take) < 2 and ready:
                    take.append(ready.pop(0))
                others = [t for t in others if all(t.uid != q.uid for q in take)]
                ins = [p] + take + (others if last else [])
                if not last and len(ins) == 1:            # nothing to add yet: this producer joins the next partial sum
                    others.append(p)
                    continue
                node = G.Node('add', ins, [n.outputs[0].shape], name=(n.name or 'add') + ('' if last else '/part%d' % k))
                if last:
                    node.outputs = [n.outputs[0]]          # downstream readers keep their tensor
                    replace[n.uid] = node
                else:
                    insert_after.setdefault(p.node.uid, []).append(node)
                partial = node.outputs[0]
        if not replace:
            return nodes
        out = []
        for n in nodes:
            out.append(replace.get(n.uid, n))
            out.extend(insert_after.get(n.uid, []))
        return out

    # ---- helpers ------------------------------------------------------------------------------------
    def new_buf(self, shape, kind='act'):
        items = int(np.prod(shape))
        items = (items + 3) // 4 * 4
        b = Buf(items, kind)
        self.plan.bufs.append(b)
        return b

    def new_value(self, shape):
        return Value(shape, self.new_buf(shape))

    def n_consumers(self, t):
        return len(self.consumers.get(t.uid, [])) + self.out_uids.get(t.uid, 0)

    def sole_consumer(self, t, op=None):
        """The single consumer node of t (None if several, or t is a model output)."""
        if self.out_uids.get(t.uid, 0):
            return None
        cs = self.consumers.get(t.uid, [])
        if len(cs) != 1:
            return None
        n = cs[0][0]
        if op is not None and n.op != op:
            return None
        return n

    def emit(self, kind, ins, outs, attrs=None, params=None, name=None):
        s = Step(kind, ins, outs, attrs, params, name)
        self.plan.steps.append(s)
        for v in outs.values():
            if v is not None:
                self.producer[id(v)] = s
        return s

    def available(self, t):
        return t.uid in self.val

    def materialize(self, t):
        """Value for tensor t, running a pending element-wise chain if needed."""
        v = self.val[t.uid]
        if isinstance(v, _UpView):                       # a consumer that needs the up-sampled tensor in memory
            v = self.val[t.uid] = self._realize_up(v)
        if isinstance(v, _PoolView):
            v = self.val[t.uid] = self._realize_pool(v)
        if isinstance(v, _Lazy) and isinstance(v.base, _UpView):
            v = _Lazy(self._realize_up(v.base), bn=v.bn, relu=v.relu)
        if isinstance(v, _Lazy) and isinstance(v.base, _PoolView):
            v = _Lazy(self._realize_pool(v.base), bn=v.bn, relu=v.relu)
        if isinstance(v, _Lazy):
            out = self.new_value(v.base.shape)
            self.emit('eltwise', dict(a=v.base), dict(y=out), dict(op=0, relu=int(v.relu)),
                      dict(bn=v.bn) if v.bn is not None else {}, name='materialize')
            self.val[t.uid] = out
            return out
        return v

    def lazy_or_value(self, t):
        return self.val[t.uid]

    def _realize_up(self, uv):
        """Write a virtual up-sampled tensor out after all (once): the stand-alone up-sampling launch."""
        if uv.full is None:
            uv.full = self.new_value(uv.shape)
            self.emit('upsample_add', dict(b=uv.low), dict(y=uv.full), name='upsample')
        return uv.full

    def _realize_pool(self, pv):
        """Write a virtual pooled tensor out after all (once): the stand-alone pooling launch."""
        if pv.full is None:
            pv.full = self.new_value(pv.shape)
            self.emit('pool', dict(x=pv.base), dict(y=pv.full), dict(pv.attrs), name='pool')
        return pv.full

    def _skinny_conv_node(self, n, cin):
        """True when conv node n (stride 1) runs on the skinny-conv kernel, which can resample its input on load (R12)."""
        a, shape = n.attrs, n.outputs[0].shape
        return n.op == 'conv' and len(shape) >= 3 and (a.get('sh', 1), a.get('sw', 1)) == (1, 1) and \
            split_k_rule(shape[-3] * shape[-2], a['kh'] * a['kw'] * cin, a['filters'], cin, a['kh'], a['kw'])

    def _only_reader_through_bn_relu(self, t):
        """The single node that reads t through nothing but BatchNormalization / ReLU (each with one reader), or None."""
        while True:
            if self.out_uids.get(t.uid, 0):
                return None
            cs = self.consumers.get(t.uid, [])
            if len(cs) != 1:
                return None
            n = cs[0][0]
            if n.op not in ('bn', 'relu'):
                return n
            t = n.outputs[0]

    # ---- R11: UpSampling2D in front of a residual unit is not written out -----------------------------------------
    def _virtual_upsample_ok(self, node):
        """True when every reader of this UpSampling2D((2, 2)) output reaches, through BatchNormalization / ReLU only, a
        plain 1x1 convolution or a separable convolution: those read the half-resolution tensor directly (the 1x1
        convolution runs BEFORE the up-sampling, the depthwise convolution up-samples on load).  DEEPHAR_UP_COMMUTE=0: off."""
        t = node.outputs[0]
        if os.environ.get('DEEPHAR_UP_COMMUTE', '1') == '0' or len(t.shape) < 3 or node.attrs.get('size', (2, 2)) != (2, 2):
            return False
        seen, stack, ends = set(), [t], 0
        while stack:
            x = stack.pop()
            if self.out_uids.get(x.uid, 0):
                return False
            for n, _ in self.consumers.get(x.uid, []):
                if n.op in ('bn', 'relu'):
                    if n.uid not in seen:
                        seen.add(n.uid)
                        stack.append(n.outputs[0])
                elif n.op == 'conv' and (n.attrs['kh'], n.attrs['kw'], n.attrs.get('sh', 1), n.attrs.get('sw', 1)) == (1, 1, 1, 1):
                    ends += 1
                elif self._skinny_conv_node(n, t.shape[-1]) and os.environ.get('DEEPHAR_RESAMPLE_ON_LOAD', '1') != '0':
                    ends += 1          # R12: the skinny-conv kernel up-samples on load (dh_conv_args.x_resample = 1)
                elif n.op == 'sepconv' and (n.attrs.get('sh', 1), n.attrs.get('sw', 1)) == (1, 1):
                    ends += 1
                else:
                    return False
        return ends > 0

    # consumers that read a tensor through a (pointer, pixel pitch) view whatever its pitch: they may share a producer's
    # output with ONE concatenate, which then needs no copy (R4b).  Convolutions are NOT in the list: no
    # shipped model needs it, and a 1x1 convolution on the LDS-DMA GEMM whose Cin is not a multiple of its K-step fills
    # the padded k slots of a pixel from the floats that follow it in memory -- inside a shared slab that is the
    # neighbouring tensor, possibly not written yet (0 * NaN from an uninitialised arena)
    _VIEW_READERS = ('softmax2d', 'jointprob', 'globalmax2d')

    def _concat_home(self, t):
        """The concatenate whose buffer t's producer writes into (R4): t's only consumer -- or [r05, R4b] the only
        concatenate among consumers that all read strided views.  SPNet's heat-map head (spnet.py:24-48): `pred_maps` goes
        to the channel soft-max AND into concatenate([fw_maps, pred_maps]); written where the concatenation wants it, the
        soft-argmax kernel reads it there (pitch 2 J) and the copy launch disappears.  DEEPHAR_CONCAT_SHARED=0: off."""
        cat = self.sole_consumer(t, 'concat')
        if cat is not None or self.out_uids.get(t.uid, 0) or os.environ.get('DEEPHAR_CONCAT_SHARED', '1') == '0':
            return cat
        cs = [n for n, _ in self.consumers.get(t.uid, [])]
        cats = [n for n in cs if n.op == 'concat']
        if len(cats) != 1 or any(n.op not in self._VIEW_READERS for n in cs if n.op != 'concat'):
            return None
        return cats[0]

    def out_value_for(self, t, shape=None):
        """Where the producer of t should write: inside a concat buffer if t's only consumer is a concat."""
        shape = shape or t.shape
        cat = self._concat_home(t)
        if cat is not None and all(len(x.shape) == len(shape) for x in cat.inputs):
            cv = self.concat_val.get(cat.uid)
            if cv is None:
                target = self.out_value_for(cat.outputs[0])
                cv = target
                self.concat_val[cat.uid] = cv
            off = 0
            for x in cat.inputs:
                if x.uid == t.uid:
                    break
                off += x.shape[-1]
            # the same tensor listed twice in one concat cannot be a view
            if sum(1 for x in cat.inputs if x.uid == t.uid) == 1:
                return Value(shape, cv.buf, cv.coff + off, cv.ld)
        return self.new_value(shape)

    # ---- main loop ------------------------------------------------------------------------------------
    def run(self):
        for t in self.g_inputs:
            v = Value(t.shape, self.new_buf(t.shape, 'input'))
            self.val[t.uid] = v
            self.plan.inputs.append(v)
        for node in self.nodes:
            if node.uid in self.absorbed:
                continue
            if node.op in ('conv', 'sepconv') and self._defer_for_upsampled_residual(node):
                continue
            getattr(self, 'op_' + node.op)(node)
            self.processed.add(node.uid)
            self._flush_deferred()
        assert not self.deferred, 'deferred convolutions never became ready'
        for t in self.g_outputs:
            v = self.materialize(t)
            v.buf.pinned = True
            self.plan.outputs.append(v)
        self._merge_sibling_pointwise()
        self._merge_siblings_into_joint_buffers()
        self._merge_sibling_pools()
        self._pool_into_segmented_conv()
        self._collect_params()
        from . import schedule
        schedule.finalize(self.plan, self.nstreams, self.stream_policy)
        return self.plan

    # ---- R10: sibling 1x1 convolutions that fill neighbouring channel slabs of one buffer ------------------------
    def _merge_sibling_pointwise(self):
        """[r05] SPNet's heat-map head (spnet.py:24-48): `_fw_maps` and `_conv1` are two 1x1 convolutions of the SAME activated
        tensor, num_joints columns each, concatenated right away -- R4 already makes them write neighbouring channel slabs of
        the concatenation buffer.  They become ONE launch over both weight matrices side by side: 16 + 16 (17 + 17) columns
        are one 32-column tile (two 16-column tiles on the skinny kernel), each output column keeps its own K order, so the
        result is bit-identical and a launch is saved per prediction block.  (Round 4 tried the same on the 64 -> 96 | 48
        pairs of the entry flow -- widths that tile worse together than apart; this rule only merges slabs that stay
        inside the family rule of their parts and carry no epilogue.)  DEEPHAR_MERGE_HEADS=0 switches it off."""
        if os.environ.get('DEEPHAR_MERGE_HEADS', '1') == '0':
            return
        same = ('sh', 'sw', 'Cin', 'pre_relu', 'post_relu', 'up2', 'res2_down')

        def mergeable(a, b):
            if a.kind != 'conv' or b.kind != 'conv' or set(a.ins) != {'x'} or set(b.ins) != {'x'} or \
                    set(a.outs) != {'y'} or set(b.outs) != {'y'}:
                return False
            if any(a.attrs.get(k) != b.attrs.get(k) for k in same) or \
                    a.attrs['up2'] or a.attrs['res2_down'] or a.attrs.get('sh', 1) != 1 or a.attrs.get('sw', 1) != 1:
                return False
            pointwise = all(st.attrs['kh'] == 1 and st.attrs['kw'] == 1 for st in (a, b))
            if not pointwise:
                # [r06, R10b] K x K siblings of one tensor -- the three bare convolutions over the (T, J) plane that open
                # every action head, spnet.py:109-112: 3x1, 3x3, 3x5 into one concatenation -- run as ONE convolution with
                # the largest window, the smaller kernels centred in it between zero taps (ConcatParam): every product
                # beside the part's own taps is an exact zero, the taps keep their relative order, so each output column
                # still sees its own chain of fused multiply-adds.  Odd kernels under symmetric 'SAME' padding only, on the
                # general kernel (Cin not a multiple of 16: the K x K families order K by chunks of channels).
                for st in (a, b):
                    if st.attrs['kh'] % 2 == 0 or st.attrs['kw'] % 2 == 0 or st.attrs['pt'] != st.attrs['kh'] // 2 or \
                            st.attrs['pl'] != st.attrs['kw'] // 2 or st.attrs['Cin'] % 16 == 0:
                        return False
                if os.environ.get('DEEPHAR_MERGE_KXK', '1') == '0':
                    return False
            if set(a.params) - {'w', 'pre_bn'} or set(b.params) - {'w', 'pre_bn'} or \
                    a.params.get('pre_bn') is not b.params.get('pre_bn'):
                return False
            xa, xb, ya, yb = a.ins['x'], b.ins['x'], a.outs['y'], b.outs['y']
            if xa.buf is not xb.buf or (xa.coff, xa.ld, xa.shape) != (xb.coff, xb.ld, xb.shape):
                return False
            if ya.buf is not yb.buf or ya.ld != yb.ld or ya.shape[:-1] != yb.shape[:-1] or ya.coff + ya.C != yb.coff:
                return False
            px = ya.shape[-3] * ya.shape[-2] if len(ya.shape) >= 3 else 1
            kh, kw = max(a.attrs['kh'], b.attrs['kh']), max(a.attrs['kw'], b.attrs['kw'])
            fam = lambda st, c, h, w_: split_k_rule(px, h * w_ * st.attrs['Cin'], c, st.attrs['Cin'], h, w_)
            return fam(a, ya.C, a.attrs['kh'], a.attrs['kw']) == fam(b, yb.C, b.attrs['kh'], b.attrs['kw']) == \
                fam(a, ya.C + yb.C, kh, kw)

        steps = self.plan.steps
        i = 0
        while i + 1 < len(steps):
            a = steps[i]
            # the sibling may sit a few steps further down (the soft-argmax read-out of the first head is emitted between
            # them): it only reads x, which exists before a, and buffers are still logical here (no re-use yet), so it can
            # be pulled up to a's place
            j = next((j for j in range(i + 1, min(i + 8, len(steps))) if mergeable(a, steps[j]) or mergeable(steps[j], a)), None)
            if j is None:
                i += 1
                continue
            b = steps[j]
            first, second = (a, b) if mergeable(a, b) else (b, a)
            ya, yb = first.outs['y'], second.outs['y']
            y = Value(ya.shape[:-1] + (ya.C + yb.C,), ya.buf, ya.coff, ya.ld)
            parts = []
            for st in (first, second):
                w = st.params['w']
                parts += w.parts if isinstance(w, ConcatParam) else [w]
            wcat = ConcatParam(parts)
            params = dict(first.params, w=wcat)
            attrs = dict(first.attrs, Cout=ya.C + yb.C, kh=wcat.kh, kw=wcat.kw, pt=wcat.kh // 2, pl=wcat.kw // 2,
                         K=wcat.kh * wcat.kw * first.attrs['Cin'])
            if (wcat.kh, wcat.kw) == (1, 1):
                attrs.update(pt=first.attrs['pt'], pl=first.attrs['pl'])
            merged = Step('conv', dict(first.ins), dict(y=y), attrs, params, '%s+%s' % (first.name, second.name))
            del steps[j]
            steps[i] = merged
            for v in (ya, yb, y):
                self.producer[id(v)] = merged

    # ---- R10c: sibling 1x1 convolutions with outputs of their own become one launch into a joint buffer ----------------
    def _merge_siblings_into_joint_buffers(self):
        """[r06] Two plain 1x1 convolutions of the SAME tensor under the SAME prologue whose results are separate tensors --
        the shortcut and the first convolution of a 'normal' residual unit (common.py:33-52: `shortcut = conv2d(relu(BN(x)))`
        beside `conv2d(relu(BN(x)), out / div)`), the replica heat-map head beside the forward / heat-map pair (spnet.py:
        32-38) -- are ONE launch over both weight matrices into a joint buffer; every reader takes its channel run as a
        (pointer, pitch) view.  The parts may disagree about what follows them: a BatchNormalization becomes a per-column
        affine of the merged launch (identity columns for the part without one, ConcatAffine), and a ReLU that only one part
        has moves into the prologue of that part's readers (convolutions reading it as their input: relu on load = relu on
        store).  Same products, same chains: bit-identical.  Only inside one kernel family (the skinny-conv rule must say
        the same for the parts and the whole: round 4 measured that the wide pairs of the entry flow tile worse together
        than apart).  DEEPHAR_MERGE_SIBLINGS=0 switches it off."""
        if os.environ.get('DEEPHAR_MERGE_SIBLINGS', '1') == '0':
            return
        steps = self.plan.steps

        def plain(st):
            return st.kind == 'conv' and set(st.ins) == {'x'} and set(st.outs) == {'y'} and \
                (st.attrs['kh'], st.attrs['kw'], st.attrs.get('sh', 1), st.attrs.get('sw', 1)) == (1, 1, 1, 1) and \
                not st.attrs['up2'] and not st.attrs['res2_down'] and not (set(st.params) - {'w', 'pre_bn', 'post_bn'})

        def readers(buf):
            return [(q, role, v) for q in steps for role, v in q.ins.items() if v is not None and v.buf is buf]

        def family(st, c):
            y = st.outs['y']
            px = y.shape[-3] * y.shape[-2] if len(y.shape) >= 3 else 1
            return split_k_rule(px, st.attrs['K'], c, st.attrs['Cin'])

        def compatible(a, b):
            if not plain(a) or not plain(b) or a.params.get('pre_bn') is not b.params.get('pre_bn') or \
                    a.attrs['pre_relu'] != b.attrs['pre_relu'] or a.attrs['Cin'] != b.attrs['Cin']:
                return False
            xa, xb, ya, yb = a.ins['x'], b.ins['x'], a.outs['y'], b.outs['y']
            if xa.buf is not xb.buf or (xa.coff, xa.ld, xa.shape) != (xb.coff, xb.ld, xb.shape):
                return False
            if ya.buf is yb.buf or ya.shape[:-1] != yb.shape[:-1] or ya.buf.pinned or yb.buf.pinned:
                return False
            for st in (a, b):                 # each part fills ITS buffer completely, and nobody else writes into it
                y = st.outs['y']
                if y.coff != 0 or y.ld != y.C or y.buf.items != (int(np.prod(y.shape)) + 3) // 4 * 4:
                    return False
                if any(v is not None and v.buf is y.buf for q in steps if q is not st for v in q.outs.values()):
                    return False
            # this pass is for the skinny-conv family (16-column tiles: any widths tile together as well as apart)
            # (round 6 re-measured the wide case on the one pair whose joint width is a whole number of 32-column tiles,
            # SPNet-NTU's res1 shortcut | conv1 = 192 + 96 columns at 64 x 64 x 256 frames: 613.7 us merged against 392.3 +
            # 214.8 us apart, 22.26 against 22.21 ms per step -- the second read of the input comes from the memory-side cache)
            if not (family(a, ya.C) and family(b, yb.C) and family(a, ya.C + yb.C)):
                return False
            for st in (a, b):                 # readers must take (pointer, pitch) views
                for q, role, v in readers(st.outs['y'].buf):
                    if q.kind not in ('conv', 'sam'):
                        return False
            if a.attrs['post_relu'] != b.attrs['post_relu']:      # the ReLU of one part moves to that part's readers
                st = a if a.attrs['post_relu'] else b
                for q, role, v in readers(st.outs['y'].buf):
                    if q.kind != 'conv' or role != 'x' or 'pre_bn' in q.params:
                        return False
            return True

        # candidates share the input view and the prologue: only those are compared (a sibling may sit far down the list --
        # the replica head is emitted with its action head -- and is pulled up to the first one's place: it reads nothing
        # but x, which exists there)
        def key(st):
            x = st.ins['x']
            return (id(x.buf), x.coff, x.ld, x.shape, id(st.params.get('pre_bn')), st.attrs['pre_relu'])

        i = 0
        while i + 1 < len(steps):
            a = steps[i]
            j = None
            if plain(a):
                ka = key(a)
                j = next((j for j in range(i + 1, len(steps)) if plain(steps[j]) and key(steps[j]) == ka and
                          compatible(a, steps[j])), None)
            if j is None:
                i += 1
                continue
            b = steps[j]
            ya, yb = a.outs['y'], b.outs['y']
            ca, cb = ya.C, yb.C
            post_relu = int(a.attrs['post_relu'] and b.attrs['post_relu'])
            if a.attrs['post_relu'] != b.attrs['post_relu']:
                st = a if a.attrs['post_relu'] else b
                for q, role, v in readers(st.outs['y'].buf):
                    q.attrs['pre_relu'] = 1
            joint = self.new_buf(ya.shape[:-1] + (ca + cb,))
            for old, base in ((ya.buf, 0), (yb.buf, ca)):
                for q in steps:
                    for v in list(q.ins.values()) + list(q.outs.values()):
                        if v is not None and v.buf is old:
                            v.buf, v.coff, v.ld = joint, v.coff + base, ca + cb
                self.plan.bufs.remove(old)
            parts = []
            for st in (a, b):
                w = st.params['w']
                parts += w.parts if isinstance(w, ConcatParam) else [w]
            params = dict(w=ConcatParam(parts))
            if 'pre_bn' in a.params:
                params['pre_bn'] = a.params['pre_bn']
            if 'post_bn' in a.params or 'post_bn' in b.params:
                params['post_affine'] = ConcatAffine([(ca, a.params.get('post_bn')), (cb, b.params.get('post_bn'))])
            y = Value(ya.shape[:-1] + (ca + cb,), joint, 0, ca + cb)
            merged = Step('conv', dict(a.ins), dict(y=y), dict(a.attrs, Cout=ca + cb, post_relu=post_relu), params,
                          '%s+%s' % (a.name, b.name))
            del steps[j]
            steps[i] = merged
            for v in (ya, yb, y):
                self.producer[id(v)] = merged

    # ---- R13: sibling poolings that fill neighbouring channel slabs of one buffer ---------------------------------
    def _merge_sibling_pools(self):
        """[r06] The action head pools its pose features and its appearance features with the same window and concatenates the
        results right away (spnet.py:126-139: `x1 = maxpooling2d(x, (2, 2), strides=(time_stride, 2))`, `x2 = maxpooling2d(...)`,
        `concat_tensorlist([x1, x2, xa])`): two launches over two [frames, joints, 160] tensors.  Their producers write the two
        channel runs of ONE joint buffer instead and a single pooling launch covers both -- pooling is per channel, so every
        output element is the maximum of the same values.  Bit-identical; one launch less per action head (the latency
        regime: a node costs ~5 us whatever it moves).  DEEPHAR_MERGE_POOLS=0 switches it off."""
        if os.environ.get('DEEPHAR_MERGE_POOLS', '1') == '0':
            return
        steps = self.plan.steps

        def sole_buffer(st):
            """the pool's input fills its own buffer, one step writes it, the pool alone reads it"""
            x = st.ins['x']
            if x.coff != 0 or x.ld != x.C or x.buf.pinned or x.buf.kind != 'act' or \
                    x.buf.items != (int(np.prod(x.shape)) + 3) // 4 * 4:
                return False
            writers = [q for q in steps for v in q.outs.values() if v is not None and v.buf is x.buf]
            readers = [q for q in steps for v in q.ins.values() if v is not None and v.buf is x.buf]
            # (the producer must be able to write a channel run of a wider buffer: convolutions do -- ZeroPadding2D, in
            #  front of the pooling when the joint count is not a multiple of four, writes dense rows)
            return len(writers) == 1 and readers == [st] and writers[0].kind == 'conv' and \
                writers[0].outs.get('y') is not None and writers[0].outs['y'].buf is x.buf and \
                all(v.coff == 0 and v.ld == x.C for v in writers[0].outs.values() if v is not None and v.buf is x.buf)

        i = 0
        while i < len(steps):
            a = steps[i]
            j = None
            if a.kind == 'pool' and sole_buffer(a):
                xa, ya = a.ins['x'], a.outs['y']
                for jj in range(i + 1, len(steps)):
                    b = steps[jj]
                    if b.kind != 'pool' or b.attrs != a.attrs:
                        continue
                    xb, yb = b.ins['x'], b.outs['y']
                    if yb.buf is ya.buf and yb.ld == ya.ld and yb.coff == ya.coff + ya.C and yb.shape[:-1] == ya.shape[:-1] and \
                            xb.shape[:-1] == xa.shape[:-1] and xb.buf is not xa.buf and sole_buffer(b) and \
                            not any(v is not None and v.buf is ya.buf for q in steps[i + 1:jj] for v in q.ins.values()):
                        j = jj
                        break
            if j is None:
                i += 1
                continue
            b = steps[j]
            xb, yb = b.ins['x'], b.outs['y']
            ca, cb = xa.C, xb.C
            joint = self.new_buf(xa.shape[:-1] + (ca + cb,))
            for old, base in ((xa.buf, 0), (xb.buf, ca)):
                for q in steps:
                    for v in list(q.ins.values()) + list(q.outs.values()):
                        if v is not None and v.buf is old:
                            v.buf, v.coff, v.ld = joint, v.coff + base, ca + cb
                self.plan.bufs.remove(old)
            x = Value(xa.shape[:-1] + (ca + cb,), joint, 0, ca + cb)
            y = Value(ya.shape[:-1] + (ya.C + yb.C,), ya.buf, ya.coff, ya.ld)
            merged = Step('pool', dict(x=x), dict(y=y), dict(a.attrs), {}, '%s+%s' % (a.name, b.name))
            # at the LATER pool's place: both producers have run by then, and nothing in between reads the first slab
            steps[j] = merged
            del steps[i]
            for v in (ya, yb, y):
                self.producer[id(v)] = merged
            # (the merged step may pair again with a third sibling further down; i now holds the next step)

    # ---- R14: a pooling that fills the first channels of a concatenation read by one skinny convolution -------------
    def _pool_into_segmented_conv(self):
        """[r06] `concat_tensorlist([x1, x2, xa])` in front of the action head's second residual unit (spnet.py:126-141): x1, x2
        are poolings (one launch out of a joint buffer after R13), xa the features of the previous head; the only reader is
        the unit's merged shortcut | conv1 launch (R10c), a layer of the skinny-conv kernel.  That kernel takes the maximum of
        the window while it gathers the first c_split channels of its input and reads the rest where they are
        (dh_conv2d_seg_f32): the pooled tensor is never written, one launch less per head.  Bit-identical.
        DEEPHAR_POOL_SEGMENTS=0 switches it off."""
        if os.environ.get('DEEPHAR_POOL_SEGMENTS', '1') == '0':
            return
        steps = self.plan.steps
        i = 0
        while i < len(steps):
            p = steps[i]
            i += 1
            a = p.attrs
            if p.kind != 'pool' or a.get('mode', 0) != 0 or (a['kh'], a['kw'], a['sw'], a['pt'], a['pl']) != (2, 2, 2, 0, 0) or \
                    a['sh'] not in (1, 2):
                continue
            x, y = p.ins['x'], p.outs['y']
            if len(x.shape) < 3 or x.coff != 0 or x.ld != x.C or x.buf.pinned or y.coff != 0 or y.buf.pinned or x.C % 4 or \
                    x.shape[-2] != 2 * y.shape[-2] or x.shape[-3] != a['sh'] * y.shape[-3]:
                continue
            if any(v is not None and v.buf is x.buf for q in steps if q is not p for v in q.ins.values()):
                continue                                   # somebody else reads the un-pooled tensor
            readers = [(q, r) for q in steps for r, v in q.ins.items() if v is not None and v.buf is y.buf]
            if len(readers) != 1 or readers[0][1] != 'x':
                continue
            q = readers[0][0]
            cx = q.ins['x']
            qa = q.attrs
            if q.kind != 'conv' or qa.get('x_resample') or qa.get('up2') or qa.get('res2_down') or 'ypool' in q.outs or \
                    cx.coff != 0 or cx.ld != y.ld or cx.C < y.C or cx.shape[:-1] != y.shape[:-1] or qa['Cin'] != cx.C:
                continue
            oy = q.outs['y']
            if not split_k_rule(oy.shape[-3] * oy.shape[-2] if len(oy.shape) >= 3 else 1, qa['K'], qa['Cout'], qa['Cin'], qa['kh'],
                                qa['kw']):
                continue
            # the other writers of the concatenation fill channels behind the pooled run (or there are none)
            others = [v for w in steps if w is not p for v in w.outs.values() if v is not None and v.buf is y.buf]
            if any(v.coff < y.C or v.ld != y.ld for v in others) or (cx.C > y.C) != bool(others) or (cx.C - y.C) % 4:
                continue
            q.ins['x'] = x
            if cx.C > y.C:
                q.ins['x2'] = Value(cx.shape[:-1] + (cx.C - y.C,), y.buf, y.C, y.ld)
            qa['seg'] = dict(c_split=y.C, pool_sh=a['sh'])
            steps.remove(p)
            i -= 1

    # ---- R3: add([a, UpSampling2D(b)]) as the second residual of the convolution that produces a --------------
    def _upsampled_residual(self, t):
        """t = a convolution's output after its own BN / first residual add.  -> (add node, upsample node, b) when t's
        only consumer is add([t, UpSampling2D(b)]) with b at half resolution and the same channels, else None."""
        a2 = self.sole_consumer(t, 'add')
        if a2 is None or len(a2.inputs) != 2 or len(t.shape) < 3 or os.environ.get('DEEPHAR_RES2_DOWN', '1') == '0':
            return None                          # (the switch exists for A/B measurements of the rule)
        other = [x for x in a2.inputs if x.uid != t.uid]
        if len(other) != 1:
            return None
        up = next((n for n in self.nodes if n.op == 'upsample' and n.outputs[0].uid == other[0].uid), None)
        if up is None or self.sole_consumer(up.outputs[0], 'add') is not a2:
            return None
        b = up.inputs[0]
        if tuple(b.shape[:-3]) != tuple(t.shape[:-3]) or b.shape[-1] != t.shape[-1] or t.shape[-3] % 2 or \
                t.shape[-2] % 2 or (b.shape[-3], b.shape[-2]) != (t.shape[-3] // 2, t.shape[-2] // 2):
            return None
        return a2, up, b

    def _epilogue_tail(self, out_t):
        """Side-effect free walk conv -> bn -> (add with ONE available other) like _epilogue; -> (the tensor after it, the
        nodes walked over), tensor None when a ReLU or a two-residual add makes the second residual slot unavailable."""
        t, chain = out_t, []
        n = self.sole_consumer(t, 'bn')
        if n is not None:
            t = n.outputs[0]
            chain.append(n)
        if self.sole_consumer(t, 'relu') is not None:
            return None, chain
        n = self.sole_consumer(t, 'add')
        if n is not None:
            others = [x for x in n.inputs if x.uid != t.uid]
            if len(others) == len(n.inputs) - 1 and len(others) == 1 and all(self.available(x) for x in others):
                return n.outputs[0], chain + [n]
            if self._upsampled_residual(t) is None:
                return None, chain
        return t, chain

    def _defer_for_upsampled_residual(self, node):
        """True (and the node is parked) when this convolution can take add([., UpSampling2D(b)]) as its second residual
        but b does not exist yet: it is emitted right after the step that produces b."""
        if not all(self.available(x) for x in node.inputs):
            return False
        if self._node_is_skinny(node):
            return False
        t, chain = self._epilogue_tail(node.outputs[0])
        hit = self._upsampled_residual(t) if t is not None else None
        if hit is None or self.available(hit[2]):
            return False
        self.deferred.setdefault(hit[2].uid, []).append(node)
        # the epilogue chain (absorbed again, for good, when the convolution is emitted) and the UpSampling2D / add pair
        # must not run on their own in the meantime
        self.absorbed.update(n.uid for n in chain + [hit[0], hit[1]])
        return True

    def _flush_deferred(self):
        ready = [uid for uid in self.deferred if uid in self.val]
        while ready:
            for uid in ready:
                for node in self.deferred.pop(uid):
                    getattr(self, 'op_' + node.op)(node)
                    self.processed.add(node.uid)
            ready = [uid for uid in self.deferred if uid in self.val]

    # ---- element-wise laziness (R1) --------------------------------------------------------------------
    def op_bn(self, node):
        src = self.val[node.inputs[0].uid]
        if isinstance(src, _Lazy):
            src = self.materialize(node.inputs[0])
        self.val[node.outputs[0].uid] = _Lazy(src, bn=node.layers['bn'], relu=False)

    def op_relu(self, node):
        src = self.val[node.inputs[0].uid]
        if isinstance(src, _Lazy):
            if src.relu:
                self.val[node.outputs[0].uid] = src
            else:
                self.val[node.outputs[0].uid] = _Lazy(src.base, bn=src.bn, relu=True)
        else:
            self.val[node.outputs[0].uid] = _Lazy(src, relu=True)

    # ---- convolutions ------------------------------------------------------------------------------------
    def _prologue(self, t):
        v = self.val[t.uid]
        if isinstance(v, _Lazy):
            return v.base, v.bn, v.relu
        return v, None, False

    @staticmethod
    def _node_is_skinny(node):
        """True when dh_conv2d_f32 runs this conv / sep-conv node's (pointwise) convolution on the split-K kernel, whose
        epilogue has no half-resolution residual (conv_igemm.hip: conv_is_skinny + res2_down -> DH_EUNSUPPORTED)."""
        shape, a = node.outputs[0].shape, node.attrs
        if len(shape) < 3:
            return False
        cin = node.inputs[0].shape[-1]
        K = cin if node.op == 'sepconv' else a['kh'] * a['kw'] * cin
        kh, kw = (1, 1) if node.op == 'sepconv' else (a['kh'], a['kw'])
        return split_k_rule(shape[-3] * shape[-2], K, a['filters'], cin, kh, kw)

    def _epilogue(self, out_t, skinny=False):
        """Walk conv -> bn -> relu -> add -> (upsample -> add) while each link has a single consumer.  `skinny`: the
        convolution runs on the split-K kernel, which cannot read a half-resolution residual (R3 falls back to the
        up-sampling epilogue / upsample_add)."""
        epi = dict(post_bn=None, post_relu=False, res1=None, res2=None, up2=False, res2_down=False)
        t = out_t
        n = self.sole_consumer(t, 'bn')
        if n is not None:
            epi['post_bn'] = n.layers['bn']
            self.absorbed.add(n.uid)
            t = n.outputs[0]
        n = self.sole_consumer(t, 'relu')
        if n is not None:
            epi['post_relu'] = True
            self.absorbed.add(n.uid)
            t = n.outputs[0]
        def take(x):
            """Residual operand x into a free slot: a tensor that only exists at half resolution (R11: the shortcut of an
            up-scaling unit) becomes the half-resolution second residual, anything else goes to res1, then res2."""
            v = self.val.get(x.uid)
            if isinstance(v, _UpView) and v.full is None and epi['res2'] is None and not skinny and len(x.shape) >= 3 and \
                    x.shape[-3] % 2 == 0 and x.shape[-2] % 2 == 0:
                epi['res2'], epi['res2_down'] = v.low, True
                return
            val = self.materialize(x)
            if epi['res1'] is None:
                epi['res1'] = val
            else:
                assert epi['res2'] is None
                epi['res2'] = val

        if not epi['post_relu']:
            n = self.sole_consumer(t, 'add')
            if n is not None:
                others = [x for x in n.inputs if x.uid != t.uid]
                if len(others) == len(n.inputs) - 1 and 1 <= len(others) <= 2 and \
                        all(self.available(x) for x in others):
                    for x in sorted(others, key=lambda x: not isinstance(self.val.get(x.uid), _UpView)):
                        take(x)                 # (a half-resolution operand first: it can only go to the second slot)
                    self.absorbed.add(n.uid)
                    t = n.outputs[0]
            if epi['res2'] is None and not skinny:
                hit = self._upsampled_residual(t)
                if hit is not None and self.available(hit[2]):
                    a2, up, b = hit
                    epi['res2'] = self.materialize(b)
                    epi['res2_down'] = True
                    self.absorbed.update((a2.uid, up.uid))
                    t = a2.outputs[0]
            if (epi['res1'] is None) != (epi['res2'] is None) and self.split_adds:
                # a second two-operand add behind the first (SPNet: add([residual_unit(x), lateral]), spnet.py:303 on top of
                # common.py:67): the free residual slot takes it -- (conv + shortcut) + lateral, the reference's order (with
                # a half-resolution shortcut in the second slot, R11: (conv + lateral) + shortcut)
                n2 = self.sole_consumer(t, 'add')
                if n2 is not None and len(n2.inputs) == 2:
                    other = [x for x in n2.inputs if x.uid != t.uid]
                    if len(other) == 1 and self.available(other[0]):
                        take(other[0])
                        self.absorbed.add(n2.uid)
                        t = n2.outputs[0]
            if epi['res2'] is None:
                u = self.sole_consumer(t, 'upsample')
                if u is not None and len(t.shape) >= 3:
                    a = self.sole_consumer(u.outputs[0], 'add')
                    if a is not None and len(a.inputs) == 2:
                        other = [x for x in a.inputs if x.uid != u.outputs[0].uid]
                        if len(other) == 1 and self.available(other[0]):
                            epi['res2'] = self.materialize(other[0])
                            epi['up2'] = True
                            self.absorbed.add(u.uid)
                            self.absorbed.add(a.uid)
                            t = a.outputs[0]
        return epi, t

    def _emit_conv_before_upsampling(self, uv, pre_bn, pre_relu, param, a, out_t, name):
        """R11: conv1x1(UpSampling2D(x)) = UpSampling2D(conv1x1(x)), and so for the BatchNormalization / ReLU around it: the
        convolution runs on the half-resolution tensor and its result stays a virtual up-sampled tensor (the residual slot
        of the unit's separable convolution reads it at half resolution)."""
        low = uv.low
        t, post_bn, post_relu = out_t, None, False
        n = self.sole_consumer(t, 'bn')
        if n is not None:
            post_bn = n.layers['bn']
            self.absorbed.add(n.uid)
            t = n.outputs[0]
        n = self.sole_consumer(t, 'relu')
        if n is not None:
            post_relu = True
            self.absorbed.add(n.uid)
            t = n.outputs[0]
        y = self.new_value(tuple(t.shape[:-3]) + (t.shape[-3] // 2, t.shape[-2] // 2, t.shape[-1]))
        attrs = dict(kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, Cin=low.C, Cout=a['filters'], K=low.C, pre_relu=int(pre_relu),
                     post_relu=int(post_relu), up2=0, res2_down=0)
        params = dict(w=param)
        if pre_bn is not None:
            params['pre_bn'] = pre_bn
        if post_bn is not None:
            params['post_bn'] = post_bn
        self.emit('conv', dict(x=low), dict(y=y), attrs, params, name)
        self.val[t.uid] = _UpView(y, t.shape)

    def _emit_conv(self, x, pre_bn, pre_relu, param, a, out_t, name):
        resample, cin = 0, x.shape[-1]
        will_be_skinny = len(out_t.shape) >= 3 and (a.get('sh', 1), a.get('sw', 1)) == (1, 1) and \
            split_k_rule(out_t.shape[-3] * out_t.shape[-2], a['kh'] * a['kw'] * cin, a['filters'], cin, a['kh'], a['kw'])
        if isinstance(x, _UpView):
            if x.full is None and (a['kh'], a['kw'], a.get('sh', 1), a.get('sw', 1)) == (1, 1, 1, 1) and len(out_t.shape) >= 3:
                return self._emit_conv_before_upsampling(x, pre_bn, pre_relu, param, a, out_t, name)
            if x.full is None and will_be_skinny and os.environ.get('DEEPHAR_RESAMPLE_ON_LOAD', '1') != '0':
                x, resample = x.low, 1           # R12: the skinny-conv kernel reads the half-resolution tensor up-sampled
            else:
                x = self._realize_up(x)
        elif isinstance(x, _PoolView):
            if x.full is None and will_be_skinny:
                x, resample = x.base, 2 + int(x.attrs.get('mode', 0))      # R12: ... or pools 2 x 2 windows on load
            else:
                x = self._realize_pool(x)
        skinny = len(out_t.shape) >= 3 and split_k_rule(out_t.shape[-3] * out_t.shape[-2], a['kh'] * a['kw'] * x.C,
                                                        a['filters'], x.C, a['kh'], a['kw'])
        epi, final_t = self._epilogue(out_t, skinny)
        y = self.out_value_for(final_t)
        attrs = dict(kh=a['kh'], kw=a['kw'], sh=a.get('sh', 1), sw=a.get('sw', 1), pt=a['pt'], pl=a['pl'],
                     Cin=x.C, Cout=a['filters'], K=a['kh'] * a['kw'] * x.C, pre_relu=int(pre_relu),
                     post_relu=int(epi['post_relu']), up2=int(epi['up2']), res2_down=int(epi['res2_down']))
        if resample:
            attrs['x_resample'] = resample
        ins = dict(x=x)
        if epi['res1'] is not None:
            ins['res1'] = epi['res1']
        if epi['res2'] is not None:
            ins['res2'] = epi['res2']
        params = dict(w=param)
        if pre_bn is not None:
            params['pre_bn'] = pre_bn
        if epi['post_bn'] is not None:
            params['post_bn'] = epi['post_bn']
        self.emit('conv', ins, dict(y=y), attrs, params, name)
        self.val[final_t.uid] = y

    def op_conv(self, node):
        x, pre_bn, pre_relu = self._prologue(node.inputs[0])
        self._emit_conv(x, pre_bn, pre_relu, node.layers['conv'].params[0], node.attrs, node.outputs[0],
                        node.name)

    def op_sepconv(self, node):
        x, pre_bn, pre_relu = self._prologue(node.inputs[0])
        layer = node.layers['sepconv']
        a = node.attrs
        pw = dict(kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, filters=a['filters'])
        # depthwise and pointwise stay two launches: a fused kernel (depthwise evaluated as the A operand of the MFMA
        # GEMM) was built in round 2, bit-identical and 7-40 % slower -- nothing issues beside an fp32 MFMA on gfx950
        # (profiles/r02_sepconv_fusion_study.md; the kernel lives in the history at commit 5d88aa8); a one-launch kernel
        # for the 8 x 8 level (frame, depthwise result and A operand all resident in LDS) was built in round 4,
        # bit-identical and neutral: 23-25 us against 11.5 + 15.5 us, step time unchanged (profiles/r04_sepconv8_study.md)
        mid = self.new_value(node.inputs[0].shape)
        params = dict(w=layer.params[0])
        if pre_bn is not None:
            params['pre_bn'] = pre_bn
        up_in = 0
        if isinstance(x, _PoolView):
            x = self._realize_pool(x)
        if isinstance(x, _UpView):               # R11: the depthwise half up-samples on load (dh_dw_args.up_in)
            if x.full is None and (a.get('sh', 1), a.get('sw', 1)) == (1, 1):
                x, up_in = x.low, 1
            else:
                x = self._realize_up(x)
        self.emit('dwconv', dict(x=x), dict(y=mid),
                  dict(kh=a['kh'], kw=a['kw'], pt=a['pt'], pl=a['pl'], pre_relu=int(pre_relu), up_in=up_in), params,
                  node.name + '/dw')
        self._emit_conv(mid, None, False, layer.params[1], pw, node.outputs[0], node.name + '/pw')

    # ---- glue ------------------------------------------------------------------------------------------
    def op_add(self, node):
        vals = [self.materialize(t) for t in node.inputs]
        y = self.out_value_for(node.outputs[0])
        acc = vals[0]
        rest = vals[1:]
        while rest:
            chunk, rest = rest[:2], rest[2:]
            dst = y if not rest else self.new_value(node.outputs[0].shape)
            ins = dict(a=acc, b=chunk[0])
            if len(chunk) > 1:
                ins['c'] = chunk[1]
            self.emit('eltwise', ins, dict(y=dst), dict(op=0, relu=0), name=node.name or 'add')
            acc = dst
        self.val[node.outputs[0].uid] = y

    def op_mul(self, node):
        a = self.materialize(node.inputs[0])
        b = self.materialize(node.inputs[1])
        y = self.out_value_for(node.outputs[0])
        self.emit('eltwise', dict(a=a, b=b), dict(y=y), dict(op=1, relu=0, bcast_b=int(b.C == 1 and a.C != 1)),
                  name=node.name or 'mul')
        self.val[node.outputs[0].uid] = y

    def op_scale(self, node):
        a = self.materialize(node.inputs[0])
        y = self.out_value_for(node.outputs[0])
        self.emit('eltwise', dict(a=a), dict(y=y), dict(op=0, relu=0, scale_const=float(node.attrs['k'])),
                  name=node.name or 'scale')
        self.val[node.outputs[0].uid] = y

    def op_sigmoid(self, node):
        a = self.materialize(node.inputs[0])
        y = self.out_value_for(node.outputs[0])
        self.emit('eltwise', dict(a=a), dict(y=y), dict(op=2, relu=0), name=node.name or 'sigmoid')
        self.val[node.outputs[0].uid] = y

    def op_concat(self, node):
        cv = self.concat_val.get(node.uid)
        if cv is None:
            cv = self.out_value_for(node.outputs[0])
            self.concat_val[node.uid] = cv
        off = 0
        for t in node.inputs:
            v = self.materialize(t)
            c = t.shape[-1]
            if not (v.buf is cv.buf and v.coff == cv.coff + off and v.ld == cv.ld):
                dst = Value(t.shape, cv.buf, cv.coff + off, cv.ld)
                self.emit('copy', dict(x=v), dict(y=dst), name=node.name or 'concat')
            off += c
        self.val[node.outputs[0].uid] = cv

    def op_slice(self, node):
        v = self.materialize(node.inputs[0])
        a = node.attrs
        self.val[node.outputs[0].uid] = Value(node.outputs[0].shape, v.buf, v.coff + a['start'], v.ld)

    def op_reshape(self, node):
        v = self.materialize(node.inputs[0])
        shape = node.outputs[0].shape
        if not v.dense:
            # e.g. [.., C, 1] view of a [.., C] slab living inside a concat buffer: keep ld if only the
            # trailing unit dim changes, else copy to a dense buffer first
            if shape[-1] == 1 and tuple(shape[:-1]) == tuple(v.shape) and False:
                pass
            d = self.new_value(v.shape)
            self.emit('copy', dict(x=v), dict(y=d), name='densify')
            v = d
        self.val[node.outputs[0].uid] = Value(shape, v.buf, 0, shape[-1])

    def op_pool(self, node):
        x = self.materialize(node.inputs[0])
        a = node.attrs
        # R12 [r06]: a 2 x 2 / stride-2 pooling (max, or max +- min: layers.py:411-425) whose only reader -- through
        # BatchNormalization / ReLU -- is a convolution on the skinny-conv kernel is not written out: that kernel takes the
        # maximum (+ minimum) of the four pixels while it loads its input (dh_conv_args.x_resample = 2 / 3).  The action
        # head's `conv2h` on max_min_pooling(x1) (spnet.py:77-84): one launch less per head.
        if os.environ.get('DEEPHAR_RESAMPLE_ON_LOAD', '1') != '0' and len(x.shape) >= 3 and a.get('mode', 0) in (0, 1) and \
                (a['kh'], a['kw'], a['sh'], a['sw'], a['pt'], a['pl']) == (2, 2, 2, 2, 0, 0) and x.shape[-3] % 2 == 0 and \
                x.shape[-2] % 2 == 0:
            reader = self._only_reader_through_bn_relu(node.outputs[0])
            if reader is not None and self._skinny_conv_node(reader, x.C):
                self.val[node.outputs[0].uid] = _PoolView(x, node.outputs[0].shape, a)
                return
        y = self.out_value_for(node.outputs[0])
        # R7: MaxPooling2D((2, 2)) of a convolution's output at 32 columns is written by that convolution's epilogue as a
        # second output (dh_conv_args.y_pool) -- the stand-alone pool reads the whole tensor back from HBM
        # (reception.py:105-116: every hourglass level is used at full AND at half resolution).  Bit-identical.  Neutral
        # while the pool ran beside other work on a second stream (round 2); on ONE stream, where the forward is the sum
        # of its kernels, it is worth 1-2 % on the MPII model.  DEEPHAR_FUSE_POOL=0 switches it off.
        # [r06] also at 16 and 8 columns (a wave's 32 output rows are two / four whole image rows there: it pools alone) --
        # one dependent launch less per down-scaling unit of SPNet's pyramids (DEEPHAR_FUSE_POOL_SMALL=0: 32 columns only).
        prod = self.producer.get(id(x))
        if os.environ.get('DEEPHAR_FUSE_POOL', '1') != '0' and prod is not None and prod.kind == 'conv' and \
                prod.outs.get('y') is x and 'ypool' not in prod.outs and not prod.attrs.get('up2') and \
                a.get('mode', 0) == 0 and (a['kh'], a['kw'], a['sh'], a['sw'], a['pt'], a['pl']) == (2, 2, 2, 2, 0, 0) and \
                (x.shape[-2] == 32 or (x.shape[-2] in (8, 16) and (x.shape[-3] * x.shape[-2]) % 32 == 0 and
                                       os.environ.get('DEEPHAR_FUSE_POOL_SMALL', '1') != '0')) and \
                x.shape[-3] % 2 == 0 and x.C % 4 == 0 and x.ld % 4 == 0 and y.ld % 4 == 0 and \
                x.coff % 4 == 0 and y.coff % 4 == 0 and \
                not (prod.attrs['kh'] * prod.attrs['kw'] > 1 and prod.attrs['Cin'] % 32 == 16) and \
                not split_k_rule(x.shape[-3] * x.shape[-2], prod.attrs['K'], prod.attrs['Cout'], prod.attrs['Cin'],
                                 prod.attrs['kh'], prod.attrs['kw']):
            prod.outs['ypool'] = y
            prod.attrs['pool2'] = 1
            self.producer[id(y)] = prod
            self.val[node.outputs[0].uid] = y
            return
        self.emit('pool', dict(x=x), dict(y=y), dict(node.attrs), name=node.name or 'pool')
        self.val[node.outputs[0].uid] = y

    def op_upsample(self, node):
        b = self.materialize(node.inputs[0])
        t = node.outputs[0]
        if self._virtual_upsample_ok(node):      # R11: not written out; its readers take the half-resolution tensor
            self.val[t.uid] = _UpView(b, t.shape)
            return
        a_node = self.sole_consumer(t, 'add')
        if a_node is not None and len(a_node.inputs) == 2:
            other = [x for x in a_node.inputs if x.uid != t.uid]
            if len(other) == 1 and self.available(other[0]):
                a = self.materialize(other[0])
                y = self.out_value_for(a_node.outputs[0])
                self.emit('upsample_add', dict(a=a, b=b), dict(y=y), name='upsample_add')
                self.absorbed.add(a_node.uid)
                self.val[a_node.outputs[0].uid] = y
                return
        y = self.out_value_for(t)
        self.emit('upsample_add', dict(b=b), dict(y=y), name='upsample')
        self.val[t.uid] = y

    def op_zeropad(self, node):
        x = self.materialize(node.inputs[0])
        if not x.dense:
            d = self.new_value(x.shape)
            self.emit('copy', dict(x=x), dict(y=d), name='densify')
            x = d
        y = self.new_value(node.outputs[0].shape)
        self.emit('zeropad', dict(x=x), dict(y=y), dict(pt=node.attrs.get('pt', 0), pl=node.attrs.get('pl', 0)),
                  name=node.name or 'zeropad')
        self.val[node.outputs[0].uid] = y

    def op_depthsum(self, node):
        d = self.materialize(node.inputs[0])
        h = self.materialize(node.inputs[1])
        z = self.out_value_for(node.outputs[0])
        self.emit('depthsum', dict(d=d, h=h), dict(z=z), name=node.name or 'depthsum')
        self.val[node.outputs[0].uid] = z

    # ---- decoder (R5) ------------------------------------------------------------------------------------
    def _sam(self, h_t, alpha, softmax_node):
        """One soft-argmax kernel covering every decoder read-out of the maps `h_t`.  Several channel
        soft-max nodes on the same maps with the same temperature (sSAM's own + the td_ChannelSoftmax of
        action.py:202) are one computation."""
        h = self.materialize(h_t)
        outs, attrs = {}, dict(alpha=float(alpha), conf_scale=1.0)
        twins = []
        if softmax_node is not None:
            twins = [n for n, _ in self.consumers.get(h_t.uid, [])
                     if n.op == 'softmax2d' and n.attrs['alpha'] == alpha and n.uid not in self.absorbed and
                     n.uid not in self.processed]
            if softmax_node not in twins:
                twins.append(softmax_node)
        prob_users = []
        for sm in twins:
            p_t = sm.outputs[0]
            need = bool(self.out_uids.get(p_t.uid, 0))
            for n, _ in self.consumers.get(p_t.uid, []):
                if n.op == 'expect2d':
                    if 'xy' not in outs:
                        outs['xy'] = self.out_value_for(n.outputs[0])
                    self.val[n.outputs[0].uid] = outs['xy']      # twins alias the first read-out
                    self.absorbed.add(n.uid)
                elif n.op == 'jointprob' and n.attrs.get('scale', 1.0) == 1.0:
                    if 'conf_prob' not in outs:
                        outs['conf_prob'] = self.out_value_for(n.outputs[0])
                    self.val[n.outputs[0].uid] = outs['conf_prob']
                    self.absorbed.add(n.uid)
                else:
                    need = True
            if need:
                prob_users.append(p_t)
            else:
                self.val[p_t.uid] = None  # never read
            if sm is not softmax_node:
                self.absorbed.add(sm.uid)
        # [r06] multiply([p, c]) right behind a read-out whose coordinates and confidence have no other reader (the replica
        # read-out in front of an action head, spnet.py:108) is folded into it: xy receives (x, y) * confidence
        if os.environ.get('DEEPHAR_FOLD_POSE_MUL', '1') != '0' and len(twins) == 1 and 'xy' in outs and 'conf_prob' in outs:
            xy_t = next((n.outputs[0] for n, _ in self.consumers.get(twins[0].outputs[0].uid, []) if n.op == 'expect2d'), None)
            cf_t = next((n.outputs[0] for n, _ in self.consumers.get(twins[0].outputs[0].uid, []) if n.op == 'jointprob'), None)
            mul = self.sole_consumer(xy_t, 'mul') if xy_t is not None else None
            if mul is not None and cf_t is not None and self.sole_consumer(cf_t, 'mul') is mul and \
                    [t.uid for t in mul.inputs] == [xy_t.uid, cf_t.uid] and outs['xy'].dense and outs['conf_prob'].dense:
                for v in (outs['xy'], outs['conf_prob']):             # the two intermediate tensors are never written
                    if v.buf in self.plan.bufs:
                        self.plan.bufs.remove(v.buf)
                outs['xy'] = self.out_value_for(mul.outputs[0])
                del outs['conf_prob']
                attrs['xy_times_conf'] = 1
                self.val[mul.outputs[0].uid] = outs['xy']
                self.val[xy_t.uid] = self.val[cf_t.uid] = None
                self.absorbed.add(mul.uid)
        if prob_users:
            outs['prob'] = self.out_value_for(prob_users[0]) if len(prob_users) == 1 else \
                self.new_value(prob_users[0].shape)
            for p_t in prob_users:
                self.val[p_t.uid] = outs['prob']
        # siblings reading the raw maps
        for n, _ in self.consumers.get(h_t.uid, []):
            if n.uid in self.absorbed or n.uid in self.processed or n in twins:
                continue
            if n.op == 'jointprob' and 'conf_raw' not in outs:
                outs['conf_raw'] = self.out_value_for(n.outputs[0])
                attrs['conf_scale'] = float(n.attrs.get('scale', 1.0))
                self.val[n.outputs[0].uid] = outs['conf_raw']
                self.absorbed.add(n.uid)
            elif n.op == 'globalmax2d' and 'gmax' not in outs:
                outs['gmax'] = self.new_value(n.outputs[0].shape)
                self.val[n.outputs[0].uid] = outs['gmax']
                self.absorbed.add(n.uid)
        self.emit('sam', dict(h=h), outs, attrs, name='softargmax2d')

    def op_softmax2d(self, node):
        self._sam(node.inputs[0], node.attrs['alpha'], node)

    def op_jointprob(self, node):
        # not claimed by a sibling soft-max: stand-alone confidence read-out
        h = self.materialize(node.inputs[0])
        y = self.out_value_for(node.outputs[0])
        self.emit('sam', dict(h=h), dict(conf_raw=y), dict(alpha=1.0, conf_scale=float(node.attrs['scale'])),
                  name='jointprob')
        self.val[node.outputs[0].uid] = y

    def op_globalmax2d(self, node):
        h = self.materialize(node.inputs[0])
        y = self.new_value(node.outputs[0].shape)
        self.emit('sam', dict(h=h), dict(gmax=y), dict(alpha=1.0, conf_scale=1.0), name='globalmax2d')
        self.val[node.outputs[0].uid] = y

    def op_expect2d(self, node):
        raise NotImplementedError('softargmax2d must be applied to the output of act_channel_softmax')

    def _fused_decoder(self, node):
        """R5b: the two soft-argmax read-outs (joint maps, context maps) feeding a context aggregation and the
        aggregation itself as ONE launch (dh_softargmax2d_context_f32) when nothing else reads the intermediate
        coordinates / confidences and the maps are the two channel runs [c0, c0+J), [c0+J, c0+J+J*nctx) of one tensor."""
        nctx = node.attrs['nctx']
        if any(self.n_consumers(t) != 1 for t in node.inputs) or not 1 <= nctx <= 3:
            return None
        vals = [self.val.get(t.uid) for t in node.inputs]
        if any(v is None or isinstance(v, _Lazy) for v in vals):
            return None
        ys, yc, pc = vals
        s_s, s_c = self.producer.get(id(ys)), self.producer.get(id(yc))
        if s_s is None or s_c is None or s_s is s_c or s_s.kind != 'sam' or s_c.kind != 'sam' or \
                self.producer.get(id(pc)) is not s_c or s_s.outs.get('xy') is not ys or s_c.outs.get('xy') is not yc or \
                s_c.outs.get('conf_raw') is not pc:
            return None
        if set(k for k, v in s_s.outs.items() if v is not None) - {'xy', 'conf_raw'} or \
                set(k for k, v in s_c.outs.items() if v is not None) != {'xy', 'conf_raw'}:
            return None
        if s_s.attrs != s_c.attrs or s_s.attrs.get('alpha') != 1.0:
            return None
        hs, hc = s_s.ins['h'], s_c.ins['h']
        J = hs.C
        if hs.buf is not hc.buf or hs.ld != hc.ld or hc.coff != hs.coff + J or hc.C != J * nctx or J % 4 or hs.ld % 4 or \
                hs.coff % 4 or hs.shape[:-1] != hc.shape[:-1]:
            return None
        # the fused step is emitted HERE, after everything planned so far: nothing already emitted may read what the two
        # read-outs wrote (under another topological order a consumer of the joint confidences could sit in between)
        first = min(self.plan.steps.index(s_s), self.plan.steps.index(s_c))
        written = [v for st in (s_s, s_c) for v in st.outs.values() if v is not None]
        for q in self.plan.steps[first + 1:]:
            if q is s_s or q is s_c:
                continue
            if any(w is not None and any(w.buf is v.buf for v in written) for w in q.ins.values()):
                return None
        return s_s, s_c, Value(hs.shape[:-1] + (J * (1 + nctx),), hs.buf, hs.coff, hs.ld)

    def op_context_agg(self, node):
        hit = self._fused_decoder(node)
        if hit is not None:
            s_s, s_c, h = hit
            for st in (s_s, s_c):
                self.plan.steps.remove(st)
            for v in (s_s.outs['xy'], s_c.outs['xy'], s_c.outs['conf_raw']):
                if v.buf in self.plan.bufs and not any(w is not None and w.buf is v.buf for q in self.plan.steps
                                                        for w in list(q.ins.values()) + list(q.outs.values())):
                    self.plan.bufs.remove(v.buf)
            y = self.out_value_for(node.outputs[0])
            outs = dict(y=y)
            if s_s.outs.get('conf_raw') is not None:
                outs['conf_raw'] = s_s.outs['conf_raw']
            attrs = dict(node.attrs)
            attrs.update(J=s_s.ins['h'].C, sam_alpha=s_s.attrs['alpha'], conf_scale=s_s.attrs['conf_scale'])
            self.emit('sam_ctx', dict(h=h), outs, attrs, name=node.name)
            self.val[node.outputs[0].uid] = y
            return
        ys, yc, pc = [self.materialize(t) for t in node.inputs]
        for v in (ys, yc, pc):
            assert v.dense, 'context aggregation expects dense operands'
        y = self.out_value_for(node.outputs[0])
        self.emit('context_agg', dict(ys=ys, yc=yc, pc=pc), dict(y=y), dict(node.attrs), name=node.name)
        self.val[node.outputs[0].uid] = y

    def op_depthmean(self, node):
        """reception.pose_regression_3d (reception.py:193-222) takes the mean of the same D x J maps over depth and over
        the pixels: both read-outs are one launch (one pass over the maps) when both nodes hang on the same tensor."""
        h = self.materialize(node.inputs[0])
        role_of = lambda n: 'hxy' if n.attrs['axis'] == 'd' else 'hz'
        outs = {role_of(node): self.new_value(node.outputs[0].shape)}
        self.val[node.outputs[0].uid] = outs[role_of(node)]
        for n, _ in self.consumers.get(node.inputs[0].uid, []):
            if n is not node and n.op == 'depthmean' and n.uid not in self.absorbed and n.uid not in self.processed and \
                    role_of(n) not in outs and (n.attrs['D'], n.attrs['J']) == (node.attrs['D'], node.attrs['J']):
                outs[role_of(n)] = self.new_value(n.outputs[0].shape)
                self.val[n.outputs[0].uid] = outs[role_of(n)]
                self.absorbed.add(n.uid)
        self.emit('depth_means', dict(h=h), outs, dict(D=node.attrs['D'], J=node.attrs['J']),
                  name='depth_means_' + '_'.join(sorted(outs)))

    def op_softargmax1d(self, node):
        hz = self.materialize(node.inputs[0])
        assert hz.dense
        outs = dict(z=self.out_value_for(node.outputs[0]))
        for n, _ in self.consumers.get(node.inputs[0].uid, []):
            if n.op == 'globalmax1d' and n.uid not in self.absorbed and n.uid not in self.processed:
                outs['vz'] = self.new_value(n.outputs[0].shape)
                self.val[n.outputs[0].uid] = outs['vz']
                self.absorbed.add(n.uid)
                break
        self.emit('softargmax1d', dict(hz=hz), outs, name='softargmax1d')
        self.val[node.outputs[0].uid] = outs['z']

    def op_globalmax1d(self, node):
        hz = self.materialize(node.inputs[0])
        y = self.new_value(node.outputs[0].shape)
        self.emit('softargmax1d', dict(hz=hz), dict(vz=y), name='globalmax1d')
        self.val[node.outputs[0].uid] = y

    def op_kronecker(self, node):
        hm = self.materialize(node.inputs[0])
        x = self.materialize(node.inputs[1])
        y = self.out_value_for(node.outputs[0])
        self.emit('kronecker', dict(hm=hm, x=x), dict(y=y), name=node.name or 'kronecker')
        self.val[node.outputs[0].uid] = y

    def op_globalmaxmin(self, node):
        x = self.materialize(node.inputs[0])
        t = node.outputs[0]
        sm = self.sole_consumer(t, 'softmax')
        final = sm.outputs[0] if sm is not None else t
        if sm is not None:
            self.absorbed.add(sm.uid)
        y = self.new_value(final.shape)
        self.emit('globalmaxmin', dict(x=x), dict(y=y), dict(softmax=int(sm is not None)), name=node.name)
        self.val[final.uid] = y

    def op_softmax(self, node):
        raise NotImplementedError('stand-alone softmax (only used after global_max_min_pooling)')

    # ---- bookkeeping ------------------------------------------------------------------------------------
    def _collect_params(self):
        seen, out = set(), []
        for n in self.nodes:
            for layer in n.layers.values():
                for p in layer.params:
                    if id(p) not in seen:
                        seen.add(id(p))
                        out.append(p)
        self.plan.params = out


def build_plan(inputs, outputs, nstreams=1, gemm_precision='f32', stream_policy='list'):
    if gemm_precision not in ('f32', 'bf16x3'):
        raise ValueError("gemm_precision must be 'f32' or 'bf16x3', got %r" % (gemm_precision,))
    if stream_policy not in ('list', 'tail'):
        raise ValueError("stream_policy must be 'list' or 'tail', got %r" % (stream_policy,))
    plan = Planner(inputs, outputs, nstreams, stream_policy).run()
    plan.gemm_precision = gemm_precision
    return plan
