"""Keras 2.1.4 weight-file layout of a model built with deephar_amd (host logic, no arithmetic).

The reference restores its published weights with `model.load_weights(path)` -- by topological ORDER for the
ReceptionNet / merge files (exp/mpii/eval_mpii_singleperson.py:54, exp/h36m/eval_h36m.py:53,
exp/pennaction/eval_penn_ar_pe_merge.py:62) and `by_name=True` for SPNet (exp/pennaction/eval_penn_multitask.py:76,
exp/ntu/eval_ntu_multitask.py:66).  Order-based loading only works if the loader walks the model exactly like
Keras does, so this module rebuilds, from the graph IR, what Keras would have seen:

  * the layer graph at each nesting level (an IR node is one Keras layer; a call of a nested `Model` is one
    layer; the few fused decoder ops expand into the chain of Keras layers the reference builds for them,
    including its frozen helper layers that own weights in the file but not in this engine);
  * `Model.layers` order (keras/engine/topology.py, Container.__init__): layers are numbered by a depth-first
    walk from the outputs, a node's depth is its longest path to an output, a layer sits at the largest depth of
    its nodes, and layers are listed by decreasing depth, ties by traversal number;
  * `layer.weights` order: trainable weights first, then non-trainable ones -- per layer for plain layers, over
    ALL inner layers for a nested Model (so a nested Model lists every kernel / beta before any moving mean);
    frozen layers report everything as non-trainable, in the same relative order.

`layout(model)` returns the groups `save_weights` would write; `load_hdf5` / `save_hdf5` use it.

Verified group-for-group against the reference's builders for the families the reference loads BY ORDER (ReceptionNet
2-D / 3-D, merge models; 11 configurations).  SPNet is loaded by the reference with by_name=True, which only needs the
top-level layer names and the per-layer weight order (verified); the relative ORDER of SPNet's parallel pose / action
branches is not reproduced exactly (its fused decoder ops span several Keras layers at the top level), so an SPNet
file written here loads in Keras with by_name=True, like the reference's own SPNet files.
"""
import sys

import numpy as np

from . import graph as G


class KLayer:
    def __init__(self, name, kind, trainable=True, params=(), ntrain=0, inner=None, model=None):
        self.name, self.kind, self.trainable = name, kind, trainable
        self.params = list(params)      # leaf: Keras order, the last `ntrain` entries are non-trainable
        self.ntrain = ntrain
        self.inner = inner              # nested model: KView
        self.model = model
        self.nodes = []


class KNode:
    def __init__(self, layer):
        self.layer = layer
        self.inputs = []                # KNodes producing the input tensors, in call order
        layer.nodes.append(self)


class KView:
    def __init__(self, name, outputs):
        self.name, self.outputs = name, outputs     # outputs: KNodes producing the model outputs
        self._layers = None

    @property
    def layers(self):
        if self._layers is None:
            self._layers = _keras_order(self.outputs)
        return self._layers


def _keras_order(outputs):
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
    index, order, finished = {}, [], set()

    def visit(node):
        if id(node) in finished:
            return
        if id(node.layer) not in index:
            index[id(node.layer)] = len(index)
        for i in node.inputs:
            visit(i)
        finished.add(id(node))
        order.append(node)

    for o in outputs:
        visit(o)
    ndepth, ldepth, layers = {}, {}, {}
    for node in reversed(order):
        lay = node.layer
        d = max(ndepth.setdefault(id(node), 0), ldepth.get(id(lay), 0))
        ldepth[id(lay)] = d
        layers[id(lay)] = lay
        ndepth[id(node)] = d
        for i in node.inputs:
            ndepth[id(i)] = max(d + 1, ndepth.get(id(i), 0))
    return sorted(layers.values(), key=lambda l: (-ldepth[id(l)], index[id(l)]))


# ---- IR op -> the Keras layers it stands for ------------------------------------------------------------
class Frozen:
    """A weight the reference's Keras graph owns and this engine does not: the fixed kernels of the soft-argmax
    / aggregation helper layers.  `kind` + the node shape are enough to regenerate its value for export."""

    def __init__(self, name, kind, meta):
        self.name, self.kind, self.meta = name, kind, meta

    def __repr__(self):
        return 'Frozen(%s)' % self.name

    def value(self):
        m = self.meta
        if self.kind == 'sam_dw':           # layers.lin_interpolation_2d (layers.py:160-200), utils/math.py:6-20
            H, W, C, axis = m
            lin = (np.tile(np.linspace(0.0, 1.0, num=W), (H, 1)) if axis == 0
                   else np.tile(np.linspace(0.0, 1.0, num=H), (W, 1)).T).astype(np.float32)
            return np.repeat(lin[:, :, None, None], C, axis=2)
        if self.kind == 'sam_pw':
            return np.eye(m[2], dtype=np.float32)[None, None]
        if self.kind == 'lin1d':            # layers.lin_interpolation_1d (layers.py:131-157)
            D, C = m
            w = np.zeros((D, C, C), np.float32)
            lin = np.linspace(1 / (2 * D), 1 - 1 / (2 * D), num=D)
            for i in range(C):
                w[:, i, i] = lin
            return w
        if self.kind == 'ctx_dense':        # blocks.build_context_aggregation (blocks.py:221-233)
            J, nctx = m
            w = np.zeros((J * nctx, J), np.float32)
            for j in range(J):
                w[j * nctx:(j + 1) * nctx, j] = 1.0
            return w
        raise ValueError(self.kind)


def _leaf_layers(n):
    """The Keras layers one IR node stands for, as a tiny graph: [(key, name, frozen weights, [input keys])];
    input key None = the node's own inputs; the last entry produces the node's output.  Ops not listed are
    exactly one weight-less Keras layer (Activation / Lambda / merge / pooling / ...)."""
    op = n.op
    if op == 'expect2d':
        # layers.softargmax2d (layers.py:122-129) = lin_interpolation_2d per axis: a frozen SeparableConv2D
        # named name+'_x' / '_y' (or custom_sam_<k>), three squeeze/expand Lambdas, then concatenate
        H, W, C = n.inputs[0].shape[-3:]
        base = n.name
        chain = []
        for ax, sfx in ((0, '_x'), (1, '_y')):
            nm = (base + sfx) if base else 'custom_sam'
            fr = [Frozen(nm + '/depthwise_kernel:0', 'sam_dw', (H, W, C, ax)),
                  Frozen(nm + '/pointwise_kernel:0', 'sam_pw', (H, W, C, ax))]
            chain += [('c%d' % ax, nm, fr, [None]), ('a%d' % ax, 'lambda', (), ['c%d' % ax]),
                      ('b%d' % ax, 'lambda', (), ['a%d' % ax]), ('e%d' % ax, 'lambda', (), ['b%d' % ax])]
        return chain + [('cat', 'concatenate', (), ['e0', 'e1'])]
    if op == 'context_agg':
        J = n.outputs[0].shape[-2]
        fr = [Frozen('dense/kernel:0', 'ctx_dense', (J, n.attrs['nctx']))]
        return [('d', 'dense', fr, [None]), ('o', 'lambda', (), ['d'])]
    if op == 'softargmax1d':
        D, C = n.inputs[0].shape[-2:]
        fr = [Frozen('conv1d/kernel:0', 'lin1d', (D, C))]
        return [('a', 'activation', (), [None]), ('c', 'conv1d', fr, ['a']), ('o', 'lambda', (), ['c'])]
    return [('o', n.name or op, (), [None])]


def view(model):
    """KView of a deephar_amd Model (cached on the model)."""
    cached = getattr(model, '_kview', None)
    if cached is not None:
        return cached
    stop = {t.uid for t in model.inputs}
    nodes = G.topo_nodes(model.outputs, stop=stop)
    prod = {}                       # tensor uid -> KNode
    wiring = []                     # (KNode, [input tensor uids])
    leaf_layers = {}                # id(graph Layer) -> KLayer  (a shared layer is ONE Keras layer)
    calls = {}                      # id(CallRec) -> KNode
    model_layers = {}               # id(nested Model) -> KLayer
    for t in model.inputs:
        prod[t.uid] = KNode(KLayer(t.name or 'input', 'input'))
    for n in nodes:
        recs = n.attrs.get('_calls')
        if recs:
            rec = recs[0]
            if id(rec) in calls:
                continue
            m = rec.model
            lay = model_layers.get(id(m))
            if lay is None:
                lay = model_layers[id(m)] = KLayer(m.name, 'model', getattr(m, 'trainable', True),
                                                   inner=view(m), model=m)
            kn = calls[id(rec)] = KNode(lay)
            for t in rec.outputs:
                prod[t.uid] = kn
            wiring.append((kn, [t.uid for t in rec.inputs]))
            continue
        local = {}
        specs = _leaf_layers(n)
        for key, lname, frozen, ins in specs:
            glayers = list(n.layers.values())
            if glayers and not frozen and key == specs[-1][0]:
                gl = glayers[0]
                lay = leaf_layers.get(id(gl))
                if lay is None:
                    nt = 2 if gl.cls == 'BatchNormalization' else 0
                    lay = leaf_layers[id(gl)] = KLayer(gl.name, 'leaf', gl.trainable, gl.params, nt)
            else:
                lay = KLayer(lname, 'leaf', False, list(frozen), 0)
            kn = KNode(lay)
            local[key] = kn
            ext = [t.uid for t in n.inputs] if None in ins else []
            wiring.append((kn, ext, [local[k] for k in ins if k is not None]))
        for o in n.outputs:
            prod[o.uid] = local[specs[-1][0]]
    for w in wiring:
        kn, ext = w[0], w[1]
        kn.inputs = [prod[u] for u in ext] + (list(w[2]) if len(w) > 2 else [])
    kv = KView(model.name, [prod[t.uid] for t in model.outputs])
    model._kview = kv
    return kv


def layer_weights(lay):
    """`layer.weights` in Keras order: list of Param / Frozen."""
    def tr(l):
        if not l.trainable:
            return []
        if l.kind == 'model':
            return [w for x in l.inner.layers for w in tr(x)]
        return l.params[:len(l.params) - l.ntrain]

    def ntr(l):
        if l.kind == 'model':
            w = [w for x in l.inner.layers for w in ntr(x)]
            return w if l.trainable else [w_ for x in l.inner.layers for w_ in tr(x)] + w
        return l.params[len(l.params) - l.ntrain:] if l.trainable else list(l.params)

    return tr(lay) + ntr(lay)


def layout(model):
    """[(Keras layer name, [Param | Frozen, ...])] for every weight-owning top-level layer, in the order
    `save_weights` writes them and `load_weights(by_name=False)` consumes them."""
    out = []
    for lay in view(model).layers:
        ws = layer_weights(lay)
        if ws:
            out.append((lay.name, ws))
    return out


# ---- HDF5 import / export -----------------------------------------------------------------------------------
def _file_groups(f):
    root = f['model_weights'] if 'model_weights' in f else f        # `model.save` nests them one level down
    names = [n.decode('utf8') if isinstance(n, bytes) else str(n) for n in np.atleast_1d(root.attrs['layer_names'])]
    out = []
    for name in names:
        g = root[name]
        wn = g.attrs.get('weight_names')
        wn = [] if wn is None else [w.decode('utf8') if isinstance(w, bytes) else str(w) for w in np.atleast_1d(wn)]
        out.append((name, g, wn))
    return out


def _assign(lname, targets, g, wnames, path):
    if len(targets) != len(wnames):
        raise ValueError('%s: layer "%s" expects %d weight tensors, the file group holds %d' %
                         (path, lname, len(targets), len(wnames)))
    pairs = []
    for t, wn in zip(targets, wnames):
        if isinstance(t, Frozen):
            continue
        a = np.asarray(g[wn])
        if tuple(a.shape) != t.shape:
            raise ValueError('%s: %s has shape %s, model weight %s expects %s' % (path, wn, a.shape, t.key, t.shape))
        pairs.append((t, a))
    return pairs


def load_hdf5(model, path, by_name=False):
    """keras.engine.topology.load_weights_from_hdf5_group / ..._by_name on a Keras 2.x weight file.
    Order mode: the file's weight-owning groups are paired, in order, with the model's weight-owning layers in
    Keras' `Model.layers` order (count and shapes are checked).  Name mode: every file group whose name matches
    a top-level layer (a nested Model, or a layer by its Keras name) is loaded, everything else is left alone.
    Returns the number of tensors set."""
    from . import hdf5
    f = hdf5.File(path)
    groups = [(n, g, wn) for (n, g, wn) in _file_groups(f) if wn]
    if by_name:
        index = {}
        for lay in view(model).layers:
            ws = layer_weights(lay)
            if ws:
                index.setdefault(lay.name, []).append(ws)
        pairs = []
        for name, g, wn in groups:
            for ws in index.get(name, []):
                pairs += _assign(name, ws, g, wn, path)
        return _commit(pairs)
    lay = layout(model)
    if len(lay) != len(groups):
        raise ValueError('%s holds %d weight-owning layers, model %s has %d (Keras order: %s ...)' %
                         (path, len(groups), model.name, len(lay), [n for n, _ in lay][:6]))
    return _commit([pr for (name, ws), (_, g, wn) in zip(lay, groups) for pr in _assign(name, ws, g, wn, path)])


def _commit(pairs):
    """Nothing is assigned unless the whole file validated (K.batch_set_value at the end of Keras' loader)."""
    for t, a in pairs:
        t.set(a)
    return len(pairs)


def save_hdf5(model, path):
    """`model.save_weights(path)` in Keras 2.1.4's layout (groups in `Model.layers` order, `layer_names` /
    `weight_names` attributes, frozen helper kernels regenerated) so that the reference's own `load_weights`
    accepts the file.  Only weight-owning layers are written (Keras' loader filters the others out anyway)."""
    from . import hdf5
    lay = layout(model)
    tree = {hdf5.ATTRS: {'layer_names': [n.encode('utf8') for n, _ in lay], 'backend': b'tensorflow',
                         'keras_version': b'2.1.4'}}
    used = set()
    for name, ws in lay:
        if name in used:
            raise ValueError('two weight-owning top-level layers are both named "%s"' % name)
        used.add(name)
        sub = tree.setdefault(name, {})
        names, seen = [], {}
        for w in ws:
            if isinstance(w, Frozen):
                wn, val = w.name, w.value()
            else:
                if w.value is None:
                    raise RuntimeError('cannot save: weight %s is unset' % w.key)
                wn, val = '%s/%s:0' % (w.key.split('/')[-2], w.name), w.value
            k = seen.get(wn, 0)
            seen[wn] = k + 1
            if k:                                   # TensorFlow uniquifies repeated scope names the same way
                head, tail = wn.split('/', 1)
                wn = '%s_%d/%s' % (head, k, tail)
            hdf5.put_path(sub, wn, val)
            names.append(wn.encode('utf8'))
        sub[hdf5.ATTRS] = {'weight_names': names}
    hdf5.write_file(path, tree)
