"""Symbolic graph IR at Keras-layer granularity.

The reference builds its models with the Keras functional API (deephar/layers.py, deephar/models/*.py).
This module provides just enough of that API surface -- symbolic tensors, layers that own named weights,
nested Models that can be called on new tensors -- for the builders in deephar_amd/models to be written in
the reference's own vocabulary, while the result is a small static op graph that the planner
(deephar_amd/engine/planner.py) fuses and lowers to gfx950 kernel launches.  No arithmetic happens here.

Shapes exclude the batch dimension, like keras.backend.int_shape(x)[1:].  Spatial ops act on the last three
dims (H, W, C); any dims in front of them (the T of a clip, what the reference wraps in TimeDistributed,
layers.py:63-104) are folded into the batch by the executor.
"""
import contextlib
import itertools

import numpy as np

_uid = itertools.count()


class Param:
    """One weight tensor of a layer (Keras names: kernel, depthwise_kernel, pointwise_kernel, gamma, beta,
    moving_mean, moving_variance)."""

    def __init__(self, key, name, shape, role):
        self.key = key          # '<scope>/<layer>/<weight>'
        self.name = name        # weight name
        self.shape = tuple(int(s) for s in shape)
        self.role = role        # 'conv' | 'depthwise' | 'beta' | 'mean' | 'var' | 'gamma' | 'frozen'
        self.value = None       # np.float32 array once set
        self.version = 0        # bumped on every set (device copies are refreshed lazily)
        self.fan_in = None

    def set(self, value):
        value = np.asarray(value, dtype=np.float32)
        if tuple(value.shape) != self.shape:
            raise ValueError('weight %s expects shape %s, got %s' % (self.key, self.shape, value.shape))
        self.value = np.ascontiguousarray(value)
        self.version += 1


class Layer:
    """A weight-owning layer record (Conv2D / SeparableConv2D / BatchNormalization)."""

    def __init__(self, cls, name, scope, params):
        self.cls = cls
        self.name = name
        self.scope = scope
        self.params = params    # list[Param] in Keras `layer.weights` order
        self.trainable = True
        self.uid = next(_uid)   # creation order (= the reference's source order inside a builder)

    @property
    def weights(self):
        return self.params

    def get_weights(self):
        return [p.value for p in self.params]

    def set_weights(self, values):
        if len(values) != len(self.params):
            raise ValueError('layer %s expects %d weight arrays, got %d' % (self.name, len(self.params),
                                                                              len(values)))
        for p, v in zip(self.params, values):
            p.set(v)


class Tensor:
    """Symbolic tensor.  `shape` excludes the batch dim."""

    def __init__(self, shape, node=None, index=0, name=None):
        self.shape = tuple(int(s) for s in shape)
        self.node = node
        self.index = index
        self.name = name
        self.uid = next(_uid)

    def __repr__(self):
        return 'Tensor(%s, shape=%s, op=%s)' % (self.name or self.uid, self.shape,
                                               self.node.op if self.node else 'input')

    # Keras-style slicing used through Lambda in the reference (reception.py:171-172): channel slices only
    def channels(self, start, stop):
        return emit('slice', [self], [self.shape[:-1] + (stop - start,)], dict(start=start, stop=stop))[0]


class Node:
    def __init__(self, op, inputs, out_shapes, attrs=None, layers=None, name=None):
        self.op = op
        self.inputs = list(inputs)
        self.attrs = dict(attrs or {})
        self.layers = dict(layers or {})   # role -> Layer
        self.name = name
        self.uid = next(_uid)
        self.outputs = [Tensor(s, self, i) for i, s in enumerate(out_shapes)]

    def __repr__(self):
        return 'Node(%s %s)' % (self.op, self.name or self.uid)


# ---------------------------------------------------------------------------------------------------------
# Naming scopes: weights are keyed '<scope>/<layer>/<weight>'; un-named layers get '<class>_<n>' with a
# counter per (scope, class) in creation order (convention documented in oracle/naming.py as well).
# ---------------------------------------------------------------------------------------------------------
class _BuildState:
    def __init__(self):
        self.scopes = []
        self.counters = {}

    @property
    def scope(self):
        return self.scopes[-1] if self.scopes else ''


_state = _BuildState()


def reset_naming():
    """Forget all auto-name counters (call before building an independent model in the same process)."""
    _state.scopes = []
    _state.counters = {}


@contextlib.contextmanager
def name_scope(name):
    _state.scopes.append(name)
    for k in [k for k in _state.counters if k[0] == name]:
        del _state.counters[k]
    try:
        yield
    finally:
        _state.scopes.pop()


def current_scope():
    return _state.scope


def auto_name(cls):
    key = (_state.scope, cls)
    _state.counters[key] = _state.counters.get(key, 0) + 1
    return '%s_%d' % (cls, _state.counters[key])


def make_layer(cls, name, specs):
    """specs: list of (weight_name, shape, role)."""
    scope = _state.scope
    lname = name or auto_name({'Conv2D': 'conv2d', 'SeparableConv2D': 'separable_conv2d',
                               'BatchNormalization': 'batch_normalization'}[cls])
    params = [Param('%s/%s/%s' % (scope, lname, w), w, shape, role) for (w, shape, role) in specs]
    return Layer(cls, lname, scope, params)


def emit(op, inputs, out_shapes, attrs=None, layers=None, name=None):
    for t in inputs:
        if not isinstance(t, Tensor):
            raise TypeError('op %s expects symbolic tensors, got %r' % (op, type(t)))
    return Node(op, inputs, out_shapes, attrs, layers, name).outputs


def Input(shape, name=None):
    return Tensor(tuple(shape), None, 0, name or 'input')


def topo_nodes(outputs, stop=None):
    """All nodes reachable from `outputs`, in a deterministic topological order (DFS post-order following
    input order, i.e. the order the builder would have executed them).  Tensors whose uid is in `stop` are
    treated as sources: their producers are not visited."""
    stop = stop or set()
    order, seen = [], set()
    stack = [(t.node, False) for t in reversed(outputs) if t.node is not None and t.uid not in stop]
    while stack:
        node, done = stack.pop()
        if done:
            order.append(node)
            continue
        if node.uid in seen:
            continue
        seen.add(node.uid)
        stack.append((node, True))
        for t in reversed(node.inputs):
            if t.node is not None and t.node.uid not in seen and t.uid not in stop:
                stack.append((t.node, False))
    return order


def clone_subgraph(inputs, outputs, new_inputs, relead=None, allow_unused=False):
    """Re-instantiate the sub-graph inputs->outputs on `new_inputs` (a nested Model being called on a new
    tensor, e.g. reception.py:93,131 or TimeDistributed(model) in action.py:124-125).  Layers (weights) are
    shared; leading dims of the new inputs (clip length T) are propagated to every cloned tensor.
    `inputs` may be internal tensors (a cut through the graph).  relead=(T, T') rewrites a leading frame axis
    of length T to T' on every cloned tensor (valid for frame-independent sub-graphs, deephar_amd/parallel.py)."""
    if len(inputs) != len(new_inputs):
        raise ValueError('model expects %d inputs, got %d' % (len(inputs), len(new_inputs)))
    mapping = {}
    lead = None

    def fix(shape):
        if relead is not None and len(shape) > 0 and shape[0] == relead[0]:
            return (relead[1],) + tuple(shape[1:])
        return tuple(shape)

    for old, new in zip(inputs, new_inputs):
        k = len(new.shape) - len(old.shape)
        if k < 0 or tuple(new.shape[k:]) != fix(old.shape):
            raise ValueError('input shape mismatch: model expects %s, got %s' % (old.shape, new.shape))
        if lead is None:
            lead = tuple(new.shape[:k])
        elif lead != tuple(new.shape[:k]):
            raise ValueError('inconsistent leading dims across inputs')
        mapping[old.uid] = new
    for node in topo_nodes(outputs, stop={t.uid for t in inputs}):
        ins = []
        for t in node.inputs:
            if t.uid not in mapping:
                raise ValueError('graph is disconnected: %r is not reachable from the model inputs' % t)
            ins.append(mapping[t.uid])
        new = Node(node.op, ins, [lead + fix(o.shape) for o in node.outputs], node.attrs, node.layers, node.name)
        for o, n in zip(node.outputs, new.outputs):
            mapping[o.uid] = n
    res = []
    for o in outputs:
        if o.uid not in mapping:
            raise ValueError('output %r not produced from the model inputs' % o)
        res.append(mapping[o.uid])
    return res
