"""A ~400-line stand-in for the slice of Keras 2.1.4 / TensorFlow that the reference's hot-path code touches,
backed by PyTorch-CPU.  TEST INFRASTRUCTURE ONLY (never imported by the product, never shipped to the GPU box
code path): it exists so that the reference's OWN builder code -- deephar/layers.py, activations.py,
models/{blocks,reception,action,common,spnet}.py, imported unmodified from /root/reference -- can be executed in
this container and its outputs committed as golden vectors (tests/golden/make_reference_golden.py).

What this pins and what it does not: everything the reference's Python decides (graph wiring, layer order, slicing
indices, constants such as alpha / 4*hs / time_stride / padding amounts, frozen soft-argmax weights, output order)
is executed verbatim; what Keras/TensorFlow decide (TF-"SAME" padding, BatchNormalization epsilon = 1e-3, pooling
and up-sampling semantics, K.epsilon() = 1e-7) is restated here from the Keras 2.1.4 defaults.

Symbolic tensors are evaluated lazily (a small DAG of closures); static shapes come from running every op once on
a batch-1 dummy.  Layers and backend functions are polymorphic: called on a KTensor they extend the graph, called
on a torch tensor (inside TimeDistributed / nested Models at run time) they compute.
"""
import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

DTYPE = [torch.float32]          # mutable: switch to float64 for the accuracy arbiter
_layer_registry = []             # every weight-owning layer in creation order
_name_counters = {}


def set_dtype(dt):
    DTYPE[0] = dt


def reset():
    _layer_registry.clear()
    _name_counters.clear()


def _auto_name(prefix):
    _name_counters[prefix] = _name_counters.get(prefix, 0) + 1
    return '%s_%d' % (prefix, _name_counters[prefix])


# ------------------------------------------------------------------------------------------------- tensors
class KTensor:
    def __init__(self, fn, parents, name=None):
        self.fn, self.parents, self.name = fn, list(parents), name
        self._dummy = fn(*[p._dummy for p in self.parents]) if fn is not None else None

    @property
    def shape(self):
        return (None,) + tuple(self._dummy.shape[1:])

    def _eval(self, feed, cache):
        k = id(self)
        if k in feed:
            return feed[k]
        if k not in cache:
            cache[k] = self.fn(*[p._eval(feed, cache) for p in self.parents])
        return cache[k]

    # arithmetic used inside the reference's Lambda functions
    def _bin(self, other, op, rev=False):
        if isinstance(other, KTensor):
            return KTensor((lambda a, b: op(b, a)) if rev else op, [self, other])
        return KTensor((lambda a: op(other, a)) if rev else (lambda a: op(a, other)), [self])

    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: a * b, rev=True)
    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: a + b, rev=True)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: a - b, rev=True)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __neg__(self): return KTensor(lambda a: -a, [self])
    def __getitem__(self, idx): return KTensor(lambda a: a[idx], [self])


def _is_sym(x):
    return isinstance(x, KTensor) or (isinstance(x, (list, tuple)) and len(x) > 0 and isinstance(x[0], KTensor))


def _apply(fn, x):
    """fn works on torch tensor(s); x is a tensor / KTensor or a list of them."""
    if isinstance(x, (list, tuple)):
        if isinstance(x[0], KTensor):
            return KTensor(lambda *v: fn(list(v)), list(x))
        return fn(list(x))
    if isinstance(x, KTensor):
        return KTensor(fn, [x])
    return fn(x)


class KNode:
    """One call of a layer (keras.engine.topology.Node): who produced the inputs, what came out."""

    def __init__(self, layer, inputs, outputs):
        self.outbound_layer, self.input_tensors, self.output_tensors = layer, list(inputs), list(outputs)
        layer.inbound_nodes.append(self)
        for i, o in enumerate(self.output_tensors):
            o._kh = (layer, len(layer.inbound_nodes) - 1, i)


def Input(shape=None, name=None, **kw):
    t = KTensor(None, [], name)
    t._dummy = torch.zeros((1,) + tuple(shape), dtype=DTYPE[0])
    lay = Layer.__new__(Layer)
    lay.name = name or _auto_name('input')
    lay.trainable, lay.weights, lay.built, lay.inbound_nodes = False, None, True, []
    KNode(lay, [], [t])
    return t


# ------------------------------------------------------------------------------------------------- layers
class Layer:
    prefix = 'layer'

    def __init__(self, name=None, **kw):
        self.name = name or _auto_name(self.prefix)
        self.trainable = bool(kw.get('trainable', True))
        self.weights = None          # list of torch tensors once built
        self.built = False
        self.inbound_nodes = []

    def build(self, x):
        pass

    def compute(self, x):
        raise NotImplementedError

    def __call__(self, x):
        if not self.built:
            d = [t._dummy for t in x] if isinstance(x, (list, tuple)) and isinstance(x[0], KTensor) else \
                (x._dummy if isinstance(x, KTensor) else x)
            self.build(d)
            self.built = True
            if self.weights:
                _layer_registry.append(self)
        out = _apply(self.compute, x)
        if isinstance(out, KTensor):
            out._layer = self
            KNode(self, x if isinstance(x, (list, tuple)) else [x], [out])
        return out

    def get_weights(self):
        return [w.detach().cpu().numpy().copy() for w in (self.weights or [])]

    def set_weights(self, ws):
        assert len(ws) == len(self.weights), (self.name, len(ws), len(self.weights))
        for i, w in enumerate(ws):
            w = np.asarray(w)
            assert tuple(w.shape) == tuple(self.weights[i].shape), (self.name, w.shape, self.weights[i].shape)
            self.weights[i] = torch.from_numpy(np.ascontiguousarray(w)).to(DTYPE[0])

    def count_params(self):
        return int(sum(w.numel() for w in (self.weights or [])))


def _same_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def _pad_nchw(xc, h, w, k, s, value=0.0):
    pt, pb = _same_pad(h, k[0], s[0])
    pl, pr = _same_pad(w, k[1], s[1])
    return F.pad(xc, (pl, pr, pt, pb), value=value)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2D(Layer):
    prefix = 'conv2d'

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', use_bias=True, name=None, **kw):
        super().__init__(name)
        assert not use_bias or kw.get('activity_regularizer') is not None or True
        self.filters, self.k, self.s, self.padding, self.use_bias = filters, _pair(kernel_size), _pair(strides), padding, use_bias
        assert not use_bias, 'the hot path only uses bias-free convolutions'

    def build(self, x):
        self.weights = [torch.zeros(self.k + (x.shape[-1], self.filters), dtype=DTYPE[0])]

    def compute(self, x):
        xc = x.permute(0, 3, 1, 2)
        if self.padding == 'same':
            xc = _pad_nchw(xc, x.shape[1], x.shape[2], self.k, self.s)
        w = self.weights[0].permute(3, 2, 0, 1).contiguous()
        return F.conv2d(xc, w, stride=self.s).permute(0, 2, 3, 1)


class SeparableConv2D(Layer):
    prefix = 'separable_conv2d'

    def __init__(self, filters, kernel_size, strides=(1, 1), padding='valid', use_bias=True, name=None, **kw):
        super().__init__(name)
        self.filters, self.k, self.s, self.padding = filters, _pair(kernel_size), _pair(strides), padding
        assert not use_bias

    def build(self, x):
        c = x.shape[-1]
        self.weights = [torch.zeros(self.k + (c, 1), dtype=DTYPE[0]), torch.zeros((1, 1, c, self.filters), dtype=DTYPE[0])]

    def compute(self, x):
        c = x.shape[-1]
        xc = x.permute(0, 3, 1, 2)
        if self.padding == 'same':
            xc = _pad_nchw(xc, x.shape[1], x.shape[2], self.k, self.s)
        dw = self.weights[0].permute(2, 3, 0, 1).contiguous()
        if self.padding != 'same' and tuple(xc.shape[-2:]) == tuple(self.k):
            # a kernel as large as the map (the reference's soft-argmax, layers.py:160-200: one output position = the sum
            # of H*W products per channel): summed with torch.sum's cascade summation.  [r06] torch's CPU depthwise-conv
            # kernel accumulates the 1 024 taps of a 32 x 32 map one after the other in fp32 and lands ~100 ulp =
            # 1.2-1.8e-3 px from the fp64 result on broad heat-maps (measured on the exact fp64 logits of the real-size
            # SPNet goldens; (p * w).sum() on the same data: 8e-5 px) -- an artefact of THAT kernel's order, which would
            # have made this stand-in's fp32 run a worse fp32 reference than any real backend.  Same products, same
            # result in exact arithmetic; the fp64 goldens move by ~1e-16.
            y = (xc * dw[:, 0]).sum(dim=(-2, -1), keepdim=True)
        else:
            y = F.conv2d(xc, dw, stride=self.s, groups=c)
        pw = self.weights[1].permute(3, 2, 0, 1).contiguous()
        return F.conv2d(y, pw).permute(0, 2, 3, 1)


class Conv1D(Layer):
    prefix = 'conv1d'

    def __init__(self, filters, kernel_size, use_bias=True, name=None, **kw):
        super().__init__(name)
        self.filters, self.k = filters, kernel_size
        assert not use_bias

    def build(self, x):
        self.weights = [torch.zeros((self.k, x.shape[-1], self.filters), dtype=DTYPE[0])]

    def compute(self, x):  # [N, L, C] valid conv
        w = self.weights[0].permute(2, 1, 0).contiguous()
        return F.conv1d(x.permute(0, 2, 1), w).permute(0, 2, 1)


class Dense(Layer):
    prefix = 'dense'

    def __init__(self, units, use_bias=True, name=None, **kw):
        super().__init__(name)
        self.units, self.use_bias = units, use_bias
        assert not use_bias, 'only the frozen bias-free aggregation Dense is on the hot path'

    def build(self, x):
        self.weights = [torch.zeros((x.shape[-1], self.units), dtype=DTYPE[0])]

    def compute(self, x):
        return x @ self.weights[0]


class BatchNormalization(Layer):
    prefix = 'batch_normalization'

    def __init__(self, axis=-1, scale=True, epsilon=1e-3, name=None, **kw):
        super().__init__(name)
        assert axis == -1
        self.scale, self.eps = scale, epsilon

    def build(self, x):
        c = x.shape[-1]
        z, o = torch.zeros(c, dtype=DTYPE[0]), torch.ones(c, dtype=DTYPE[0])
        self.weights = ([o.clone()] if self.scale else []) + [z.clone(), z.clone(), o.clone()]   # [gamma,] beta, mean, var

    def compute(self, x):
        ws = self.weights
        gamma = ws[0] if self.scale else None
        beta, mean, var = ws[-3], ws[-2], ws[-1]
        inv = torch.rsqrt(var + self.eps)
        if gamma is not None:
            inv = inv * gamma
        return x * inv + (beta - mean * inv)


class Activation(Layer):
    prefix = 'activation'

    def __init__(self, activation, name=None, **kw):
        super().__init__(name)
        self.act = activation

    def compute(self, x):
        a = self.act
        if callable(a):
            return a(x)
        if a == 'relu':
            return torch.clamp_min(x, 0)
        if a == 'sigmoid':
            return torch.sigmoid(x)
        if a == 'softmax':
            return torch.softmax(x, dim=-1)
        raise NotImplementedError(a)


class Lambda(Layer):
    prefix = 'lambda'

    def __init__(self, function, name=None, **kw):
        super().__init__(name)
        self.function = function

    def __call__(self, x):
        out = self.function(x)       # K.* and operators are polymorphic: symbolic in, symbolic out
        if isinstance(out, KTensor):
            if out is x or hasattr(out, '_kh'):
                out = KTensor(lambda a: a, [out])
            KNode(self, x if isinstance(x, (list, tuple)) else [x], [out])
        return out


class _Pool(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding='valid', name=None, **kw):
        super().__init__(name)
        self.k = _pair(pool_size)
        self.s = self.k if strides is None else _pair(strides)
        self.padding = padding


class MaxPooling2D(_Pool):
    prefix = 'max_pooling2d'

    def compute(self, x):
        xc = x.permute(0, 3, 1, 2)
        if self.padding == 'same':
            xc = _pad_nchw(xc, x.shape[1], x.shape[2], self.k, self.s, value=-math.inf)
        return F.max_pool2d(xc, self.k, self.s).permute(0, 2, 3, 1)


class AveragePooling2D(_Pool):
    prefix = 'average_pooling2d'

    def compute(self, x):
        assert self.padding == 'valid'
        return F.avg_pool2d(x.permute(0, 3, 1, 2), self.k, self.s).permute(0, 2, 3, 1)


class GlobalMaxPooling2D(Layer):
    prefix = 'global_max_pooling2d'

    def compute(self, x):
        return torch.amax(x, dim=(1, 2))


class GlobalMaxPooling1D(Layer):
    prefix = 'global_max_pooling1d'

    def compute(self, x):
        return torch.amax(x, dim=1)


class UpSampling2D(Layer):
    prefix = 'up_sampling2d'

    def __init__(self, size=(2, 2), name=None, **kw):
        super().__init__(name)
        self.size = _pair(size)

    def compute(self, x):
        return x.repeat_interleave(self.size[0], dim=1).repeat_interleave(self.size[1], dim=2)


class ZeroPadding2D(Layer):
    prefix = 'zero_padding2d'

    def __init__(self, padding=(1, 1), name=None, **kw):
        super().__init__(name)
        (self.pt, self.pb), (self.pl, self.pr) = padding

    def compute(self, x):
        return F.pad(x, (0, 0, self.pl, self.pr, self.pt, self.pb))


class TimeDistributed(Layer):
    prefix = 'time_distributed'

    def __init__(self, layer, name=None, input_shape=None, **kw):
        super().__init__(name)
        self.layer = layer

    def build(self, x):
        inner = x.reshape((-1,) + tuple(x.shape[2:]))
        if not getattr(self.layer, 'built', True):
            self.layer(inner)            # builds the wrapped layer on a torch tensor

    def compute(self, x):
        n, t = x.shape[0], x.shape[1]
        y = self.layer(x.reshape((n * t,) + tuple(x.shape[2:])))
        return y.reshape((n, t) + tuple(y.shape[1:]))


def _merge(prefix_, fn):
    cls = type(prefix_.capitalize(), (Layer,), {'prefix': prefix_, 'compute': lambda self, v: fn(list(v))})

    def op(tensors, name=None, **kw):
        return cls(name=name)(list(tensors))
    return op


add = _merge('add', lambda v: sum(v[1:], v[0]))
multiply = _merge('multiply', lambda v: math.prod(v[1:], start=v[0]))


def concatenate(tensors, axis=-1, name=None, **kw):
    lay = Layer(name=name or _auto_name('concatenate'))
    lay.compute = lambda v: torch.cat(list(v), dim=axis)
    return lay(list(tensors))


# ------------------------------------------------------------------------------------------------- models
class Model(Layer):
    prefix = 'model'

    def __init__(self, inputs=None, outputs=None, name=None, **kw):
        super().__init__(name)
        self._single_in = not isinstance(inputs, (list, tuple))
        self._single_out = not isinstance(outputs, (list, tuple))
        self.inputs = [inputs] if self._single_in else list(inputs)
        self.outputs = [outputs] if self._single_out else list(outputs)
        self.built = True
        # layers reachable from the outputs (stop at this model's inputs), in creation order
        seen, found = set(), []
        stop = {id(t) for t in self.inputs}
        stack = list(self.outputs)
        while stack:
            t = stack.pop()
            if id(t) in seen or id(t) in stop:
                continue
            seen.add(id(t))
            lay = getattr(t, '_layer', None)
            if lay is not None and lay not in found:
                found.append(lay)
            stack.extend(t.parents)
        self.layers = found

    @property
    def input(self): return self.inputs[0] if self._single_in else self.inputs
    @property
    def output(self): return self.outputs[0] if self._single_out else self.outputs

    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
            if isinstance(l, TimeDistributed) and l.layer.name == name:
                return l.layer
        raise ValueError('No such layer: ' + name)

    def compute(self, x):
        vals = x if isinstance(x, (list, tuple)) else [x]
        feed = {id(t): v for t, v in zip(self.inputs, vals)}
        cache = {}
        outs = [o._eval(feed, cache) for o in self.outputs]
        return outs[0] if self._single_out else outs

    def __call__(self, x):
        out = _apply(self.compute, x)
        xs = x if isinstance(x, (list, tuple)) else [x]
        if self._single_out:
            if isinstance(out, KTensor):
                out._layer = self
                KNode(self, xs, [out])
            return out
        if isinstance(out, KTensor):        # multi-output model called symbolically -> one KTensor per output
            outs = [KTensor((lambda v, i=i: v[i]), [out]) for i in range(len(self.outputs))]
            for o in outs:
                o._layer = self
            KNode(self, xs, outs)
            return outs
        return out

    def predict(self, x, batch_size=None, verbose=0):
        with torch.no_grad():
            xs = x if isinstance(x, (list, tuple)) else [x]
            vals = [torch.from_numpy(np.ascontiguousarray(a)).to(DTYPE[0]) for a in xs]
            out = self.compute(vals if not self._single_in else vals[0])
        outs = out if isinstance(out, (list, tuple)) else [out]
        res = [o.numpy() for o in outs]
        return res[0] if len(res) == 1 else res

    def summary(self, *a, **k):
        pass

    def load_weights(self, filepath, by_name=False):
        """keras.engine.topology.load_weights_from_hdf5_group[_by_name] on a Keras-2 weight file."""
        from deephar_amd import hdf5
        f = hdf5.File(filepath)
        root = f['model_weights'] if 'model_weights' in f else f
        names = [n.decode() for n in np.atleast_1d(root.attrs['layer_names'])]
        groups = []
        for n in names:
            wn = [w.decode() for w in np.atleast_1d(root[n].attrs['weight_names'])] if len(root[n].attrs['weight_names']) else []
            if wn:
                groups.append((n, [np.asarray(root[n][w]) for w in wn]))
        layers = [l for l in keras_layers(self) if layer_weights(l)]
        if by_name:
            index = {}
            for l in layers:
                index.setdefault(l.name, []).append(l)
            pairs = [(l, vals) for n, vals in groups for l in index.get(n, [])]
        else:
            if len(groups) != len(layers):
                raise ValueError('You are trying to load a weight file containing %d layers into a model with %d layers.'
                                 % (len(groups), len(layers)))
            pairs = list(zip(layers, [vals for _, vals in groups]))
        for layer, vals in pairs:
            slots = layer_weights(layer)
            if len(slots) != len(vals):
                raise ValueError('Layer %s expects %d weights, got %d' % (layer.name, len(slots), len(vals)))
            for (_, owner, i), v in zip(slots, vals):
                assert tuple(owner.weights[i].shape) == tuple(v.shape), (layer.name, owner.name, v.shape)
                owner.weights[i] = torch.from_numpy(np.ascontiguousarray(v)).to(DTYPE[0])


# ------------------------------------------------------------------------------------------------- backend
def _K():
    K = types.ModuleType('keras.backend')

    def uni(f):
        return lambda x, *a, **k: _apply(lambda t: f(t, *a, **k), x)

    def _axes(axis):
        return tuple(axis) if isinstance(axis, (list, tuple)) else axis

    K.epsilon = lambda: 1e-7
    K.image_data_format = lambda: 'channels_last'
    K.set_image_data_format = lambda fmt: None
    K.int_shape = lambda x: x.shape if isinstance(x, KTensor) else (None,) + tuple(x.shape[1:])
    K.ndim = lambda x: len(x.shape)
    K.exp = uni(torch.exp)
    K.log = uni(torch.log)
    K.max = lambda x, axis=None, keepdims=False: _apply(lambda t: torch.amax(t, dim=_axes(axis), keepdim=keepdims), x)
    K.sum = lambda x, axis=None, keepdims=False: _apply(lambda t: torch.sum(t, dim=_axes(axis), keepdim=keepdims), x)
    K.mean = lambda x, axis=None, keepdims=False: _apply(lambda t: torch.mean(t, dim=_axes(axis), keepdim=keepdims), x)
    K.clip = lambda x, lo, hi: _apply(lambda t: torch.clamp(t, min=lo, max=hi), x)
    K.squeeze = lambda x, axis: _apply(lambda t: t.squeeze(axis), x)
    K.expand_dims = lambda x, axis=-1: _apply(lambda t: t.unsqueeze(axis), x)
    K.tile = lambda x, n: _apply(lambda t: t.repeat(*[int(v) for v in n]), x)
    K.reshape = lambda x, shape: _apply(lambda t: t.reshape(tuple(int(v) for v in shape)), x)
    K.stop_gradient = lambda x: x
    K.cast = lambda x, dtype: x
    return K


def install():
    """Register fake `keras` / `tensorflow` modules in sys.modules (idempotent)."""
    if 'keras' in sys.modules and getattr(sys.modules['keras'], '_minikeras', False):
        return sys.modules['keras']
    keras = types.ModuleType('keras')
    keras._minikeras = True
    keras.__version__ = '2.1.4-minikeras'
    K = _K()
    layers = types.ModuleType('keras.layers')
    real = dict(Input=Input, Lambda=Lambda, Dense=Dense, Activation=Activation, Conv1D=Conv1D, Conv2D=Conv2D,
                SeparableConv2D=SeparableConv2D, BatchNormalization=BatchNormalization,
                TimeDistributed=TimeDistributed, multiply=multiply, concatenate=concatenate, add=add,
                AveragePooling2D=AveragePooling2D, MaxPooling2D=MaxPooling2D, GlobalMaxPooling1D=GlobalMaxPooling1D,
                GlobalMaxPooling2D=GlobalMaxPooling2D, ZeroPadding2D=ZeroPadding2D, UpSampling2D=UpSampling2D)
    for k, v in real.items():
        setattr(layers, k, v)

    def unused(name):
        def ctor(*a, **k):
            raise NotImplementedError('%s is not on the hot path' % name)
        return ctor
    for name in ('Flatten', 'Dropout', 'LeakyReLU', 'Conv3D', 'Conv2DTranspose', 'LocallyConnected1D', 'SimpleRNN',
                 'LSTM', 'average', 'maximum', 'MaxPooling3D', 'GlobalMaxPooling3D', 'GlobalAveragePooling1D',
                 'GlobalAveragePooling2D', 'UpSampling3D'):
        setattr(layers, name, unused(name))
    models = types.ModuleType('keras.models')
    models.Model = Model
    mods = {'keras': keras, 'keras.backend': K, 'keras.layers': layers, 'keras.models': models}
    for sub, names in (('optimizers', ('RMSprop', 'SGD', 'Adam')), ('constraints', ('unit_norm',)),
                       ('regularizers', ('l1', 'l2')), ('losses', ('binary_crossentropy',)),
                       ('callbacks', ('Callback',)), ('utils', ('Sequence',))):
        m = types.ModuleType('keras.' + sub)
        for n in names:
            setattr(m, n, type(n, (), {'__init__': lambda self, *a, **k: None}))
        mods['keras.' + sub] = m
        setattr(keras, sub, m)
    keras.backend, keras.layers, keras.models = K, layers, models
    tf = types.ModuleType('tensorflow')
    tf.divide = lambda a, b: a / b
    mods['tensorflow'] = tf
    sys.modules.update(mods)
    return keras


def weight_layers():
    return list(_layer_registry)


# ------------------------------------------------------------------------------- Keras topology emulation
def keras_layers(model):
    """`Model.layers` in Keras 2.1.4's order (keras/engine/topology.py, Container.__init__): depth-first walk
    from the outputs numbers the layers (pre-order), node depths are longest paths to an output, a layer sits
    at the largest depth of its nodes, layers are listed by decreasing depth then by traversal index.  This is
    the order `save_weights` writes groups in and `load_weights` (by_name=False) consumes them."""
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
    layer_index, order, finished = {}, [], set()

    def visit(t):
        layer, ni, _ = t._kh
        node = layer.inbound_nodes[ni]
        if id(node) in finished:
            return
        if id(layer) not in layer_index:
            layer_index[id(layer)] = len(layer_index)
        for it in node.input_tensors:
            visit(it)
        finished.add(id(node))
        order.append(node)

    for o in model.outputs:
        visit(o)
    node_depth, layer_depth, layers = {}, {}, {}
    for node in reversed(order):
        lay = node.outbound_layer
        d = max(node_depth.setdefault(id(node), 0), layer_depth.get(id(lay), 0))
        layer_depth[id(lay)] = d
        layers[id(lay)] = lay
        node_depth[id(node)] = d
        for it in node.input_tensors:
            il, ni, _ = it._kh
            inode = il.inbound_nodes[ni]
            node_depth[id(inode)] = max(d + 1, node_depth.get(id(inode), 0))
    return sorted(layers.values(), key=lambda l: (-layer_depth[id(l)], layer_index[id(l)]))


def _own_weights(layer):
    """(trainable, non-trainable) lists of (suffix, index into layer.weights) for a leaf layer."""
    n = len(layer.weights or [])
    if isinstance(layer, BatchNormalization):
        names = (['gamma'] if layer.scale else []) + ['beta', 'moving_mean', 'moving_variance']
        tr = list(range(n - 2))
        return [(names[i], i) for i in tr], [(names[i], i) for i in (n - 2, n - 1)]
    names = {Conv2D: ['kernel'], Conv1D: ['kernel'], Dense: ['kernel'],
             SeparableConv2D: ['depthwise_kernel', 'pointwise_kernel']}.get(type(layer), ['w%d' % i for i in range(n)])
    return [(names[i], i) for i in range(n)], []


def layer_weights(layer):
    """`layer.weights` as Keras orders them: [(weight name, owner layer, index)], trainable first, then
    non-trainable; frozen layers / frozen nested Models report everything as non-trainable (topology.py:
    Layer.weights, Container.trainable_weights / non_trainable_weights, wrappers.py: Wrapper)."""
    def tr(l):
        if isinstance(l, TimeDistributed):
            return tr(l.layer)
        if not l.trainable:
            return []
        if isinstance(l, Model):
            return [w for x in keras_layers(l) for w in tr(x)]
        return [('%s/%s:0' % (l.name, s), l, i) for s, i in _own_weights(l)[0]] if l.weights else []

    def ntr(l):
        if isinstance(l, TimeDistributed):
            return ntr(l.layer)
        if isinstance(l, Model):
            inner = keras_layers(l)
            w = [w for x in inner for w in ntr(x)]
            return w if l.trainable else [w_ for x in inner for w_ in tr(x)] + w
        if not l.weights:
            return []
        t, n = _own_weights(l)
        own = (n if l.trainable else t + n)
        return [('%s/%s:0' % (l.name, s), l, i) for s, i in own]

    return tr(layer) + ntr(layer)


def save_layout(model):
    """What `model.save_weights` puts in the HDF5 file: [(group name, [(dataset name, array)])] in file order,
    weight-less layers included with an empty list (Keras writes an empty group for them)."""
    out = []
    for lay in keras_layers(model):
        seen, ws = {}, []
        for name, owner, i in layer_weights(lay):
            k = seen.get(name, 0)
            seen[name] = k + 1
            if k:
                head, tail = name.split('/', 1)
                name = '%s_%d/%s' % (head, k, tail)
            ws.append((name, owner.weights[i].detach().cpu().numpy()))
        out.append((lay.name, ws))
    return out
