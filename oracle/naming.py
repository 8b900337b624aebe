"""Weight naming shared by convention (not by code) with the product package.  TEST INFRASTRUCTURE.

Keys are '<scope>/<layer>/<weight>' where scope is the nested keras Model the layer lives in ('' at top
level), layer is the explicit Keras name when the reference passes one, else '<class>_<n>' with a counter
per (scope, class) starting at 1 in the reference's source order.  Weight names are Keras':
kernel | depthwise_kernel, pointwise_kernel | gamma, beta, moving_mean, moving_variance.
"""
import numpy as np
import torch


class Weights:
    """Read-only view over a {key: np.ndarray} dict handing out torch tensors; tracks naming scopes."""

    def __init__(self, arrays, dtype=torch.float32):
        self.arrays = arrays
        self.dtype = dtype
        self.scopes = []
        self.counters = {}
        self.used = set()

    def reset(self):
        self.scopes = []
        self.counters = {}
        self.used = set()

    # -- scopes -------------------------------------------------------------------------------
    def push(self, name):
        self.scopes.append(name)
        # a nested model called twice (e.g. TimeDistributed re-use) restarts its own counters
        for k in [k for k in self.counters if k[0] == name]:
            del self.counters[k]

    def pop(self):
        self.scopes.pop()

    @property
    def scope(self):
        return self.scopes[-1] if self.scopes else ''

    def auto(self, cls):
        key = (self.scope, cls)
        self.counters[key] = self.counters.get(key, 0) + 1
        return '%s_%d' % (cls, self.counters[key])

    # -- access -------------------------------------------------------------------------------
    def get(self, layer, weight, shape=None):
        key = '%s/%s/%s' % (self.scope, layer, weight)
        if key not in self.arrays:
            raise KeyError('oracle asked for weight %r which the model does not define' % key)
        a = self.arrays[key]
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError('weight %r has shape %s, oracle expects %s' % (key, a.shape, shape))
        self.used.add(key)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.dtype)

    def unused(self):
        return sorted(set(self.arrays) - self.used)
