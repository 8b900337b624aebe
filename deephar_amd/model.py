"""`Model`: the Keras-Model-shaped object the reference's callers hold.

Surface used by exp/ and exp/common/*_tools.py and reproduced here with the same meaning:
  predict(x, batch_size=32, verbose=0)   mpii_tools.py:25,56,86  h36m_tools.py:46  penn_tools.py:18,55,124
  load_weights(path, by_name=False)      eval_mpii_singleperson.py:54, eval_penn_multitask.py:76
  outputs / output / input / inputs      eval_mpii_singleperson.py:56-61, eval_speed2d.py:62-68
  input_shape, get_input_shape_at(0)     h36m_tools.py:20, mpii_tools.py:67
  get_layer(name), layers, name, summary action.py:117-179
  __call__(tensor)                        nested models: reception.py:93,131; action.py:124-125,357,366
Execution is the gfx950 engine (deephar_amd/engine); there is no CPU execution path.
"""
import numpy as np

from . import graph as G


class CallRec:
    """One call of a nested Model on new tensors (a Keras Node); kept on the cloned nodes so that
    deephar_amd/keras_compat.py can rebuild the layer graph Keras would have seen."""

    def __init__(self, model, inputs, outputs):
        self.model, self.inputs, self.outputs = model, list(inputs), list(outputs)


_ENGINE_CHOICES = {'stream_policy': ('list', 'tail'), 'gemm_precision': ('f32', 'bf16x3')}


class Model:
    def __init__(self, inputs, outputs, name=None):
        self._single_in = not isinstance(inputs, (list, tuple))
        self._single_out = not isinstance(outputs, (list, tuple))
        self.inputs = [inputs] if self._single_in else list(inputs)
        self.outputs = [outputs] if self._single_out else list(outputs)
        for t in self.inputs + self.outputs:
            if not isinstance(t, G.Tensor):
                raise TypeError('Model inputs/outputs must be symbolic tensors, got %r' % (t,))
        self.name = name or 'model_%d' % next(G._uid)
        self.trainable = True
        self._plan = None
        self._exec = None
        # >1: independent branches run on parallel hipGraph branches (engine/schedule.py).  One stream is the default
        # since round 3: the kernels of two branches time-slice the SIMDs instead of filling each other's bubbles
        # (profiles/r03_concurrency_study.md); with the hourglass' up-sampled residuals folded into the producing
        # convolutions (planner R3) one stream is 2 % faster than two on the MPII model, 12 % on the PennAction merge
        # model and 7 % on SPNet-NTU (frame-sharded stages)
        self.num_streams = max(1, int(__import__('os').environ.get('DEEPHAR_STREAMS', '1')))
        # how steps are spread when num_streams > 1: 'list' = the list scheduler (branches of an hourglass; throughput
        # regime), 'tail' = two streams, the second runs a suffix of the step list -- SPNet's action stream beside its pose
        # stream -- for the latency regime of a couple of clips per call (engine/schedule.py: assign_streams_tail)
        self.stream_policy = __import__('os').environ.get('DEEPHAR_STREAM_POLICY', 'list')
        # uint8 inputs are raw frames: predict() normalises them on the GPU exactly like the reference's loaders do
        # on the host (utils/transform.normalize_channels(frame, channel_power), transform.py:212-231)
        self.channel_power = 1
        # 'f32' (default): every conv on the fp32 matrix path, an exact k-ordered fmaf chain.  'bf16x3': pointwise / K x K
        # GEMM-shaped convs split their fp32 operands exactly into three bf16 parts and run six partial products on the
        # bf16 matrix cores with fp32 accumulation (csrc/gemm1x1s.hip): same accuracy class, ~1.7x faster, not bit-identical
        self.gemm_precision = __import__('os').environ.get('DEEPHAR_GEMM', 'f32')
        # validates connectivity early (raises like Keras' "graph disconnected")
        self._nodes = G.topo_nodes(self.outputs)
        reach = {t.uid for t in self.inputs}
        for n in self._nodes:
            for t in n.inputs:
                if t.uid not in reach:
                    raise ValueError('Graph disconnected: %r is not derived from the model inputs' % t)
            for o in n.outputs:
                reach.add(o.uid)

    # ---- engine options: changing one after the first predict re-plans (plan, bound arenas and graphs are dropped) ---
    def _engine_option(name):                  # noqa: N805  (class-body helper)
        def get(self):
            return self.__dict__['_opt_' + name]

        def set_(self, value):
            if name in _ENGINE_CHOICES and value not in _ENGINE_CHOICES[name]:      # (a typo fails here, not at the first predict)
                raise ValueError('%s must be one of %s, got %r' % (name, '/'.join(map(repr, _ENGINE_CHOICES[name])), value))
            if self.__dict__.get('_opt_' + name, value) != value and self.__dict__.get('_plan') is not None:
                self._plan, self._exec = None, None
            self.__dict__['_opt_' + name] = value
        return property(get, set_)

    gemm_precision = _engine_option('gemm_precision')
    num_streams = _engine_option('num_streams')
    stream_policy = _engine_option('stream_policy')
    del _engine_option

    # ---- Keras-like attributes -----------------------------------------------------------------------
    @property
    def input(self):
        return self.inputs[0] if len(self.inputs) == 1 else self.inputs

    @property
    def output(self):
        return self.outputs[0] if len(self.outputs) == 1 else self.outputs

    @property
    def input_shape(self):
        shapes = [(None,) + t.shape for t in self.inputs]
        return shapes[0] if len(shapes) == 1 else shapes

    @property
    def output_shape(self):
        shapes = [(None,) + t.shape for t in self.outputs]
        return shapes[0] if len(shapes) == 1 else shapes

    def get_input_shape_at(self, index):
        if index != 0:
            raise ValueError('only node index 0 exists')
        return self.input_shape

    @property
    def layers(self):
        """Top-level layers in graph order: nested Models (by call) and weight-owning layers."""
        seen, out = set(), []
        for n in self._nodes:
            path = n.attrs.get('_models')
            objs = [path[0]] if path else list(n.layers.values())
            for o in objs:
                if id(o) not in seen:
                    seen.add(id(o))
                    out.append(o)
        return out

    def get_layer(self, name=None, index=None):
        if index is not None:
            return self.layers[index]
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError('No such layer: %s' % name)

    @property
    def params(self):
        seen, out = set(), []
        for n in self._nodes:
            for layer in n.layers.values():
                for p in layer.params:
                    if id(p) not in seen:
                        seen.add(id(p))
                        out.append(p)
        return out

    @property
    def weights(self):
        return self.params

    def count_params(self):
        return int(sum(np.prod(p.shape) for p in self.params))

    def get_weights(self):
        return [p.value for p in self.params]

    def set_weights(self, values):
        ps = self.params
        if len(values) != len(ps):
            raise ValueError('model %s expects %d weight arrays, got %d' % (self.name, len(ps), len(values)))
        for p, v in zip(ps, values):
            p.set(v)

    def summary(self, print_fn=print):
        print_fn('Model "%s": %d graph nodes, %d parameters' % (self.name, len(self._nodes), self.count_params()))
        for l in self.layers:
            if isinstance(l, Model):
                print_fn('  %-28s Model   %10d params' % (l.name, l.count_params()))
            else:
                print_fn('  %-28s %-18s %s' % (l.name, l.cls, [p.shape for p in l.params]))

    # ---- functional call (nested model) --------------------------------------------------------------
    def __call__(self, x):
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        outs = G.clone_subgraph(self.inputs, self.outputs, xs)
        rec = CallRec(self, xs, outs)
        tagged = set()
        stack = [t.node for t in outs if t.node is not None]
        stop = {t.uid for t in xs}
        while stack:
            n = stack.pop()
            if n.uid in tagged:
                continue
            tagged.add(n.uid)
            n.attrs['_models'] = [self] + list(n.attrs.get('_models', []))
            n.attrs['_calls'] = [rec] + list(n.attrs.get('_calls', []))
            for t in n.inputs:
                if t.uid not in stop and t.node is not None:
                    stack.append(t.node)
        return outs[0] if self._single_out else outs

    # ---- weights I/O -------------------------------------------------------------------------------------
    def load_weights(self, filepath, by_name=False):
        from . import weights as W
        W.load_weights(self, filepath, by_name=by_name)
        if self._exec is not None:
            self._exec.refresh_weights()

    def save_weights(self, filepath):
        from . import weights as W
        W.save_weights(self, filepath)

    def export_plan(self, filepath, batch_size, uint8=False):
        """Write the bound, autotuned launch list + weight image of this model for `batch_size` items: the blob the
        C-level executor of the library runs without Python (dh_plan_create / dh_forward, include/deephar_hip.h;
        INTEGRATION.md shows a C host).  uint8=True: the plan takes raw uint8 frames (what `predict` does with uint8
        arrays: normalised on the GPU with `self.channel_power`).  Returns the number of bytes written."""
        from .engine.serialize import dump_plan
        blob = dump_plan(self, int(batch_size), u8_norm=self.channel_power if uint8 else None)
        with open(filepath, 'wb') as f:
            f.write(blob)
        return len(blob)

    # ---- execution -----------------------------------------------------------------------------------------
    @property
    def plan(self):
        if self._plan is None:
            from .engine.planner import build_plan
            self._plan = build_plan(self.inputs, self.outputs, nstreams=self.num_streams,
                                    gemm_precision=self.gemm_precision, stream_policy=self.stream_policy)
        return self._plan

    @property
    def executor(self):
        if self._exec is None:
            from .engine.executor import Executor
            self._exec = Executor(self.plan, stream_role=getattr(self, '_stream_role', 'compute'))
        return self._exec

    def predict(self, x, batch_size=32, verbose=0):
        """Forward pass on the GPU in chunks of `batch_size` (keras Model.predict semantics): returns one
        np.float32 array per model output, or a bare array when the model has a single output.
        uint8 arrays are taken as raw frames: 4x fewer bytes cross PCIe and the /255, -0.5, x2 normalisation of the
        reference's data loaders is applied inside the first convolution (`self.channel_power` as in DataConfig)."""
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        if len(xs) != len(self.inputs):
            raise ValueError('model %s expects %d input arrays, got %d' % (self.name, len(self.inputs), len(xs)))
        xs = [np.asarray(a) for a in xs]
        total = xs[0].shape[0]
        for a, t in zip(xs, self.inputs):
            if tuple(a.shape[1:]) != t.shape or a.shape[0] != total:
                raise ValueError('input array has shape %s, model expects (N,)+%s' % (a.shape, t.shape))
        raw = [a.dtype == np.uint8 for a in xs]
        if any(raw) and not all(raw):
            raise ValueError('either all inputs are uint8 frames or none is')
        u8_norm = self.channel_power if all(raw) and raw else None
        if total == 0:                       # nothing to run (and nothing that needs the GPU)
            outs = [np.zeros((0,) + t.shape, np.float32) for t in self.outputs]
            return outs[0] if len(outs) == 1 else outs
        ex = self.executor
        bs = int(min(batch_size or total, total))
        outs = ex.run_pipelined(xs, bs, u8_norm=u8_norm, verbose=verbose)
        return outs[0] if len(outs) == 1 else outs


def concatenate(tensors, axis=-1, name=None):
    """keras.layers.concatenate on symbolic outputs (re-wrapping idiom, eval_mpii_singleperson.py:56-61)."""
    from . import layers
    return layers.concatenate(tensors, axis=axis, name=name)
