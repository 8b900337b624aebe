"""Deterministic stand-ins shared by tests/golden/make_golden.py (which feeds them to the REFERENCE's
exp/common/*_tools.py) and tests/test_evaltools.py (which feeds them to deephar_amd.evaltools): a model whose
`predict` is a cheap function of its input, and duck-typed datasets with the handful of methods the tools call."""
import numpy as np


class StubModel:
    """outputs: list of ('pose', J, D) / ('action', C) specs; predictions are float32 like Keras'."""

    def __init__(self, input_shape, specs, ds=None):
        self.input_shape = (None,) + tuple(input_shape)
        self.specs = specs
        self.outputs = [None] * len(specs)
        self.ds = ds
        self.calls = 0

    def get_input_shape_at(self, i):
        return self.input_shape

    def predict(self, x, batch_size=None, verbose=0):
        self.calls += 1
        x = x[0] if isinstance(x, (list, tuple)) else x
        x = np.asarray(x[:], dtype=np.float64)
        n = len(x)
        lead = x.shape[1:len(self.input_shape) - 3]            # (T,) for clips
        feat = x.reshape((n,) + lead + (-1,)).mean(axis=-1)      # [N(, T)]
        shift = 0.0 if self.ds is None else 0.05 * len(self.ds.custom)
        outs = []
        for k, spec in enumerate(self.specs):
            if spec[0] == 'pose':
                _, J, D = spec
                j = np.arange(J * D, dtype=np.float64).reshape(J, D)
                y = 0.5 + 0.35 * np.sin(3.1 * feat[..., None, None] * (k + 1) + 0.7 * j + shift)
                if D >= 3:
                    y[..., -1] = 0.5 + 0.5 * np.cos(5.0 * feat[..., None] + np.arange(J))      # confidence-like
            else:
                C = spec[1]
                f = feat.reshape(n, -1).mean(axis=-1)
                logits = 2.0 * np.sin((k + 2) * 1.7 * f[:, None] + np.arange(C))
                e = np.exp(logits - logits.max(axis=-1, keepdims=True))
                y = e / e.sum(axis=-1, keepdims=True)
            outs.append(y.astype(np.float32))
        return outs if len(outs) > 1 else outs[0]


class _Conf:
    fixed_hflip = 0


class FakeClipDataset:
    """Penn/NTU-shaped test split: `n` sequences, each cut into 3 clips of T frames; get_data honours
    dataconf.fixed_hflip (mirrors the frames) and an optional bbox (scales them)."""

    def __init__(self, n, T, num_actions, label_key, seed=0):
        rng = np.random.default_rng(seed)
        self.n, self.T, self.C, self.key = n, T, num_actions, label_key
        self.video = rng.uniform(-1, 1, (n, 3 * T, 8, 8, 3))
        self.labels = rng.integers(0, num_actions, n)
        self.dataconf = _Conf()
        self.use_gt_bbox = True
        self.bbox_seen = []

    def get_length(self, mode):
        return self.n

    def get_shape(self, key):
        assert key == self.key
        return (self.C,)

    def get_clip_index(self, i, mode, subsamples=(1,)):
        return [list(range(c * self.T, (c + 1) * self.T)) for c in range(3)]

    def get_data(self, i, mode, frame_list=None, bbox=None):
        if i == 2 and frame_list[0] == self.T:
            raise IOError('unreadable clip')                     # the tools must skip it and go on
        fr = self.video[i, frame_list]
        if self.dataconf.fixed_hflip:
            fr = fr[:, :, ::-1]
        if bbox is not None:
            self.bbox_seen.append(tuple(int(v) for v in bbox))
            fr = fr * (1.0 + 0.001 * float(bbox[2] - bbox[0]))
        onehot = np.zeros(self.C)
        onehot[self.labels[i]] = 1
        return {'frame': fr, self.key: onehot}


class FakeBoxDataset:
    """MPII-shaped: frames / afmat / bbox are live views that change when custom boxes are set."""

    def __init__(self, n, seed=0):
        rng = np.random.default_rng(seed)
        self.base = rng.uniform(-1, 1, (n, 8, 8, 3))
        self.box0 = np.concatenate([rng.uniform(0, 100, (n, 2)), rng.uniform(150, 300, (n, 2))], axis=1)
        self.custom = []
        self.log = []

    def set_custom_bboxes(self, mode, boxes):
        self.log.append(('set', mode, len(boxes)))
        self.custom = [] if len(boxes) == 0 else np.array(boxes, dtype=np.float64).copy()

    def clear_custom_bboxes(self, mode):
        self.log.append(('clear', mode))
        self.custom = []

    def boxes(self):
        return self.box0 if len(self.custom) == 0 else self.custom

    class _View:
        def __init__(self, fn):
            self.fn = fn

        def __getitem__(self, k):
            return self.fn()[k]

        def __len__(self):
            return len(self.fn())

    def frames(self):
        return self._View(lambda: self.base * (1 + 0.001 * (self.boxes()[:, 2] - self.boxes()[:, 0]))[:, None, None, None])

    def afmat(self):
        def make():
            b = self.boxes()
            A = np.zeros((len(b), 3, 3))
            A[:, 0, 0] = 1 / (b[:, 2] - b[:, 0])
            A[:, 1, 1] = 1 / (b[:, 3] - b[:, 1])
            A[:, 0, 2] = -b[:, 0] / (b[:, 2] - b[:, 0])
            A[:, 1, 2] = -b[:, 1] / (b[:, 3] - b[:, 1])
            A[:, 2, 2] = 1
            return A
        return self._View(make)

    def bbox(self):
        return self._View(lambda: self.boxes().copy())


class FakeSequence:
    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return self.ds.n

    def __getitem__(self, i):
        d = self.ds.get_data(i, 0, frame_list=list(range(self.ds.T)))
        return [d['frame'][None]], [d[self.ds.key][None]]
