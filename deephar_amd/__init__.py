"""deephar_amd -- MI355X (gfx950) native engine for the pose-regression hot path of dluvizon/deephar.

Host side (this package, Python): graph builders with the reference's `deephar.models` signatures, a
Keras-Model-shaped `Model` (predict / load_weights / outputs / get_layer), a fusing planner and an executor
that drives hand-written HIP kernels through the C-ABI of csrc/libdeephar_hip.so (include/deephar_hip.h).
There is no CPU execution path: without the HIP library and an AMD GPU, `predict` raises.
"""
from . import graph  # noqa: F401
from . import layers  # noqa: F401
from .model import Model, concatenate  # noqa: F401
from . import models  # noqa: F401
from . import weights  # noqa: F401

__version__ = '0.1.0'
