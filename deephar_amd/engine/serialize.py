"""Serialise a bound plan for the C-level executor of the library (include/deephar_hip.h: dh_plan_create / dh_forward /
dh_plan_destroy -- SURVEY.md 8b's plan / execute pair): a host written in C or C++ runs the model without Python.

Blob (little endian), version 2 (version 1 = float inputs only, no `u8 bytes` / `dtype`, still read by the library):
    header   'DHPL' u32 version | i32 batch n | u64 arena bytes | u64 weight bytes | u32 #inputs | u32 #outputs | u32 #steps
             | u64 u8 bytes (size of region 3: the byte staging buffers of a uint8-input plan, 0 otherwise)
    inputs   per input : u64 tagged pointer of the buffer dh_forward copies the caller's data into (region 1: the plan's
             float input inside the arena; region 3: the byte staging buffer of a uint8 plan) | u64 elements per batch
             item | u32 dtype (0 float32, 1 uint8) | u32 0
    outputs  per output: u64 arena byte offset | u64 pixels per batch item | u32 channels | u32 pixel pitch (floats)
    steps    per launch: u32 function id | u32 payload bytes | payload
               struct functions: the argument struct, byte for byte, then one u64 per further argument
               scalar functions: one u64 per argument (ints sign-extended, floats as their bit pattern in the low half)
             every device pointer (struct field or scalar) is written as 0 (NULL) or (region << 60) | byte offset,
             region 1 = the activation arena, 2 = the weight image, 3 = the uint8 input staging
    weights  the weight image (packed conv / depthwise kernels, BN affines, grids), every tensor 256-byte aligned
Launch order is the plan's step order, a valid single-stream schedule of the graph; tilings are the bound plan's
(autotuned) ones, so dh_forward reproduces Model.predict bit for bit.
"""
import ctypes as C
import struct

import numpy as np

from .. import _lib

MAGIC, VERSION = b'DHPL', 2
FUNCTIONS = ['dh_conv2d_f32', 'dh_dwconv2d_f32', 'dh_pool2d_f32', 'dh_upsample2x_add_f32', 'dh_eltwise_f32',
             'dh_softargmax2d_f32', 'dh_context_aggregation_f32', 'dh_depth_means_f32', 'dh_softargmax1d_f32',
             'dh_kronecker_f32', 'dh_global_maxmin_softmax_f32', 'dh_copy_channels_f32', 'dh_zeropad2d_f32',
             'dh_depth_from_maps_f32', 'dh_softargmax2d_context_f32', 'dh_normalize_u8_f32', 'dh_conv2d_dw_group_f32',
             'dh_conv2d_pair_f32', 'dh_conv2d_seg_f32']
ARENA, WEIGHTS, BYTES = 1, 2, 3


class _Regions:
    def __init__(self, bp):
        self.base = bp.base
        self.end = bp.base + bp.arena.numel() * 4
        st = bp.store
        tensors = [e[0] for e in st.conv.values()] + [e[0] for e in st.dw.values()] + list(st.const.values())
        for e in st.bn.values():
            tensors += [e[0], e[1]]
        self.known = sorted(((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), t) for t in tensors),
                            key=lambda r: r[0])
        self.offset = {}          # tensor data_ptr -> byte offset in the weight image
        self.image = []
        self.size = 0
        # uint8-input plans: the byte staging buffers (one per model input) form region 3, 256-byte aligned each
        self.u8 = []              # (lo, hi, region offset)
        self.u8_size = 0
        for buf, _, _ in (bp.u8 or {}).values():
            self.u8.append((buf.data_ptr(), buf.data_ptr() + buf.numel(), self.u8_size))
            self.u8_size = (self.u8_size + buf.numel() + 255) & ~255

    def tag(self, ptr):
        ptr = int(ptr or 0)
        if ptr == 0:
            return 0
        if self.base <= ptr < self.end:
            return (ARENA << 60) | (ptr - self.base)
        for lo, hi, off in self.u8:
            if lo <= ptr < hi:
                return (BYTES << 60) | (off + ptr - lo)
        for lo, hi, t in self.known:
            if lo <= ptr < hi:
                if lo not in self.offset:
                    self.size = (self.size + 255) & ~255
                    self.offset[lo] = self.size
                    self.image.append((self.size, t))
                    self.size += hi - lo
                return (WEIGHTS << 60) | (self.offset[lo] + ptr - lo)
        raise ValueError('pointer 0x%x belongs neither to the arena nor to the weight store' % ptr)


def _is_ptr_type(t):
    return t is C.c_void_p


def _scalars(sig, args, reg):
    assert len(sig) == len(args), (len(sig), len(args))
    parts = []
    for t, a in zip(sig, args):
        if _is_ptr_type(t):
            parts.append(struct.pack('<Q', reg.tag(a)))
        elif t is C.c_float:
            parts.append(struct.pack('<fI', float(a), 0))
        else:
            parts.append(struct.pack('<q', int(a)))
    return b''.join(parts)


def dump_plan(model, batch, u8_norm=None):
    """-> bytes.  `model`'s plan bound (and autotuned) for `batch`; needs a HIP device (the weight image is read back).
    u8_norm: None for float inputs; a channel_power (as for Executor.bind) for a plan that takes raw uint8 frames and
    normalises them on the GPU (inside the first convolution where the planner could fuse it)."""
    ex = model.executor
    torch = __import__('torch')
    with torch.cuda.device(ex.device), torch.cuda.stream(ex.stream):
        if ex.bound:
            ex.sync_weights()
        bp = ex.bind(batch, u8_norm=u8_norm)
    ex.stream.synchronize()
    lib = _lib.load()
    names = {n: i for i, n in enumerate(FUNCTIONS)}
    by_addr = {C.cast(getattr(lib, n), C.c_void_p).value: n for n in FUNCTIONS}
    reg = _Regions(bp)
    steps = []
    for idx, (fn, args, step) in enumerate(bp.calls):
        if idx in bp.noop_calls:                              # merged into an earlier launch (BoundPlan.group_launches)
            continue
        name = by_addr.get(C.cast(fn, C.c_void_p).value)
        if name is None:
            raise ValueError('step %s (%s) has no serialised form' % (step.kind, step.name))
        sig = _lib.SIGNATURES[name][1][:-1]                 # without the trailing stream
        if len(sig) and isinstance(getattr(sig[0], '_type_', None), type) and issubclass(sig[0]._type_, C.Structure):
            payload, nstruct = b'', 0
            while nstruct < len(sig) and isinstance(getattr(sig[nstruct], '_type_', None), type) and \
                    issubclass(sig[nstruct]._type_, C.Structure):     # (dh_conv2d_dw_group_f32 takes two structs)
                obj = args[nstruct]._obj
                raw = bytearray(bytes(obj))
                for fname, ftype in obj._fields_:
                    if _is_ptr_type(ftype):
                        off = getattr(type(obj), fname).offset
                        raw[off:off + 8] = struct.pack('<Q', reg.tag(getattr(obj, fname)))
                payload += bytes(raw)
                nstruct += 1
            payload += _scalars(sig[nstruct:], args[nstruct:], reg)
        else:
            payload = _scalars(sig, args, reg)
        steps.append(struct.pack('<II', names[name], len(payload)) + payload)
    plan = bp.plan
    head = MAGIC + struct.pack('<IiQQIIIQ', VERSION, bp.n, bp.arena.numel() * 4, (reg.size + 255) & ~255,
                               len(plan.inputs), len(plan.outputs), len(steps), reg.u8_size)
    ins = b''
    for v in plan.inputs:
        if bp.u8 is not None:
            ins += struct.pack('<QQII', reg.tag(bp.u8[id(v.buf)][0].data_ptr()), int(np.prod(v.shape)), 1, 0)
        else:
            ins += struct.pack('<QQII', reg.tag(bp.ptr(v)), int(np.prod(v.shape)), 0, 0)
    outs = b''
    for v in plan.outputs:
        if v.coff % 1 or v.ld < v.C:
            raise ValueError('output view not serialisable')
        outs += struct.pack('<QQII', bp.ptr(v) - bp.base, v.npix, v.C, v.ld)
    image = bytearray((reg.size + 255) & ~255)
    for off, t in reg.image:
        b = t.detach().cpu().contiguous().numpy().tobytes()
        image[off:off + len(b)] = b
    return head + ins + outs + b''.join(steps) + bytes(image)
