"""ctypes binding of libdeephar_hip.so (the C-ABI declared in include/deephar_hip.h).

The product path has no CPU fallback: if the shared library is missing this module raises, loudly.
Build it with `python -m deephar_amd.csrc.build` (or __graft_entry__.build()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libdeephar_hip.so')

c_f = C.POINTER(C.c_float)
i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p


class ConvArgs(C.Structure):
    _fields_ = [(n, vp) for n in ('x', 'w', 'y', 'pre_scale', 'pre_shift', 'post_scale', 'post_shift',
                                   'res1', 'res2', 'in_lut')] + \
               [(n, i32) for n in ('N', 'H', 'W', 'Cin', 'ldx', 'OH', 'OW', 'Cout', 'ldy', 'KH', 'KW', 'SH',
                                   'SW', 'PT', 'PL', 'K', 'Kp', 'Np', 'ldr1', 'ldr2', 'pre_relu', 'post_relu',
                                   'up2', 'x_u8', 'w_split', 'res2_down', 'ldyp', 'x_resample')] + [('y_pool', vp)]


class ConvSeg(C.Structure):                      # == struct dh_conv_seg
    _fields_ = [('x2', vp)] + [(n, i32) for n in ('ldx2', 'c_split', 'pool_sh', 'reserved')]


class DwArgs(C.Structure):
    _fields_ = [(n, vp) for n in ('x', 'w', 'y', 'pre_scale', 'pre_shift')] + \
               [(n, i32) for n in ('N', 'H', 'W', 'C', 'ldx', 'ldy', 'KH', 'KW', 'PT', 'PL', 'pre_relu', 'up_in')]


class PoolArgs(C.Structure):
    _fields_ = [(n, vp) for n in ('x', 'y')] + \
               [(n, i32) for n in ('N', 'H', 'W', 'C', 'ldx', 'OH', 'OW', 'ldy', 'KH', 'KW', 'SH', 'SW', 'PT',
                                   'PL', 'mode')]


class EltArgs(C.Structure):
    _fields_ = [(n, vp) for n in ('a', 'b', 'c', 'y', 'scale', 'shift')] + \
               [(n, i32) for n in ('lda', 'ldb', 'ldc', 'ldy')] + [('npix', i64)] + \
               [(n, i32) for n in ('C', 'relu', 'op', 'bcast_b')]


class SamArgs(C.Structure):
    _fields_ = [(n, vp) for n in ('h', 'gx', 'gy', 'xy', 'conf_raw', 'conf_prob', 'prob', 'gmax')] + \
               [(n, i32) for n in ('F', 'H', 'W', 'C', 'ldh', 'ldxy', 'ldcr', 'ldcp', 'ldp')] + \
               [('alpha', C.c_float), ('conf_scale', C.c_float), ('xy_times_conf', i32)]


# name -> (restype, argtypes); every symbol include/deephar_hip.h declares
SIGNATURES = {
    'dh_version': (C.c_int, []),
    'dh_error_string': (C.c_char_p, [C.c_int]),
    'dh_device_info': (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    'dh_conv2d_packed_dims': (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2),
    'dh_conv2d_pack_weights_host': (C.c_int, [vp, vp] + [C.c_int] * 4),
    'dh_conv2d_pack_weights_split_host': (C.c_int, [vp, vp] + [C.c_int] * 4),
    'dh_conv2d_num_tile_cfgs': (C.c_int, []),
    'dh_conv2d_num_split_tile_cfgs': (C.c_int, []),
    'dh_conv2d_pick_tile_cfg': (C.c_int, [C.c_int, C.c_int]),
    'dh_conv2d_uses_split_k': (C.c_int, [C.POINTER(ConvArgs)]),
    'dh_conv2d_uses_first_layer_kernel': (C.c_int, [C.POINTER(ConvArgs)]),
    'dh_conv2d_split_eligible': (C.c_int, [C.POINTER(ConvArgs)]),
    'dh_conv2d_halo_eligible': (C.c_int, [C.POINTER(ConvArgs)]),
    'dh_conv2d_num_halo_tile_cfgs': (C.c_int, []),
    'dh_conv2d_f32': (C.c_int, [C.POINTER(ConvArgs), C.c_int, vp]),
    'dh_normalize_u8_f32': (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, vp]),
    'dh_dwconv2d_f32': (C.c_int, [C.POINTER(DwArgs), vp]),
    'dh_conv2d_dw_group_f32': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(DwArgs), vp]),
    'dh_conv2d_pair_f32': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), vp]),
    'dh_conv2d_seg_f32': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvSeg), vp]),
    'dh_pool2d_f32': (C.c_int, [C.POINTER(PoolArgs), vp]),
    'dh_upsample2x_add_f32': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int] + [C.c_int] * 4 + [vp]),
    'dh_eltwise_f32': (C.c_int, [C.POINTER(EltArgs), vp]),
    'dh_softargmax2d_f32': (C.c_int, [C.POINTER(SamArgs), vp]),
    'dh_softargmax2d_context_f32': (C.c_int, [C.POINTER(SamArgs), C.c_int, C.c_int, C.c_float, vp, C.c_int, vp]),
    'dh_context_aggregation_f32': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp]),
    'dh_depth_means_f32': (C.c_int, [vp, C.c_int, vp, vp] + [C.c_int] * 4 + [vp]),
    'dh_softargmax1d_f32': (C.c_int, [vp, vp, vp, C.c_int, vp] + [C.c_int] * 3 + [vp]),
    'dh_kronecker_f32': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int] + [C.c_int] * 4 + [vp]),
    'dh_global_maxmin_softmax_f32': (C.c_int, [vp, C.c_int, vp] + [C.c_int] * 4 + [vp]),
    'dh_copy_channels_f32': (C.c_int, [vp, C.c_int, vp, C.c_int, i64, C.c_int, vp]),
    'dh_zeropad2d_f32': (C.c_int, [vp, vp] + [C.c_int] * 8 + [vp]),
    'dh_depth_from_maps_f32': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int] + [C.c_int] * 3 + [vp]),
    'dh_plan_create': (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    'dh_plan_destroy': (C.c_int, [vp]),
    'dh_plan_batch': (C.c_int, [vp]),
    'dh_plan_num_inputs': (C.c_int, [vp]),
    'dh_plan_num_outputs': (C.c_int, [vp]),
    'dh_plan_input_items': (C.c_int64, [vp, C.c_int]),
    'dh_plan_input_is_u8': (C.c_int, [vp, C.c_int]),
    'dh_plan_output_items': (C.c_int64, [vp, C.c_int]),
    'dh_forward': (C.c_int, [vp, C.POINTER(vp), C.c_int, C.POINTER(vp), vp]),
    'dh_forward_host': (C.c_int, [vp, C.POINTER(vp), C.c_int, C.POINTER(vp)]),
    'dh_graph_begin_capture': (C.c_int, [vp]),
    'dh_graph_end_capture': (C.c_int, [vp, C.POINTER(vp)]),
    'dh_graph_launch': (C.c_int, [vp, vp]),
    'dh_graph_destroy': (C.c_int, [vp]),
    'dh_event_create': (C.c_int, [C.POINTER(vp)]),
    'dh_event_record': (C.c_int, [vp, vp]),
    'dh_event_synchronize': (C.c_int, [vp]),
    'dh_event_elapsed_ms': (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    'dh_event_destroy': (C.c_int, [vp]),
    'dh_stream_synchronize': (C.c_int, [vp]),
    'dh_stream_create': (C.c_int, [C.POINTER(vp)]),
    'dh_stream_destroy': (C.c_int, [vp]),
    'dh_event_create_sync': (C.c_int, [C.POINTER(vp)]),
    'dh_stream_wait_event': (C.c_int, [vp, vp]),
    'dh_stream_spin_us': (C.c_int, [vp, C.c_int]),
}

_lib = None


class DeepharHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, a different SONAME from the system
    # libamdhip64.so.7 this library links against).  Streams and device pointers only mean something inside
    # ONE runtime, so torch must be in the process (its runtime in the global symbol scope) before this
    # library is loaded; loaded the other way round the two runtimes coexist and every launch on a torch
    # stream fails.  A host without torch simply gets the system runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get('DEEPHAR_HIP_LIB', LIB_PATH)     # override: A/B runs of two builds of the library
    if not os.path.exists(path):
        raise DeepharHipError(
            'libdeephar_hip.so not found at %s -- the HIP back-end is mandatory (no CPU fallback). '
            'Build it with `python -m deephar_amd.csrc.build`.' % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().dh_error_string(rc).decode()
        raise DeepharHipError('%s failed: %s (rc=%d)' % (what or 'deephar_hip call', msg, rc))
