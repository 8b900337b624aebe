"""Host-side weight re-layout for the gfx950 kernels (delegates to the C-ABI so that other hosts get the
exact same packing: dh_conv2d_pack_weights_host)."""
import ctypes as C

import numpy as np

from .. import _lib


def pack_conv(w_hwio):
    """Keras HWIO conv kernel [kh,kw,cin,cout] -> ([Kp/4][Np][4] float32 flat array, Kp, Np)."""
    lib = _lib.load()
    w = np.ascontiguousarray(w_hwio, dtype=np.float32)
    kh, kw, cin, cout = w.shape
    kp, np_ = C.c_int(), C.c_int()
    _lib.check(lib.dh_conv2d_packed_dims(kh, kw, cin, cout, C.byref(kp), C.byref(np_)), 'packed_dims')
    out = np.empty(kp.value * np_.value, dtype=np.float32)
    _lib.check(lib.dh_conv2d_pack_weights_host(w.ctypes.data, out.ctypes.data, kh, kw, cin, cout), 'pack')
    return out, kp.value, np_.value


HALO_CHUNK = 16      # channels per resident chunk of csrc/conv_halo.hip


def halo_order(w_hwio):
    """HWIO kernel -> the [1, 1, K, Cout] kernel whose K runs chunk-major, [Cin/16][KH][KW][16], the order in which the
    halo-resident K x K kernel sums (dh_conv_args.w_split = 2, include/deephar_hip.h: dh_conv2d_halo_eligible)."""
    w = np.ascontiguousarray(w_hwio, dtype=np.float32)
    kh, kw, cin, cout = w.shape
    if cin % HALO_CHUNK:
        raise ValueError('chunk-major packing needs Cin %% %d == 0, got %d' % (HALO_CHUNK, cin))
    w = w.reshape(kh, kw, cin // HALO_CHUNK, HALO_CHUNK, cout).transpose(2, 0, 1, 3, 4)
    return np.ascontiguousarray(w).reshape(1, 1, kh * kw * cin, cout)


def pack_conv_halo(w_hwio):
    """Keras HWIO conv kernel -> ([Kp/4][Np][4] float32, Kp, Np) with K chunk-major (see halo_order)."""
    return pack_conv(halo_order(w_hwio))


def unpack_conv(packed, kh, kw, cin, cout):
    """Inverse of pack_conv (tests)."""
    k = kh * kw * cin
    kp = (k + 31) // 32 * 32
    np_ = (cout + 31) // 32 * 32
    a = packed.reshape(kp // 4, np_, 4).transpose(0, 2, 1).reshape(kp, np_)
    return a[:k, :cout].reshape(kh, kw, cin, cout)


def pack_conv_split(w_hwio):
    """Keras HWIO conv kernel -> split-bf16 packing for dh_conv_args.w_split = 1: every weight as three bf16 parts
    (exact: w = w1 + w2 + w3), laid out [Kp/8][3][Np][8]; returned as a float32-typed flat array of 1.5 * Kp * Np
    words (the bf16 bit patterns, two per word), Kp, Np.  Delegates to dh_conv2d_pack_weights_split_host."""
    lib = _lib.load()
    w = np.ascontiguousarray(w_hwio, dtype=np.float32)
    kh, kw, cin, cout = w.shape
    kp, np_ = C.c_int(), C.c_int()
    _lib.check(lib.dh_conv2d_packed_dims(kh, kw, cin, cout, C.byref(kp), C.byref(np_)), 'packed_dims')
    out = np.empty(3 * kp.value * np_.value, dtype=np.uint16)
    _lib.check(lib.dh_conv2d_pack_weights_split_host(w.ctypes.data, out.ctypes.data, kh, kw, cin, cout), 'pack split')
    return out.view(np.float32), kp.value, np_.value


def unpack_conv_split(packed, kh, kw, cin, cout):
    """Inverse of pack_conv_split (tests): the sum of the three parts, which must reproduce the weights exactly."""
    k = kh * kw * cin
    kp = (k + 31) // 32 * 32
    np_ = (cout + 31) // 32 * 32
    u = packed.view(np.uint16).reshape(kp // 8, 3, np_, 8).astype(np.uint32) << 16
    parts = u.view(np.float32)                                  # [kg, part, n, 8]
    tot = (parts[:, 0].astype(np.float64) + parts[:, 1] + parts[:, 2]).transpose(0, 2, 1).reshape(kp, np_)
    return tot[:k, :cout].reshape(kh, kw, cin, cout), parts
