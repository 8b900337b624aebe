#!/usr/bin/env python
"""Tiny HDF5 files written by the REAL libhdf5 (h5py), used to pin deephar_amd/hdf5.py's reader.
Needs h5py, which the main interpreter of this image lacks:

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py

keras_tiny.h5         h5py defaults (superblock v0, v1 object headers, symbol-table groups): a Keras-2.1.4-shaped
                      weight file -- `layer_names` / `weight_names` fixed-length-string attributes, nested
                      '<layer>/<weight>:0' datasets, 40 groups (several symbol nodes), plus a gzip+shuffle chunked
                      dataset, an int dataset, a scalar and a variable-length-string attribute
keras_tiny_latest.h5  libver='latest' (superblock v3, v2 object headers, link messages) under `model_weights/`
hdf5_expected.npz     every array, keyed '<file>:<path>'
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(2018)
ref = {}

with h5py.File(os.path.join(HERE, 'keras_tiny.h5'), 'w') as f:
    names = ['Stem', 'rBlock1'] + ['layer_%d' % i for i in range(38)]
    f.attrs['layer_names'] = [n.encode() for n in names]
    f.attrs['backend'] = b'tensorflow'
    f.attrs['keras_version'] = b'2.1.4'
    f.attrs['vl'] = np.array(['soft', 'argmax'], dtype=h5py.string_dtype())
    for i, n in enumerate(names):
        g = f.create_group(n)
        wn = [] if i % 4 == 3 else ['conv2d_%d/kernel:0' % i, 'batch_normalization_%d/beta:0' % i,
                                    'batch_normalization_%d/moving_mean:0' % i]
        g.attrs['weight_names'] = [w.encode() for w in wn]
        for w in wn:
            a = rng.standard_normal((1, 1, i % 3 + 1, 4) if 'kernel' in w else (4,)).astype(np.float32)
            g.create_dataset(w, data=a)
            ref['tiny:%s/%s' % (n, w)] = a
    a = np.arange(600, dtype=np.float64).reshape(12, 50)
    f.create_dataset('chunked', data=a, chunks=(5, 16), compression='gzip', shuffle=True)
    ref['tiny:chunked'] = a
    f.create_dataset('ints', data=np.arange(12, dtype=np.int32).reshape(3, 4))
    ref['tiny:ints'] = np.arange(12, dtype=np.int32).reshape(3, 4)
    f.create_dataset('scalar', data=np.float32(3.5))

with h5py.File(os.path.join(HERE, 'keras_tiny_latest.h5'), 'w', libver='latest') as f:
    g = f.create_group('model_weights')
    g.attrs['layer_names'] = [b'x', b'y']
    for n in ('x', 'y'):
        s = g.create_group(n)
        s.attrs['weight_names'] = [('%s/w:0' % n).encode()]
        a = rng.standard_normal((5, 2)).astype(np.float32)
        s.create_dataset('%s/w:0' % n, data=a)
        ref['latest:model_weights/%s/%s/w:0' % (n, n)] = a

np.savez(os.path.join(HERE, 'hdf5_expected.npz'), **ref)
print('wrote', len(ref), 'arrays')
