#!/usr/bin/env python
"""Convert a Keras 2.1.4 `save_weights` HDF5 file into the .npz that deephar_amd.Model.load_weights reads.
Needs h5py (not in the main interpreter of this image; /opt/conda/bin/python3.9 has it).

    /opt/conda/bin/python3.9 tools/h5_to_npz.py weights_PE_MPII_cvpr18_19-09-2017.h5 weights_PE_MPII.npz

Keys are 'h5:<index>:<keras weight name>' in the file's layer_names / weight_names order (nested Models are one
group holding all their inner weights)."""
import sys

import numpy as np


def convert(src, dst):
    import h5py
    out, i = {}, 0
    with h5py.File(src, 'r') as f:
        root = f['model_weights'] if 'model_weights' in f else f
        for lname in root.attrs['layer_names']:
            g = root[lname]
            for wname in g.attrs['weight_names']:
                key = wname.decode() if isinstance(wname, bytes) else wname
                out['h5:%d:%s' % (i, key)] = np.asarray(g[wname])
                i += 1
    np.savez(dst, **out)
    return i


if __name__ == '__main__':
    n = convert(sys.argv[1], sys.argv[2])
    print('wrote %d tensors to %s' % (n, sys.argv[2]))
