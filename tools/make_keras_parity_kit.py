#!/usr/bin/env python
"""Everything needed to pin the Keras/TF layer numerics the day a machine with keras==2.1.4 + tensorflow==1.6 is at
hand: writes, for each golden configuration of tests/refgolden.py,

    <out>/<tag>.h5        the deterministic synthetic weights in Keras' own save_weights layout (deephar_amd.hdf5)
    <out>/<tag>_x.npy     the seeded input
    <out>/run_in_keras.py a script for THAT machine: builds the reference's model from its own sources, load_weights
                          (by order, or by name for SPNet -- exactly what the reference's eval scripts do), predict,
                          and stores <out>/keras_outputs.npz

Copy keras_outputs.npz to tests/golden/ and tests/test_gpu_models.py::test_hip_matches_real_keras_outputs compares the
HIP engine with it (1e-3 px); until then that test is skipped and the parity of those numerics stays "restated".

    python tools/make_keras_parity_kit.py /tmp/kit            (CPU only; ~150 MB)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

RUNNER = r'''#!/usr/bin/env python
"""Run with keras==2.1.4 / tensorflow==1.6 and the deephar checkout on PYTHONPATH (see tools/make_keras_parity_kit.py)."""
import os, sys
import numpy as np
from deephar.config import ModelConfig
from deephar.models import reception, action, spnet
from deephar.utils import pa16j2d, pa17j3d
HERE = os.path.dirname(os.path.abspath(__file__))

def build(tag):
    if tag == 'rec2d':
        return reception.build((256, 256, 3), 16, dim=2, num_context_per_joint=2, num_blocks=2, ksize=(5, 5),
                               concat_pose_confidence=False), False
    if tag == 'rec3d':
        return reception.build((256, 256, 3), 17, dim=3, num_blocks=2, depth_maps=16, ksize=(5, 5),
                               export_heatmaps=True), False
    if tag in ('merge2d', 'merge3d'):
        dim, J, ver = (2, 16, 'v1') if tag == 'merge2d' else (3, 20, 'v2')
        kw = dict(num_context_per_joint=2, num_blocks=2, ksize=(5, 5)) if dim == 2 else \
            dict(num_blocks=2, depth_maps=8, ksize=(5, 5))
        pe = reception.build((128, 128, 3), J, dim=dim, **kw)
        return action.build_merge_model(pe, 15, (128, 128, 3), 4, J, 2, pose_dim=dim, depth_maps=8,
                                        pose_net_version=ver, output_poses=True), False
    spnet.__dict__.pop('act_cnt', None)      # the reference numbers its action blocks with a process-global counter
    T, lay, nact, apyr, feats = {'spnet3d': (4, pa17j3d, 60, [1, 2], 192), 'spnet2d': (16, pa16j2d, 15, [2], 160)}[tag]
    cfg = ModelConfig((T, 128, 128, 3), lay, num_actions=[nact], num_pyramids=2, action_pyramids=apyr, num_levels=4,
                      pose_replica=False, num_pose_features=feats, num_visual_features=feats)
    return spnet.build(cfg), True

out = {}
for tag in %(tags)r:
    model, by_name = build(tag)
    model.load_weights(os.path.join(HERE, tag + '.h5'), by_name=by_name)
    y = model.predict(np.load(os.path.join(HERE, tag + '_x.npy')).astype('float32'))
    y = y if isinstance(y, list) else [y]
    for i, a in enumerate(y):
        out['%%s/%%d' %% (tag, i)] = np.asarray(a, dtype='float32')
    print(tag, [a.shape for a in y])
np.savez_compressed(os.path.join(HERE, 'keras_outputs.npz'), **out)
'''


def main(out, tags=None):
    from refgolden import CASES, build_case
    os.makedirs(out, exist_ok=True)
    tags = list(tags or CASES)
    for tag in tags:
        m, x, _ = build_case(tag)
        m.save_weights(os.path.join(out, tag + '.h5'))
        np.save(os.path.join(out, tag + '_x.npy'), x.astype(np.float32))
        print(tag, 'weights %.1f MB' % (os.path.getsize(os.path.join(out, tag + '.h5')) / 1e6), 'input', x.shape)
    with open(os.path.join(out, 'run_in_keras.py'), 'w') as f:
        f.write(RUNNER % dict(tags=tags))
    print('wrote', out)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'keras_parity_kit', sys.argv[2:] or None)
