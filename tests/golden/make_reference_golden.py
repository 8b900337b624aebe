#!/usr/bin/env python
"""Golden vectors from the reference's OWN model code (this container only; /root/reference is not on the GPU box).

The reference's builders (deephar/layers.py, activations.py, models/*.py) are imported unmodified and executed on
oracle/refrun/minikeras.py, a PyTorch-CPU stand-in for the few Keras/TF entry points they use.  Weights are the
product's deterministic synthetic weights, handed to the reference model layer by layer (by name where the
reference names its layers, by creation order inside each nested Model otherwise -- the same rule Keras'
load_weights uses), so a wiring or ordering difference between the product builders and the reference shows up
either as a shape mismatch here or as a numeric mismatch in tests/test_reference_golden.py.

    python tests/golden/make_reference_golden.py            ->  tests/golden/reference_models.npz
    python tests/golden/make_reference_golden.py --smooth   ->  tests/golden/reference_models_smooth.npz
        (SPNet on the well-conditioned vectors of tests/wellcond.py: video clips + heat-map heads fitted on the
         oracle; the fitted head kernels are stored beside the outputs, '<tag>/head/<layer name>')
    python tests/golden/make_reference_golden.py --real     ->  tests/golden/reference_models_real.npz
        (the BASELINE configurations at their REAL size: ReceptionNet 8 blocks 2-D / 3-D at 256 px, the merge model of
         eval_penn_ar_pe_merge.py at T = 16 / 4 blocks / 256 px, SPNet-NTU at T = 32 / 256 px with fitted heads, and
         [r06] the model of the reference's own speed protocol, eval_speed2d.py:31-43: SPNet-Penn, 6 pyramids, actions
         on all six, pose_replica, T = 8 at 256 px, fitted heads)
    ... --real --only=spnet2d_speed_s,spnet3d_32_s           (recompute just these cases of the file)
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_models.npz')

from oracle.refrun import minikeras as mk   # noqa: E402


def load_reference():
    mk.install()
    pkg = types.ModuleType('deephar')
    pkg.__path__ = [REF + '/deephar']        # do not execute deephar/__init__.py (it imports the data loaders)
    sys.modules['deephar'] = pkg
    mods = {}
    for name in ('layers', 'activations', 'config', 'models.blocks', 'models.reception', 'models.action',
                 'models.common', 'models.spnet'):
        mods[name] = importlib.import_module('deephar.' + name)
    return mods


def _scopes(model):
    """{scope name: [weight layers in creation order]} for the nested Models of a mini-keras model; '' = top level."""
    reg = {id(l): i for i, l in enumerate(mk.weight_layers())}
    out = {'': []}

    def unwrap(l):
        return l.layer if isinstance(l, mk.TimeDistributed) else l

    def walk(m, scope):
        for l in m.layers:
            l = unwrap(l)
            if isinstance(l, mk.Model):
                walk(l, l.name)
            elif l.weights:
                out.setdefault(scope, [])
                if l not in out[scope]:
                    out[scope].append(l)
    walk(model, '')
    for k in out:
        out[k].sort(key=lambda l: reg[id(l)])
    return out


def transfer_weights(product_model, ref_model):
    """Hand the product's weights to the reference model; returns the number of tensors set."""
    by_scope = {}
    for n in product_model._nodes:
        for layer in n.layers.values():
            lst = by_scope.setdefault(layer.scope, [])
            if layer not in lst:
                lst.append(layer)
    ref_scopes = _scopes(ref_model)
    count = 0
    for scope, players in by_scope.items():
        players.sort(key=lambda l: l.uid)
        rlayers = ref_scopes.get(scope)
        assert rlayers is not None, 'reference model has no nested model %r' % scope
        rnames = {l.name: l for l in rlayers}
        if all(pl.name in rnames for pl in players):
            pairs = [(pl, rnames[pl.name]) for pl in players]            # explicit names (SPNet, sepconv_l*)
        else:
            # auto-named layers: creation order inside the nested model, frozen helper layers have no weights in
            # the product and never appear in a scope that has product layers
            assert len(players) == len(rlayers), (scope, len(players), len(rlayers),
                                                  [l.name for l in players], [l.name for l in rlayers])
            pairs = list(zip(players, rlayers))
        for pl, rl in pairs:
            rl.set_weights([p.value for p in pl.params])     # shape-checked inside
            count += len(pl.params)
    return count


def run_both(ref_model, x):
    outs = {}
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        mk.set_dtype(dt)
        for l in mk.weight_layers():          # re-cast the weights
            l.weights = [w.to(dt) for w in l.weights]
        y = ref_model.predict(x.astype(np.float64 if dt == torch.float64 else np.float32))
        outs[tag] = y if isinstance(y, list) else [y]
    mk.set_dtype(torch.float32)
    return outs


def draw(tag, shape):
    """inputs are not stored: tests regenerate them from the tag (tests/refgolden.py: case_input)"""
    seed = int.from_bytes(tag.encode(), 'little') % (2 ** 31)
    return np.random.default_rng(seed).uniform(-1, 1, shape)


TAGS = ('rec2d', 'rec3d', 'merge2d', 'merge3d', 'spnet3d', 'spnet2d', 'spnet2dr')
SMOOTH_TAGS = ('spnet3d_s', 'spnet2d_s', 'spnet2dr_s')
OUT_SMOOTH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_models_smooth.npz')
REAL_TAGS = ('rec2d_8', 'rec3d_8', 'merge2d_16', 'spnet3d_32_s', 'spnet2d_speed_s')
OUT_REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_models_real.npz')


def build_pair(R, tag):
    """(reference model on mini-keras, product model, input) for one golden case."""
    from deephar_amd import graph
    from deephar_amd import config as pconfig
    from deephar_amd import utils as putils
    from deephar_amd.models import reception as prec, action as pact, spnet as pspn
    mk.reset()
    graph.reset_naming()
    if tag == 'rec2d':      # ReceptionNet 2-D with context (cfg 1/2 family), 2 blocks
        kw = dict(num_context_per_joint=2, num_blocks=2, ksize=(5, 5), concat_pose_confidence=False)
        ref = R['models.reception'].build((256, 256, 3), 16, dim=2, **kw)
        prod = prec.build((256, 256, 3), 16, dim=2, **kw)
        return ref, prod, draw(tag, (2, 256, 256, 3))
    if tag == 'rec3d':      # ReceptionNet 3-D (cfg 3 family), 2 blocks, with exported heat-maps
        kw = dict(num_blocks=2, depth_maps=16, ksize=(5, 5), export_heatmaps=True)
        ref = R['models.reception'].build((256, 256, 3), 17, dim=3, **kw)
        prod = prec.build((256, 256, 3), 17, dim=3, **kw)
        return ref, prod, draw(tag, (2, 256, 256, 3))
    if tag == 'rec2d_8':    # configs[1] at its real depth (reception.py:277-312 loop indices 3 .. 8)
        kw = dict(num_context_per_joint=2, num_blocks=8, ksize=(5, 5), concat_pose_confidence=False)
        ref = R['models.reception'].build((256, 256, 3), 16, dim=2, **kw)
        prod = prec.build((256, 256, 3), 16, dim=2, **kw)
        return ref, prod, draw(tag, (2, 256, 256, 3))
    if tag == 'rec3d_8':    # configs[2] at its real depth
        kw = dict(num_blocks=8, depth_maps=16, ksize=(5, 5))
        ref = R['models.reception'].build((256, 256, 3), 17, dim=3, **kw)
        prod = prec.build((256, 256, 3), 17, dim=3, **kw)
        return ref, prod, draw(tag, (2, 256, 256, 3))
    if tag == 'merge2d_16':  # configs[3] exactly as exp/pennaction/eval_penn_ar_pe_merge.py:51-57 builds it (action.py:127-153 at 4 blocks)
        pe_kw = dict(num_blocks=4, num_context_per_joint=2, ksize=(5, 5), concat_pose_confidence=False)
        ref_pe = R['models.reception'].build((256, 256, 3), 16, dim=2, **pe_kw)
        ref = R['models.action'].build_merge_model(ref_pe, 15, (256, 256, 3), 16, 16, 4, pose_dim=2,
                                                   pose_net_version='v1', full_trainable=False)
        prod_pe = prec.build((256, 256, 3), 16, dim=2, **pe_kw)
        prod = pact.build_merge_model(prod_pe, 15, (256, 256, 3), 16, 16, 4, pose_dim=2, pose_net_version='v1',
                                      full_trainable=False)
        return ref, prod, draw(tag, (1, 16, 256, 256, 3))
    if tag in ('merge2d', 'merge3d'):   # merge action model, 2-D v1 (cfg 4 family) and 3-D v2
        dim, J, ver = (2, 16, 'v1') if tag == 'merge2d' else (3, 20, 'v2')
        T = 4
        if dim == 2:
            pe_kw = dict(num_context_per_joint=2, num_blocks=2, ksize=(5, 5))
        else:
            pe_kw = dict(num_blocks=2, depth_maps=8, ksize=(5, 5))
        ref_pe = R['models.reception'].build((128, 128, 3), J, dim=dim, **pe_kw)
        ref = R['models.action'].build_merge_model(ref_pe, 15, (128, 128, 3), T, J, 2, pose_dim=dim, depth_maps=8,
                                                   pose_net_version=ver, output_poses=True)
        prod_pe = prec.build((128, 128, 3), J, dim=dim, **pe_kw)
        prod = pact.build_merge_model(prod_pe, 15, (128, 128, 3), T, J, 2, pose_dim=dim, depth_maps=8,
                                      pose_net_version=ver, output_poses=True)
        return ref, prod, draw(tag, (2, T, 128, 128, 3))
    # SPNet: NTU-like 3-D (T=4, time_stride 1), Penn-like 2-D (T=16, time_stride 2, frame/joint padding) and the
    # shipped PennAction multitask configuration (exp/pennaction/eval_penn_multitask.py:36-40: T=8, 6 pyramids,
    # actions on pyramids 5 and 6, pose_replica=True -> '<pb>_heatmaps_conv1_replica' feeds the action stream)
    smooth = tag.endswith('_s')
    T, lay, nact, pyr, apyr, feats, res, rep = {'spnet3d': (4, 'pa17j3d', 60, 2, [1, 2], 192, 128, False),
                                                'spnet2d': (16, 'pa16j2d', 15, 2, [2], 160, 128, False),
                                                'spnet2dr': (8, 'pa16j2d', 15, 6, [5, 6], 160, 128, True),
                                                # configs[4]: T = 32 -> time_stride 2 (spnet.py:100), 256 px
                                                'spnet3d_32': (32, 'pa17j3d', 60, 2, [1, 2], 192, 256, False),
                                                # exp/pennaction/eval_speed2d.py:31-43: the model the reference times
                                                'spnet2d_speed': (8, 'pa16j2d', 15, 6, [1, 2, 3, 4, 5, 6], 160, 256, True)}[
        tag[:-2] if smooth else tag]
    R['models.spnet'].__dict__.pop('act_cnt', None)       # the reference's process-global counter
    rcfg = R['config'].ModelConfig((T, res, res, 3), getattr(putils, lay), num_actions=[nact], num_pyramids=pyr,
                                   action_pyramids=apyr, num_levels=4, pose_replica=rep,
                                   num_pose_features=feats, num_visual_features=feats)
    ref = R['models.spnet'].build(rcfg)
    pcfg = pconfig.ModelConfig((T, res, res, 3), getattr(putils, lay), num_actions=[nact], num_pyramids=pyr,
                               action_pyramids=apyr, num_levels=4, pose_replica=rep, num_pose_features=feats,
                               num_visual_features=feats)
    prod = pspn.build(pcfg)
    if smooth:
        import refgolden
        return ref, prod, refgolden.smooth_input(tag, res)[0].astype(np.float64)
    return ref, prod, draw(tag, (1, T, res, res, 3))


def keras_file_layout(ref_model):
    """What Keras' save_weights would write for the reference model (mini-keras' statement of Container.layers /
    layer.weights order): [[group, [[dataset name, shape, crc32 of the float32 bytes], ...]], ...], weight-owning
    groups only.  tests/test_keras_weights.py holds the product's own derivation (deephar_amd/keras_compat.py)
    against this, frozen helper kernels included."""
    import zlib
    out = []
    for name, ws in mk.save_layout(ref_model):
        if ws:
            out.append([name, [[wn, list(a.shape), zlib.crc32(np.ascontiguousarray(a, dtype=np.float32).tobytes())]
                               for wn, a in ws]])
    return out


def check_weight_files(tag, ref_model, product_model):
    """Both directions through real files: reference layout -> .h5 -> product.load_weights, and
    product.save_weights(.h5) -> Keras-order assignment into the reference model."""
    import tempfile
    from deephar_amd import hdf5
    by_name = tag.startswith('spnet')          # the reference loads SPNet files with by_name=True
    want = {p.key: p.value.copy() for p in product_model.params}
    with tempfile.TemporaryDirectory() as d:
        rl = mk.save_layout(ref_model)
        tree = {hdf5.ATTRS: {'layer_names': [n.encode() for n, _ in rl], 'backend': b'tensorflow',
                             'keras_version': b'2.1.4'}}
        for n, ws in rl:
            sub = tree.setdefault(n, {})
            for wn, a in ws:
                hdf5.put_path(sub, wn, a.astype(np.float32))
            sub[hdf5.ATTRS] = {'weight_names': [wn.encode() for wn, _ in ws]}
        hdf5.write_file(d + '/ref.h5', tree)
        for p in product_model.params:
            p.value = None
        product_model.load_weights(d + '/ref.h5', by_name=by_name)
        bad = [p.key for p in product_model.params if p.value is None or not np.array_equal(p.value, want[p.key])]
        assert not bad, (tag, 'reference file -> product', bad[:3])
        if not by_name:
            product_model.save_weights(d + '/prod.h5')
            f = hdf5.File(d + '/prod.h5')
            names = [n.decode() for n in f.attrs['layer_names']]
            ref_groups = [(n, ws) for n, ws in rl if ws]
            assert len(names) == len(ref_groups), (tag, len(names), len(ref_groups))
            for name, (rn, ws) in zip(names, ref_groups):      # Keras pairs groups and layers by ORDER
                wn = [w.decode() for w in f[name].attrs['weight_names']]
                assert len(wn) == len(ws), (tag, name, rn)
                for w, (_, a) in zip(wn, ws):
                    assert np.array_equal(np.asarray(f[name][w]), a.astype(np.float32)), (tag, name, w)
    print(tag, 'weight files: reference->product %s, product->reference %s' %
          ('by name' if by_name else 'by order', 'n/a (by-name family)' if by_name else 'by order'))


def main(smooth=False, real=False, only=None):
    """only: comma list of tags -- just these cases are recomputed, the other arrays of the output file are kept as they
    are (a full run reproduces them bit for bit; this keeps an added or refitted case to minutes)."""
    import json
    import time
    from deephar_amd import weights
    R = load_reference()
    g = {}
    layouts = {}
    tags = REAL_TAGS if real else SMOOTH_TAGS if smooth else TAGS
    out = OUT_REAL if real else OUT_SMOOTH if smooth else OUT
    if only:
        assert smooth or real, '--only is for the outputs-only files (--smooth / --real)'
        assert all(t in tags for t in only), (only, tags)
        old = np.load(out)
        g = {k: old[k] for k in old.files if k.split('/')[0] not in only}
        tags = [t for t in tags if t in only]
    for tag in tags:
        t0 = time.time()
        ref_model, product_model, x = build_pair(R, tag)
        weights.init_synthetic(product_model, seed=0)
        if tag.endswith('_s'):
            import refgolden
            import wellcond
            heads = wellcond.fit_spnet_heads(product_model, refgolden.spnet_ocfg(tag), x.astype(np.float32),
                                             refgolden.smooth_input(tag)[1],
                                             per_joint=tag[:-2] in refgolden.FIT_PER_JOINT)
            for name, k in heads.items():
                g['%s/head/%s' % (tag, name)] = k
        n = transfer_weights(product_model, ref_model)
        assert n == len(product_model.params), (tag, n, len(product_model.params))
        layouts[tag] = keras_file_layout(ref_model)
        check_weight_files(tag, ref_model, product_model)
        outs = run_both(ref_model, x)
        for k, arrs in outs.items():
            for i, a in enumerate(arrs):
                g['%s/%s/%d' % (tag, k, i)] = a.astype(np.float64 if k == 'f64' else np.float32)
        g['%s/nout' % tag] = np.array(len(outs['f64']))
        print(tag, 'outputs', [a.shape for a in outs['f64']], 'weights set', n, '%.0f s' % (time.time() - t0), flush=True)
    if not smooth and not real:
        with open(os.path.join(os.path.dirname(OUT), 'keras_layouts.json'), 'w') as fh:
            json.dump(layouts, fh, separators=(',', ':'))
    np.savez_compressed(out, **g)
    print('wrote', out, '%.1f MB' % (os.path.getsize(out) / 1e6))


if __name__ == '__main__':
    only = next((a.split('=', 1)[1].split(',') for a in sys.argv if a.startswith('--only=')), None)
    main(smooth='--smooth' in sys.argv, real='--real' in sys.argv, only=only)
