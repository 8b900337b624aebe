"""Keras-HDF5 weight import/export: the pure-Python HDF5 reader/writer, Keras' layer /
weight ordering rebuilt from the graph IR, order- and name-based loading.  CPU only.

Anchors: files written by the real libhdf5 (tests/golden/make_hdf5_fixtures.py) for the reader; the reference's
own builders executed on mini-keras for the layout (tests/golden/keras_layouts.json, produced by
tests/golden/make_reference_golden.py, which also round-trips real files in both directions)."""
import json
import os
import shutil
import subprocess
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from refgolden import build_case          # noqa: E402
from deephar_amd import hdf5, keras_compat as KC, weights     # noqa: E402

GOLD = os.path.join(HERE, 'golden')


def test_reader_on_libhdf5_files():
    exp = np.load(os.path.join(GOLD, 'hdf5_expected.npz'))
    f = hdf5.File(os.path.join(GOLD, 'keras_tiny.h5'))
    names = [n.decode() for n in f.attrs['layer_names']]
    assert names[:2] == ['Stem', 'rBlock1'] and len(names) == 40
    assert f.attrs['backend'] == b'tensorflow' and f.attrs['keras_version'] == b'2.1.4'
    assert [v.decode() for v in f.attrs['vl']] == ['soft', 'argmax']
    seen = 0
    for n in names:
        for w in f[n].attrs['weight_names']:
            a = np.asarray(f[n][w.decode()])
            assert a.dtype == np.float32 and np.array_equal(a, exp['tiny:%s/%s' % (n, w.decode())])
            seen += 1
    assert seen == 90
    assert np.array_equal(np.asarray(f['chunked']), exp['tiny:chunked'])        # gzip + shuffle, edge chunks
    assert np.array_equal(np.asarray(f['ints']), exp['tiny:ints'])
    assert float(np.asarray(f['scalar'])) == 3.5
    assert 'Stem' in f and 'nope' not in f
    with pytest.raises(KeyError):
        f['Stem/conv2d_0/missing']
    g = hdf5.File(os.path.join(GOLD, 'keras_tiny_latest.h5'))                   # v2 object headers, link messages
    assert sorted(g['model_weights'].keys()) == ['x', 'y']
    for n in 'xy':
        assert np.array_equal(np.asarray(g['model_weights/%s/%s/w:0' % (n, n)]),
                              exp['latest:model_weights/%s/%s/w:0' % (n, n)])
    with pytest.raises(hdf5.HDF5Error):
        hdf5.File(os.path.join(GOLD, 'hdf5_expected.npz'))


def test_writer_round_trip_and_limits(tmp_path):
    rng = np.random.default_rng(0)
    tree = {hdf5.ATTRS: {'layer_names': [b'a', b'bb'], 'backend': b'tensorflow', 'n': np.int32(7)}}
    want = {}
    for i in range(300):                                    # > 8 members: one large symbol node
        a = rng.standard_normal((2, i % 5 + 1)).astype(np.float32)
        hdf5.put_path(tree, 'a/conv2d_%d/kernel:0' % i, a)
        want['a/conv2d_%d/kernel:0' % i] = a
    tree['bb'] = {hdf5.ATTRS: {'weight_names': []}, 'e': np.zeros((0, 3), np.float32), 'd': np.arange(4.0)}
    p = str(tmp_path / 'w.h5')
    hdf5.write_file(p, tree)
    assert hdf5.is_hdf5(p)
    f = hdf5.File(p)
    assert [n.decode() for n in f.attrs['layer_names']] == ['a', 'bb'] and int(f.attrs['n']) == 7
    for k, a in want.items():
        assert np.array_equal(np.asarray(f[k]), a)
    assert np.asarray(f['bb/e']).shape == (0, 3) and np.asarray(f['bb/d']).dtype == np.float64
    assert len(f['bb'].attrs['weight_names']) == 0
    with pytest.raises(hdf5.HDF5Error):                     # Keras 2.1.4's own 64 KB attribute limit
        hdf5.write_file(p, {hdf5.ATTRS: {'layer_names': [b'x' * 64] * 2000}})


@pytest.mark.skipif(not os.path.exists('/opt/conda/bin/python3.9'), reason='no interpreter with h5py')
def test_libhdf5_reads_our_files(tmp_path):
    m, _, _ = build_case('merge3d')
    p = str(tmp_path / 'm.h5')
    m.save_weights(p)
    code = ("import h5py,sys,zlib,numpy as np;f=h5py.File(sys.argv[1],'r');"
            "print(len(f.attrs['layer_names']), sum(len(f[n].attrs['weight_names']) for n in f.attrs['layer_names']),"
            "zlib.crc32(b''.join(np.ascontiguousarray(f[n][w][()]).tobytes() for n in f.attrs['layer_names'] "
            "for w in f[n].attrs['weight_names'])))")
    out = subprocess.run(['/opt/conda/bin/python3.9', '-c', code, p], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lay = KC.layout(m)
    crc = zlib.crc32(b''.join((w.value() if isinstance(w, KC.Frozen) else w.value).astype(np.float32).tobytes()
                              for _, ws in lay for w in ws))
    assert out.stdout.split() == [str(len(lay)), str(sum(len(ws) for _, ws in lay)), str(crc)]


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.float32).tobytes())


@pytest.mark.parametrize('tag', ['rec2d', 'rec3d', 'merge2d', 'merge3d'])
def test_layout_matches_keras_order_of_reference_models(tag):
    """Group order, weight order inside nested Models (all trainables, then all moving statistics), the frozen
    helper layers and their regenerated values -- against what Keras would write for the reference's model."""
    gold = json.load(open(os.path.join(GOLD, 'keras_layouts.json')))[tag]
    m, _, _ = build_case(tag)                      # same synthetic weights as the generator transferred
    lay = KC.layout(m)
    assert len(lay) == len(gold)
    nfrozen = 0
    for (name, ws), (gname, gws) in zip(lay, gold):
        assert len(ws) == len(gws), (name, gname)
        for w, (gw, gshape, gcrc) in zip(ws, gws):
            val = w.value() if isinstance(w, KC.Frozen) else w.value
            nfrozen += isinstance(w, KC.Frozen)
            assert list(val.shape) == gshape and _crc(val) == gcrc, (name, gname, gw)
    assert nfrozen >= 4


@pytest.mark.parametrize('tag', ['spnet3d', 'spnet2d'])
def test_by_name_layout_of_spnet(tag):
    """SPNet files are loaded with by_name=True (eval_ntu_multitask.py:66): every weight-owning reference layer
    must exist under the same name at the top level, with the same weights in the same order."""
    gold = dict((g, ws) for g, ws in json.load(open(os.path.join(GOLD, 'keras_layouts.json')))[tag])
    m, _, _ = build_case(tag)
    mine = {}
    for lay in KC.view(m).layers:
        ws = KC.layer_weights(lay)
        if ws:
            mine[lay.name] = ws
    assert set(mine) == set(gold)
    for name, ws in mine.items():
        assert [_crc(w.value() if isinstance(w, KC.Frozen) else w.value) for w in ws] == [c for _, _, c in gold[name]]


@pytest.mark.parametrize('tag,by_name', [('rec2d', False), ('merge2d', False), ('spnet2d', True), ('spnet2d', False)])
def test_save_load_round_trip(tag, by_name, tmp_path):
    m, _, _ = build_case(tag)
    want = {p.key: p.value.copy() for p in m.params}
    p = str(tmp_path / 'w.h5')
    m.save_weights(p)
    m2, _, _ = build_case(tag)
    for q in m2.params:
        q.value = None
    m2.load_weights(p, by_name=by_name)
    for q in m2.params:
        assert np.array_equal(q.value, want[q.key]), q.key


def test_order_loading_rejects_other_architectures(tmp_path):
    m, _, _ = build_case('rec2d')
    p = str(tmp_path / 'w.h5')
    m.save_weights(p)
    other, _, _ = build_case('rec3d')
    with pytest.raises(ValueError):
        other.load_weights(p)                       # group count / shapes differ
    before = [q.value.copy() for q in other.params]
    with pytest.raises(ValueError):
        other.load_weights(p, by_name=True)         # 'RegMap1' exists in both but with another head width
    assert all(np.array_equal(a, q.value) for a, q in zip(before, other.params))    # nothing half-loaded


def test_nested_model_lists_trainables_before_statistics():
    m, _, _ = build_case('rec2d')
    name, ws = KC.layout(m)[0]
    assert name == 'Stem'
    roles = [w.name for w in ws]
    first_stat = roles.index('moving_mean')
    assert all(r in ('moving_mean', 'moving_variance') for r in roles[first_stat:])
    assert all(r in ('kernel', 'beta', 'depthwise_kernel', 'pointwise_kernel') for r in roles[:first_stat])


def test_hdf5_round_trip_fuzz(tmp_path):
    """Random trees (nesting, many members, odd names, float32/float64/int32, empty and scalar datasets, string and
    numeric attributes) written by deephar_amd.hdf5 read back identically; with h5py around, libhdf5 agrees."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    names = st.text(alphabet='abcXYZ019_:.-', min_size=1, max_size=12)
    dtypes = st.sampled_from([np.float32, np.float64, np.int32])
    shapes = st.lists(st.integers(0, 5), min_size=0, max_size=4).map(tuple)

    @st.composite
    def arrays(draw):
        dt, shp = draw(dtypes), draw(shapes)
        n = int(np.prod(shp)) if shp else 1
        vals = draw(st.lists(st.integers(-1000, 1000), min_size=n, max_size=n))
        return np.array(vals, dtype=dt).reshape(shp)

    @st.composite
    def trees(draw, depth=0):
        t = {}
        for k in draw(st.lists(names, min_size=0, max_size=6, unique=True)):
            if depth < 2 and draw(st.booleans()):
                t[k] = draw(trees(depth + 1))
            else:
                t[k] = draw(arrays())
        if draw(st.booleans()):
            t[hdf5.ATTRS] = {'weight_names': [n.encode() for n in draw(st.lists(names, max_size=5))],
                             'scalar': np.float32(draw(st.integers(-5, 5))), 'tag': draw(names).encode()}
        return t

    def check(node, tree):
        keys = sorted(k for k in tree if k != hdf5.ATTRS)
        assert sorted(node.keys()) == keys
        if hdf5.ATTRS in tree:
            at = tree[hdf5.ATTRS]
            assert [w for w in np.atleast_1d(node.attrs['weight_names'])] == at['weight_names'] or \
                (len(at['weight_names']) == 0 and len(node.attrs['weight_names']) == 0)
            assert float(node.attrs['scalar']) == float(at['scalar']) and node.attrs['tag'] == at['tag']
        for k in keys:
            if isinstance(tree[k], dict):
                check(node[k], tree[k])
            else:
                got = np.asarray(node[k])
                assert got.dtype == tree[k].dtype and got.shape == tree[k].shape and np.array_equal(got, tree[k])

    counter = [0]

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(trees())
    def run(tree):
        counter[0] += 1
        p = str(tmp_path / ('f%d.h5' % counter[0]))
        hdf5.write_file(p, tree)
        check(hdf5.File(p), tree)
    run()
    assert counter[0] >= 30


@pytest.mark.skipif(not os.path.exists('/opt/conda/bin/python3.9'), reason='no interpreter with h5py')
def test_libhdf5_reads_odd_datasets_we_write(tmp_path):
    """Scalar, empty, int and float64 datasets, nested groups with odd names, string-list / scalar attributes."""
    tree = {hdf5.ATTRS: {'layer_names': [b'g:0', b'h.1'], 'n': np.int32(3), 'tag': b'x'},
            'g:0': {'s': np.float32(2.5), 'e': np.zeros((0, 4), np.float32), 'i': np.arange(6, dtype=np.int32).reshape(2, 3),
                    hdf5.ATTRS: {'weight_names': [b's', b'e', b'i']}},
            'h.1': {'deep': {'er': {'d': np.linspace(0, 1, 7)}}, hdf5.ATTRS: {'weight_names': []}}}
    p = str(tmp_path / 'odd.h5')
    hdf5.write_file(p, tree)
    code = ("import h5py,sys,numpy as np;f=h5py.File(sys.argv[1],'r');g=f['g:0'];"
            "print(float(g['s'][()]), g['e'].shape, g['i'][()].sum(), f['h.1/deep/er/d'][()].sum(), "
            "[x.decode() for x in f.attrs['layer_names']], int(f.attrs['n']), len(f['h.1'].attrs['weight_names']), "
            "[x.decode() for x in g.attrs['weight_names']])")
    out = subprocess.run(['/opt/conda/bin/python3.9', '-c', code, p], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "2.5 (0, 4) 15 3.5 ['g:0', 'h.1'] 3 0 ['s', 'e', 'i']", out.stdout
