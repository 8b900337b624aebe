"""Run the reference's own callers unmodified: `install()` registers this package under the module names they
import (`deephar`, `deephar.models`, `deephar.config`, `deephar.utils`, `deephar.measures`, ...) together with the
three Keras names the evaluation scripts and exp/common/*_tools.py touch (`keras.models.Model`,
`keras.layers.concatenate` / `Input`, `keras.callbacks.Callback`, `keras.backend.epsilon`).

    import deephar_amd.compat as compat; compat.install()
    sys.path.append('exp/common'); from mpii_tools import eval_singleperson_pckh      # the reference's file, as is

What stays out of scope keeps failing loudly: `deephar.data` exposes the dataset class names but instantiating
one raises (datasets / augmentation: SURVEY.md section 2), `deephar.trainer` / `losses` are not provided, and
`keras.utils.data_utils.get_file` only resolves files that already exist locally (no network here).
An installed real `keras` / `deephar` is never shadowed unless force=True.
"""
import os
import sys
import types

_REGISTERED = []
_DATASETS = ('MpiiSinglePerson', 'Human36M', 'PennAction', 'Ntu', 'BatchLoader')


def _out_of_scope(name):
    class _Missing(object):
        def __init__(self, *a, **k):
            raise NotImplementedError('deephar.data.%s: dataset loading / augmentation is outside the MI355X '
                                      'hot-path scope; feed arrays to Model.predict instead' % name)
    _Missing.__name__ = name
    return _Missing


def _get_file(fname, origin=None, md5_hash=None, cache_subdir='datasets', **kw):
    """keras.utils.data_utils.get_file without the download: look where Keras would have cached the file."""
    for root in (os.getcwd(), os.path.join(os.path.expanduser('~'), '.keras', cache_subdir)):
        path = os.path.join(root, fname)
        if os.path.exists(path):
            return path
    raise IOError('%s not found locally and there is no network to fetch %s' % (fname, origin))


def install(force=False):
    """Idempotent.  Returns the list of module names that were registered."""
    import deephar_amd
    from deephar_amd import config, layers, measures, models, utils
    from deephar_amd.models import action, blocks, common, reception, spnet
    from deephar_amd.utils import bbox, camera, io, pose, transform

    done = []

    def put(name, mod):
        sys.modules[name] = mod
        _REGISTERED.append(name)
        done.append(name)

    def new(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        return m

    if force or 'deephar' not in sys.modules:
        data = new('deephar.data', **{n: _out_of_scope(n) for n in _DATASETS})
        pkg = new('deephar', models=models, config=config, utils=utils, measures=measures, layers=layers, data=data,
                  __path__=[])
        put('deephar', pkg)
        for name, mod in (('deephar.models', models), ('deephar.models.reception', reception),
                          ('deephar.models.action', action), ('deephar.models.spnet', spnet),
                          ('deephar.models.blocks', blocks), ('deephar.models.common', common),
                          ('deephar.config', config), ('deephar.measures', measures), ('deephar.layers', layers),
                          ('deephar.utils', utils), ('deephar.utils.bbox', bbox), ('deephar.utils.camera', camera),
                          ('deephar.utils.io', io), ('deephar.utils.pose', pose),
                          ('deephar.utils.transform', transform), ('deephar.data', data)):
            put(name, mod)
    if force or 'keras' not in sys.modules:
        kmodels = new('keras.models', Model=deephar_amd.Model)
        klayers = new('keras.layers', concatenate=deephar_amd.concatenate, Input=layers.Input)
        kcb = new('keras.callbacks', Callback=type('Callback', (object,), {}))
        kback = new('keras.backend', epsilon=lambda: 1e-7, image_data_format=lambda: 'channels_last',
                    set_image_data_format=lambda fmt: None)
        kdu = new('keras.utils.data_utils', get_file=_get_file)
        kutils = new('keras.utils', data_utils=kdu, __path__=[])
        keras = new('keras', models=kmodels, layers=klayers, callbacks=kcb, backend=kback, utils=kutils,
                    __version__='2.1.4-deephar_amd', __path__=[])
        for name, mod in (('keras', keras), ('keras.models', kmodels), ('keras.layers', klayers),
                          ('keras.callbacks', kcb), ('keras.backend', kback), ('keras.utils', kutils),
                          ('keras.utils.data_utils', kdu)):
            put(name, mod)
    return done


def uninstall():
    while _REGISTERED:
        sys.modules.pop(_REGISTERED.pop(), None)
