"""TEST INFRASTRUCTURE -- run the reference's own Python, unmodified, on top of import shims.

Only usable where /root/reference exists (the build container).  Used by
tests/test_oracle.py and tests/golden/make_golden.py to pin the restatements in
oracle/ against the literal reference code.  Never imported by the product.

What is patched (never by editing the reference):
  * sys.path gets oracle/shims first: torchvision (ResNet restated), skimage (on scipy.ndimage),
    pydensecrf (on oracle/crf_ref.py), pycocotools.mask (on oracle/annot_ref.py), attrdict, imgaug (identity augmenters)
  * permissive stubs for neptune, cv2, pycocotools.coco/cocoeval, lightgbm, xgboost, imageio, pydot_ng,
    IPython
  * sklearn.externals.joblib -> joblib ; collections.Iterable -> collections.abc.Iterable
  * yaml.load default Loader (PyYAML 6) ; env NEPTUNE_API_TOKEN / CONFIG_PATH
  * sys.dont_write_bytecode so nothing is written into /root/reference
"""
import collections
import collections.abc
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get('MSC_REFERENCE_ROOT', '/root/reference')
_HERE = os.path.dirname(os.path.abspath(__file__))
_installed = False


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'src', 'unet_models.py'))


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError('reference tree not found at %s' % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    repo_root = os.path.dirname(_HERE)
    if repo_root not in sys.path:
        sys.path.insert(0, repo_root)
    shims = os.path.join(_HERE, 'shims')
    if shims not in sys.path:
        sys.path.insert(0, shims)
    import pycocotools.mask            # the shim package; its coco / cocoeval submodules become stubs below
    import _anystub
    _anystub.install('neptune', 'cv2',
                     'pycocotools.coco', 'pycocotools.cocoeval', 'lightgbm', 'xgboost', 'imageio',
                     'pydot_ng', 'IPython', 'IPython.display')
    import joblib
    import sklearn.externals as ext
    ext.joblib = joblib
    sys.modules['sklearn.externals.joblib'] = joblib
    if not hasattr(collections, 'Iterable'):
        collections.Iterable = collections.abc.Iterable
    import yaml
    if not getattr(yaml.load, '_msc_patched', False):
        _orig = yaml.load

        def _load(stream, Loader=yaml.SafeLoader, **kw):
            return _orig(stream, Loader=Loader, **kw)
        _load._msc_patched = True
        yaml.load = _load
    os.environ.setdefault('NEPTUNE_API_TOKEN', 'offline')
    os.environ.setdefault('CONFIG_PATH', os.path.join(REFERENCE_ROOT, 'neptune.yaml'))
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    _installed = True


def ref(module):
    """Import `src.<module>` from the reference tree, e.g. ref('unet_models')."""
    install()
    return importlib.import_module('src.' + module)
