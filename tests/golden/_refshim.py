"""
Import shim that lets the *unmodified* reference (mittagessen/kraken, mounted
read-only at /root/reference) run on CPU in the authoring container, where a
number of its optional dependencies are absent.  It is used ONLY by
``make_golden.py`` (fixture generation) -- nothing in the product, the GPU
tests, ``smoke()`` or ``bench.py`` imports it, and it is a no-op on a box
without /root/reference.

The stubs stand in for packages that the hot path never touches for
arithmetic (serialisation, geometry, XML, training); see SURVEY.md App. C.
"""
import importlib
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get('KRAKEN_REFERENCE', '/root/reference')

_STUBS = [
    'coremltools', 'coremltools.proto', 'coremltools.proto.NeuralNetwork_pb2',
    'coremltools.models', 'coremltools.models.neural_network',
    'lightning', 'lightning.fabric', 'lightning.pytorch', 'lightning.pytorch.callbacks',
    'shapely', 'shapely.geometry', 'shapely.ops', 'shapely.validation',
    'skimage', 'skimage.draw', 'skimage.filters', 'skimage.graph', 'skimage.measure',
    'skimage.morphology', 'skimage.transform', 'skimage.filters.thresholding',
    'torchvision', 'torchvision.transforms', 'torchvision.transforms.v2',
    'torchvision.transforms.v2.functional', 'torchvision.transforms.functional',
    'lxml', 'lxml.etree', 'iso639', 'iso639.exceptions', 'jsonschema',
    'torchmetrics', 'torchmetrics.text', 'torchmetrics.classification',
    'torchmetrics.aggregation', 'htrmopo', 'google.protobuf.message',
]


class _Stub(types.ModuleType):
    """Module whose every missing attribute is an inert MagicMock."""

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        m = mock.MagicMock(name=f'{self.__name__}.{name}')
        setattr(self, name, m)
        return m


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'kraken'))


def install():
    """Registers the stubs and puts the reference on sys.path (idempotent)."""
    if not available():
        raise RuntimeError(f'reference not found under {REFERENCE_ROOT}')
    for name in _STUBS:
        try:
            importlib.import_module(name)
            continue
        except Exception:
            pass
        mod = _Stub(name)
        mod.__path__ = []  # behave like a package
        sys.modules[name] = mod
        if '.' in name:
            parent, child = name.rsplit('.', 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], child, mod)
    # names the reference subclasses at import time must be real classes
    st = sys.modules['skimage.transform']
    if isinstance(st, _Stub):
        st.PiecewiseAffineTransform = type('PiecewiseAffineTransform', (), {})
        st.AffineTransform = type('AffineTransform', (), {})
        sys.modules['skimage.graph'].MCP_Connect = type('MCP_Connect', (), {})
    tv = sys.modules['torchvision.transforms']
    if isinstance(tv, _Stub):
        tv.Compose = type('Compose', (), {'__init__': lambda self, t=None: setattr(self, 'transforms', t)})
    gm = sys.modules['google.protobuf.message']
    if isinstance(gm, _Stub):
        gm.DecodeError = type('DecodeError', (Exception,), {})
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return True
