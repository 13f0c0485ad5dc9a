"""
Executes INTEGRATION.md section 3's in-tree binding (cut out of the markdown) over the REFERENCE's own modules and compares the
krk_layer table it builds with the one kraken_amd.vgsl.layer_table builds for the same network.  Run by
tests/test_host_cpu.py::test_integration_stub_builds_the_same_layer_table in a process of its own (it imports the reference).
"""
import ctypes as C
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.golden import _refshim  # noqa: E402

_refshim.install()
from kraken.lib import vgsl as ref_vgsl  # noqa: E402  (the reference)

import kraken_amd  # noqa: E402
from kraken_amd import vgsl as kv  # noqa: E402
from tests.helpers import layer_cases  # noqa: E402

FIELDS = ('op', 'cout', 'kh', 'kw', 'sh', 'sw', 'dh', 'dw', 'act', 'direction')


def stub_namespace():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text[text.index('## 3. In-tree binding'):]
    block = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    assert 'def layer_table(net)' in block and 'def build_plan(net' in block and 'def rec_predict(' in block
    block = block.replace("C.CDLL('libkraken_amd.so')", f"C.CDLL({os.path.join(ROOT, 'kraken_amd', 'libkraken_amd.so')!r})")
    ns: dict = {}
    exec(compile(block, 'INTEGRATION.md#3', 'exec'), ns)
    return ns


def arrays_of(d, keep):
    """The weight arrays an entry points at, identified through the kept host arrays."""
    by_addr = {a.ctypes.data: a for ws in keep for a in ws}
    return [by_addr[p] for p in d.w if p]


def main():
    ns = stub_namespace()
    cases = {}
    for f in ('layers.npz', 'groups.npz', 'breadth.npz', 'forms_r5.npz'):
        cases.update(layer_cases(f))
    checked = 0
    for name, c in sorted(cases.items()):
        if c['spec'].startswith('[1,0,'):
            continue                                   # variable height: kraken_amd routes the height collapse differently
        net = ref_vgsl.TorchVGSLModel(vgsl=c['spec'])
        net.load_state_dict({k: torch.as_tensor(v) for k, v in c['sd'].items()})
        net.eval()
        theirs, keep_t = ns['layer_table'](net)
        mine_model = kraken_amd.TorchVGSLModel(vgsl=c['spec'])
        mine_model.load_state_dict({k: torch.as_tensor(v) for k, v in c['sd'].items()})
        ours, keep_o = kv.layer_table(mine_model.nn._specs, mine_model.nn)
        assert len(theirs) == len(ours), (name, len(theirs), len(ours))
        for i, (a, b) in enumerate(zip(theirs, ours)):
            fa, fb = [getattr(a, f) for f in FIELDS], [getattr(b, f) for f in FIELDS]
            assert fa == fb, (name, i, dict(zip(FIELDS, fa)), dict(zip(FIELDS, fb)))
            wa, wb = arrays_of(a, keep_t), arrays_of(b, keep_o)
            assert len(wa) == len(wb), (name, i, len(wa), len(wb))
            for x, y in zip(wa, wb):
                assert x.shape == y.shape and x.dtype == y.dtype == np.float32 and np.array_equal(x, y), (name, i, x.shape, y.shape)
        checked += 1
    # the struct the stub declares is the library's (size and field offsets of kraken_amd._lib.KrkLayer)
    from kraken_amd import _lib
    assert C.sizeof(ns['_Layer']) == C.sizeof(_lib.KrkLayer)
    assert [(n, getattr(ns['_Layer'], n).offset) for n, _ in ns['_Layer']._fields_] == \
        [(n, getattr(_lib.KrkLayer, n).offset) for n, _ in _lib.KrkLayer._fields_]
    print(f'integration stub ok: {checked} networks')


if __name__ == '__main__':
    main()
