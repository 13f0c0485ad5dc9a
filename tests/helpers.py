"""Shared helpers of the test-suite (fixture loading, seeded inputs)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth_input(n, w, seed=1234, h=48, c=1):
    """The seeded synthetic batch of SURVEY.md section 8d (same generator as tests/golden/make_golden.py)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, c, h, w, generator=g)


def arr_to_tuples(flat, counts):
    """inverse of make_golden.tuples_to_arr -> list of lists of (label, start, end, conf)"""
    out, k = [], 0
    for c in counts.tolist():
        out.append([(int(r[0]), int(r[1]), int(r[2]), float(r[3])) for r in flat[k:k + c]])
        k += c
    return out


def layer_cases(fname='layers.npz'):
    z = load_golden(fname)
    cases = json.loads(str(z['cases']))
    out = {}
    for name, meta in cases.items():
        sd = {k.split('/sd/')[1]: z[k] for k in z.files if k.startswith(f'{name}/sd/')}
        entry = dict(meta)
        entry['sd'] = sd
        entry['x'] = z[f'{name}/x']
        if f'{name}/vgsl' in z.files:
            entry['vgsl'] = str(z[f'{name}/vgsl'])      # the reference's named spec (user_metadata['vgsl'])
        if meta['lens'] is None:
            entry['y'] = z[f'{name}/y']
            if f'{name}/olens_probe' in z.files:      # seq_lens behind the network for meta['lens_probe'], from the reference's batched call
                entry['olens_probe'] = z[f'{name}/olens_probe']
        else:
            entry['ys'] = [z[f'{name}/y{i}'] for i in range(len(meta['lens']))]
            entry['olens'] = z[f'{name}/olens'] if f'{name}/olens' in z.files else None
        out[name] = entry
    return out


def build_model(spec, sd=None, codec=None, seed=None):
    import kraken_amd
    if seed is not None:
        torch.manual_seed(seed)
    m = kraken_amd.TorchVGSLModel(vgsl=spec, codec=codec)
    if sd is not None:
        missing, unexpected = m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
    return m


def wavy_line(rng, h, w):
    """Dark strokes on a light page around a wandering centre line (the generator of tests/golden/make_golden.py:transforms_fixture)."""
    arr = np.full((h, w), 255, np.uint8)
    yc = (h / 2 + 0.15 * h * np.sin(np.arange(w) / rng.uniform(20.0, 60.0))).astype(int)
    for x in range(0, w, 3):
        if rng.rand() < 0.6:
            lo = max(yc[x] - rng.randint(2, max(h // 3, 3)), 0)
            hi = min(yc[x] + rng.randint(2, max(h // 3, 3)), h)
            arr[lo:hi, x:x + 2] = rng.randint(0, 90)
    return arr
