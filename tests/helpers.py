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


def portable_weights(model, seed: int = 0) -> None:
    """
    Overwrites every parameter of a VGSL model (the reference's or ours: same state-dict names) with values that are the same bits on
    every machine: element-wise uniform draws, torch's default bounds (convolutions U(+-0.1) like kraken's init_weights, LSTMs
    U(+-1/sqrt(hidden)), linear layers Xavier-uniform, forget-gate biases 1).  kraken's own init is orthogonal for the LSTMs
    (model.py:465-475) -- a LAPACK QR whose bits depend on the CPU and thread count once a matrix is wide enough (the 800 x 960 input
    weights of the height-120 spec differed between the authoring container and the GPU box), so a fixture made with it does not travel.
    """
    import math
    sd = model.state_dict()
    for k, name in enumerate(sorted(sd)):
        p = sd[name]
        g = torch.Generator().manual_seed(seed * 1000 + k)
        if '.co.' in name:
            bound = 0.1
        elif 'weight_ih' in name or 'weight_hh' in name or 'bias_ih' in name or 'bias_hh' in name:
            bound = 1.0 / math.sqrt(p.shape[0] // 4)
        elif 'weight_ip' in name or 'weight_fp' in name or 'weight_op' in name:      # ocropy peephole weights (layers.py:60-70)
            bound = 0.5
        elif p.dim() == 2:
            bound = math.sqrt(6.0 / (p.shape[0] + p.shape[1]))
        else:
            bound = 0.0
        v = (torch.rand(p.shape, generator=g) * 2 - 1) * bound
        if 'bias_ih' in name or 'bias_hh' in name:
            h = p.shape[0] // 4
            v[h:2 * h] = 1.0
        with torch.no_grad():
            p.copy_(v.to(p.device))

