"""Dev tool (GPU, torch ops only -- not the product path): what each of the split-bf16 plan's cross terms buys.

The bf16x3 plan computes every product as a_hi*b_hi + a_lo*b_hi + a_hi*b_lo (a = activation, b = weight; hi = bf16(x),
lo = bf16(x - hi), fp32 accumulate).  This script emulates exactly that arithmetic with fp32 torch operators on operands that are
exactly representable in bf16 (so every product is exact in fp32, as on the matrix cores) and lets ONE launch group at a time drop
`a_lo*b_hi`, `a_hi*b_lo` or both.  For each of the 11 launch groups x 3 choices it reports
  * max |d logit| against the fp32 forward of the same torch operators on the 256 lines of BASELINE config 2, and
  * how many of the 2304 golden lines of equal width (config 2: 256, config 3's shard: 2048; kraken's own strings,
    tests/golden/bench_lines*.npz) decode to a different string, and how many of those lie outside the tie window (1e-4),
next to the all-terms plan (the product's arithmetic) and the all-dropped plan (the opt-in 1-term `bf16` plan).
    python tools/term_budget.py [--lines 2304] > profiles/r06_term_budget.txt
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kraken_amd  # noqa: E402
from kraken_amd.specs import BENCH_A, bench_codec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--lines', type=int, default=2304)
args = ap.parse_args()
dev = 'cuda' if torch.cuda.is_available() else 'cpu'     # (cpu: a slow self-check of this script only)
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False

GROUPS = ['conv1', 'taps', 'x3p_a', 'x3p_b', 'xproj1', 'rec1', 'xproj2', 'rec2', 'xproj3', 'rec3', 'linear']
KEEP_ALL = frozenset()


def split(t):
    hi = t.bfloat16().float()
    return hi, (t - hi).bfloat16().float()


def prod(op, a, b, drop, exact):
    """op(a, b) in the plan's arithmetic; drop: subset of {'lo_hi', 'hi_lo'} (activation part _ weight part)."""
    if exact:
        return op(a, b)
    ah, al = split(a)
    bh, bl = split(b)
    y = op(ah, bh)
    if 'lo_hi' not in drop:
        y = y + op(al, bh)
    if 'hi_lo' not in drop:
        y = y + op(ah, bl)
    return y


class Net:
    def __init__(self):
        torch.manual_seed(0)
        m = kraken_amd.TorchVGSLModel(vgsl=BENCH_A, codec=bench_codec())
        self.codec = m.codec
        self.sd = {k: v.to(dev) for k, v in m.state_dict().items()}

    @torch.no_grad()
    def forward(self, x, drops, exact=False):
        sd = self.sd
        d = lambda g: drops.get(g, KEEP_ALL)      # noqa: E731

        def conv(x, name, g, pool):
            w, b = sd[f'nn.{name}.co.weight'], sd[f'nn.{name}.co.bias']
            pad = (w.shape[2] // 2, w.shape[3] // 2)
            y = prod(lambda a, ww: F.conv2d(a, ww, None, padding=pad), x, w, d(g), exact) + b.view(1, -1, 1, 1)
            y = F.relu(y)
            return F.max_pool2d(y, 2, 2) if pool else y
        x = conv(x, 'C_0', 'conv1', True)
        x = conv(x, 'C_3', 'taps', True)
        x = conv(x, 'C_6', 'x3p_a', True)
        x = conv(x, 'C_9', 'x3p_b', False)
        n, c, h, w = x.shape
        x = x.permute(0, 3, 2, 1).reshape(n, w, h * c)             # S1(1x0)1,3: feature = h * C + c, time-major rows per line
        for li, name in enumerate(('L_12', 'L_14', 'L_16')):
            outs = []
            for suffix in ('', '_reverse'):
                wih, whh = sd[f'nn.{name}.layer.weight_ih_l0{suffix}'], sd[f'nn.{name}.layer.weight_hh_l0{suffix}']
                bias = sd[f'nn.{name}.layer.bias_ih_l0{suffix}'] + sd[f'nn.{name}.layer.bias_hh_l0{suffix}']
                xp = prod(lambda a, ww: a @ ww.t(), x, wih, d(f'xproj{li + 1}'), exact) + bias
                T, H = x.shape[1], whh.shape[1]
                hcur = torch.zeros(n, H, device=dev)
                ccur = torch.zeros(n, H, device=dev)
                ys = [None] * T
                order = range(T - 1, -1, -1) if suffix else range(T)
                for t in order:
                    z = xp[:, t] + prod(lambda a, ww: a @ ww.t(), hcur, whh, d(f'rec{li + 1}'), exact)
                    i, f, g, o = z.chunk(4, dim=1)
                    ccur = torch.sigmoid(f) * ccur + torch.sigmoid(i) * torch.tanh(g)
                    hcur = torch.sigmoid(o) * torch.tanh(ccur)
                    ys[t] = hcur
                outs.append(torch.stack(ys, dim=1))
            x = torch.cat(outs, dim=2)
        w, b = sd['nn.O_18.lin.weight'], sd['nn.O_18.lin.bias']
        return prod(lambda a, ww: a @ ww.t(), x, w, d('linear'), exact) + b          # (n, T, classes)


def strings(net, logits):
    lab = logits.argmax(dim=2).cpu().numpy()
    out = []
    inv = {v[0]: k for k, v in bench_codec().items()}
    for row in lab:
        keep = row[np.concatenate(([True], row[1:] != row[:-1]))]
        out.append(''.join(inv[int(v)] for v in keep if v != 0))
    return out


def synth(n, w, seed):
    return torch.rand(n, 1, 48, w, generator=torch.Generator().manual_seed(seed))


def main():
    net = Net()
    z2 = np.load(os.path.join(ROOT, 'tests', 'golden', 'bench_lines.npz'), allow_pickle=False)
    z3 = np.load(os.path.join(ROOT, 'tests', 'golden', 'bench_lines_r6.npz'), allow_pickle=False)
    want = json.loads(str(z2['cfg2_strings'])) + json.loads(str(z3['cfg3_strings']))
    margin = np.concatenate([z2['cfg2_margin'], z3['cfg3_margin']])
    batches = [synth(256, 1200, 1234)]
    for lo in range(0, 2048, 256):
        batches.append(torch.cat([synth(16, 1200, 30000 + (lo + k) // 16) for k in range(0, 256, 16)]))
    nb = max(1, min(len(batches), args.lines // 256))
    batches, want, margin = batches[:nb], want[:256 * nb], margin[:256 * nb]
    exact0 = net.forward(batches[0].to(dev), {}, exact=True)

    def evaluate(drops):
        t0 = time.perf_counter()
        got, dl = [], None
        for k, xb in enumerate(batches):
            y = net.forward(xb.to(dev), drops)
            if k == 0:
                dl = float((y - exact0).abs().max())
            got += strings(net, y)
        diff = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
        return dl, len(diff), sum(1 for i in diff if margin[i] >= 1e-4), time.perf_counter() - t0

    got0 = strings(net, exact0)
    print(f'# fp32 torch forward against kraken\'s strings on config 2: {sum(a == b for a, b in zip(got0, want[:256]))} / 256 identical')
    print(f'# {len(want)} golden lines of width 1200 (config 2: 256, config 3: {len(want) - 256}); |d logit| on the 256 lines of config 2 against the fp32 forward')
    print(f'{"launch group":<10} {"dropped":<14} {"max|dlogit|":>12} {"lines differing":>16} {"outside 1e-4 window":>20}')
    dl, nd, no, dt = evaluate({})
    print(f'{"(none)":<10} {"-- bf16x3 --":<14} {dl:12.2e} {nd:16d} {no:20d}      [{dt:.1f} s]', flush=True)
    both = frozenset(('lo_hi', 'hi_lo'))
    dl, nd, no, dt = evaluate({g: both for g in GROUPS})
    print(f'{"(all)":<10} {"both = bf16":<14} {dl:12.2e} {nd:16d} {no:20d}', flush=True)
    for g in GROUPS:
        for name, drop in (('a_lo*b_hi', frozenset(('lo_hi',))), ('a_hi*b_lo', frozenset(('hi_lo',))), ('both', both)):
            dl, nd, no, dt = evaluate({g: drop})
            print(f'{g:<10} {name:<14} {dl:12.2e} {nd:16d} {no:20d}', flush=True)


if __name__ == '__main__':
    main()
