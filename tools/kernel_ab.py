"""Dev tool (GPU): per-launch-group times of BENCH-A ALONE (one batch in flight) under several probe-switch settings, interleaved
in ONE process (the switches of `struct Probes`, capi.hip, are read once per forward call), plus the difference of the logits
between the settings.  Usage:
    python tools/kernel_ab.py "KRK_CONV_X3P=0" "KRK_CONV_X3P=1" [--rounds 7] [--spec A|B] [--n 256] [--w 1200]
Each argument is a comma-separated list of NAME=VALUE; the first setting is the reference for the logit difference."""
import os
import sys

sys.path.insert(0, '.')
import torch  # noqa: E402

import kraken_amd  # noqa: E402
from kraken_amd.engine import RecognitionEngine  # noqa: E402
from kraken_amd.specs import BENCH_A, BENCH_B, bench_codec  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
opts = {a.split('=')[0][2:]: (a.split('=') + ['1'])[1] for a in sys.argv[1:] if a.startswith('--')}
rounds = int(opts.get('rounds', 7))
N, W = int(opts.get('n', 256)), int(opts.get('w', 1200))
spec = BENCH_B if opts.get('spec', 'A') == 'B' else BENCH_A
settings = [dict(kv.split('=') for kv in a.split(',') if kv) for a in args] or [{}]
names = sorted({k for s in settings for k in s})


def apply(s):
    for k in names:
        if k in s:
            os.environ[k] = s[k]
        else:
            os.environ.pop(k, None)


torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=spec, codec=bench_codec()).to('cuda')
m.nn.set_precision(opts.get('precision', 'bf16x3'))
x = torch.rand(N, 1, 48, W, generator=torch.Generator().manual_seed(1)).cuda()
lens = None
if 'ragged' in opts:
    lens = torch.randint(W // 3, W + 1, (N,), generator=torch.Generator().manual_seed(2))
    lens[0] = W
eng = RecognitionEngine(m, device=0, max_batch=N, max_width=W, slots=1)
eng.set_profiling(True)
best = [None] * len(settings)
for r in range(rounds + 1):
    for i, s in enumerate(settings):
        apply(s)
        eng.submit(x, None if lens is None else lens.numpy().astype('int32'))
        eng.collect()
        if r == 0:
            continue                          # warm-up round
        t = [(n_, ms) for n_, ms, _ in eng.layer_times()[0]]
        best[i] = t if best[i] is None else [(a[0], min(a[1], b[1])) for a, b in zip(best[i], t)]
ref = None
for i, s in enumerate(settings):
    apply(s)
    y, _ = m.nn(x, lens)
    y = y.float().cpu()
    if ref is None:
        ref = y
    d = (y - ref).abs().max().item()
    tot = sum(ms for _, ms in best[i])
    print(' '.join(f'{k}={v}' for k, v in s.items()) or '(default)', '| total %.3f ms | max|dlogit| vs first %.2e |' % (tot, d),
          ' '.join('%s=%.3f' % (n_, ms) for n_, ms in best[i]), flush=True)
