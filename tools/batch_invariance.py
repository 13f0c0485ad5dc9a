"""Dev tool (GPU): do a line's logits depend on the batch it travels in?  60 ragged 1-channel lines (the set of
test_one_channel_bbox_lines_...) through BENCH-A: alone, in input-order batches, in width-sorted batches; per plan and
recurrent kernel.   python tools/batch_invariance.py"""
import os
import sys

sys.path.insert(0, '.')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402
from kraken_amd.transforms import ImageInputTransforms  # noqa: E402
from tests.helpers import build_model  # noqa: E402
from tests.specs import BENCH_A, bench_codec  # noqa: E402
from tests.helpers import wavy_line as _wavy_line  # noqa: E402

rng = np.random.RandomState(9)
ts = ImageInputTransforms(1, 48, 0, 1, (16, 0), valid_norm=True)
lines = []
for i in range(60):
    h, w = int(rng.randint(30, 90)), int(rng.randint(200, 1000))
    if i == 17:
        continue
    lines.append(ts(Image.fromarray(_wavy_line(rng, h, w), 'L')))
widths = [t.shape[2] for t in lines]


def run(m, idx):
    W = max(widths[i] for i in idx)
    x = torch.zeros(len(idx), 1, 48, W)
    for k, i in enumerate(idx):
        x[k, :, :, :widths[i]] = lines[i]
    _, olens, logits, _ = m.nn.recognize(x.cuda(), torch.tensor([widths[i] for i in idx]), want_logits=True)
    lg = logits.cpu().numpy()
    return {i: lg[k, :, :olens[k]] for k, i in enumerate(idx)}


for prec in ('f32', 'bf16x3'):
    for v in ((0,) if prec == 'f32' else (1, 3, 4)):
        os.environ['KRK_LSTM_V'] = str(v)
        m = build_model(BENCH_A, codec=bench_codec(), seed=0).to('cuda')
        m.nn.set_precision(prec)
        alone = {}
        for i in range(len(lines)):
            alone.update(run(m, [i]))
        order = list(range(len(lines)))
        byw = sorted(order, key=lambda i: widths[i])
        for name, seq in (('input order', order), ('width-sorted', byw)):
            worst = 0.0
            for lo in range(0, len(seq), 32):
                got = run(m, seq[lo:lo + 32])
                for i, z in got.items():
                    worst = max(worst, float(np.abs(z - alone[i]).max()))
            print(f'{prec} KRK_LSTM_V={v}: batches of 32 in {name}: max |logit - alone| = {worst:.3e}', flush=True)
