"""Dev tool (GPU): the line-preparation kernels of the API path on their own -- krk_prep_lines (crop + LANCZOS resize + pad + invert) and the
CenterNormalizer dewarp (krk_dewarp_measure / krk_dewarp_apply) -- on 256 synthetic lines of one source height, timed with events;
run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/collect: profiles/r06_prep_kernels.txt).
    python tools/prep_probe.py --src-h 48 [--model-h 48] [--width 1168] [--lines 256] [--reps 10] [--case prep_L|prep_RGB|dewarp]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kraken_amd  # noqa: E402
from kraken_amd import _lib  # noqa: E402
from kraken_amd.engine import RecognitionEngine  # noqa: E402
from tests.helpers import wavy_line  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--src-h', type=int, default=48)
ap.add_argument('--model-h', type=int, default=48)
ap.add_argument('--width', type=int, default=1168)
ap.add_argument('--lines', type=int, default=256)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--case', default='all')
args = ap.parse_args()
dev = torch.device('cuda:0')
lib = _lib.load()
rng = np.random.RandomState(3)
n, sh, H, W = args.lines, args.src_h, args.model_h, args.width
lines = [wavy_line(rng, sh, W) for _ in range(8)]
gray = np.vstack([lines[i % 8] for i in range(n)])


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for ch, name in ((1, 'prep_L'), (3, 'prep_RGB'), (4, 'prep_RGBX')):
    if args.case not in ('all', name):
        continue
    page = gray if ch == 1 else np.stack([gray, np.roll(gray, 3, axis=1), np.minimum(gray, 200)] + ([np.full_like(gray, 255)] if ch == 4 else []), axis=2)
    ps, ch = ch, min(ch, 3)                       # 'prep_RGBX': a 3-channel model reading Pillow's own R, G, B, X rows (krk_prep_lines_fmt)
    pg = torch.from_numpy(np.ascontiguousarray(page)).to(dev)
    ow = int(W * H / sh)
    rows = [(0, i * sh, W, (i + 1) * sh, ow) for i in range(n)]
    bx = torch.tensor(rows, dtype=torch.int32, device=dev)
    wmax = ow + 32
    out = torch.empty((n, ch, H, wmax), device=dev)
    flags = torch.empty((n,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    if ps == 4:
        ms = timed(lambda: _lib.check(lib.krk_prep_lines_fmt(pg.data_ptr(), gray.shape[0], W, W * ps, ps, ch, bx.data_ptr(), n, sh, H, 16, wmax,
                                                             out.data_ptr(), flags.data_ptr(), st)), args.reps)
    else:
        ms = timed(lambda: _lib.check(lib.krk_prep_lines(pg.data_ptr(), gray.shape[0], W, ch, bx.data_ptr(), n, sh, H, 16, wmax,
                                                         out.data_ptr(), flags.data_ptr(), st)), args.reps)
    print(f'{name}: {n} lines {sh} x {W} -> {H} x {ow}: {ms:.3f} ms per batch ({n / ms:.1f} k lines/s); bytes in {page.nbytes / 1e6:.1f} MB, out {out.numel() * 4 / 1e6:.1f} MB')

if args.case in ('all', 'dewarp'):
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=f'[1,{H},0,1 Cr3,13,32 Mp2,2 Cr3,13,32 S1(1x0)1,3 Lbx16 O1c9]', codec={'a': [1]})
    m.nn.set_precision('bf16x3')
    m.to('cuda')
    eng = RecognitionEngine(m, device=0, max_batch=n, max_width=int(W * H / 8) + 64, slots=1)
    crops = [lines[i % 8] for i in range(n)]
    tm, ta = [], []
    for rep in range(args.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r, ok, ink = eng.measure_dewarp(crops)
        t1 = time.perf_counter()
        ticket = eng.submit_dewarped(r, ok & ink, 16)
        eng.slots[ticket].stream.synchronize()
        t2 = time.perf_counter()
        eng.collect(ticket)
        if rep:
            tm.append((t1 - t0) * 1e3)
            ta.append((t2 - t1) * 1e3)
    print(f'dewarp: {n} lines {sh} x {W} -> height {H}: measure (pack + upload + 8 kernels + read-back) {np.median(tm):.3f} ms, apply + tiny recognition {np.median(ta):.3f} ms; '
          f'r = {sorted(set(int(v) for v in r))[:6]}, used {int((ok & ink).sum())}')
    eng.close()
