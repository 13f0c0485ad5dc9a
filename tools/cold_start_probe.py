"""Dev tool (GPU): where the FIRST rpred() pass of a fresh process spends its time (6 k lines/s on the first pass of a
2048-line page against 48 k warm).  Wraps the engine's construction and its first calls with wall-clock timers.
    python tools/cold_start_probe.py [--lines 2048] [--mode L|RGB] [--passes 3]"""
import argparse
import sys
import time
import warnings
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
t_import = time.perf_counter()
import torch  # noqa: E402

import bench  # noqa: E402
import kraken_amd  # noqa: E402
from kraken_amd import _lib, engine as E, rpred as R, vgsl as V  # noqa: E402
from kraken_amd.models import TorchSeqRecognizer  # noqa: E402
from kraken_amd.specs import BENCH_A, BENCH_A_RGB, DEFAULT_H120, bench_codec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--lines', type=int, default=2048)
ap.add_argument('--mode', default='L')
ap.add_argument('--passes', type=int, default=3)
ap.add_argument('--width', type=int, default=1200)
ap.add_argument('--height', type=int, default=48, help='48: BENCH-A; 120: kraken\'s default recognition spec')
ap.add_argument('--workers', type=int, default=8, help='num_line_workers (1: everything on the main thread, so that --cprofile sees the line preparation)')
ap.add_argument('--cprofile', type=int, default=0, help='print the top N functions (cumulative) of every pass')
ap.add_argument('--sort', default='cumulative', help='cProfile order: cumulative | tottime')
a = ap.parse_args()
print('import %.0f ms' % (1e3 * (time.perf_counter() - t_import)))

acc = {}
import gc  # noqa: E402
_gc_t0 = [0.0]


def _gc_cb(phase, info):          # how long the cyclic collector ran inside a pass, per generation
    if phase == 'start':
        _gc_t0[0] = time.perf_counter()
    else:
        e = acc.setdefault(f'gc generation {info["generation"]} ({info["collected"]} collected)', [0, 0.0])
        e[0] += 1
        e[1] += time.perf_counter() - _gc_t0[0]


gc.callbacks.append(_gc_cb)


def timed(obj, name, label=None):
    f = getattr(obj, name)
    label = label or f'{getattr(obj, "__name__", obj)}.{name}'

    def w(*x, **k):
        t0 = time.perf_counter()
        try:
            return f(*x, **k)
        finally:
            e = acc.setdefault(label, [0, 0.0])
            e[0] += 1
            e[1] += time.perf_counter() - t0
    setattr(obj, name, w)


timed(E.RecognitionEngine, '__init__', 'engine.__init__')
timed(E._Slot, '__init__', 'slot.__init__ (plan + streams + pinned result buffers)')
timed(E._Slot, 'ensure_results', 'slot.ensure_results')
timed(E._Slot, 'ensure_stage', 'slot.ensure_stage')
timed(E._Slot, 'ensure_crops', 'slot.ensure_crops')
timed(E._Slot, 'ensure_boxes', 'slot.ensure_boxes')
timed(E.RecognitionEngine, 'page_buffer', 'engine.page_buffer (pinned page)')
timed(E.RecognitionEngine, 'upload_page_buffer', 'engine.upload_page_buffer')
timed(E.RecognitionEngine, 'upload_page', 'engine.upload_page')
timed(E.RecognitionEngine, 'submit_boxes', 'engine.submit_boxes')
timed(E.RecognitionEngine, 'measure_dewarp_begin', 'engine.measure_dewarp_begin')
timed(E.RecognitionEngine, 'submit_dewarped', 'engine.submit_dewarped')
timed(E.RecognitionEngine, 'collect', 'engine.collect')
timed(V.HipSequential, 'plan', 'HipSequential.plan (krk_plan_create per slot)')
timed(R._RecognitionRun, '_advance', 'run._advance (everything below + line descriptors)')
timed(R._RecognitionRun, '_submit_dewarp', 'run._submit_dewarp')
timed(R._RecognitionRun, '_dewarp_finish_begun', 'run._dewarp_finish_begun')
timed(R._RecognitionRun, '_absorb', 'run._absorb (records)')
timed(R._RecognitionRun, '_strip_on_device', 'run._strip_on_device (page band upload)')
timed(R.LinePipeline, '_collect_one', 'pipe._collect_one (collect + decode)')
timed(R.LinePipeline, 'dewarp_begin', 'pipe.dewarp_begin')
timed(R.LinePipeline, 'dewarp_finish', 'pipe.dewarp_finish (waits for the measurement)')
timed(R.LinePipeline, 'submit_boxes', 'pipe.submit_boxes')
timed(R, '_decode_lines', 'rpred._decode_lines')

torch.manual_seed(0)
t0 = time.perf_counter()
spec = BENCH_A if a.mode == 'L' else BENCH_A_RGB
if a.height == 120:
    spec = DEFAULT_H120 if a.mode == 'L' else DEFAULT_H120.replace('[1,120,0,1 ', '[1,120,0,3 ')
m = kraken_amd.TorchVGSLModel(vgsl=spec, codec=bench_codec())
m.seg_type, m.model_type = 'bbox', ['recognition']
m.to('cuda:0')
net = TorchSeqRecognizer(m, device='cuda:0')
torch.cuda.synchronize()
print('model build + to(cuda) %.0f ms' % (1e3 * (time.perf_counter() - t0)))
page, seg = bench._page_of_lines(a.lines, a.width - 32, a.height, a.mode)
for p in range(a.passes):
    acc.clear()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        prof = None
        if a.cprofile:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        it = R.rpred(net, page, seg, bidi_reordering=False, num_line_workers=a.workers)
        first = next(it)
        t_first = time.perf_counter() - t0
        n = 1 + sum(1 for _ in it)
        dt = time.perf_counter() - t0
        if prof is not None:
            prof.disable()
    print(f'pass {p}: {n} lines in {1e3 * dt:.1f} ms = {n / dt:.0f} lines/s; first record after {1e3 * t_first:.1f} ms')
    for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f'    {1e3 * s:8.1f} ms  {c:4d} x  {k}')
    if prof is not None:
        import io
        import pstats
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats(a.sort).print_stats(a.cprofile)
        print('\n'.join(ln for ln in buf.getvalue().splitlines()[6:] if ln.strip()))
