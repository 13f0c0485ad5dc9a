"""
Host-side cost of the reference-API path (kraken_amd/rpred.py) WITHOUT a GPU: the RecognitionEngine is replaced by a stub
that returns a synthetic compact decode result at once, so what is timed is everything the host does per line -- bounds
checks, crop descriptors, batching, codec, cut arithmetic, record objects, ordering.  That is the ceiling of the API
path in lines/s per host process, whatever the device does.   python tools/api_host_profile.py [--lines 2048] [--profile]
"""
import argparse
import cProfile
import pstats
import sys
import time
import types
import warnings
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from kraken_amd import rpred as R  # noqa: E402
from kraken_amd.specs import bench_codec  # noqa: E402
from kraken_amd.codec import PytorchCodec  # noqa: E402
from kraken_amd.vgsl import DecodedBatch  # noqa: E402
from kraken_amd import ctc_decoder as _ctc  # noqa: E402


class StubEngine:
    def __init__(self, chars=85, T=150):
        self.rng = np.random.default_rng(0)
        self.chars, self.T = chars, T
        self.q = {}
        self.k = 0
        self.last_flags = None

    def free_slots(self):
        return 3 - len(self.q)

    def upload_page(self, arr):
        return arr

    def page_buffer(self, shape):
        self.buf = np.empty(shape, np.uint8)
        return self.buf

    def upload_page_buffer(self):
        return self.buf

    def submit_boxes(self, page, boxes, pad, want_probs=False):
        self.k += 1
        self.q[self.k] = len(boxes)
        return self.k

    def collect(self, t):
        n = self.q.pop(t)
        k, T = self.chars, self.T
        labels = self.rng.integers(1, 256, (n, T), dtype=np.int32)
        starts = np.tile(np.arange(T, dtype=np.int32), (n, 1))
        confs = self.rng.random((n, T), dtype=np.float32)
        self.last_flags = np.ones(n, np.uint8)
        return DecodedBatch(labels, starts, starts.copy(), confs, np.full(n, k, np.int32)), np.full(n, T, np.int32)

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lines', type=int, default=2048)
    ap.add_argument('--workers', type=int, default=16)
    ap.add_argument('--profile', action='store_true')
    ap.add_argument('--bidi', action='store_true')
    args = ap.parse_args()
    page, seg = bench._page_of_lines(args.lines, 1200 - 32, 48, 'RGB')
    eng = StubEngine()
    R._engine_for = lambda net, temperature: eng
    R._fused_ok = lambda net: True
    net = types.SimpleNamespace(nn=types.SimpleNamespace(input=(1, 3, 48, 0), one_channel_mode='L', use_legacy_polygons=False,
                                                         nn=types.SimpleNamespace(recognize=None)),
                                seg_type="bbox", codec=PytorchCodec(bench_codec()), decoder=_ctc.greedy_decoder, temperature=1.0)

    def go():
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return list(R.rpred(net, page, seg, bidi_reordering=args.bidi, num_line_workers=args.workers))
    go()
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        go()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(22)
    best = 0
    for _ in range(3):
        t0 = time.perf_counter()
        recs = go()
        best = max(best, len(recs) / (time.perf_counter() - t0))
    print(f'host-only API path: {best:.0f} lines/s ({args.lines} lines, {len(recs[0].prediction)} code points per line)')


if __name__ == '__main__':
    main()
