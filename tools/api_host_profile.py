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
        # like the engine: two alternating buffers that are kept (pinned there; here at least already touched)
        need = int(np.prod(shape))
        pool = self.__dict__.setdefault('_pg', [None, None])
        self._pg_i = 1 - self.__dict__.get('_pg_i', 0)
        if pool[self._pg_i] is None or pool[self._pg_i].size < need:
            pool[self._pg_i] = np.zeros(need + need // 4, np.uint8)
        self.buf = pool[self._pg_i][:need].reshape(shape)
        return self.buf

    def upload_page_buffer(self):
        return self.buf

    def submit_boxes(self, page, boxes, pad, want_probs=False):
        self.k += 1
        self.q[self.k] = len(boxes)
        return self.k

    # the dewarp path of 1-channel models (--mode L): measurement and normalisation are device work, the host sees their handles
    in_height = 48
    slots = (0, 1, 2)

    def measure_dewarp_begin(self, crops, pool=None, ahead=0, page=None):
        n = len(crops)
        if page is None and pool is not None:           # packed crops: the packing copies are host work
            buf = np.empty(sum(a.size for a in crops), np.uint8)
            off = 0
            for a in crops:
                buf[off:off + a.size] = np.ascontiguousarray(a).reshape(-1)
                off += a.size
        self._dw_n = n

        class H:
            def result(_s):
                return np.full(n, 12, np.int32), np.ones(n, bool), np.ones(n, bool)
        return H()

    def submit_dewarped(self, r, use, pad, want_probs=False):
        self.k += 1
        self.q[self.k] = self._dw_n
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
    ap.add_argument('--mode', default='RGB', choices=['RGB', 'L'], help='RGB: 3-channel model, rectangular crops; L: 1-channel model, dewarped lines')
    ap.add_argument('--no-rows', action='store_true', help='np.asarray(im) instead of Pillow\'s row table (kraken_amd.pilmem)')
    args = ap.parse_args()
    if args.no_rows:
        R.PAGE_ROWS = False
    page, seg = bench._page_of_lines(args.lines, 1200 - 32, 48, args.mode)
    eng = StubEngine()
    R._engine_for = lambda net, temperature: eng
    R._fused_ok = lambda net: True
    net = types.SimpleNamespace(nn=types.SimpleNamespace(input=(1, 3 if args.mode == 'RGB' else 1, 48, 0), one_channel_mode='L', use_legacy_polygons=False,
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
