"""
Host logic of the recognition generators (kraken_amd/rpred.py) without a GPU: a recogniser object with the reference's
legacy interface (``.nn.input``, ``.predict``, ``.outputs``, ``.codec`` -- SURVEY.md seam B5) takes the per-line path,
so ordering, tag routing, ignore rules, empty-record rules, chunking and record arithmetic are checked on the CPU.
Mirrors the behaviours the reference pins in tests/test_rpred.py:366-462.
"""
import types
import warnings
from collections import defaultdict

import numpy as np
import pytest
import torch
from PIL import Image

from kraken_amd import rpred as R
from kraken_amd.codec import PytorchCodec
from kraken_amd.containers import BBoxLine, Segmentation
from kraken_amd.vgsl import DecodedBatch


class FakeNet:
    """Emits one character per 16 input columns; remembers what it was called with."""

    def __init__(self, char='a', height=48, channels=3, seg_type='bbox'):
        self.nn = types.SimpleNamespace(input=(1, channels, height, 0), one_channel_mode='L', use_legacy_polygons=False)
        self.seg_type = seg_type
        self.char = char
        self.calls = []
        self.outputs = None
        self.codec = None

    def predict(self, x):
        assert x.dim() == 4 and x.shape[0] == 1
        w = x.shape[3]
        t = max(w // 8, 1)
        self.calls.append(tuple(x.shape))
        self.outputs = np.zeros((1, 4, t), np.float32)
        n = max(w // 16, 1)
        return [[(self.char, 2 * i, min(2 * i + 1, t - 1), 0.5 + 0.01 * i) for i in range(n) if 2 * i < t]]


def page(w=640, h=400, seed=0):
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8), 'RGB')


def seg(boxes, tags=None, script_detection=False, direction='horizontal-lr'):
    lines = [BBoxLine(id=f'l{i}', bbox=list(b), tags=None if tags is None else tags[i]) for i, b in enumerate(boxes)]
    return Segmentation(type='bbox', imagename='p.png', lines=lines, text_direction=direction,
                        script_detection=script_detection)


def run(*a, **k):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        return list(R.mm_rpred(*a, **k))


def test_records_come_back_in_input_order_with_reference_cut_arithmetic():
    boxes = [(0, 10 * i, 100 + 37 * i, 10 * i + 30) for i in range(11)]        # widths 100 .. 470, unsorted by design below
    boxes = boxes[::2] + boxes[1::2]
    net = FakeNet('a')
    im = page()
    recs = run(defaultdict(lambda: net), im, seg(boxes), pad=16, bidi_reordering=False, num_line_workers=3)
    assert len(recs) == len(boxes) and [r.line.id for r in recs] == [f'l{i}' for i in range(len(boxes))]
    for r, b in zip(recs, boxes):
        bw, bh = b[2] - b[0], b[3] - b[1]
        w_in = int(bw * 48 / bh) + 32                       # fixed-height resize + 16 px padding on both sides
        t = w_in // 8
        net_scale, in_scale = w_in / t, bw / (w_in - 32)
        assert len(r.prediction) == max(w_in // 16, 1) or 2 * (len(r.prediction) - 1) < t
        for k, cut in enumerate(r.cuts):
            lo = int(round(min(max((2 * k * net_scale - 16) * in_scale, 0), bw - 1)))
            hi = int(round(min(max((min(2 * k + 1, t - 1) * net_scale - 16) * in_scale, 0), bw - 1)))
            assert cut == [[b[0] + lo, b[1]], [b[0] + lo, b[3]], [b[0] + hi, b[3]], [b[0] + hi, b[1]]]
        np.testing.assert_allclose(r.confidences, [0.5 + 0.01 * k for k in range(len(r.prediction))], rtol=1e-6)


def test_tag_routing_ignore_and_missing_models():
    """reference tests/test_rpred.py:366-440: per-tag models, tags_ignore -> empty record, missing model -> error."""
    boxes = [(0, 0, 200, 40), (0, 50, 300, 90), (0, 100, 250, 140)]
    tags = [{'type': [{'type': 'foobar'}]}, {'type': [{'type': 'default'}]}, {'type': [{'type': 'foobar'}]}]
    a, b = FakeNet('a'), FakeNet('b')
    recs = run({'default': a, 'foobar': b}, page(), seg(boxes, tags, True), bidi_reordering=False)
    assert [set(r.prediction) for r in recs] == [{'b'}, {'a'}, {'b'}]
    assert len(a.calls) == 1 and len(b.calls) == 2
    # ignored tag: empty record, its model is never called
    a, b = FakeNet('a'), FakeNet('b')
    recs = run({'default': a}, page(), seg(boxes, tags, True), bidi_reordering=False, tags_ignore=['foobar'])
    assert [r.prediction for r in recs][0] == '' and recs[2].prediction == '' and set(recs[1].prediction) == {'a'}
    assert recs[0].cuts == () or list(recs[0].cuts) == []
    # default factory serves every tag
    a = FakeNet('a')
    recs = run(defaultdict(lambda: a), page(), seg(boxes, tags, True), bidi_reordering=False)
    assert all(set(r.prediction) == {'a'} for r in recs) and len(a.calls) == 3
    # a tag without a model and without a default
    with pytest.raises(Exception):
        run({'default': FakeNet()}, page(), seg(boxes, tags, True))
    # no tags in the data and no default model (reference: ValueError)
    with pytest.raises(ValueError):
        run({('type', 'default'): FakeNet()}, page(), seg(boxes))


def test_empty_record_rules_and_iterator_protocol():
    im = page()
    flat = Image.new('RGB', (640, 400), (255, 255, 255))
    boxes = [(0, 0, 200, 40), (10, 10, 10, 50), (700, 0, 800, 40), (0, 60, 200, 100)]      # ok, zero width, outside (the reference's lexicographic test), ok
    net = FakeNet('a')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        it = R.mm_rpred(defaultdict(lambda: net), im, seg(boxes), bidi_reordering=False)
        assert len(it) == 4
        recs = [next(it) for _ in range(4)]
        with pytest.raises(StopIteration):
            next(it)
    assert [bool(r.prediction) for r in recs] == [True, False, False, True]
    # a flat (all white) page gives only empty records: max == min after the transform (kraken/rpred.py:221)
    recs = run(defaultdict(lambda: net), flat, seg(boxes[:1]), bidi_reordering=False)
    assert recs[0].prediction == ''


def test_lines_of_different_heights_are_never_padded_into_one_batch():
    """variable-height specs ([1,0,0,1 ...]) keep each line's own height."""
    net = FakeNet('a', height=0, channels=1)
    boxes = [(0, 0, 200, 40), (0, 50, 300, 80), (0, 100, 250, 160)]
    recs = run(defaultdict(lambda: net), page().convert('L'), seg(boxes), bidi_reordering=False)
    assert [c[2] for c in net.calls] == [40, 30, 60]
    assert all(r.prediction for r in recs)


def test_many_lines_are_processed_in_chunks_and_stay_ordered(monkeypatch):
    monkeypatch.setattr(R, 'ENGINE_SLOTS', 2)
    rng = np.random.default_rng(3)
    boxes = [(0, 0, int(w), 30) for w in rng.integers(40, 600, 300)]
    net = FakeNet('a')
    recs = run(defaultdict(lambda: net), page(), seg(boxes), bidi_reordering=False, batch_size=16)
    assert [r.line.id for r in recs] == [f'l{i}' for i in range(300)]
    assert len(net.calls) == 300


def test_decode_lines_table_path_equals_the_codec():
    from tests.specs import bench_codec
    rng = np.random.default_rng(1)
    n, t = 9, 21
    counts = np.array([0, 21, 3, 0, 7, 1, 21, 0, 5], np.int32)
    labels = rng.integers(1, 300, size=(n, t)).astype(np.int32)          # labels above 255 are not decodable: skipped
    starts = np.sort(rng.integers(0, 150, size=(n, t)), axis=1).astype(np.int32)
    ends = starts + 1
    confs = rng.random((n, t)).astype(np.float32)
    batch = DecodedBatch(labels, starts, ends, confs, counts)
    olens = np.arange(100, 100 + n)
    single = PytorchCodec(bench_codec())
    got = R._decode_lines(single, batch, olens)
    want = single.decode_batch(batch)
    for g, w, ol in zip(got, want, olens):
        assert g.text == ''.join(x[0] for x in w) and g.out_width == ol
        assert g.starts.tolist() == [x[1] for x in w] and g.ends.tolist() == [x[2] for x in w]
        np.testing.assert_allclose(g.confs, [x[3] for x in w], rtol=0, atol=0)
    # every label decodable: the batch-wide path (one utf-32 decode, a line's string a slice of it), garbage behind the counts
    clean = rng.integers(1, 200, size=(n, t)).astype(np.int32)
    for i, k in enumerate(counts):
        clean[i, k:] = rng.integers(-5, 5000, size=t - k)
    batch2 = DecodedBatch(clean, starts, ends, confs, counts)
    got2, want2 = R._decode_lines(single, batch2, olens), single.decode_batch(batch2)
    for g, w in zip(got2, want2):
        assert g.text == ''.join(x[0] for x in w) and len(g.text) == len(g.starts) == len(w)
        assert g.starts.tolist() == [x[1] for x in w] and g.confs.tolist() == [np.float32(x[3]) for x in w]
    multi = PytorchCodec({'a': [1], 'bc': [2], 'd': [3, 4]})
    small = DecodedBatch(np.array([[1, 2, 3, 4], [3, 1, 0, 0]], np.int32), np.array([[0, 2, 4, 6], [1, 3, 0, 0]], np.int32),
                         np.array([[1, 3, 5, 7], [2, 4, 0, 0]], np.int32), np.full((2, 4), 0.5, np.float32), np.array([4, 2], np.int32))
    got = R._decode_lines(multi, small, [8, 8])
    assert [g.text for g in got] == ['abcd', 'a'] and got[0].starts.tolist() == [0, 2, 2, 4]


def test_new_api_records_carry_logits_and_line_images():
    """lib/vgsl/rpred.py:155-160: return_logits / return_line_image; custom decoders take the per-line path."""
    calls = []

    class Model:
        input = (1, 3, 48, 0)
        use_legacy_polygons = False
        codec = types.SimpleNamespace(decode=lambda locs: [('x', s, e, c) for _, s, e, c in locs])

        class nn:   # the network operator: logits (1, C, 1, T)
            @staticmethod
            def __call__(x, lens=None):
                raise AssertionError

        def __init__(self):
            self.nn = lambda x, lens=None: (calls.append(tuple(x.shape)) or torch.zeros(x.shape[0], 5, 1, x.shape[3] // 8), None)

    def my_decoder(outputs, lens):
        return [[(1, 0, 1, 0.75), (2, 3, 4, 0.5)] for _ in range(outputs.shape[0])]

    cfg = types.SimpleNamespace(batch_size=2, temperature=1.0, padding=16, bidi_reordering=False, num_line_workers=0,
                                return_logits=True, return_line_image=True, decoder=my_decoder)
    boxes = [(0, 0, 200, 40), (0, 50, 300, 90)]
    recs = list(R.recognition_pred(Model(), page(), seg(boxes), cfg))
    assert [r.prediction for r in recs] == ['xx', 'xx'] and len(calls) == 2
    assert recs[0].logits == [('x', 0, 1, 0.75), ('x', 3, 4, 0.5)]
    assert recs[1].image.size == (300, 40)
    cfg.return_logits = cfg.return_line_image = False
    recs = list(R.recognition_pred(Model(), page(), seg(boxes), cfg))
    assert recs[0].logits is None and recs[0].image is None


def test_lazy_lists_behave_like_the_lists_the_reference_stores():
    """cuts / confidences are computed on first use (kraken_amd.rpred.LazyList): same values, list semantics for readers."""
    import pickle
    calls = []

    def make():
        calls.append(1)
        return [[[0, 1], [0, 5], [3, 5], [3, 1]], [[4, 1], [4, 5], [9, 5], [9, 1]], [[10, 1], [10, 5], [12, 5], [12, 1]]]
    lz = R.LazyList(make, 3)
    assert len(lz) == 3 and bool(lz) and not calls                      # length and truthiness do not build anything
    assert lz[1] == [[4, 1], [4, 5], [9, 5], [9, 1]] and len(calls) == 1
    assert lz[0:2] == make()[0:2] and lz[-1][0] == [10, 1]               # slices (kraken/serialization.py:211) and negatives
    assert [c[0][0] for c in lz] == [0, 4, 10] and lz == make()
    assert len(calls) == 3                                               # the two explicit make() calls above, not the reads
    assert pickle.loads(pickle.dumps(lz)) == make() and isinstance(pickle.loads(pickle.dumps(lz)), list)
    assert not R.LazyList(list, 0) and list(R.LazyList(list, 0)) == []
    # through the generator: lazy without bidi reordering, plain lists with it (reordering walks them right away)
    boxes = [(0, 0, 200, 30), (0, 40, 300, 70)]
    net = FakeNet('a')
    for bidi in (False, True):
        recs = run(defaultdict(lambda: net), page(), seg(boxes), pad=16, bidi_reordering=bidi)
        assert all(len(r.cuts) == len(r.prediction) == len(r.confidences) for r in recs)
        assert all(isinstance(r.cuts, R.LazyList) != bidi for r in recs)
        assert [list(r.cuts) for r in recs] == [[list(c) for c in r.cuts] for r in recs]


# ------------------------------------------------------------------ engine ownership (abandoned runs, shared engines)
class _StubEngine:
    """The RecognitionEngine surface LinePipeline uses, on the host: submit_staged queues, collect answers one char per line."""
    made = []

    def __init__(self, model, device=0, max_batch=32, max_width=256, slots=3, temperature=1.0):
        self.slots = [types.SimpleNamespace(busy=False) for _ in range(slots)]
        self.temperature = temperature
        self.in_use = self.closed = False
        self.q, self.k, self.resets, self.last_flags = {}, 0, 0, None
        self.fail_next_collect = False
        _StubEngine.made.append(self)

    def free_slots(self):
        return len(self.slots) - len(self.q)

    def stage(self, n, w, height=None):
        self._staged = n
        return np.zeros((n, 3, 48, w), np.float32)

    def submit_staged(self, lens=None, want_probs=False):
        assert self.free_slots() > 0, 'all slots busy'
        self.k += 1
        self.q[self.k] = np.asarray(lens)
        return self.k

    def collect(self, ticket):
        lens = self.q.pop(ticket)
        if self.fail_next_collect:
            self.fail_next_collect = False
            raise RuntimeError('device error')
        n = len(lens)
        one = np.ones((n, 1), np.int32)
        return DecodedBatch(one, 0 * one, one, np.full((n, 1), 0.5, np.float32), np.ones(n, np.int32)), (lens // 8).astype(np.int32)

    def last_probs(self):
        return None

    def reset(self):
        self.q.clear()
        self.resets += 1

    def close(self):
        self.closed = True


def _stub_recogniser(monkeypatch):
    import kraken_amd.engine as E
    monkeypatch.setattr(E, 'RecognitionEngine', _StubEngine)
    monkeypatch.setattr(R, '_fused_ok', lambda net: True)
    monkeypatch.setattr(R, 'DEVICE_PREP', False)
    _StubEngine.made.clear()
    w = torch.nn.Parameter(torch.zeros(1))
    hs = types.SimpleNamespace(precision=2, _weights_version=lambda: (w.data_ptr(), w._version), _engines={})
    hs.__dict__['_engines'] = {}
    vgsl = types.SimpleNamespace(nn=hs, input=(1, 3, 48, 0), one_channel_mode='L', use_legacy_polygons=False,
                                 parameters=lambda: iter([types.SimpleNamespace(is_cuda=True, device=types.SimpleNamespace(index=0))]))
    net = types.SimpleNamespace(nn=vgsl, seg_type='bbox', codec=PytorchCodec({'a': [1]}), temperature=1.0)
    return net, hs, w


def test_an_abandoned_run_hands_back_a_clean_engine(monkeypatch):
    """A generator dropped mid-page (break / next(...) once) must not leave the model's cached engine with busy slots."""
    net, hs, _ = _stub_recogniser(monkeypatch)
    boxes = [(0, 0, 100 + i, 30) for i in range(200)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        it = R.mm_rpred(defaultdict(lambda: net), page(), seg(boxes), bidi_reordering=False, batch_size=32)
        assert next(it).prediction == 'a'
        eng = _StubEngine.made[0]
        assert eng.in_use and len(eng.q) > 0                  # batches still in flight
        del it                                               # abandoned: the run's finaliser hands the engine back
        import gc
        gc.collect()                                         # (lazy cut lists refer back to the run: a cycle)
        assert not eng.in_use and not eng.q and eng.resets >= 1
        recs = list(R.mm_rpred(defaultdict(lambda: net), page(), seg(boxes), bidi_reordering=False, batch_size=32))
    assert len(recs) == 200 and all(r.prediction == 'a' for r in recs)
    assert len(_StubEngine.made) == 1 and not eng.in_use     # the same engine served the second page


def test_two_live_runs_on_one_model_do_not_share_an_engine(monkeypatch):
    net, hs, w = _stub_recogniser(monkeypatch)
    boxes = [(0, 0, 100 + i, 30) for i in range(120)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        a = R.mm_rpred(defaultdict(lambda: net), page(), seg(boxes), bidi_reordering=False, batch_size=32)
        b = R.mm_rpred(defaultdict(lambda: net), page(), seg(boxes), bidi_reordering=False, batch_size=32)
        ra, rb = [next(a)], [next(b)]
        assert len(_StubEngine.made) == 2 and all(e.in_use for e in _StubEngine.made)
        # a different temperature is an argument of the calls, not another engine
        net.temperature = 0.5
        ra += list(a)
        rb += list(b)
        assert len(ra) == len(rb) == 120
        c = list(R.mm_rpred(defaultdict(lambda: net), page(), seg(boxes[:40]), bidi_reordering=False))
        assert len(c) == 40 and len(_StubEngine.made) == 2 and _StubEngine.made[0].temperature == 0.5
        # weights updated in place: the cached engines are stale -> closed, a fresh one is built
        with torch.no_grad():
            w.add_(1.0)
        list(R.mm_rpred(defaultdict(lambda: net), page(), seg(boxes[:40]), bidi_reordering=False))
    assert len(_StubEngine.made) == 3 and _StubEngine.made[0].closed and _StubEngine.made[1].closed


def test_a_failed_batch_frees_the_engine(monkeypatch):
    net, hs, _ = _stub_recogniser(monkeypatch)
    boxes = [(0, 0, 100 + i, 30) for i in range(100)]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', DeprecationWarning)
        it = R.mm_rpred(defaultdict(lambda: net), page(), seg(boxes), bidi_reordering=False, batch_size=32)
        next(it)
        _StubEngine.made[0].fail_next_collect = True
        with pytest.raises(RuntimeError):
            list(it)
        assert not _StubEngine.made[0].in_use and not _StubEngine.made[0].q
        assert len(list(R.mm_rpred(defaultdict(lambda: net), page(), seg(boxes), bidi_reordering=False))) == 100


def test_width_buckets_cover_every_line_once_widest_first_and_respect_both_limits():
    """BASELINE config 4's bucketing rule (reference: width-sorted fixed-size batches, kraken/lib/vgsl/rpred.py:129-131)."""
    from kraken_amd.rpred import width_buckets
    rng = np.random.RandomState(40)
    widths = rng.randint(400, 2401, size=1024)
    for max_lines, px in ((256, 0), (128, 0), (512, 307200), (64, 50000), (7, 0)):
        b = width_buckets(widths, max_lines, px)
        assert sorted(np.concatenate(b).tolist()) == list(range(1024))
        tops = [int(widths[i].max()) for i in b]
        assert tops == sorted(tops, reverse=True)                       # widest bucket first
        for i, idx in enumerate(b):
            assert 1 <= len(idx) <= max_lines
            assert px == 0 or len(idx) == 1 or len(idx) * int(widths[idx].max()) <= px
            if i + 1 < len(b):
                assert widths[idx].min() >= widths[b[i + 1]].max()      # buckets partition the sorted order
    assert width_buckets([], 8) == []
    assert [x.tolist() for x in width_buckets([5, 5, 5], 2)] == [[2], [0, 1]]


# ------------------------------------------------------------------ the dewarp pipeline's state machine on the host (round 6)
class _DewarpStubEngine(_StubEngine):
    """The dewarp surface of RecognitionEngine with the RULES of the real one as assertions: a batch's second half goes into the slot
    its measurement was begun on, nothing else is submitted in between, `ahead=1` only behind a begun batch, never more batches in
    flight than slots.  Every 11th line's band "leaves the padded stack" (ok False: host transform), every 17th line is flat."""
    in_height, in_channels, device = 48, 1, 0

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.begun = None          # [n] of the batch whose measurement is in flight
        self.log = []
        self.uploads = 0

    def upload_rows(self, table, y0, y1, pins=None):
        self.uploads += 1
        return types.SimpleNamespace(rows=(y0, y1))

    def measure_dewarp_begin(self, crops, pool=None, ahead=0, page=None):
        assert page is not None, 'the lines of this test are boxes of the uploaded page'
        assert (ahead == 1) == (self.begun is not None), ('ahead', ahead, self.begun)
        assert self.free_slots() >= 1 + ahead, 'no slot for the measurement'
        n = len(crops)
        r = np.full(n, 12, np.int32)
        idx = (np.asarray(crops)[:, 1] + page.rows[0]) // 40    # line index: the boxes are 40 rows apart, relative to the band's first row
        ok, ink = idx % 61 != 3, idx % 17 != 5
        prev, self.begun_next = self.begun, [n, ok, ink]
        self.log.append(('begin', n, ahead))
        stub = self

        class H:
            def result(_):
                return r, ok, ink
        if prev is None:
            self.begun = self.begun_next
        else:
            self.pending_begin = self.begun_next
        return H()

    def submit_dewarped(self, r, use, pad, want_probs=False):
        assert self.begun is not None, 'second half without a first'
        n = self.begun[0]
        assert len(r) == n and len(use) == n
        self.log.append(('finish', n))
        self.begun = self.__dict__.pop('pending_begin', None)
        t = self.submit_staged(np.full(n, 8 * 30))
        self.__dict__.setdefault('flags_of', {})[t] = np.asarray(use).astype(np.int32)      # krk_dewarp_apply's "holds ink" flags
        return t

    def collect(self, ticket):
        self.last_flags = self.__dict__.get('flags_of', {}).pop(ticket, None)
        return super().collect(ticket)

    def submit_staged(self, lens=None, want_probs=False):
        self.log.append(('staged', len(lens)))
        return super().submit_staged(lens, want_probs)


@pytest.mark.parametrize('two_models', [False, True])
def test_dewarp_batches_are_pipelined_across_chunks_without_breaking_the_engines_rules(monkeypatch, two_models):
    """Descriptor-only lines go batch by batch; a dewarp batch's measurement stays in flight until the next batch has begun
    (_dw_begun), the end of the page or a submission of another kind: the stub engine asserts the real engine's rules on every call.
    Lines for the host transform are submitted when no batch is begun; flat lines give empty records; every line gets its record,
    in input order, with one or two recognisers sharing the page."""
    import kraken_amd.engine as E
    monkeypatch.setattr(E, 'RecognitionEngine', _DewarpStubEngine)
    monkeypatch.setattr(R, '_fused_ok', lambda net: True)
    monkeypatch.setattr(R, 'PIN_PAGES', False)
    _StubEngine.made.clear()

    def recogniser(ch):
        w = torch.nn.Parameter(torch.zeros(1))
        hs = types.SimpleNamespace(precision=2, _weights_version=lambda: (w.data_ptr(), w._version))
        hs.__dict__['_engines'] = {}
        vgsl = types.SimpleNamespace(nn=hs, input=(1, 1, 48, 0), one_channel_mode='L', use_legacy_polygons=False,
                                     parameters=lambda: iter([types.SimpleNamespace(is_cuda=True, device=types.SimpleNamespace(index=0))]))
        return types.SimpleNamespace(nn=vgsl, seg_type='bbox', codec=PytorchCodec({ch: [1]}), temperature=1.0)
    a, b = recogniser('a'), recogniser('b')
    n = 230
    rng = np.random.default_rng(2)
    px = rng.integers(0, 255, (40 * n, 300), dtype=np.uint8)
    for i in range(5, n, 17):
        px[40 * i:40 * i + 40] = 255                                         # the flat lines are white (any other uniform value is recognised)
    im = Image.fromarray(px, 'L')
    boxes = [(0, 40 * i, 200 + i % 50, 40 * i + 36) for i in range(n)]
    tags = [{'type': [{'type': 'foo' if two_models and (i // 40) % 2 else 'default'}]} for i in range(n)]
    nets = {'default': a, 'foo': b} if two_models else defaultdict(lambda: a)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        recs = list(R.mm_rpred(nets, im, seg(boxes, tags=tags, script_detection=True), bidi_reordering=False, batch_size=32))
    assert len(recs) == n
    engines = _StubEngine.made
    assert len(engines) == (2 if two_models else 1)
    for e in engines:
        assert e.begun is None and not e.q and not e.in_use                  # nothing left begun or in flight
        begins = [x for x in e.log if x[0] == 'begin']
        assert len(begins) == sum(1 for x in e.log if x[0] == 'finish') >= 2
        # batches are begun behind their predecessor (ahead = 1) except the first and those behind a batch with lines for the host
        # transform (the pipeline is drained for them)
        assert sum(x[2] for x in begins) >= (2 if not two_models else 1), begins
    for i, r in enumerate(recs):
        want = 'b' if two_models and (i // 40) % 2 else 'a'
        if i % 17 == 5 and i % 61 != 3:
            assert r.prediction == '', i                                      # flat line: empty record
        elif i % 61 == 3:
            assert r.prediction in ('', want), i                              # host transform (random pixels: recognised by the stub or flat)
        else:
            assert r.prediction == want, i
