"""
Line-recognition generators on top of the pipelined HIP engine.

Two front ends of the reference funnel into the same three calls (``nn(x, lens)`` -> softmax -> decoder -> codec):

* the legacy ``kraken.rpred.rpred`` / ``mm_rpred`` generators (reference kraken/rpred.py:57-370): same constructor
  arguments, the same iterator protocol (``len()``, ``next()`` yields one ``ocr_record`` per line, in input order),
  per-tag model routing (``_resolve_type_to_model`` :373-391), the same empty-record rules (ignored tags, failed
  extraction, zero-sized crops, failed tensor conversion, flat lines -- :193-223) and the same position arithmetic
  (``_scale_val`` :329-330);
* the batched API ``TorchVGSLModel.predict(im, segmentation)`` = ``VGSLRecognitionInference._recognition_pred``
  (reference kraken/lib/vgsl/rpred.py:56-232): config-driven (``batch_size``, ``temperature``, ``padding``,
  ``num_line_workers``, ``return_logits``, ``return_line_image``, ``bidi_reordering``, ``decoder``), records carry
  ``logits`` / ``image`` when asked (``recognition_pred`` below).

What changes underneath: the reference runs one line (legacy) or one padded batch (new API) at a time through torch
and decodes on the host.  Here both front ends drive a ``LinePipeline``: lines are prepared (crop, dewarp / resize,
pad) on a pool of host threads one CHUNK ahead of the device, width-sorted into buckets so that a batch pads to a
similar width (BASELINE config 4: "length bucketing + packed LSTM"), staged into pinned memory and submitted to a
``RecognitionEngine`` that keeps several batches in flight; records are assembled from the compact label tuples with
vectorised scaling.  Because the kernels implement masked padding, a line's result does not depend on its batch mates
(it equals the reference's per-line ``rpred`` result), so the device batch size is a tuning knob of THIS implementation
(``ENGINE_BATCH``) and not the caller's ``batch_size``.

Line extraction: bounding-box lines are cropped here (the reference's ``extract_polygons`` bbox branch,
kraken/lib/segmentation.py:1630-1643); baseline/polygon extraction is CPU geometry outside the hot path and is
delegated to kraken when it is installed.  The module-level name ``extract_polygons`` is kept patchable like in the
reference (tests/test_newpolygons.py:64-102 mocks it).
"""
import collections.abc
import dataclasses
import logging
import math
import os
import warnings
from collections import defaultdict, deque
from concurrent.futures import ThreadPoolExecutor
from functools import partial
from typing import Optional, Union

import numpy as np
import torch
from PIL import Image

from . import ctc_decoder as _ctc
from . import pilmem
from .containers import BaselineOCRRecord, BBoxOCRRecord
from .transforms import ImageInputTransforms

__all__ = ['mm_rpred', 'rpred', 'recognition_pred', 'extract_polygons', 'LinePipeline']

logger = logging.getLogger(__name__)

ENGINE_BATCH = 256     # lines per device batch (results do not depend on it: masked padding, see the module docstring)
ENGINE_SLOTS = 3       # device batches in flight per recogniser
DEVICE_PREP = True     # crop / resize / pad / invert eligible lines on the device (krk_prep_lines) instead of with PIL
DEVICE_DEWARP = True   # ... and the CenterNormalizer dewarp of 1-channel bbox lines (krk_dewarp_measure / krk_dewarp_apply) instead of scipy
PAGE_ROWS = True       # upload the page's rows straight from Pillow's memory (kraken_amd.pilmem) instead of through np.asarray(im)
DEWARP_BATCH_PIXELS = 32 * 1024 * 1024   # pixels per device dewarp batch (krk_dewarp_measure: 24 bytes of fp64 scratch per pixel, 32-bit offsets)
PIN_PAGES = os.environ.get('KRK_PIN_PAGES', '1') == '1'     # page-lock Pillow's blocks in place for the run (hipHostRegister): the band copies are asynchronous (KRK_PIN_PAGES=0: pageable copies)
PREP_THREADS = 4       # host threads preparing line images when the caller does not say (``num_line_workers``); PIL holds the GIL in its
                       # conversions: 2..6 threads give the same throughput, 16 and more lose 40 % to contention


class KrakenInputException(Exception):
    pass


try:
    from kraken.lib.exceptions import KrakenInputException  # type: ignore # noqa: F811
except Exception:  # pragma: no cover
    pass


def extract_polygons(im, bounds, legacy: bool = False):
    """
    Yields (line image, line) for every line of `bounds`.  bbox segmentations are cropped with PIL
    (kraken/lib/segmentation.py:1630-1643); baseline segmentations need kraken's polygon extractor.
    """
    if bounds.type == 'baselines':
        try:
            from kraken.lib.segmentation import extract_polygons as _kraken_extract
        except Exception as e:  # pragma: no cover
            raise NotImplementedError('baseline line extraction is outside the hot path and needs kraken '
                                      '(kraken.lib.segmentation.extract_polygons)') from e
        yield from _kraken_extract(im, bounds, legacy=legacy)
        return
    angle = 90 if bounds.text_direction.startswith('vertical') else 0
    for line in bounds.lines:
        box = list(line.bbox)
        if box < [0, 0, 0, 0] or box[::2] >= [im.size[0], im.size[0]] or box[1::2] >= [im.size[1], im.size[1]]:
            logger.error('bbox {} is outside of image bounds {}'.format(box, im.size))
            raise ValueError('Line outside of image bounds')
        yield im.crop(box).rotate(angle, expand=True), line


_EXTRACT_POLYGONS = extract_polygons      # a patched module attribute (tests mock it) switches the device-side crop off


def _line_type(tags: Optional[dict], default: str = 'default') -> str:
    if tags is None:
        return default
    first = tags.get('type', [{'type': default}])[0]
    t = first.get('type')
    return t if t is not None else default


def _pick_model(tags, model_map, default=None):
    """tag -> (tag name, recogniser); mirrors _resolve_type_to_model (kraken/rpred.py:373-391)."""
    tag = None
    if tags is not None:
        try:
            tag = _line_type(tags)
        except Exception:
            pass
    if not tag and default:
        return 'default', default
    if tag in model_map:
        return tag, model_map[tag]
    if tag and default:
        return tag, default
    raise KrakenInputException(f'No model for type {tag}')


def _is_bitonal(im) -> bool:
    cols = im.getcolors(2)
    return cols is not None and len(cols) == 2


@dataclasses.dataclass
class _Pending:
    idx: int
    line: object
    tag: str
    net: object
    tensor: Optional[torch.Tensor]      # prepared on the host ... or None:
    box_size: tuple
    image: object = None
    width: int = 0                      # network input width (resized line + padding)
    box: Optional[tuple] = None         # ... prepared on the device: (x0, y0, x1, y1, resized width) into the uploaded page
    mode: str = ''                      # PIL mode of the page the device crops from
    crop: object = None                 # ... or a uint8 line image cut out on the host, resized / padded / inverted on the device


class _PageCrop:
    """A dewarp line that is a box of the page the device holds: shape like the uint8 array a host-side crop would be, its first
    pixel looked up on demand (only a UNIFORM line needs it, LinePipeline.dewarp_finish)."""
    __slots__ = ('box', 'shape', 'size', '_first')

    def __init__(self, box, first):
        self.box = box
        self.shape = (box[3] - box[1], box[2] - box[0])
        self.size = self.shape[0] * self.shape[1]
        self._first = first

    def first(self) -> int:
        return self._first()


class LazyList(collections.abc.Sequence):
    """
    A read-only list that is computed on first use.  The per-code-point ``cuts`` of a record are ~13 Python objects per
    code point (a list of four [x, y] pairs each): built eagerly they cost more host time per line than everything else the
    API path does, and most consumers (text export, string comparison) never look at them.  Behaves like the list the
    reference stores (indexing, slicing, iteration, len, ==, pickling as a plain list).
    """
    __slots__ = ('_make', '_n', '_items')

    def __init__(self, make, n: int):
        self._make, self._n, self._items = make, n, None

    def _get(self) -> list:
        if self._items is None:
            self._items = self._make()
            self._make = None
        return self._items

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())

    def __eq__(self, other):
        return self._get() == (other._get() if isinstance(other, LazyList) else other)

    def __repr__(self):
        return repr(self._get())

    def __reduce__(self):
        return (list, (self._get(),))


@dataclasses.dataclass
class LineResult:
    """What the device path returns for one line: code points with their network-output positions."""
    text: str
    starts: np.ndarray        # network time steps (inclusive first / last step of every code point)
    ends: np.ndarray
    confs: np.ndarray         # float32
    out_width: int            # valid output steps of this line (the reference's ``outputs.shape[2]`` / ``olen``)
    probs: object = None      # (C, out_width) softmax (torch tensor) when asked for


# ---------------------------------------------------------------------------------------------------- device side
def _fused_ok(net) -> bool:
    """The engine runs greedy best-path decoding on the device: only for recognisers whose decoder IS greedy_decoder."""
    vgsl = getattr(net, 'nn', None)
    hs = getattr(vgsl, 'nn', None)
    if hs is None or not hasattr(hs, 'recognize'):
        return False
    # The engine sizes its slots for ONE line of the widest width (krk_plan_out_dims with N = 1): a general Reshape must divide
    # every batch's (lines, width), not that probe's, and a network that changes the number of lines has no per-line output at all
    # (krk_recognize refuses it) -- such recognisers take the synchronous path
    if any(sp.kind == 'reshape' and sp.params.get('general') for sp in getattr(vgsl, 'layer_specs', ())):
        return False
    if any(sp.kind == 'add' and sp.params.get('axis') == 0 for sp in getattr(vgsl, 'layer_specs', ())):
        return False
    return getattr(net, 'decoder', _ctc.greedy_decoder) is _ctc.greedy_decoder


def _engine_for(net, temperature: float):
    """
    Checks out a RecognitionEngine of a recogniser for ONE consumer (a LinePipeline); None for models the engine does not take
    (variable height).  Engines are cached on the model per (device, arithmetic, weights version) -- the softmax temperature is
    an argument of every call, not a property of the plans -- and handed out exclusively: a second live pipeline on the same
    model gets an engine of its own, and an engine that comes back from an abandoned run (generator dropped mid-page, exception
    between submit and collect) is reset before it is reused.  ``_release_engine`` hands it back.
    """
    from .engine import RecognitionEngine
    vgsl = net.nn
    hs = vgsl.nn
    if vgsl.input[2] <= 0:
        return None
    p = next(vgsl.parameters())
    if not p.is_cuda:
        vgsl.to('cuda')
        p = next(vgsl.parameters())
    dev = p.device.index if p.device.index is not None else torch.cuda.current_device()
    key = (dev, hs.precision, hs._weights_version())
    cache = hs.__dict__.setdefault('_engines', {})
    for k in [k for k in cache if k[1:] != key[1:]]:  # weights updated in place / other arithmetic: those plans are stale;
        for old in cache.pop(k):                      # engines of ANOTHER device stay (one model used on two devices in turn)
            if old.in_use:
                old.stale = True                      # closed by the pipeline that still holds it
            else:
                old.close()
    pool = cache.setdefault(key, [])
    eng = next((e for e in pool if not e.in_use and not e.closed), None)
    if eng is None:
        eng = RecognitionEngine(vgsl, device=dev, max_batch=32, max_width=256, slots=ENGINE_SLOTS, temperature=temperature)
        if hs.precision != key[1]:                    # the engine's plans fell back to exact f32 (new_plan): file it under what it is
            key = (dev, hs.precision, hs._weights_version())
            pool = hs.__dict__.setdefault('_engines', {}).setdefault(key, [])
        eng.stale = False
        pool.append(eng)
        for old in pool[:-4]:                         # evicted: an idle engine is closed now, a held one by its holder
            if old.in_use:
                old.stale = True
            else:
                old.close()
        del pool[:-4]
    else:
        eng.reset()
    eng.temperature = float(temperature)
    eng.in_use = True
    return eng


def _release_engine(eng):
    if eng is None or not eng.in_use:
        return
    eng.in_use = False
    if getattr(eng, 'stale', False):
        eng.close()
    elif not eng.closed:
        eng.reset()


def _decode_lines(codec, batch, olens, probs=None) -> list:
    """DecodedBatch -> LineResult per line.  Single-label codecs take the vectorised table path."""
    n = len(batch.counts)
    lut = codec._single_lut() if hasattr(codec, '_single_lut') else False
    out = []
    if lut is False:
        for i, rec in enumerate(codec.decode_batch(batch) if hasattr(codec, 'decode_batch') else
                                [codec.decode(t) for t in batch.tuples()]):
            out.append(LineResult(''.join(x[0] for x in rec), np.array([x[1] for x in rec], dtype=np.int64),
                                  np.array([x[2] for x in rec], dtype=np.int64),
                                  np.array([x[3] for x in rec], dtype=np.float32), int(olens[i])))
    else:
        labels = np.asarray(batch.labels)
        t = labels.shape[1] if labels.ndim == 2 else 0
        counts = np.minimum(np.maximum(np.asarray(batch.counts), 0), t)
        # entries past a line's count are uninitialised device memory: clip before the table lookup
        cps = np.where((labels >= 0) & (labels < len(lut)), lut[np.clip(labels, 0, len(lut) - 1)], 0).astype('<u4') if t else labels
        # the whole batch at once where nothing is special: which lines hold an undecodable label, and ONE utf-32 decode of all rows
        # (a line's string is a slice of it) -- the per-line mask / all() / tobytes().decode() were half of this function's time
        odd = ((cps == 0) & (np.arange(t)[None, :] < counts[:, None])).any(axis=1) if t else np.zeros(n, dtype=bool)
        whole = None
        if t:
            try:
                whole = np.ascontiguousarray(cps).tobytes().decode('utf-32-le')
            except UnicodeDecodeError:
                whole = None                               # (a code point utf-32 refuses: the per-line decode says which line)
        strict = getattr(codec, 'strict', False)
        starts, ends, confs, counts_l, olens_l = batch.starts, batch.ends, batch.confs, counts.tolist(), [int(v) for v in olens]
        for i in range(n):
            k = counts_l[i]
            if whole is not None and not odd[i]:
                out.append(LineResult(whole[i * t:i * t + k], starts[i, :k], ends[i, :k], confs[i, :k], olens_l[i]))
                continue
            c = cps[i, :k]
            ok = c != 0                                    # undecodable labels are skipped (codec.decode, non-strict)
            if strict and not ok.all():
                codec.decode(batch.tuples()[i])            # raises the reference's exception
            if ok.all():
                s, e, cf = starts[i, :k], ends[i, :k], confs[i, :k]
            else:
                c, s, e, cf = c[ok], starts[i, :k][ok], ends[i, :k][ok], confs[i, :k][ok]
            out.append(LineResult(c.tobytes().decode('utf-32-le'), s, e, cf, olens_l[i]))
    if probs is not None:
        for i, r in enumerate(out):
            r.probs = probs[i, :, :r.out_width].clone()
    return out


def width_buckets(widths, max_lines: int = ENGINE_BATCH, px_budget: int = 0) -> list:
    """
    Length bucketing (BASELINE config 4; the reference's rule is `sorted by width, fixed batch size`, kraken/lib/vgsl/rpred.py:129-131,
    and pads every batch to its widest line).  Returns index arrays into ``widths``, width-sorted, one per device batch.  A batch
    closes at ``max_lines`` lines or -- with ``px_budget`` > 0 -- when one more line would push ``lines x widest line`` (the pixels
    the convolutions compute, padding included) over the budget: narrow lines then travel in larger batches than wide ones and every
    batch costs the device about the same.  Batches are returned WIDEST FIRST.  Results do not depend on the bucketing (masked padding).
    """
    w = np.asarray(widths)
    order = np.argsort(w, kind='stable')
    out, lo = [], 0
    while lo < len(order):
        hi = min(lo + max_lines, len(order))
        if px_budget > 0:
            # widths ascend: the widest line of [lo, k) is w[order[k-1]]; largest k with (k - lo) * w[order[k-1]] <= budget
            cost = (np.arange(1, hi - lo + 1)) * w[order[lo:hi]]
            fit = int(np.searchsorted(cost, px_budget, side='right'))
            hi = lo + max(fit, 1)
        out.append(order[lo:hi])
        lo = hi
    # widest first: a batch's recurrent layers are a T-step latency chain on a quarter of the CUs, so the job's LAST batch is
    # exposed by its own recurrence -- 1.0 ms for the narrowest bucket of config 4 against 2.8 ms for the widest
    return out[::-1]


class LinePipeline:
    """
    Recognises prepared line tensors of ONE recogniser: width-bucketed batches through its RecognitionEngine, several
    batches in flight.  ``submit(items)`` takes ``(key, tensor (C, H, W))`` pairs and returns immediately;
    ``drain(block)`` yields ``(key, LineResult)`` for finished batches.
    """

    def __init__(self, net, temperature: float = 1.0, batch_size: int = ENGINE_BATCH, want_probs: bool = False, pool=None):
        self.net = net
        self.pool = pool              # optional ThreadPoolExecutor: the copies into the pinned staging buffer run on it
        self.batch_size = max(1, int(batch_size))
        self.px_budget = 0            # optional cap on lines x widest line per device batch (width_buckets)
        self.want_probs = want_probs
        self.temperature = float(temperature)
        self.engine = _engine_for(net, temperature) if _fused_ok(net) else None
        self._tickets = deque()       # (ticket, keys)
        self._done = deque()

    @property
    def fused(self) -> bool:
        return self.engine is not None

    def close(self):
        """Hands the engine back (abandoning batches still in flight); the pipeline cannot be used afterwards."""
        eng, self.engine = self.__dict__.get('engine'), None
        self.__dict__.get('_tickets', deque()).clear()
        try:
            _release_engine(eng)
        except Exception:      # interpreter shutdown, device already gone
            pass

    def __del__(self):
        self.close()

    def _parts(self, items: list, width_of) -> list:
        """The device batches of one submission: width-sorted buckets (`width_buckets`), widest first."""
        return [[items[i] for i in idx] for idx in width_buckets([width_of(it) for it in items], self.batch_size, self.px_budget)]

    def submit(self, items: list):
        """items: [(key, tensor)]; all tensors share (C, H).  Width-sorted so that a batch pads to similar widths."""
        if not items:
            return
        if self.engine is None:
            self._run_sync(items)
            return
        for part in self._parts(items, lambda kt: kt[1].shape[2]):
            while self.engine.free_slots() == 0:
                self._collect_one()
            widths = [t.shape[2] for _, t in part]
            x = self.engine.stage(len(part), max(widths))

            def stage(lo_hi, x=x, part=part, widths=widths):
                for i in range(*lo_hi):
                    t = part[i][1]
                    x[i, :, :, :widths[i]] = t.numpy() if isinstance(t, torch.Tensor) else t
                    x[i, :, :, widths[i]:] = 0.0                 # the staging array is reused: clear only the padding
            n = len(part)
            if self.pool is not None and n >= 32:                # numpy copies release the GIL: ~5 GB/s per thread
                step = -(-n // 8)
                list(self.pool.map(stage, [(a, min(a + step, n)) for a in range(0, n, step)]))
            else:
                stage((0, n))
            ticket = self.engine.submit_staged(np.asarray(widths, dtype=np.int32), want_probs=self.want_probs)
            self._tickets.append((ticket, [k for k, _ in part]))

    def submit_boxes(self, page_dev, items: list, pad: int):
        """items: [(key, (x0, y0, x1, y1, resized width))]: crops of an uploaded page, prepared on the device."""
        for part in self._parts(items, lambda kb: kb[1][4]):
            while self.engine.free_slots() == 0:
                self._collect_one()
            ticket = self.engine.submit_boxes(page_dev, np.asarray([b for _, b in part], dtype=np.int32), pad,
                                              want_probs=self.want_probs)
            self._tickets.append((ticket, [k for k, _ in part]))

    def submit_crops(self, items: list, pad: int):
        """items: [(key, uint8 array (h, w[, 3]))]: line images cut out on the host, prepared (resize, pad, invert) on the device."""
        h = self.engine.in_height
        for part in self._parts(items, lambda ka: int(ka[1].shape[1] * h / max(ka[1].shape[0], 1))):
            while self.engine.free_slots() == 0:
                self._collect_one()
            ticket = self.engine.submit_crops([a for _, a in part], pad, want_probs=self.want_probs, pool=self.pool)
            self._tickets.append((ticket, [k for k, _ in part]))

    def submit_dewarp(self, items: list, pad: int):
        """
        items: [(key, uint8 array (h, w))] of ONE batch: 1-channel bbox lines, dewarped (CenterNormalizer), padded and inverted on
        the device.  Returns ({key: network input width}, keys that must take the host transform instead).
        """
        return self.dewarp_finish(items, self.dewarp_begin(items), pad)

    def dewarp_begin(self, items: list, ahead: int = 0, page=None, top: int = 0):
        """Upload + measurement of one dewarp batch, not waited for (engine.measure_dewarp_begin); ``ahead=1``: a batch that was
        begun before is still to be finished (``dewarp_finish``) -- nothing else may be submitted in between.  With ``page`` (device
        tensor whose first row is row ``top`` of the image) the items are _PageCrop boxes read from it."""
        while self.engine.free_slots() < 1 + ahead:
            self._collect_one()
        if page is not None:
            boxes = np.asarray([(a.box[0], a.box[1] - top, a.box[2], a.box[3] - top) for _, a in items], dtype=np.int64)
            return self.engine.measure_dewarp_begin(boxes, pool=self.pool, ahead=ahead, page=page)
        return self.engine.measure_dewarp_begin([a for _, a in items], pool=self.pool, ahead=ahead)

    def dewarp_finish(self, items: list, measured, pad: int):
        r, ok, ink = measured.result()
        use = ok & ink
        h = self.engine.in_height
        widths, host, keys = {}, set(), []
        if bool(use.all()):
            # the common batch -- every line has ink and a full band -- without a per-line branch (same double arithmetic as below)
            keys = [k for k, _ in items]
            w = np.fromiter((a.shape[1] for _, a in items), dtype=np.float64, count=len(items))
            wd = ((h * 1.0 / (2 * r.astype(np.int64))) * w).astype(np.int64) + 2 * pad
            widths = dict(zip(keys, wd.tolist()))
            items = ()
        for (k, a), rr, o, i in zip(items, r, ok, ink):
            # a UNIFORM crop (ink False: max == min) is the reference's flat line only when it is white: any other value becomes a
            # non-flat tensor once the white padding is added (kraken/rpred.py:221 tests the PADDED tensor) and is recognised --
            # solid black crops do occur on binarised pages.  Those take the reference's own transform on the host.
            to_host = (i and not o) or (not i and a.size > 0 and (a.first() if isinstance(a, _PageCrop) else int(a.flat[0])) != 255)
            if to_host:
                host.add(k)                                  # (band outside the padded stack: the reference's own code decides)
            elif o and i:
                widths[k] = int(h * 1.0 / (2 * int(rr)) * a.shape[1]) + 2 * pad
            keys.append(None if to_host else k)
        ticket = self.engine.submit_dewarped(r, use, pad, want_probs=self.want_probs)
        self._tickets.append((ticket, keys))
        return widths, host

    def _collect_one(self):
        ticket, keys = self._tickets.popleft()
        batch, olens = self.engine.collect(ticket)
        flags = self.engine.last_flags
        probs = self.engine.last_probs() if self.want_probs else None
        for i, (k, r) in enumerate(zip(keys, _decode_lines(self.net.codec, batch, olens, probs))):
            if k is None:
                continue                                     # a slot of the batch whose line took the host transform instead
            # a line without a single non-white pixel is the reference's "flat line": empty record (kraken/rpred.py:221)
            self._done.append((k, None if flags is not None and not flags[i] else r))

    def _run_sync(self, items: list):
        """Recognisers the engine does not take (custom decoder, variable height, foreign objects): per line, like the reference."""
        for k, t in items:
            preds = self.net.predict(t.unsqueeze(0))[0]
            outw = self.net.outputs.shape[2]
            probs = None
            if self.want_probs:
                o = self.net.outputs
                probs = torch.as_tensor(o)[0, :, :outw].clone()
            self._done.append((k, LineResult(''.join(x[0] for x in preds), np.array([x[1] for x in preds], dtype=np.int64),
                                             np.array([x[2] for x in preds], dtype=np.int64),
                                             np.array([x[3] for x in preds], dtype=np.float32), int(outw), probs)))

    def pending(self) -> int:
        return len(self._tickets)

    def drain(self, block: bool = False):
        """Yields finished (key, LineResult); with ``block`` waits for the oldest batch in flight first."""
        if block and self._tickets:
            self._collect_one()
        while self._done:
            yield self._done.popleft()


# ------------------------------------------------------------------------------------------------------- host side
class _RecognitionRun:
    """Shared machinery of both front ends: chunked preparation, per-recogniser pipelines, record assembly."""

    def _init_run(self, im, bounds, pad, bidi_reordering, temperature_of, want_probs=False, return_image=False,
                  workers: int = PREP_THREADS, engine_batch: Optional[int] = None):
        self.im, self.bounds, self.pad = im, bounds, pad
        if hasattr(im, 'load'):
            im.load()                                  # worker threads crop concurrently: decode the page once, up front
        self.bidi_reordering = bidi_reordering
        self.len = len(bounds.lines)
        self._valid_norm = bounds.type != 'baselines'
        self._record_cls = BBoxOCRRecord if bounds.type != 'baselines' else BaselineOCRRecord
        self._results: dict[int, object] = {}
        self._cursor = 0          # next line index to hand out
        self._prepared = 0        # lines already extracted / queued
        self._pipes: dict[int, LinePipeline] = {}
        self._pending: dict[int, _Pending] = {}
        self._temperature_of = temperature_of
        self._want_probs = want_probs
        self._return_image = return_image
        self._workers = max(1, int(workers or 1))
        self._pool = ThreadPoolExecutor(max_workers=workers) if workers and workers > 1 else None
        # device batch: large enough to fill the chip, small enough that a page still splits into a few batches whose
        # preparation overlaps the device work of their predecessors
        b = engine_batch or ENGINE_BATCH
        self._batch = int(min(b, max(32, math.ceil(self.len / (2 * ENGINE_SLOTS)))))
        self._chunk = self._batch * ENGINE_SLOTS
        self._pages: dict = {}     # PIL mode -> the page as a device tensor (device-side line preparation)
        self._gray = None          # the page as one uint8 'L' array, converted band by band (bbox lines of a dewarping model)
        self._gray_done: set = set()
        # Pillow's row table of the page (None: np.asarray path).  '1' / 'L' pages are one byte per pixel, 'RGB' / 'RGBX' / 'RGBA'
        # four: R, G, B, X -- uploaded as they are, the kernels take the pixel stride (and Pillow's 'L' conversion for 1-channel models)
        self._rows = pilmem.image_rows(im) if (PAGE_ROWS and DEVICE_PREP and hasattr(im, 'getpixel')) else None

    # -- device-side preparation (krk_prep_lines): rectangular crops of a fixed-height model, no dewarp ------------
    def _transform_on_device_ok(self, net, ts) -> bool:
        """The transform is crop -> fixed-height LANCZOS resize -> white padding -> scale -> invert: what the device kernels do.
        (Asked once per LINE: the answer for a (recogniser, transform) pair is kept for the run -- walking the network's layer list
        2048 times was 10 % of a warm page's host time.)"""
        memo = self.__dict__.setdefault('_device_ok_memo', {})
        key = (id(net), id(ts), DEVICE_PREP, DEVICE_DEWARP)
        ok = memo.get(key)
        if ok is None:
            ok = memo[key] = self._transform_on_device_ok_uncached(net, ts)
        return ok

    def _transform_on_device_ok_uncached(self, net, ts) -> bool:
        if not DEVICE_PREP:
            return self._host_path('device preparation is switched off (rpred.DEVICE_PREP)')
        pad = ts.pad
        if not (isinstance(pad, (tuple, list)) and len(pad) == 2 and int(pad[0]) > 0 and int(pad[1]) == 0):
            return self._host_path('padding other than (n > 0, 0)')
        if ts._center_norm and (ts._mode != 'L' or not DEVICE_DEWARP):
            return self._host_path('the CenterNormalizer dewarp on the device is switched off (rpred.DEVICE_DEWARP)')
        if ts._perm != (0, 1, 2) or ts._mode not in ('L', 'RGB') or ts._scale[1] != 0 or not 1 <= ts._scale[0] <= 128:
            return self._host_path('input spec outside the kernel\'s range (legacy height-in-channels layout, height > 128, fixed width)')
        if not (_fused_ok(net) and net.nn.input[2] > 0):
            return self._host_path('recogniser without the fused engine (custom decoder, variable height)')
        return True                                        # (the engine itself is created on the main thread, _advance)

    def _host_path(self, why: str) -> bool:
        """Says ONCE per run and reason that lines are prepared with PIL on the host: a 30x slower path the user should see."""
        seen = self.__dict__.setdefault('_host_reasons', set())
        if why not in seen:
            seen.add(why)
            logger.info(f'line images are prepared on the host, not on the device: {why}')
        return False

    def _device_prep_ok(self, net, ts) -> bool:
        """Rectangular crops straight from the uploaded page (krk_prep_lines): bbox segmentations of horizontal text."""
        if self.bounds.type == 'baselines' or not self.bounds.text_direction.startswith('horizontal') or ts._center_norm:
            return False                                   # (dewarped lines are cut out on the host and go through _crop_for_device)
        return self._transform_on_device_ok(net, ts)

    # -- bbox lines of a dewarping (1-channel) model: cut out of ONE grayscale copy of the page ----------------------------
    GRAY_ROWS = 256            # rows converted per work item

    def _gray_wanted(self) -> bool:
        """bbox segmentation of horizontal text, the reference's own extractor, device dewarp on: the per-line `im.crop(box)` +
        `convert('L')` + array copy of the reference (~50 us of interpreter time per line, under the GIL) becomes a numpy view."""
        if self.bounds.type == 'baselines' or not self.bounds.text_direction.startswith('horizontal'):
            return False
        if extract_polygons is not _EXTRACT_POLYGONS or not (DEVICE_PREP and DEVICE_DEWARP):
            return False
        if self._rows is not None:
            return False                               # the lines are read from the uploaded page itself (_dewarp_box_on_page)
        ts = getattr(self, 'ts', None)
        tss = list(ts.values()) if isinstance(ts, dict) else [ts]
        return any(getattr(t, '_center_norm', False) and getattr(t, '_mode', '') == 'L' for t in tss)

    def _dewarps(self) -> bool:
        """Does this run put lines through the CenterNormalizer dewarp (1-channel models on a bbox segmentation)?"""
        if self.bounds.type == 'baselines':
            return False
        ts = getattr(self, 'ts', None)
        tss = list(ts.values()) if isinstance(ts, dict) else [ts]
        return any(getattr(t, '_center_norm', False) for t in tss)

    def _ensure_gray(self, idxs):
        """Main thread, before a chunk is prepared: converts the page rows the chunk's boxes touch (pool: 256-row pieces)."""
        W, H = self.im.size
        rows = [ln.bbox for ln in (self.bounds.lines[i] for i in idxs) if getattr(ln, 'bbox', None) is not None]
        rows = [(b[1], b[3]) for b in rows if 0 <= b[1] < b[3] <= H]
        if not rows:
            return
        if self._gray is None:
            self._gray = np.empty((H, W), dtype=np.uint8)
        step = self.GRAY_ROWS
        need = sorted({k for y0, y1 in rows for k in range(y0 // step, (y1 - 1) // step + 1)} - self._gray_done)

        def part(k):
            ya, yb = k * step, min((k + 1) * step, H)
            sub = self.im.crop((0, ya, W, yb))
            self._gray[ya:yb] = np.asarray(sub if sub.mode == 'L' else sub.convert('L'))
        if self._pool and len(need) > 1:
            list(self._pool.map(part, need))
        else:
            for k in need:
                part(k)
        self._gray_done.update(need)

    def _dewarp_crop_from_page(self, idx: int, line, tag: str, net, ts, want_image: bool = False):
        """
        The common case of a dewarped bbox line -- box inside the page, 2 <= height <= 192 -- as a view of the grayscale page
        (pixel for pixel `im.crop(box).convert('L')`: the conversion is point-wise).  None: the general path decides (boxes
        touching the page border are padded by PIL, invalid ones give the reference's empty records).
        """
        # (asked once per line: what does not depend on the line is worked out once per run and switch setting)
        memo = self.__dict__.get('_on_page_memo')
        if memo is None or memo[0] != (DEVICE_DEWARP, extract_polygons):
            on_page = self._rows is not None and DEVICE_DEWARP and self.bounds.type != 'baselines' and \
                self.bounds.text_direction.startswith('horizontal') and extract_polygons is _EXTRACT_POLYGONS
            memo = self._on_page_memo = ((DEVICE_DEWARP, extract_polygons), on_page, self.im.size)
        on_page, (W, H) = memo[1], memo[2]
        if (self._gray is None and not on_page) or not ts._center_norm or not self._transform_on_device_ok(net, ts):
            return None
        box = line.bbox
        if box is None or len(box) != 4:
            return None
        x0, y0, x1, y1 = box
        if not (type(x0) is int and type(y0) is int and type(x1) is int and type(y1) is int):
            x0, y0, x1, y1 = (int(v) for v in box)
            if list(box) != [x0, y0, x1, y1]:
                return None
        if not (0 <= x0 < x1 <= W and 0 <= y0 < y1 <= H):
            return None
        w, h = x1 - x0, y1 - y0
        step = self.GRAY_ROWS
        if h < 2 or h > 192 or w > 16384:
            return None
        image = self.im.crop((x0, y0, x1, y1)) if want_image else None
        if on_page:
            # a crop of the page the DEVICE holds (krk_dewarp_measure_page): no pixel is touched on the host
            return _Pending(idx, line, tag, net, None, (w, h), image=image, width=0, mode='dewarp', box=(x0, y0, x1, y1, 0))
        if any(k not in self._gray_done for k in range(y0 // step, (y1 - 1) // step + 1)):
            return None
        return _Pending(idx, line, tag, net, None, (w, h), image=image, width=0, mode='dewarp', crop=self._gray[y0:y1, x0:x1])

    def _crop_for_device(self, idx: int, line, tag: str, net, ts, box, box_size, want_image: bool = False):
        """
        A line image that was cut out on the host (baseline / polygon extraction, vertical text ...) as a uint8 array for
        krk_prep_crops -- or None: the reference's host transform takes it.  The flat-line rule is applied by the kernel's flags.
        """
        if not self._transform_on_device_ok(net, ts):
            return None
        w, h = box.size
        out_h = ts._scale[0]
        if ts._center_norm:
            # the dewarp's output width depends on the measured spread: known after krk_dewarp_measure (LinePipeline.submit_dewarp)
            if h < 2 or h > 192 or w > 16384:
                self._host_path('line outside the device dewarp\'s range (height < 2 or > 192, wider than 16384)')
                return None
            arr = np.asarray(box if box.mode == 'L' else box.convert('L'), dtype=np.uint8)
            return _Pending(idx, line, tag, net, None, box_size, image=box if want_image else None, width=0, mode='dewarp', crop=arr)
        ow = int(w * out_h / h)
        if ow <= 0:
            return None                              # Image.resize raises on an empty target: the host path reports it
        taps = lambda n_in, n_out: math.ceil(3.0 * max(1.0, n_in / n_out)) * 2 + 1      # noqa: E731
        channels = 3 if ts._mode == 'RGB' else 1
        max_rows = min(512, (160 * 1024 - (out_h + 64) * 99 * 4) // (channels * 64) - 2)      # (the kernel's LDS budget: _prepare_on_device)
        if h > max_rows or taps(h, out_h) > 96 or taps(w, ow) > 96:
            self._host_path(f'crop geometry outside the kernel\'s range (taller than {max_rows} px or scaled by more than 15)')
            return None
        im = box if box.mode == ts._mode else box.convert(ts._mode)
        arr = np.asarray(im, dtype=np.uint8)
        return _Pending(idx, line, tag, net, None, box_size, image=box if want_image else None,
                        width=ow + 2 * int(ts.pad[0]), mode=ts._mode, crop=arr)

    def _strip_on_device(self, net, mode: str, boxes):
        """
        Device tensor holding the page rows the crops `boxes` touch, and its first row.  A chunk of lines usually spans a
        band of the page: converting / uploading that band per chunk overlaps the device work of the previous chunk; when
        the band is most of the page (lines in no particular order) the whole page goes up once and is reused.
        """
        W, H = self.im.size
        y0 = max(min(b[1] for b in boxes), 0)
        y1 = min(max(b[3] for b in boxes), H)
        whole = mode in self._pages or (y1 - y0) > 0.6 * H
        if mode in self._pages:
            return self._pages[mode], 0
        if whole:
            y0, y1 = 0, H
        eng = self._pipe(net).engine
        rows = self._rows
        if rows is not None and (rows.pixelsize == 1 if mode == 'L' and self.im.mode in ('1', 'L') else
                                 rows.pixelsize == 4 and self.im.mode in ('RGB', 'RGBX', 'RGBA')):
            # the band's rows as they lie in Pillow's memory.  Colour pages travel as R, G, B, X; the kernels take the pixel stride and,
            # for a 1-channel model, Pillow's 'L' conversion (krk_prep_lines_fmt / krk_dewarp_*_page).
            # The band goes up straight from Pillow's blocks, one pageable copy per contiguous run of rows (engine.upload_rows: on these
            # hosts as fast as memmove + pinned DMA, and a fresh process pays no first-use cost of a pinned page buffer)
            dev = eng.upload_rows(rows, y0, y1, pins=self.__dict__.setdefault('_pins', {}) if PIN_PAGES else None)
            if whole:
                self._pages[mode] = dev
            return dev, y0
        # PIL -> numpy runs at ~1 GB/s per thread: the band is converted in 256-row pieces by the worker pool, every piece
        # straight into the engine's pinned upload buffer
        buf = eng.page_buffer((y1 - y0, W) if mode == 'L' else (y1 - y0, W, 3))
        step = 256

        def part(ya):
            yb = min(ya + step, y1)
            sub = self.im.crop((0, ya, W, yb))
            buf[ya - y0:yb - y0] = np.asarray(sub if sub.mode == mode else sub.convert(mode))
        cuts = range(y0, y1, step)
        if self._pool and len(cuts) > 1:
            list(self._pool.map(part, cuts))
        else:
            for c in cuts:
                part(c)
        dev = eng.upload_page_buffer()
        if whole:
            self._pages[mode] = dev
        return dev, y0

    def _prepare_on_device(self, idx: int, line, tag: str, net, ts, want_image: bool = False):
        """
        The bbox branch of extract_polygons + the fixed-height transform as a crop descriptor for krk_prep_lines.
        Returns an ocr_record (the reference's empty-record cases), a _Pending, or None: take the host path.
        """
        box = list(line.bbox)
        W, H = self.im.size
        # the reference's bounds test (lexicographic list comparison, kraken/lib/segmentation.py:1636-1640)
        if box < [0, 0, 0, 0] or box[::2] >= [W, W] or box[1::2] >= [H, H] or box[2] < box[0] or box[3] < box[1]:
            logger.warning(f'Extracting line failed: bbox {box} is outside of image bounds {self.im.size}')
            return self._empty(line, [])
        w, h = box[2] - box[0], box[3] - box[1]
        if w == 0 or h == 0:
            logger.warning(f'{line} with zero dimension. Emitting empty record.')
            return self._empty(line)
        out_h = ts._scale[0]
        ow = int(w * out_h / h)
        if ow <= 0:                                  # Image.resize raises on an empty target: "conversion failed"
            logger.warning(f'Conversion of line {line} failed. Emitting empty record..')
            return self._empty(line)
        taps = lambda n_in, n_out: math.ceil(3.0 * max(1.0, n_in / n_out)) * 2 + 1      # noqa: E731
        # the kernel keeps the crop's rows of a 64-column tile and its filter tables in LDS (prep_lines.hip; since round 6 its table rows are as long as the scale needs -- this bound still assumes the longest, (out_h + 64) x 99 ints +
        # channels x (rows + 2) x 64 bytes <= 160 KB): 512 rows at height 48, 456 for a colour line of a 120-row model
        channels = 3 if ts._mode == 'RGB' else 1
        max_rows = min(512, (160 * 1024 - (out_h + 64) * 99 * 4) // (channels * 64) - 2)
        if h > max_rows or taps(h, out_h) > 96 or taps(w, ow) > 96:
            return None                              # outside the kernel's range
        pad = int(ts.pad[0])
        return _Pending(idx, line, tag, net, None, (w, h), image=self.im.crop(box) if want_image else None,
                        width=ow + 2 * pad, box=(int(box[0]), int(box[1]), int(box[2]), int(box[3]), ow), mode=ts._mode)

    # -- record assembly ----------------------------------------------------------------------------------
    def _scale(self, val, net_scale, in_scale, max_val):
        return int(round(min(max(((val * net_scale) - self.pad) * in_scale, 0), max_val - 1)))

    def _scale_all(self, vals, net_scale, in_scale, max_val):
        """Vectorised `_scale_val` (kraken/rpred.py:329-330): np.rint rounds half to even exactly like Python's round()."""
        v = (np.asarray(vals, dtype=np.float64) * net_scale - self.pad) * in_scale
        return np.rint(np.minimum(np.maximum(v, 0), max_val - 1)).astype(np.int64)

    def _cuts(self, p: _Pending, r: LineResult):
        net_scale = p.width / r.out_width
        in_scale = p.box_size[0] / (p.width - 2 * self.pad)
        n = len(r.starts)
        if self._valid_norm:
            coords = p.line.bbox
            if self.bounds.text_direction.startswith('horizontal'):
                lo = coords[0] + self._scale_all(r.starts, net_scale, in_scale, p.box_size[0])
                hi = coords[0] + self._scale_all(r.ends, net_scale, in_scale, p.box_size[0])
                pos = np.empty((n, 4, 2), dtype=np.int64)          # [[lo, y0], [lo, y1], [hi, y1], [hi, y0]] per code point
                pos[:, 0, 0] = pos[:, 1, 0] = lo
                pos[:, 2, 0] = pos[:, 3, 0] = hi
                pos[:, 0, 1] = pos[:, 3, 1] = coords[1]
                pos[:, 1, 1] = pos[:, 2, 1] = coords[3]
            else:
                lo = coords[1] + self._scale_all(r.starts, net_scale, in_scale, p.box_size[1])
                hi = coords[1] + self._scale_all(r.ends, net_scale, in_scale, p.box_size[1])
                pos = np.empty((n, 4, 2), dtype=np.int64)          # [[x0, lo], [x1, lo], [x1, hi], [x0, hi]] per code point
                pos[:, 0, 0] = pos[:, 3, 0] = coords[0]
                pos[:, 1, 0] = pos[:, 2, 0] = coords[2]
                pos[:, 0, 1] = pos[:, 1, 1] = lo
                pos[:, 2, 1] = pos[:, 3, 1] = hi
            return pos.tolist() if n else []
        lo = self._scale_all(r.starts, net_scale, in_scale, p.box_size[0])
        hi = self._scale_all(r.ends, net_scale, in_scale, p.box_size[0])
        return np.stack([lo, hi], 1).tolist() if n else []

    def _order(self, rec):
        if self.bidi_reordering:
            return rec.logical_order(base_dir=self.bidi_reordering if self.bidi_reordering in ('L', 'R') else None)
        return rec.display_order(None)

    def _make_record(self, p: _Pending, r: LineResult):
        n = len(r.starts)
        if self.bidi_reordering or n == 0:      # reordering walks the lists right away: build them
            return self._order(self._record_cls(r.text, self._cuts(p, r), r.confs.tolist(), p.line))
        return self._order(self._record_cls(r.text, LazyList(partial(self._cuts, p, r), n), LazyList(r.confs.tolist, n), p.line))

    def _empty(self, line, cuts=()):
        return self._record_cls('', cuts, cuts, line)

    # -- pipeline -----------------------------------------------------------------------------------------
    def _pipe(self, net) -> LinePipeline:
        pipe = self._pipes.get(id(net))
        if pipe is None:
            pipe = LinePipeline(net, self._temperature_of(net), self._batch, want_probs=self._want_probs, pool=self._pool)
            self._pipes[id(net)] = pipe
        return pipe

    def _absorb(self, block_pipe: Optional[LinePipeline] = None):
        for pipe in self._pipes.values():
            for idx, r in pipe.drain(block=pipe is block_pipe):
                p = self._pending.pop(idx)
                self._results[idx] = self._make_record(p, r) if r is not None else self._empty(p.line)

    def _advance(self):
        """Prepares + submits the next chunk (the device keeps working on earlier ones meanwhile), or waits for results."""
        if self._prepared < self.len:
            # the first chunk is ONE batch: the device starts after a third of the page band has been copied and uploaded (a page is
            # a handful of batches: its first result waits for all the host work in front of the first submission)
            # (not for dewarped lines: their batches are pipelined in pairs inside a chunk, _submit_dewarp)
            # Lines the DEVICE cuts out of the uploaded page are a crop descriptor each: their chunks are ONE batch too -- prepared,
            # submitted, and the oldest batch in flight decoded, batch by batch (the dewarp's two halves are pipelined across calls)
            start = self._prepared
            if self._gray_wanted():
                self._ensure_gray(range(start, min(start + self._chunk, self.len)))
            first = self._prepare(start)
            light = isinstance(first, _Pending) and first.tensor is None and first.box is not None and first.image is None
            chunk = self._batch if light or (start == 0 and not self._dewarps()) else self._chunk
            idxs = range(start, min(start + chunk, self.len))
            # Lines the DEVICE cuts out of the uploaded page are a crop descriptor each (a few microseconds of interpreter time, all of it
            # under the GIL): the worker pool only adds hand-offs there -- a warm 2048-line page took 36.7 ms with six workers and 31.4 ms
            # on the main thread.  The pool is for lines whose pixels the host touches (PIL crops, conversions, resizes).
            if self._pool and len(idxs) > 1 and not light:
                # a future per line costs more than a crop descriptor does: hand the pool a few slices per worker instead
                rest = idxs[1:]
                k = max(1, len(rest) // (4 * self._workers))
                parts = self._pool.map(lambda lo: [self._prepare(i) for i in rest[lo:lo + k]], range(0, len(rest), k))
                items = [first] + [it for part in parts for it in part]
            else:
                items = [first] + [self._prepare(i) for i in idxs[1:]]
            self._prepared = idxs[-1] + 1
            groups: dict = {}
            for i, item in zip(idxs, items):
                if isinstance(item, _Pending):
                    self._pending[i] = item
                    shape = (('crop', item.mode) if (item.crop is not None or item.mode == 'dewarp') else ('dev', item.mode)) \
                        if item.tensor is None else tuple(item.tensor.shape[:2])
                    groups.setdefault((id(item.net), shape), []).append(item)
                else:
                    self._results[i] = item
            for (_, shape), group in groups.items():     # one (recogniser, line height) per batch: heights are never padded
                if shape[0] == 'crop' and shape[1] == 'dewarp':
                    self._submit_dewarp(group)
                    continue
                self._dewarp_flush()                     # (nothing may be submitted between the two halves of a dewarp batch)
                if shape[0] == 'crop':
                    self._pipe(group[0].net).submit_crops([(p.idx, p.crop) for p in group], self.pad)
                elif shape[0] == 'dev':
                    net = group[0].net
                    page_dev, top = self._strip_on_device(net, shape[1], [p.box for p in group])
                    self._pipe(net).submit_boxes(page_dev, [(p.idx, (p.box[0], p.box[1] - top, p.box[2], p.box[3] - top, p.box[4]))
                                                            for p in group], self.pad)
                else:
                    self._pipe(group[0].net).submit([(p.idx, p.tensor) for p in group])
            if self._prepared >= self.len:
                self._dewarp_flush()
            self._absorb()
            return
        self._dewarp_flush()
        busiest = max(self._pipes.values(), key=lambda p: p.pending(), default=None)
        self._absorb(block_pipe=busiest)

    def _submit_dewarp(self, group: list):
        """Lines of a 1-channel model on a bbox segmentation: dewarp on the device; the few lines whose band does not fit the
        reference's padded stack (or that are flat) take the reference's host transform, which also decides their record."""
        on_page = [p for p in group if p.crop is None]
        if on_page and len(on_page) < len(group):       # lines cut out on the host (boxes touching the page border ...) and boxes of
            self._submit_dewarp([p for p in group if p.crop is not None])      # the uploaded page: one kind per batch
            group = on_page
        net = group[0].net
        pipe = self._pipe(net)
        ts = self.ts[group[0].tag] if isinstance(getattr(self, 'ts', None), (dict, defaultdict)) else self.ts
        page, top = None, 0
        if on_page:
            W, H = self.im.size
            rows = min(max(p.box[3] for p in group), H) - max(min(p.box[1] for p in group), 0)
            rows = H if rows > 0.6 * H else rows                  # (_strip_on_device uploads the whole page then)
            if rows * W * (self._rows.pixelsize if self._rows is not None else 1) >= 1 << 32:
                # crop offsets into the uploaded band are 32-bit (krk_dewarp_measure_page): a band of 4 GiB or more travels as
                # packed crops cut out on the host instead
                for p in group:
                    box = self.im.crop(p.box[:4])
                    p.crop = np.asarray(box if box.mode == 'L' else box.convert('L'), dtype=np.uint8)
            else:
                page, top = self._strip_on_device(net, 'L', [p.box for p in group])
                for p in group:
                    p.crop = _PageCrop(p.box, partial(self._page_gray_at, p.box[0], p.box[1]))
        # a dewarp batch is bounded by lines AND by pixels: krk_dewarp_measure keeps 3 fp64 planes per pixel of scratch behind
        # 32-bit offsets (a batch of 256 lines at the per-line maximum of 192 x 16384 would ask for 19 GB)
        parts, cur, px = [], [], 0
        for p in group:
            n = int(p.crop.shape[0]) * int(p.crop.shape[1])
            if cur and (len(cur) >= pipe.batch_size or px + n > DEWARP_BATCH_PIXELS):
                parts.append(cur)
                cur, px = [], 0
            cur.append(p)
            px += n
        if cur:
            parts.append(cur)
        # software pipeline over the batches, ACROSS chunks: the measurement of batch k + 1 is enqueued (next slot but one) before the
        # host waits for batch k's -- the read-back between krk_dewarp_measure and krk_dewarp_apply costs the host no idle time -- and
        # the last batch of this call stays begun until the next call (or the end of the page, or a submission of another kind:
        # _dewarp_flush) finishes it.  Lines that take the host transform are submitted when no batch is begun (no other submission
        # may come between a batch's two halves).
        begun = self.__dict__.get('_dw_begun')
        if begun is not None and begun[0] is not pipe:
            self._dewarp_flush()
        two = len(pipe.engine.slots) >= 2
        for part in parts:
            items = [(p.idx, p.crop) for p in part]
            ahead = 1 if (two and self.__dict__.get('_dw_begun') is not None) else 0
            if not two:
                self._dewarp_flush()                                   # a one-slot engine: both halves of a batch back to back
            handle = pipe.dewarp_begin(items, ahead=ahead, page=page, top=top)
            self._dewarp_finish_begun()
            self._dw_begun = (pipe, part, items, handle, ts)
        if not two or self.__dict__.get('_dw_to_host'):
            self._dewarp_flush()                                       # (lines for the host transform do not wait for the end of the page)

    def _dewarp_finish_begun(self):
        """Second half (apply + recognition) of the batch whose measurement is in flight."""
        begun = self.__dict__.get('_dw_begun')
        if begun is None:
            return
        self._dw_begun = None
        pipe, part, items, handle, ts = begun
        widths, host = pipe.dewarp_finish(items, handle, self.pad)
        for p in part:
            if p.idx in widths:
                p.width = widths[p.idx]
        if host:
            self.__dict__.setdefault('_dw_to_host', []).extend((p, pipe, ts) for p in part if p.idx in host)

    def _dewarp_flush(self):
        """Finishes the begun batch and submits the lines that took the host transform: before a submission of another kind, at the
        end of the page."""
        self._dewarp_finish_begun()
        to_host, self._dw_to_host = self.__dict__.get('_dw_to_host') or [], []
        for p, pipe, ts in to_host:
            del self._pending[p.idx]
            if isinstance(p.crop, _PageCrop):
                box = self.im.crop(p.crop.box[:4])
                box = box if box.mode == 'L' else box.convert('L')
            else:
                box = Image.fromarray(np.ascontiguousarray(p.crop), 'L')
            try:
                t = ts(box)
            except Exception:
                logger.warning(f'Conversion of line {p.line} failed. Emitting empty record..')
                self._results[p.idx] = self._empty(p.line)
                continue
            if t.max() == t.min():
                logger.warning('Empty run. Emitting empty record.')
                self._results[p.idx] = self._empty(p.line)
                continue
            q = dataclasses.replace(p, tensor=t, crop=None, mode='', width=t.shape[2])
            self._pending[p.idx] = q
            pipe.submit([(q.idx, q.tensor)])

    def _page_gray_at(self, x: int, y: int) -> int:
        """Pillow's 'L' value of one page pixel (libImaging/Convert.c rgb2l for colour pages)."""
        v = self.im.getpixel((x, y))
        if isinstance(v, (tuple, list)):
            return (v[0] * 19595 + v[1] * 38470 + v[2] * 7471 + 0x8000) >> 16 if len(v) >= 3 else int(v[0])
        return int(v)

    def _fill(self):
        while self._cursor not in self._results:
            if self._prepared >= self.len and not self._pending:
                raise RuntimeError(f'line {self._cursor} was never recognised')   # cannot happen: every line yields a record
            self._advance()

    def close(self):
        """Ends the run: engines go back to their models (batches still in flight are abandoned), the thread pool stops."""
        self._dw_begun, self._dw_to_host = None, []
        for pipe in self.__dict__.get('_pipes', {}).values():
            pipe.close()                                   # (engine.reset() waits for the streams: no copy out of a pinned block is in flight)
        pins = self.__dict__.get('_pins')
        if pins:
            torch.cuda.synchronize()
            from .engine import RecognitionEngine
            RecognitionEngine.unpin_blocks(pins)
        self.__dict__.get('_pipes', {}).clear()
        pool, self._pool = self.__dict__.get('_pool'), None
        if pool:
            pool.shutdown(wait=False)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _next_record(self):
        if self._cursor >= self.len:
            self.close()
            raise StopIteration
        try:
            self._fill()
        except BaseException:
            self.close()          # a failed batch must not leave its engine's slots busy for the next page (engines are cached)
            raise
        rec = self._results.pop(self._cursor)
        self._cursor += 1
        return rec


class mm_rpred(_RecognitionRun):
    """Multi-model recogniser iterator: ``for record in mm_rpred(nets, im, bounds): ...``"""

    def __init__(self, nets, im, bounds, pad: int = 16, bidi_reordering: Union[bool, str] = True,
                 tags_ignore: Optional[list] = None, no_legacy_polygons: bool = False, batch_size: Optional[int] = None,
                 num_line_workers: int = PREP_THREADS):
        warnings.warn('`rpred.mm_rpred` is deprecated and will be removed with kraken 8. Use `RecognitionTaskModel` instead.',
                      DeprecationWarning)
        seg_types = set(r.seg_type for r in nets.values())
        default = None
        if isinstance(nets, defaultdict) and nets.default_factory:
            default = nets.default_factory()
            seg_types.add(default.seg_type)
        self._default = default
        tags_ignore = tags_ignore or []
        self.have_tags = bool(bounds.script_detection)
        if bounds.type not in seg_types or len(seg_types) > 1:
            logger.warning(f'Recognizers with segmentation types {seg_types} will be applied to segmentation of type '
                           f'{bounds.type}. This will likely result in severely degraded performace')
        modes = set(r.nn.one_channel_mode for r in nets.values())
        if '1' in modes and len(modes) > 1:
            raise ValueError('Mixing binary and non-binary recognition models is not supported.')
        if '1' in modes and not _is_bitonal(im):
            logger.warning(f'Running binary models on non-binary input image (mode {im.mode}). This will result in '
                           'severely degraded performance')
        self._init_run(im, bounds, pad, bidi_reordering, lambda net: getattr(net, 'temperature', 1.0),
                       workers=num_line_workers, engine_batch=batch_size)

        def _ts(network):
            batch, channels, height, width = network.nn.input
            return ImageInputTransforms(batch, height, width, channels, (pad, 0), self._valid_norm)

        if default is not None:
            self.ts = defaultdict(partial(_ts, default))
        else:
            self.ts = {}
        if self.have_tags:
            tags = set(_line_type(x.tags) for x in bounds.lines)
            missing = [t for t in tags if default is None and not nets.get(t) and t not in tags_ignore]
            if missing:
                raise KrakenInputException(f'Missing models for tags {set(missing)}')
            self.ts = {t: _ts(nets[t]) for t in tags if t not in tags_ignore}
        if not isinstance(self.ts, defaultdict) and not self.ts:
            raise ValueError('No tags in input data and no default model in mapping given.')
        if isinstance(self.ts, defaultdict):
            self.ts['default']                       # materialise before worker threads race on the factory
        self.nets = nets
        self.tags_ignore = tags_ignore
        self.no_legacy_polygons = no_legacy_polygons

    # ------------------------------------------------------------------ per-line preparation
    def _use_legacy_extractor(self, net) -> bool:
        if net.nn.use_legacy_polygons:
            if self.no_legacy_polygons:
                warnings.warn('Enforcing use of the new polygon extractor for models trained with old version. '
                              'Accuracy may be affected.')
                return False
            warnings.warn('Using legacy polygon extractor, as the model was not trained with the new method. '
                          'Please retrain your model to get speed improvement.')
            return True
        return False

    def _prepare(self, idx: int):
        """Returns an ocr_record (empty / failed line) or a _Pending entry to be recognised."""
        line = self.bounds.lines[idx]
        if self._valid_norm:
            line.text_direction = self.bounds.text_direction
        if self.have_tags and self.tags_ignore and _line_type(line.tags) in self.tags_ignore:
            logger.info(f'Ignoring line segment with type {_line_type(line.tags)}.')
            return self._empty(line)
        tag, net = _pick_model(line.tags, self.nets, self._default)
        ts = self.ts[tag]
        if not ts._center_norm and extract_polygons is _EXTRACT_POLYGONS and self._device_prep_ok(net, ts):
            item = self._prepare_on_device(idx, line, tag, net, ts)
            if item is not None:
                return item
        item = self._dewarp_crop_from_page(idx, line, tag, net, ts)
        if item is not None:
            return item
        legacy = self._use_legacy_extractor(net)
        seg = dataclasses.replace(self.bounds, lines=[line])
        try:
            box, line2 = next(extract_polygons(self.im, seg, legacy=legacy))
            if self._valid_norm:
                line = line2
        except ValueError as e:
            logger.warning(f'Extracting line failed: {e}')
            return self._empty(line, [])
        if 0 in box.size:
            logger.warning(f'{line} with zero dimension. Emitting empty record.')
            return self._empty(line)
        item = self._crop_for_device(idx, line, tag, net, self.ts[tag], box, box.size)
        if item is not None:
            return item
        try:
            ts_box = self.ts[tag](box)
        except Exception:
            logger.warning(f'Conversion of line {line} failed. Emitting empty record..')
            return self._empty(line)
        if ts_box.max() == ts_box.min():
            logger.warning('Empty run. Emitting empty record.')
            return self._empty(line)
        return _Pending(idx, line, tag, net, ts_box, box.size, width=ts_box.shape[2])

    # ------------------------------------------------------------------------ iterator protocol
    def __next__(self):
        return self._next_record()

    def __iter__(self):
        return self

    def __len__(self):
        return self.len


def rpred(network, im, bounds, pad: int = 16, bidi_reordering: Union[bool, str] = True,
          no_legacy_polygons: bool = False, batch_size: Optional[int] = None, num_line_workers: int = PREP_THREADS):
    """Recognises the lines of `bounds` in `im` with one recogniser (kraken/rpred.py:344-370)."""
    return mm_rpred(defaultdict(lambda: network), im, bounds, pad, bidi_reordering,
                    no_legacy_polygons=no_legacy_polygons, batch_size=batch_size, num_line_workers=num_line_workers)


# ------------------------------------------------------------------------------------------- the batched (new) API
class _ConfiguredRecognizer:
    """What LinePipeline needs of a recogniser, built from a TorchVGSLModel + its inference config."""

    def __init__(self, model, config):
        self.nn = model
        self.codec = model.codec
        self.decoder = getattr(config, 'decoder', None) or _ctc.greedy_decoder
        try:                                           # kraken's own greedy decoder is the same operator (seam B4)
            from kraken.lib.ctc_decoder import greedy_decoder as _kgreedy
            if self.decoder is _kgreedy:
                self.decoder = _ctc.greedy_decoder
        except Exception:
            pass
        self.temperature = float(getattr(config, 'temperature', 1.0) or 1.0)
        self.outputs = None

    def predict(self, line):
        """Per-line path for custom decoders (reference _rec_predict, lib/vgsl/rpred.py:210-229)."""
        logits, _ = self.nn.nn(line, None)
        probs = (logits / self.temperature).softmax(1)
        self.outputs = probs.detach().squeeze(2)
        return [self.codec.decode(locs) for locs in self.decoder(self.outputs.cpu().numpy(), None)]


class _PredRun(_RecognitionRun):
    def __init__(self, model, im, segmentation, config):
        self.model, self.config = model, config
        pad = getattr(config, 'padding', 16)
        workers = getattr(config, 'num_line_workers', PREP_THREADS)
        self.net = _ConfiguredRecognizer(model, config)
        self._init_run(im, segmentation, pad, getattr(config, 'bidi_reordering', True), lambda net: net.temperature,
                       want_probs=bool(getattr(config, 'return_logits', False)) and segmentation.type == 'baselines',
                       return_image=bool(getattr(config, 'return_line_image', False)),
                       workers=workers if workers else 0)
        batch, channels, height, width = model.input
        self.ts = ImageInputTransforms(batch, height, width, channels, (pad, 0), self._valid_norm)
        self.legacy = False
        if model.use_legacy_polygons and segmentation.type == 'baselines':
            if getattr(config, 'no_legacy_polygons', False):
                warnings.warn('Enforcing use of the new polygon extractor for models trained with old version. Accuracy '
                              'may be affected.')
            else:
                warnings.warn('Using legacy polygon extractor, as the model was not trained with the new method. Please '
                              'retrain your model to get speed improvement.')
                self.legacy = True

    def _prepare(self, idx: int):
        line = self.bounds.lines[idx]
        if extract_polygons is _EXTRACT_POLYGONS and self._device_prep_ok(self.net, self.ts):
            item = self._prepare_on_device(idx, line, 'default', self.net, self.ts, want_image=self._return_image)
            if item is not None:
                if not isinstance(item, _Pending):           # this API's empty records carry list cuts (lib/vgsl/rpred.py:105-116)
                    return self._record_cls('', [], [], line)
                return item
        item = self._dewarp_crop_from_page(idx, line, 'default', self.net, self.ts, want_image=self._return_image)
        if item is not None:
            return item
        seg = dataclasses.replace(self.bounds, lines=[line])
        try:
            box, _ = next(extract_polygons(self.im, seg, legacy=self.legacy))
        except ValueError:
            return self._record_cls('', [], [], line)
        if box is None or 0 in box.size:
            return self._record_cls('', [], [], line)
        item = self._crop_for_device(idx, line, 'default', self.net, self.ts, box, box.size, want_image=self._return_image)
        if item is not None:
            return item
        try:
            ts_box = self.ts(box)
        except Exception:
            return self._record_cls('', [], [], line)
        if ts_box.max() == ts_box.min():
            return self._record_cls('', [], [], line)
        return _Pending(idx, line, 'default', self.net, ts_box, box.size, image=box, width=ts_box.shape[2])

    def _make_record(self, p: _Pending, r: LineResult):
        logits = None
        if getattr(self.config, 'return_logits', False):
            # the reference hands the decoded tuples to bbox records (lib/vgsl/rpred.py:157) and the probability slice
            # to baseline records (:200)
            logits = r.probs if not self._valid_norm else list(zip(r.text, r.starts.tolist(), r.ends.tolist(), r.confs.tolist()))
        rec = self._record_cls(r.text, self._cuts(p, r), r.confs.tolist(), p.line, logits=logits,
                               image=p.image if self._return_image else None)
        return self._order(rec)


def recognition_pred(model, im, segmentation, config=None):
    """
    Generator of ocr_records for the lines of `segmentation`, in input order: the batched recognition API
    (``VGSLRecognitionInference._recognition_pred``, reference kraken/lib/vgsl/rpred.py:56-124) on the pipelined engine.
    """
    run = _PredRun(model, im, segmentation, config)
    try:
        while True:
            try:
                yield run._next_record()
            except StopIteration:
                return
    finally:                   # also on GeneratorExit: `next(model.predict(...))`, `break` in the consumer's loop
        run.close()


def _freeze_import_heap():
    """
    Once per process, at the END OF THIS MODULE'S IMPORT: ``gc.freeze()`` -- the objects that exist by then (torch's, PIL's, numpy's and
    this package's modules: ~10^6 of them, none of which will ever be garbage) move to the collector's permanent generation.  Without
    it the first generation-2 pass after a page -- triggered by the page's records and crop descriptors becoming long-lived -- walks
    that whole heap: 55 ms in the middle of the SECOND page of a process (profiles/r06_cold_start.txt), and again whenever the heap has
    grown by a quarter.  At import time nothing of a caller's run exists yet: freezing later (when the first engine is built) would make
    the run that builds it immortal -- an abandoned generator would never hand its engine back
    (tests/test_rpred_cpu.py::test_an_abandoned_run_hands_back_a_clean_engine).  No gc.collect() in front: that would BE the 55 ms pass.
    KRK_GC_FREEZE=0 leaves the collector alone.
    """
    if os.environ.get('KRK_GC_FREEZE', '1') != '0':
        import gc
        gc.freeze()


_freeze_import_heap()
