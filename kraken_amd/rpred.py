"""
Legacy line-recognition generators on top of the HIP path.

Drop-in for ``kraken.rpred.rpred`` / ``kraken.rpred.mm_rpred`` (reference kraken/rpred.py:57-370):
same constructor arguments, the same iterator protocol (``len()``, ``next()`` yields one
``ocr_record`` per line, in input order), per-tag model routing (``_resolve_type_to_model``
:373-391), the same empty-record rules (ignored tags, failed extraction, zero-sized crops,
failed tensor conversion, flat lines -- :193-223) and the same position arithmetic
(``_scale_val`` :329-330).

What changes underneath: the reference runs ONE line per ``next()`` through torch on the CPU and
copies the full softmax matrix to the host; here lines that share a model are queued and pushed
through ``krk_recognize`` in width-sorted batches of ``batch_size`` lines.  Because the kernels
implement masked padding, a line's result does not depend on its batch mates, so records are
identical to per-line evaluation (tests/test_rpred_mirror.py).

Line extraction: bounding-box lines are cropped here (the reference's
``extract_polygons`` bbox branch, kraken/lib/segmentation.py:1630-1643); baseline/polygon
extraction is CPU geometry outside the hot path and is delegated to kraken when it is installed.
The module-level name ``extract_polygons`` is kept patchable like in the reference
(tests/test_newpolygons.py:64-102 mocks it).
"""
import dataclasses
import logging
import warnings
from collections import defaultdict
from functools import partial
from typing import Optional, Union

import numpy as np
import torch

from .containers import BaselineOCRRecord, BBoxOCRRecord
from .transforms import ImageInputTransforms

__all__ = ['mm_rpred', 'rpred', 'extract_polygons']

logger = logging.getLogger(__name__)


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
    tensor: torch.Tensor
    box_size: tuple


class mm_rpred(object):
    """Multi-model recogniser iterator: ``for record in mm_rpred(nets, im, bounds): ...``"""

    def __init__(self, nets, im, bounds, pad: int = 16, bidi_reordering: Union[bool, str] = True,
                 tags_ignore: Optional[list] = None, no_legacy_polygons: bool = False, batch_size: int = 32):
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
        self.len = len(bounds.lines)
        self._valid_norm = bounds.type != 'baselines'
        self._record_cls = BBoxOCRRecord if bounds.type != 'baselines' else BaselineOCRRecord

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

        self.im, self.nets, self.bounds = im, nets, bounds
        self.bidi_reordering = bidi_reordering
        self.pad = pad
        self.tags_ignore = tags_ignore
        self.no_legacy_polygons = no_legacy_polygons
        self.batch_size = max(1, int(batch_size))
        self._results: dict[int, object] = {}
        self._cursor = 0          # next line index to hand out
        self._prepared = 0        # lines already extracted / queued
        self._warned_legacy = False

    # ------------------------------------------------------------------ per-line preparation
    def _empty(self, line, cuts=()):
        return self._record_cls('', cuts, cuts, line)

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
        legacy = self._use_legacy_extractor(net)
        seg = dataclasses.replace(self.bounds, lines=[line])
        try:
            box, line = next(extract_polygons(self.im, seg, legacy=legacy))
        except ValueError as e:
            logger.warning(f'Extracting line failed: {e}')
            return self._empty(line, [])
        if 0 in box.size:
            logger.warning(f'{line} with zero dimension. Emitting empty record.')
            return self._empty(line)
        try:
            ts_box = self.ts[tag](box)
        except Exception:
            logger.warning(f'Conversion of line {line} failed. Emitting empty record..')
            return self._empty(line)
        if ts_box.max() == ts_box.min():
            logger.warning('Empty run. Emitting empty record.')
            return self._empty(line)
        return _Pending(idx, line, tag, net, ts_box, box.size)

    # --------------------------------------------------------------------- batched inference
    def _scale(self, val, net_scale, in_scale, max_val):
        return int(round(min(max(((val * net_scale) - self.pad) * in_scale, 0), max_val - 1)))

    def _scale_all(self, vals, net_scale, in_scale, max_val):
        """Vectorised `_scale` (kraken/rpred.py:329-330): np.rint rounds half to even exactly like Python's round()."""
        v = (np.asarray(vals, dtype=np.float64) * net_scale - self.pad) * in_scale
        return np.rint(np.minimum(np.maximum(v, 0), max_val - 1)).astype(np.int64).tolist()

    def _finish(self, p: _Pending, preds, out_width: int):
        net_scale = p.tensor.shape[2] / out_width
        in_scale = p.box_size[0] / (p.tensor.shape[2] - 2 * self.pad)
        text = ''.join(x[0] for x in preds)
        conf = [x[3] for x in preds]
        starts, ends = [x[1] for x in preds], [x[2] for x in preds]
        if self._valid_norm:
            coords = p.line.bbox
            if self.bounds.text_direction.startswith('horizontal'):
                lo = self._scale_all(starts, net_scale, in_scale, p.box_size[0])
                hi = self._scale_all(ends, net_scale, in_scale, p.box_size[0])
                x0, y0, y1 = coords[0], coords[1], coords[3]
                pos = [[[x0 + a, y0], [x0 + a, y1], [x0 + b, y1], [x0 + b, y0]] for a, b in zip(lo, hi)]
            else:
                lo = self._scale_all(starts, net_scale, in_scale, p.box_size[1])
                hi = self._scale_all(ends, net_scale, in_scale, p.box_size[1])
                x0, x1, y0 = coords[0], coords[2], coords[1]
                pos = [[[x0, y0 + a], [x1, y0 + a], [x1, y0 + b], [x0, y0 + b]] for a, b in zip(lo, hi)]
        else:
            lo = self._scale_all(starts, net_scale, in_scale, p.box_size[0])
            hi = self._scale_all(ends, net_scale, in_scale, p.box_size[0])
            pos = [[a, b] for a, b in zip(lo, hi)]
        rec = self._record_cls(text, pos, conf, p.line)
        if self.bidi_reordering:
            return rec.logical_order(base_dir=self.bidi_reordering if self.bidi_reordering in ('L', 'R') else None)
        return rec.display_order(None)

    def _recognise(self, pending: list):
        """Runs the queued lines of ONE recogniser as padded batches."""
        net = pending[0].net
        widths = [p.tensor.shape[2] for p in pending]
        wmax = max(widths)
        x = torch.zeros((len(pending),) + tuple(pending[0].tensor.shape[:2]) + (wmax,), dtype=torch.float32)
        for i, p in enumerate(pending):
            x[i, :, :, :widths[i]] = p.tensor
        lens = torch.tensor(widths, dtype=torch.int32)
        if hasattr(net.nn.nn, 'recognize'):          # kraken_amd recogniser: fused GPU path
            batch, olens, _, _ = net.nn.nn.recognize(x, lens, temperature=getattr(net, 'temperature', 1.0))
            decoded = net.codec.decode_batch(batch) if hasattr(net.codec, 'decode_batch') else \
                [net.codec.decode(t) for t in batch.tuples()]
            outw = [int(v) for v in olens]
        else:                                        # any object with the reference's interface
            decoded, outw = [], []
            for i, p in enumerate(pending):
                decoded.append(net.predict(p.tensor.unsqueeze(0))[0])
                outw.append(net.outputs.shape[2])
        for p, preds, ow in zip(pending, decoded, outw):
            self._results[p.idx] = self._finish(p, preds, ow)

    def _fill(self):
        """Prepares and recognises lines until the record at the cursor is available."""
        while self._cursor not in self._results and self._prepared < self.len:
            queue: dict[int, list] = {}
            while self._prepared < self.len and sum(map(len, queue.values())) < self.batch_size:
                item = self._prepare(self._prepared)
                if isinstance(item, _Pending):
                    queue.setdefault(id(item.net), []).append(item)
                else:
                    self._results[self._prepared] = item
                self._prepared += 1
            for items in queue.values():
                self._recognise(items)

    # ------------------------------------------------------------------------ iterator protocol
    def __next__(self):
        if self._cursor >= self.len:
            raise StopIteration
        self._fill()
        rec = self._results.pop(self._cursor)
        self._cursor += 1
        return rec

    def __iter__(self):
        return self

    def __len__(self):
        return self.len


def rpred(network, im, bounds, pad: int = 16, bidi_reordering: Union[bool, str] = True,
          no_legacy_polygons: bool = False, batch_size: int = 32):
    """Recognises the lines of `bounds` in `im` with one recogniser (kraken/rpred.py:344-370)."""
    return mm_rpred(defaultdict(lambda: network), im, bounds, pad, bidi_reordering,
                    no_legacy_polygons=no_legacy_polygons, batch_size=batch_size)
