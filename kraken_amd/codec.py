"""
Label <-> code point codec of the recognition path.

Mirrors ``kraken.lib.codec.PytorchCodec`` (reference kraken/lib/codec.py:28-270): the
constructor forms (:61-75), ``encode`` (:106-146), ``decode`` (:148-195), ``merge``
(:197-240), ``add_labels`` (:242-264) and the validity rule (:84-97), pinned by the
reference's tests/test_codec.py.  Host-side by design (strings are a host concept); the
additions are a precomputed single-label fast path and ``decode_batch`` which turns the
compact device result of a whole batch into records without a Python loop per label when
the codec only has single-label codes.
"""
import logging
from collections import Counter
from typing import Sequence, Union

import numpy as np
from torch import IntTensor

__all__ = ['PytorchCodec', 'KrakenEncodeException', 'KrakenCodecException']

logger = logging.getLogger(__name__)


class KrakenEncodeException(Exception):
    pass


class KrakenCodecException(Exception):
    pass


try:  # share exception types with an installed kraken so `except` clauses keep working
    from kraken.lib.exceptions import KrakenCodecException, KrakenEncodeException  # type: ignore # noqa: F811
except Exception:  # pragma: no cover - kraken not installed
    pass


class PytorchCodec(object):
    """
    Translates between code point sequences and label sequences.  A code point
    sequence may map to several labels and the other way round; label 0 is the CTC blank.
    """

    def __init__(self, charset: Union[dict[str, Sequence[int]], Sequence[str], str], strict=False):
        if isinstance(charset, dict):
            self.c2l = charset
        else:
            dup = Counter(charset)
            if len(dup) < len(charset):
                raise KrakenCodecException(f'Duplicate entry in codec definition string: {dup}')
            self.c2l = {k: [v] for v, k in enumerate(sorted(charset), start=1)}
        self.c_sorted = sorted(self.c2l.keys(), key=len, reverse=True)
        self.l2c: dict[tuple[int, ...], str] = {tuple(v): k for k, v in self.c2l.items()}
        self.l2c_single = {k[0]: v for k, v in self.l2c.items() if len(k) == 1}
        self.strict = strict
        if not self.is_valid:
            raise KrakenCodecException('Codec is not valid (non-singular/non-prefix free).')
        # fast path tables
        self._multi = [k for k in self.l2c.keys() if len(k) > 1]
        self._multi_first = {k[0] for k in self._multi}

    def __len__(self) -> int:
        return len(self.l2c)

    @property
    def is_valid(self) -> bool:
        """Prefix-free in label space and one-to-one between label and code point sequences."""
        if len(self.l2c) != len(self.c2l):
            return False
        codes = sorted(self.l2c.keys())
        # in sorted order a prefix is immediately followed by its extensions
        for i, a in enumerate(codes):
            for b in codes[i + 1:]:
                if b[:len(a)] != a:
                    break
                return False
        return True

    @property
    def max_label(self) -> int:
        return max(label for labels in self.c2l.values() for label in labels)

    def encode(self, s: str) -> IntTensor:
        """Greedy longest-match encoding of a string into labels."""
        labels: list[int] = []
        pos = 0
        multi = [c for c in self.c_sorted if len(c) > 1]
        while pos < len(s):
            for code in multi:
                if s.startswith(code, pos):
                    labels.extend(self.c2l[code])
                    pos += len(code)
                    break
            else:
                ch = s[pos]
                if ch in self.c2l:
                    labels.extend(self.c2l[ch])
                elif self.strict:
                    raise KrakenEncodeException(f'Non-encodable sequence {s[pos:pos + 5]}... encountered.')
                else:
                    logger.warning(f'Non-encodable sequence {s[pos:pos + 5]}... encountered. Advancing one code point.')
                pos += 1
        return IntTensor(labels)

    def decode(self, labels: Sequence[tuple[int, int, int, float]]) -> list[tuple[str, int, int, float]]:
        """
        (label, start, end, conf) tuples -> (code point, start, end, conf) tuples.
        Single-label codes map directly; multi-label codes take min start / max end / mean
        confidence; every code point of a multi-code-point string repeats the cut and confidence;
        undecodable labels are skipped (or raise in strict mode).
        """
        labs = [int(t[0]) for t in labels]
        out: list[tuple[str, int, int, float]] = []
        i, n = 0, len(labs)
        single = self.l2c_single
        while i < n:
            lab = labs[i]
            code = single.get(lab)
            if code is not None:
                _, s, e, c = labels[i]
                for ch in code:
                    out.append((ch, s, e, c))
                i += 1
                continue
            matched = False
            if lab in self._multi_first:
                for key in self.l2c.keys():
                    k = len(key)
                    if tuple(labs[i:i + k]) == key:
                        s = labels[i][1]
                        e = labels[i + k - 1][2]
                        c = np.mean([t[3] for t in labels[i:i + k]])
                        for ch in self.l2c[key]:
                            out.append((ch, s, e, c))
                        i += k
                        matched = True
                        break
            if not matched:
                if self.strict:
                    raise KrakenEncodeException(f'Non-decodable sequence {tuple(labs[i:i + 5])}... encountered.')
                logger.debug(f'Non-decodable sequence {tuple(labs[i:i + 5])}... encountered. Advancing one label.')
                i += 1
        return out

    def decode_batch(self, batch) -> list[list[tuple[str, int, int, float]]]:
        """Decodes a whole ``DecodedBatch`` (see vgsl.py); equivalent to ``decode`` per line."""
        return [self.decode(t) for t in batch.tuples()]

    # ---- vectorised fast path ---------------------------------------------------------------
    def _single_lut(self):
        """label -> code point table, valid when every code is ONE label mapping to ONE code point."""
        lut = getattr(self, '_lut', None)
        if lut is None:
            if any(len(k) != 1 or len(v) != 1 for k, v in self.l2c.items()):
                lut = False
            else:
                lut = np.zeros(self.max_label + 1, dtype=np.uint32)      # 0 = not decodable
                for (lab,), ch in self.l2c.items():
                    lut[lab] = ord(ch)
            self._lut = lut
        return lut

    def decode_strings(self, batch) -> list[str]:
        """
        Text of every line of a ``DecodedBatch`` -- ``''.join(c for c, *_ in decode(line))`` for all lines
        without a Python loop per label: one table lookup over the compact label array, one UTF-32 decode,
        one slice per line.  Falls back to ``decode`` for codecs with multi-label / multi-code-point entries.
        """
        lut = self._single_lut()
        if lut is False:
            return [''.join(c for c, *_ in rec) for rec in self.decode_batch(batch)]
        counts = np.asarray(batch.counts)
        labels = np.asarray(batch.labels)
        n, t = labels.shape if labels.ndim == 2 else (len(counts), 0)
        keep = np.arange(t)[None, :] < counts[:, None]
        flat = labels[keep]
        cps = np.where(flat < len(lut), lut[np.minimum(flat, len(lut) - 1)], 0).astype('<u4')
        ok = cps != 0                                   # undecodable labels are skipped (non-strict mode)
        if self.strict and not ok.all():
            bad = flat[~ok][:5].tolist()
            raise KrakenEncodeException(f'Non-decodable sequence {tuple(bad)}... encountered.')
        # decodable labels per line from a prefix sum (reduceat cannot take empty / trailing segments)
        csum = np.r_[0, np.cumsum(ok, dtype=np.int64)]
        ends = np.cumsum(np.minimum(np.maximum(counts, 0), t), dtype=np.int64)
        per_line = csum[ends] - csum[np.r_[0, ends[:-1]]] if n else np.zeros(0, np.int64)
        text = cps[ok].tobytes().decode('utf-32-le')
        out, pos = [], 0
        for k in per_line.tolist():
            out.append(text[pos:pos + k])
            pos += k
        return out

    def merge(self, codec: 'PytorchCodec') -> tuple['PytorchCodec', set]:
        """
        Transforms this codec into one encoding the code point sequences of `codec`, reusing
        labels where the mapping is shared; returns the merged codec and the removed labels.
        """
        gone = {cs: enc for cs, enc in self.c2l.items() if cs not in codec.c2l}
        kept = {cs: enc for cs, enc in self.c2l.items() if cs in codec.c2l}
        rm_labels = [lab for enc in gone.values() for lab in enc]
        # labels still used by a kept mapping stay.  The scan advances by position while the list
        # shrinks underneath it -- the reference's exact (order dependent) behaviour, codec.py:219-222
        for enc in kept.values():
            i = 0
            while i < len(rm_labels):
                if rm_labels[i] in enc:
                    rm_labels.remove(rm_labels[i])
                i += 1
        # close the holes left by removed labels
        for hole in [v - i for i, v in enumerate(sorted(set(rm_labels)))]:
            kept = {k: [lab - 1 if lab > hole else lab for lab in v] for k, v in kept.items()}
        new = {cs: enc for cs, enc in codec.c2l.items() if cs not in self.c2l}
        nxt = max((0,) + tuple(lab for v in kept.values() for lab in v)) + 1
        renum = {lab: i for i, lab in enumerate(sorted({lab for v in new.values() for lab in v}), nxt)}
        for cs, enc in new.items():
            kept[cs] = [renum[lab] for lab in enc]
        return PytorchCodec(kept, self.strict), set(rm_labels)

    def add_labels(self, charset: Union[dict[str, Sequence[int]], Sequence[str], str]) -> 'PytorchCodec':
        """Returns a codec extended by `charset` (string / list: fresh labels; dict: explicit labels)."""
        c2l = self.c2l.copy()
        if isinstance(charset, dict):
            c2l.update(charset)
        else:
            c2l.update({k: [v] for v, k in enumerate(sorted(charset), start=self.max_label + 1)})
        return PytorchCodec(c2l, self.strict)

    def __repr__(self):
        return f'PytorchCodec({self.c2l})'
