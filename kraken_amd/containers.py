"""
Record / segmentation containers used by the ``rpred`` mirror.

kraken's own containers (reference kraken/containers.py: ``BBoxLine`` :152, ``BaselineLine`` :95,
``Segmentation`` :251, ``BBoxOCRRecord`` :608, ``BaselineOCRRecord`` :430) are OUT OF SCOPE of
this repository ("output container, reused as is") -- when kraken is importable
they are re-exported unchanged, so records produced here are the reference's own classes, bidi
reordering included.

On a box without kraken (the GPU test box) a minimal stand-in with the same constructor and
read accessors is provided so the hot path can still hand back records; it supports
records of left-to-right text in both orders (no UAX#9 reordering of right-to-left text, no polygon cuts).
"""
from dataclasses import dataclass, field
from typing import Any, Optional

try:  # pragma: no cover - exercised only where kraken is installed
    from kraken.containers import (BaselineLine, BaselineOCRRecord, BBoxLine, BBoxOCRRecord,  # noqa: F401
                                   Segmentation, ocr_record)
    HAVE_KRAKEN_CONTAINERS = True
except Exception:
    HAVE_KRAKEN_CONTAINERS = False

    @dataclass
    class BBoxLine:
        id: str
        bbox: Optional[tuple] = None
        text: Optional[str] = None
        base_dir: Optional[str] = None
        imagename: Optional[str] = None
        tags: Optional[dict] = None
        split: Optional[str] = None
        regions: Optional[list] = None
        language: Optional[list] = None
        type: str = 'bbox'
        text_direction: str = 'horizontal-lr'

    @dataclass
    class BaselineLine:
        id: str
        baseline: Optional[list] = None
        boundary: Optional[list] = None
        text: Optional[str] = None
        base_dir: Optional[str] = None
        imagename: Optional[str] = None
        tags: Optional[dict] = None
        split: Optional[str] = None
        regions: Optional[list] = None
        language: Optional[list] = None
        type: str = 'baselines'

    @dataclass
    class Segmentation:
        type: str
        imagename: Any
        text_direction: str
        script_detection: bool
        lines: list = field(default_factory=list)
        regions: Optional[dict] = None
        line_orders: Optional[list] = None
        language: Optional[list] = None

    class ocr_record:
        """Recognition result of one line: text, per-code-point cuts and confidences."""
        type = None

        def __init__(self, prediction, cuts, confidences, line, base_dir=None, display_order=True, logits=None,
                     image=None):
            self._prediction = prediction
            self._cuts = cuts
            self._confidences = confidences
            self._display_order = display_order
            self.line = line
            self.base_dir = base_dir
            self.logits = logits
            self.image = image
            for k in ('id', 'bbox', 'tags', 'text_direction', 'baseline', 'boundary'):
                if hasattr(line, k):
                    setattr(self, k, getattr(line, k))

        def __len__(self):
            return len(self._prediction)

        def __str__(self):
            return self._prediction

        @property
        def prediction(self):
            return self._prediction

        @property
        def cuts(self):
            return self._cuts

        @property
        def confidences(self):
            return self._confidences

        def __iter__(self):
            return iter(zip(self._prediction, self._cuts, self._confidences))

        def display_order(self, base_dir=None):
            return self

        def logical_order(self, base_dir=None):
            """
            Display -> logical order.  Text without right-to-left or digit-ordering characters reads the same in both
            orders under a left-to-right base direction, so it is returned as is; anything else needs the UAX#9
            implementation of kraken (kraken.lib.bidi), which is outside this repository's scope.
            """
            import unicodedata
            if not self._display_order:
                return self
            rtl = {'R', 'AL', 'AN', 'RLE', 'RLO', 'RLI'}
            if base_dir in (None, 'L') and not any(unicodedata.bidirectional(c) in rtl for c in self._prediction):
                rec = type(self)(self._prediction, self._cuts, self._confidences, self.line, base_dir=base_dir,
                                 display_order=False, logits=self.logits, image=self.image)
                return rec
            raise NotImplementedError('BiDi reordering of right-to-left text needs kraken.containers (kraken.lib.bidi); '
                                      'install kraken or call rpred with bidi_reordering=False')

    class BBoxOCRRecord(ocr_record):
        type = 'bbox'

    class BaselineOCRRecord(ocr_record):
        type = 'baselines'
