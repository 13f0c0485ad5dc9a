"""
Rows of a Pillow image as raw memory.

The reference hands every line to its transforms as `im.crop(box)` (kraken/rpred.py:184-341 via
kraken/lib/segmentation.py:1630-1643); this package uploads the page (or the band of it a chunk of lines touches) once
and cuts the lines out on the device.  What is left on the host is getting the page's bytes out of Pillow: `np.asarray(im)`
goes through `Image.tobytes()` -- the raw encoder packs RGBX rows into RGB, 64 KB at a time, into a list of bytes objects
that are joined and copied again, all under the GIL: ~1 GB/s, which at 170 KB of page per line was THE cost of the API path
(DESIGN.md section 3.4).  Pillow stores an image as an array of row pointers (`Imaging->image`), rows of `linesize` bytes --
1 byte per pixel for 'L', 4 (R, G, B, X) for 'RGB' -- allocated in blocks of some thousand rows; copying those rows is a
`memmove` per block run (ctypes releases the GIL: the pool's threads run in parallel), and the kernels read the 4-byte
pixels as they are (`krk_prep_lines_fmt`, pixel stride 4).

Pillow has no public accessor for that array (`ImagingCore.unsafe_ptrs` is gone in Pillow 12, the Arrow export refuses /
crashes on images of more than one block), so it is read from the `Imaging` struct behind `Image.getim()`'s capsule.  The struct
is private: its layout is not assumed but PROBED -- every known layout is tried and accepted only if all of xsize, ysize,
bands, pixelsize and linesize read back as this image's, the row pointers are non-null and `linesize` apart inside a block,
and sampled pixels equal `Image.getpixel`.  Anything else -> `None`, and the caller keeps the `np.asarray` path.
Three guards in front of the first dereference of anything the probe found (a layout whose five ints line up by accident must not
make this module read foreign memory): Pillow's major version is one the offsets were checked against (`_TESTED_MAJORS`); the row
table pointer `image` equals `image8` (1-byte pixels) or `image32` (4-byte pixels) -- Pillow keeps those three consistent in every
layout -- before the table is read; and once per process and pixel size a band of rows read through the table is compared byte for
byte with `Image.tobytes()` of the same band (`_first_use_check`).  `tests/test_host_cpu.py::test_pillow_rows_*` pin every outcome.
"""
import ctypes
from typing import NamedTuple, Optional

import numpy as np

__all__ = ['RowTable', 'image_rows', 'copy_rows']

# ('1' images hold one byte per pixel, 0 or 255: exactly what convert('L') makes of them)
_PIXELSIZE = {'1': 1, 'L': 1, 'RGB': 4, 'RGBX': 4, 'RGBA': 4}
_BANDS = {'1': 1, 'L': 1, 'RGB': 3, 'RGBX': 4, 'RGBA': 4}

# byte offsets of (bands, xsize, ysize, image, pixelsize, linesize) in struct ImagingMemoryInstance (libImaging/Imaging.h), 64-bit:
#   Pillow >= 11.3: ModeID mode; int type, depth, bands, xsize, ysize; palette*; image8**, image32**, image**, block*, blocks*;
#                   int pixelsize, linesize
#   Pillow <= 11.2: char mode[6 + 1] (+ 1 byte of padding) in front of the same fields
_LAYOUTS = ((12, 16, 20, 48, 72, 76), (16, 20, 24, 56, 80, 84))
# Pillow major versions these offsets were checked against (9.x-12.x: the struct's head has not moved apart from the mode field);
# any other version keeps the np.asarray path until someone has looked at its Imaging.h
_TESTED_MAJORS = frozenset(range(9, 13))
_VERIFIED: set = set()     # pixel sizes whose first-use band check has passed in this process


class RowTable(NamedTuple):
    rows: np.ndarray        # int64 [height]: address of every row
    linesize: int           # bytes per row
    pixelsize: int          # bytes per pixel (1: 'L'; 4: 'RGB' stored as R, G, B, X)
    width: int
    height: int
    keep: object            # the image (and its capsule): the addresses are valid while it lives and is not modified


def _capsule_pointer(capsule) -> Optional[int]:
    api = ctypes.pythonapi
    api.PyCapsule_GetName.restype = ctypes.c_char_p
    api.PyCapsule_GetName.argtypes = [ctypes.py_object]
    api.PyCapsule_GetPointer.restype = ctypes.c_void_p
    api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
    name = api.PyCapsule_GetName(capsule)
    if name is None or b'Imaging' not in name:
        return None
    return api.PyCapsule_GetPointer(capsule, name)


def _int_at(addr: int) -> int:
    return ctypes.c_int.from_address(addr).value


def image_rows(im) -> Optional[RowTable]:
    """The row table of a '1' / 'L' / 'RGB' / 'RGBX' / 'RGBA' Pillow image, or None when it cannot be read SAFELY."""
    try:
        if im.mode not in _PIXELSIZE or ctypes.sizeof(ctypes.c_void_p) != 8:
            return None
        import PIL
        if int(PIL.__version__.split('.')[0]) not in _TESTED_MAJORS:
            return None
        im.load()                                                  # files are read lazily
        w, h = im.size
        if w <= 0 or h <= 0:
            return None
        capsule = im.getim() if hasattr(im, 'getim') else getattr(im.im, 'ptr', None)
        if capsule is None or type(capsule).__name__ != 'PyCapsule':
            return None
        base = _capsule_pointer(capsule)
        if not base:
            return None
        px, bands = _PIXELSIZE[im.mode], _BANDS[im.mode]
        for o_bands, o_x, o_y, o_image, o_px, o_line in _LAYOUTS:
            if (_int_at(base + o_bands), _int_at(base + o_x), _int_at(base + o_y), _int_at(base + o_px),
                    _int_at(base + o_line)) != (bands, w, h, px, w * px):
                continue
            table = ctypes.c_void_p.from_address(base + o_image).value
            if not table:
                continue
            # char **image8, **image32, **image sit side by side; `image` is a copy of the one that fits the pixel size and the
            # other one is NULL (libImaging/Storage.c).  Checked BEFORE the table is dereferenced
            image8 = ctypes.c_void_p.from_address(base + o_image - 16).value
            image32 = ctypes.c_void_p.from_address(base + o_image - 8).value
            if (px == 1 and (image8 != table or image32)) or (px == 4 and (image32 != table or image8)):
                continue
            rows = np.frombuffer((ctypes.c_uint64 * h).from_address(table), dtype=np.uint64).astype(np.int64)
            if (rows == 0).any():
                continue
            step = np.diff(rows)
            # inside a block rows are `linesize` apart; a block holds many rows, so jumps are rare
            if h > 1 and (step != w * px).sum() > max(4, h // 64):
                continue
            t = RowTable(rows, w * px, px, w, h, (im, capsule))
            if _samples_agree(im, t) and _first_use_check(im, t):
                return t
        return None
    except Exception:
        return None


def _samples_agree(im, t: RowTable) -> bool:
    """A handful of pixels read through the table equal Image.getpixel (corners, centre, a pseudo-random walk)."""
    w, h = t.width, t.height
    pts = {(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1), (w // 2, h // 2)}
    k = 12345
    for _ in range(11):
        k = (k * 1103515245 + 12345) & 0x7fffffff
        pts.add((k % w, (k >> 8) % h))
    nb = _BANDS[im.mode]
    for x, y in pts:
        raw = (ctypes.c_ubyte * t.pixelsize).from_address(int(t.rows[y]) + x * t.pixelsize)
        want = im.getpixel((x, y))
        want = (want,) if isinstance(want, int) else tuple(want)
        if tuple(raw[:nb]) != want[:nb]:
            return False
    return True


def _first_use_check(im, t: RowTable) -> bool:
    """
    Once per process and pixel size: up to 64 rows from the middle of the image, read through the table, equal `Image.tobytes()` of
    the same band (the public, slow path this module replaces) in every byte Pillow defines (the X of R, G, B, X is padding).
    """
    if t.pixelsize in _VERIFIED:
        return True
    y0 = max(0, t.height // 2 - 32)
    y1 = min(t.height, y0 + 64)
    got = np.empty((y1 - y0) * t.linesize, np.uint8)
    copy_rows(t, y0, y1, got)
    band = im.crop((0, y0, t.width, y1))
    nb = _BANDS[im.mode]
    if im.mode == '1':
        band = band.convert('L')
    want = np.frombuffer(band.tobytes(), np.uint8).reshape(y1 - y0, t.width, nb)
    if not np.array_equal(got.reshape(y1 - y0, t.width, t.pixelsize)[:, :, :nb], want):
        return False
    _VERIFIED.add(t.pixelsize)
    return True


def copy_rows(t: RowTable, y0: int, y1: int, dst: np.ndarray, pool=None, piece: int = 4 << 20) -> None:
    """
    Rows [y0, y1) of the image into `dst` (C-contiguous uint8 of (y1 - y0) * linesize bytes, e.g. a pinned upload buffer):
    one memmove per run of rows that are contiguous in Pillow's memory, cut into pieces of ~`piece` bytes for the pool.
    """
    if not (0 <= y0 <= y1 <= t.height):
        raise ValueError('row range outside the image')
    if dst.dtype != np.uint8 or not dst.flags['C_CONTIGUOUS'] or dst.nbytes < (y1 - y0) * t.linesize:
        raise ValueError('destination must be a C-contiguous uint8 array of at least (y1 - y0) * linesize bytes')
    if y1 == y0:
        return
    ls = t.linesize
    rows = t.rows[y0:y1]
    cuts = np.flatnonzero(np.diff(rows) != ls) + 1
    starts = np.concatenate(([0], cuts))
    ends = np.concatenate((cuts, [y1 - y0]))
    per = max(1, piece // ls)
    jobs = [(a, min(a + per, e)) for s, e in zip(starts.tolist(), ends.tolist()) for a in range(s, e, per)]
    base = dst.ctypes.data

    def move(job):
        a, b = job
        ctypes.memmove(base + a * ls, int(rows[a]), (b - a) * ls)
    if pool is not None and len(jobs) > 1:
        list(pool.map(move, jobs))
    else:
        for j in jobs:
            move(j)
