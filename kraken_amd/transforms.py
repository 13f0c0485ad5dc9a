"""
Line image -> network input tensor, without torchvision.

Restates ``kraken.lib.dataset.ImageInputTransforms`` (reference kraken/lib/dataset/utils.py:54-152),
``kraken.lib.functional_im_transforms`` (:58-82) and the centre-line normaliser of
``kraken.lib.lineest`` (:14-87) with PIL / numpy / scipy.  This is the CPU stage *before* the
hot path (SURVEY.md row a9 "edge"); the synthetic benchmark configs bypass it, the legacy
``rpred`` mirror needs it.  Order of operations for the recognition case:

    mode conversion ('L' or 'RGB') -> dewarp (bbox lines of a 1-channel, fixed-height model) or
    LANCZOS resize to the network height -> white padding left/right -> [0,1] float -> invert
    (max - x, ink becomes high) -> CHW

Pinned against reference outputs in tests/golden/transforms.npz and, through the page fixture of
tests/golden/overfit.npz, against the reference's known-answer strings.
"""
import warnings
from typing import Union

import numpy as np
import torch
from PIL import Image, ImageOps
from scipy.ndimage import affine_transform, gaussian_filter, uniform_filter

__all__ = ['ImageInputTransforms', 'center_normalize', 'dewarp_tables']


def _line_centers(ink: np.ndarray, smoothness: float = 1.0, extra: float = 0.3):
    """Per-column vertical centre of the ink mass (ocropy-style), smoothed along the line."""
    h, w = ink.shape
    blur = gaussian_filter(ink, (h * 0.5, h * smoothness), mode='constant')
    blur = blur + 0.001 * uniform_filter(blur, (h * 0.5, w), mode='constant')
    ridge = np.argmax(blur, axis=0)
    ridge = gaussian_filter(ridge, h * extra)
    return np.array(ridge, 'i')


def center_normalize(gray: np.ndarray, target_height: int, spread: float = 4.0) -> np.ndarray:
    """
    Dewarps a grayscale line (0 = ink ... 255 = paper as float array) around its centre line and
    scales it to `target_height`; restates lineest.dewarp + CenterNormalizer.measure/normalize.
    """
    line = np.asarray(gray)
    top = np.amax(line)
    ink = top - line
    ink = ink * 1.0 / np.amax(ink)
    h, w = ink.shape
    center = _line_centers(ink)
    rows = np.arange(h)[:, None]
    mad = np.mean(np.abs(rows - center[None, :])[ink != 0])
    r = int(1 + spread * mad)
    # cut a band of +-r rows around the centre out of the vertically padded line
    stack = np.vstack([top * np.ones((h, w)), line, top * np.ones((h, w))])
    mid = center + h
    band = np.array([stack[mid[i] - r:mid[i] + r, i] for i in range(w)], dtype=np.dtype('f')).T
    if band.shape[0] == 0:
        band = line
    bh, bw = band.shape
    scale = target_height * 1.0 / bh
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', UserWarning)
        out = affine_transform(1.0 * band, np.ones(2) / scale, order=1,
                               output_shape=(target_height, int(scale * bw)), mode='constant', cval=top)
    return np.array(out, dtype=np.dtype('f'))


def dewarp_tables(heights) -> tuple[np.ndarray, dict]:
    """
    Gaussian weight tables of the device dewarp (csrc/dewarp.hip) for the line heights of a batch: per height the three kernels of
    lineest.CenterNormalizer.measure -- sigma = h/2 (rows), h (columns), 0.3 h (ridge) -- computed like scipy's
    ``_gaussian_kernel1d`` (radius int(4 sigma + 0.5), exp(-x^2 / 2 sigma^2) / sum) with the HOST's exp, so that the device only
    multiplies and adds.  Returns (flat float64 array, {h: (offset, r0, r1, r2)}).
    """
    parts, index, off = [], {}, 0
    for h in sorted(set(int(v) for v in heights)):
        rs = []
        for sigma in (h * 0.5, h * 1.0, h * 0.3):
            r = int(4.0 * float(sigma) + 0.5)
            x = np.arange(-r, r + 1)
            phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
            parts.append(phi / phi.sum())
            rs.append(r)
        index[h] = (off, *rs)
        off += sum(2 * r + 1 for r in rs)
    return np.concatenate(parts) if parts else np.zeros(0), index


def _to_pil_gray(arr: np.ndarray) -> Image.Image:
    """float array -> 8-bit PIL image (what kraken.lib.util.array2pil does for float input)."""
    if arr.dtype == np.dtype('bool'):
        return Image.fromarray(np.array(255 * arr, 'B'))
    return Image.fromarray(np.array(np.clip(arr, 0, 255), 'B') if arr.dtype.kind == 'f' else arr)


class ImageInputTransforms:
    """
    Callable PIL image -> float tensor (C, H, W).  Same constructor as the reference:
    ``ImageInputTransforms(batch, height, width, channels, pad, valid_norm=True,
    force_binarization=False, dtype=torch.float32)``.
    """

    def __init__(self, batch: int, height: int, width: int, channels: int,
                 pad: Union[int, tuple], valid_norm: bool = True, force_binarization: bool = False,
                 dtype: torch.dtype = torch.float32):
        if force_binarization:
            raise NotImplementedError('force_binarization (nlbin) is outside the recognition hot path')
        self._batch, self._dtype = batch, dtype
        self._pad = pad
        self._center_norm = False
        self._mode = 'RGB' if channels == 3 else 'L'
        self._perm = (0, 1, 2)
        self._scale = (height, width)
        self._channels = channels
        if height == 1 and width == 0 and channels > 3:
            # legacy [1,1,0,48] specs: the line height lives in the channel axis
            self._perm = (1, 0, 2)
            self._scale = (channels, 0)
            self._channels = 1
            self._center_norm = bool(valid_norm)
            self._mode = 'L'
        elif height > 1 and width == 0 and channels in (1, 3):
            self._center_norm = bool(valid_norm) and channels == 1
        elif height == 0 and width > 1 and channels in (1, 3):
            pass
        elif height > 0 and width > 0 and channels in (1, 3):
            self._pad = 0
        elif height == 0 and width == 0 and channels in (1, 3):
            self._pad = 0
        else:
            raise ValueError(f'Invalid input spec {batch}, {height}, {width}, {channels}, {pad}.')

    @property
    def pad(self):
        return self._pad

    def _horizontal_pad(self):
        p = self._pad
        if isinstance(p, (tuple, list)):
            return int(p[0]), (int(p[1]) if len(p) > 1 else 0)
        return int(p), int(p)

    def __call__(self, im: Image.Image) -> torch.Tensor:
        im = self.pil_stage(im)
        arr = np.array(im, dtype=np.uint8)
        t = torch.from_numpy(arr)
        t = t[None] if t.dim() == 2 else t.permute(2, 0, 1)
        t = t.to(torch.float32) / 255.0
        t = t.max() - t
        return t.permute(*self._perm).to(self._dtype)

    def pil_stage(self, im: Image.Image) -> Image.Image:
        """The transforms in front of PILToTensor (mode, dewarp / resize, padding): what the segmenter keeps as `scal_im`."""
        im = im.convert(self._mode)
        sh, sw = self._scale
        if (sh, sw) != (0, 0):
            if self._center_norm:
                arr = np.array(im.convert('L'), dtype=np.float64) if im.mode != 'L' else np.array(im).astype(np.float64)
                if im.mode == '1':
                    arr = arr * 255
                im = _to_pil_gray(center_normalize(arr, sh)).convert(self._mode)
            elif sh > 0 and sw > 0:
                im = im.resize((sw, sh), Image.Resampling.LANCZOS)
            else:
                w, h = im.size
                oh, ow = sh, sw
                if oh == 0:
                    oh = int(h * ow / w)
                elif ow == 0:
                    ow = int(w * oh / h)
                im = im.resize((ow, oh), Image.Resampling.LANCZOS)
        if self._pad:
            p = self._pad
            if isinstance(p, (tuple, list)) and len(p) == 4:
                border = tuple(int(v) for v in p)            # torchvision v2.Pad order: left, top, right, bottom
            else:
                border = self._horizontal_pad()
            im = ImageOps.expand(im, border=border, fill=255 if self._mode == 'L' else (255, 255, 255))
        return im
