"""
Forward of the baseline/region segmenter (BLLA) on the HIP path: page image -> class heatmaps.

Mirrors ``SegmentationTaskModel._compute_segmentation_map`` (reference kraken/lib/vgsl/spred.py:237-287): the
page goes through ``ImageInputTransforms(valid_norm=False)`` (resize to the network height, optional white
padding, invert), the VGSL network (strided 7x7 / 3x3 convolutions, GroupNorm, LSTMs over image rows and
columns, 1x1 heatmap head) runs in ``csrc/*.hip`` (f32 plan), and the logits are upsampled to the scaled page
(nearest, like ``F.interpolate``) and squashed (``sigmoid``) by ``krk_upsample_sigmoid``.  Vectorisation / polygonisation of
the heatmaps (``vectorize_lines`` etc.) is out of scope and stays in kraken.

BASELINE.json config 5 ("blla.mlmodel baseline segmenter over 4k x 3k page, conv U-Net forward only").
"""
from typing import Any, Sequence, Union

import numpy as np
import torch

from .transforms import ImageInputTransforms
from .vgsl import TorchVGSLModel

__all__ = ['compute_segmentation_map']


@torch.no_grad()
def compute_segmentation_map(model: TorchVGSLModel, im, input_padding: Union[int, Sequence[int]] = 0,
                             device: str = 'cuda') -> dict[str, Any]:
    """
    Returns the dictionary of the reference: 'heatmap' (classes, H, W) float32 ndarray of sigmoid
    probabilities at the scaled page's resolution, 'cls_map', 'bounding_regions', 'scale' (page size /
    heatmap size) and 'scal_im' (the scaled grayscale page as uint8 ndarray).
    """
    if model.model_type and 'segmentation' not in model.model_type:
        raise ValueError(f'Models of type {model.model_type} are not segmentation models')
    batch, channels, height, width = model.input
    padding = input_padding
    if isinstance(padding, int):
        padding = (padding,) * 4
    elif len(padding) == 2:
        padding = (padding[0], padding[0], padding[1], padding[1])
    padding = tuple(int(v) for v in padding)
    transforms = ImageInputTransforms(batch, height, width, channels, padding, valid_norm=False)
    scal_im = np.array(transforms.pil_stage(im).convert('L'))
    tensor_im = transforms(im)

    model.to(device)
    o, _ = model.nn(tensor_im.unsqueeze(0).to(device))
    # nearest upsampling to the scaled page + sigmoid: one HIP kernel (spred.py:268-272 runs F.interpolate + torch.sigmoid)
    from . import _lib
    o = o.contiguous()
    _, nc, oh, ow = o.shape
    H, W = int(scal_im.shape[0]), int(scal_im.shape[1])
    up = torch.empty((1, nc, H, W), dtype=torch.float32, device=o.device)
    _lib.check(_lib.load().krk_upsample_sigmoid(o.data_ptr(), nc, oh, ow, H, W, up.data_ptr(), torch.cuda.current_stream(o.device).cuda_stream))
    o = up
    # remove padding (same index arithmetic as the reference, spred.py:272-277)
    pad = [p if p else None for p in padding]
    pad[1] = -pad[1] if pad[1] else None
    pad[3] = -pad[3] if pad[3] else None
    o = o[:, :, pad[2]:pad[3], pad[0]:pad[1]]
    scal_im = scal_im[pad[2]:pad[3], pad[0]:pad[1]]
    o = o.squeeze().cpu().float().numpy()
    scale = np.divide(im.size, o.shape[:0:-1])
    meta = getattr(model, 'user_metadata', {}) or {}
    return {'heatmap': o,
            'cls_map': meta.get('class_mapping'),
            'bounding_regions': meta.get('bounding_regions', None),
            'scale': scale,
            'scal_im': scal_im}
