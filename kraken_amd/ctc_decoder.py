"""
CTC best-path decoder operator backed by the HIP kernels ``rowmax_*`` + ``collapse``.

Drop-in for ``kraken.lib.ctc_decoder.greedy_decoder`` (reference
kraken/lib/ctc_decoder.py:35-72), i.e. the ``config.decoder`` plug point of
RecognitionInferenceConfig (reference kraken/configs/base.py:219-235) and the ``decoder``
constructor argument of TorchSeqRecognizer (reference kraken/lib/models.py:37).

Same contract: ``outputs`` is a (C, W) or (N, C, W) tensor/ndarray of softmax outputs or raw
logits, ``seq_lens`` the valid widths; the result is a list (per line) of
``(label, start, end, max confidence)`` tuples with ``end`` inclusive, blanks (label 0)
dropped and argmax ties resolved to the lowest class index.  There is no CPU
implementation here: without a HIP device the call raises.
"""
import ctypes as C
from typing import Union

import numpy as np
import torch

from . import _lib

__all__ = ['greedy_decoder']


def greedy_decoder(outputs: Union[torch.Tensor, np.ndarray],
                   seq_lens: torch.Tensor = None) -> list[list[tuple[int, int, int, float]]]:
    lib = _lib.load()
    _lib.require_gpu()
    out = torch.as_tensor(outputs)
    if out.dim() == 2:
        out = out.unsqueeze(0)
    if out.dim() != 3:
        raise ValueError(f'expected a (C, W) or (N, C, W) score tensor, got {tuple(out.shape)}')
    N, Cc, T = out.shape
    if N == 1 and seq_lens is None:
        lens = np.array([T], dtype=np.int32)
    elif seq_lens is None:
        raise ValueError('seq_lens need to be set for batch decoding.')
    else:
        lens = np.ascontiguousarray(torch.as_tensor(seq_lens).detach().cpu().numpy().astype(np.int32).reshape(-1))
        if lens.shape[0] != N:
            raise ValueError('seq_lens needs one entry per line')
        if (lens > T).any() or (lens < 0).any():
            raise ValueError('seq_lens outside [0, W]')
    if T == 0 or Cc == 0:
        return [[] for _ in range(N)]
    if not out.is_cuda:
        out = out.to(f'cuda:{torch.cuda.current_device()}')
    if out.dtype != torch.float32:
        out = out.float()
    dev = out.device
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        labels, starts, ends = (torch.empty((N, T), **i32) for _ in range(3))
        confs = torch.empty((N, T), dtype=torch.float32, device=dev)
        counts = torch.empty((N,), **i32)
        dec = _lib.KrkDecodeOut(labels.data_ptr(), starts.data_ptr(), ends.data_ptr(), confs.data_ptr(),
                                counts.data_ptr(), T)
        sn, sc, st = out.stride()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.krk_greedy_decode(out.data_ptr(), sn, sc, st, N, Cc, T, lens.ctypes.data, 0, 1.0, None,
                                         stream, C.byref(dec)))
        packed = torch.stack([labels, starts, ends, confs.view(torch.int32)]).cpu().numpy()
        cnt = counts.cpu().tolist()
    lab, sta, end = packed[0].tolist(), packed[1].tolist(), packed[2].tolist()
    cf = packed[3].view(np.float32).tolist()
    return [list(zip(lab[n][:k], sta[n][:k], end[n][:k], cf[n][:k])) for n, k in enumerate(cnt)]
