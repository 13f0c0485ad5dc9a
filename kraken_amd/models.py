"""
Legacy recogniser object, backed by the fused HIP path.

Mirrors ``kraken.lib.models.TorchSeqRecognizer`` / ``load_any`` (reference
kraken/lib/models.py:31-185): ``forward`` (:93-119), ``predict`` (:121-136),
``predict_string`` (:138-149), ``predict_labels`` (:151-158) and the attributes the legacy
``rpred``/``mm_rpred`` generators read (``.nn``, ``.codec``, ``.seg_type``,
``.one_channel_mode``, ``.outputs`` -- SURVEY.md section 8b, seam B5).

``predict*`` call ``krk_recognize`` once (forward + softmax + best-path decode on the
GPU); only compact label tuples cross PCIe.  ``.outputs`` -- the full (N, C, T) softmax the
reference always copies to the host -- is materialised lazily on first access.
"""
from os import PathLike
from os.path import abspath, expanduser, expandvars
from typing import Optional, Union

import torch

from . import ctc_decoder as _ctc
from .vgsl import TorchVGSLModel

__all__ = ['TorchSeqRecognizer', 'load_any']


class KrakenInputException(Exception):
    pass


class KrakenInvalidModelException(Exception):
    pass


try:
    from kraken.lib.exceptions import KrakenInputException, KrakenInvalidModelException  # type: ignore # noqa: F811
except Exception:  # pragma: no cover
    pass


class _ShapeOnly:
    """``.outputs`` of a call that did not materialise the softmax: has the (N, C, T) shape, nothing else."""

    def __init__(self, shape):
        self.shape = tuple(shape)

    def __repr__(self):
        return f'<softmax not materialised, shape {self.shape}; call forward() for the values>'


class TorchSeqRecognizer(object):
    """A wrapper around a TorchVGSLModel for text recognition (GPU only)."""

    def __init__(self, nn: TorchVGSLModel, decoder=_ctc.greedy_decoder, temperature: float = 1.0,
                 train: bool = False, device: str = 'cuda'):
        self.nn = nn
        self.kind = ''
        if train is True:
            raise NotImplementedError('kraken_amd accelerates inference only; training is out of scope')
        self.nn.eval()
        self.codec = self.nn.codec
        self.decoder = decoder
        self.temperature = temperature
        self.train = train
        if nn.model_type and 'recognition' not in nn.model_type:
            raise ValueError(f'Models of type {nn.model_type} are not supported by TorchSeqRecognizer')
        self.one_channel_mode = nn.one_channel_mode
        self.seg_type = nn.seg_type
        self.device = None
        self._probs = None
        self._out_shape = None
        # The legacy recogniser has no config to carry a precision (kraken/lib/models.py:29-79 runs whatever torch's default is: fp32).
        # A model nobody chose an arithmetic for gets what kraken's default precision string '32-true' maps to in
        # prepare_for_inference: the fp32-CLASS split-bf16 plan (|d logit| 2e-5, strings identical on the pinned lines) -- not the
        # exact-f32 plan, which is a quarter as fast (round 6: rpred() on a freshly loaded model ran on it).  set_precision('f32')
        # on the model keeps the exact plan.
        hs = getattr(nn, 'nn', None)
        if hs is not None and hasattr(hs, 'set_precision') and not getattr(hs, 'precision_chosen', True):
            hs.set_precision('32-true')
            hs.precision_chosen = False            # (still nobody's explicit choice: a later prepare_for_inference decides again)
        if device:
            self.to(device)

    def to(self, device):
        """Moves the model; a 'cpu' request is rejected -- this recogniser has no CPU path."""
        if str(device).startswith('cpu'):
            raise ValueError('kraken_amd.TorchSeqRecognizer runs on HIP devices only (got device="cpu")')
        self.device = device
        self.nn.to(device)
        # the plan of a fixed-height model (packed weights on the device) is made now, not inside the first page: engines clone it
        hs = getattr(self.nn, 'nn', None)
        if hs is not None and hasattr(hs, 'plan') and self.nn.input[2] > 0 and torch.cuda.is_available():
            try:
                p = next(self.nn.parameters())
                hs.plan(p.device.index if p.device.index is not None else torch.cuda.current_device())
            except Exception:
                pass                                   # (said where the plan is needed: the first call raises or warns as before)

    # ``outputs``: (N, C, T) float32 numpy array of softmax probabilities, like the reference.
    @property
    def outputs(self):
        if self._probs is None or isinstance(self._probs, _ShapeOnly):
            return self._probs
        if isinstance(self._probs, torch.Tensor):
            self._probs = self._probs.float().cpu().numpy()
        return self._probs

    @outputs.setter
    def outputs(self, v):
        self._probs = v

    def _run(self, line: torch.Tensor, lens, want_probs: bool):
        if self.device:
            line = line.to(self.device)
        c_out = self.nn.output[2] if self.nn.output else 1
        if c_out not in (0, 1):
            raise KrakenInputException('Expected dimension 3 to be 1, actual {}'.format(self.nn.output))
        batch, olens, _, probs = self.nn.nn.recognize(line, lens, temperature=self.temperature, want_probs=want_probs)
        if want_probs:
            self._probs = probs           # device tensor; the host copy happens only if someone reads .outputs
        else:
            # rpred reads only `.outputs.shape[2]` (kraken/rpred.py:232): a shape-only stand-in keeps that contract
            # without writing N*C*T floats per call
            self._probs = _ShapeOnly((line.shape[0], self.nn.output[1] if self.nn.output else 0, int(max(olens)) if len(olens) else 0))
        return batch, olens

    def forward(self, line: torch.Tensor, lens: torch.Tensor = None):
        """(N, C, H, W) lines -> ((N, C, T) softmax ndarray, olens ndarray or None)."""
        _, olens = self._run(line, lens, True)
        return self.outputs, (olens if lens is not None else None)

    def predict(self, line: torch.Tensor, lens: Optional[torch.Tensor] = None):
        """list (per line) of (code point, start, end, confidence) tuples."""
        if self.decoder is not _ctc.greedy_decoder:
            o, olens = self.forward(line, lens)
            return [self.codec.decode(locs) for locs in self.decoder(o, olens)]
        batch, _ = self._run(line, lens, True)       # the legacy callers of predict() read .outputs (kraken/align.py:66-71)
        return self.codec.decode_batch(batch)

    def predict_string(self, line: torch.Tensor, lens: Optional[torch.Tensor] = None) -> list[str]:
        if self.decoder is not _ctc.greedy_decoder or not hasattr(self.codec, 'decode_strings'):
            return [''.join(x[0] for x in rec) for rec in self.predict(line, lens)]
        batch, _ = self._run(line, lens, False)
        return self.codec.decode_strings(batch)

    def predict_labels(self, line: torch.Tensor, lens: torch.Tensor = None):
        """list (per line) of (label, start, end, max confidence) tuples."""
        if self.decoder is not _ctc.greedy_decoder:
            o, olens = self.forward(line, lens)
            return self.decoder(o, olens)
        batch, _ = self._run(line, lens, False)
        return batch.tuples()


def load_any(fname: Union[PathLike, str], train: bool = False, device: str = 'cuda') -> TorchSeqRecognizer:
    """Loads a kraken recognition model file (.mlmodel / .safetensors) into a TorchSeqRecognizer."""
    fname = abspath(expandvars(expanduser(str(fname))))
    try:
        nn = TorchVGSLModel.load_model(fname)
    except FileNotFoundError:
        raise
    except Exception as e:
        raise KrakenInvalidModelException('File {} not loadable by any parser.'.format(fname)) from e
    seq = TorchSeqRecognizer(nn, train=train, device=device)
    seq.kind = 'vgsl'
    return seq
