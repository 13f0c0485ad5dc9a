"""
kraken_amd -- MI355X-native (gfx950) implementation of kraken's line-recognition hot path.

The package mirrors the reference's module layout for the path only:

    kraken_amd.vgsl         TorchVGSLModel, parse_vgsl        (kraken.lib.vgsl)
    kraken_amd.models       TorchSeqRecognizer, load_any      (kraken.lib.models)
    kraken_amd.ctc_decoder  greedy_decoder                    (kraken.lib.ctc_decoder)
    kraken_amd.codec        PytorchCodec                      (kraken.lib.codec)
    kraken_amd.rpred        rpred, mm_rpred                   (kraken.rpred)
    kraken_amd.blla         compute_segmentation_map          (kraken.lib.vgsl.spred, forward only)
    kraken_amd.dist         line sharding + gather over RCCL  (new; no reference analogue)

All arithmetic of the forward pass and the CTC decode lives in ``csrc/*.hip`` behind the
C ABI declared in ``include/kraken_amd.h``; importing the package does not need a GPU, running
the path does.
"""
__version__ = '0.1.0'

import os as _os

# The pipelined engine keeps several batches in flight on separate HIP streams.  The HIP runtime maps
# streams onto 4 hardware queues by default; with more streams than queues unrelated batches serialise
# behind each other (measured: 4 slots 51.6k -> 66.6k lines/s).  Must be set before the runtime starts.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from .codec import PytorchCodec  # noqa: F401
from .vgsl import TorchVGSLModel, parse_vgsl  # noqa: F401

__all__ = ['TorchVGSLModel', 'PytorchCodec', 'parse_vgsl', '__version__']
