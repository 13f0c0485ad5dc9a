"""
Data-parallel sharding of text lines over the GPUs of one node, and the single exchange step of
the path: a gather of the decoded label sequences (RCCL over xGMI).

The reference has no distributed code at all -- this module is new.  Lines are
independent units: weights (6 MB) are replicated, each rank (one process per GPU) runs forward +
decode on its shard, and only the COMPACT label tuples travel:

    1. all_gather of a 2-word header per rank  [n_lines, n_tuples]
    2. all_gather of one flat int32 buffer per rank, padded to the largest:
       [counts(n) | olens(n) | labels(k) | starts(k) | ends(k) | confidence bits(k)]

With 7 xGMI links per GPU a <= 5 MB message is latency-bound, so a single all_gather (RCCL picks
the direct algorithm) is used rather than a hand-built ring.  The backend is ``nccl`` (= RCCL on
ROCm) on GPUs and ``gloo`` in the CPU test-suite; the code path is identical.
"""
import os
from typing import Optional, Sequence

import numpy as np
import torch
import torch.distributed as td

from .vgsl import DecodedBatch

__all__ = ['init', 'shard_bounds', 'shard_indices', 'gather_decoded', 'pack_decoded', 'unpack_decoded', 'concat_decoded']


def init(backend: Optional[str] = None):
    """Initialises torch.distributed from the torchrun environment (idempotent)."""
    if td.is_initialized():
        return
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('RANK', '0')            # a plain `python bench.py --force-dist` is a one-rank job
    os.environ.setdefault('WORLD_SIZE', '1')
    # the host driver only supports dmabuf IPC (see the environment notes)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    td.init_process_group(backend=backend, init_method='env://')


def shard_bounds(n_items: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: blocks of ceil(n/world) items."""
    per = -(-n_items // world)
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def shard_indices(widths: Sequence[int], world: int, rank: int) -> np.ndarray:
    """
    Indices of the lines rank `rank` processes when lines have different widths: lines are sorted
    by width and dealt round-robin, so every rank sees the same width mix (similar work, similar
    bucket shapes).  Returned indices are ascending.
    """
    order = np.argsort(np.asarray(widths), kind='stable')
    return np.sort(order[rank::world])


def pack_decoded(batch: DecodedBatch, olens) -> np.ndarray:
    """DecodedBatch -> flat int32 [counts | olens | labels | starts | ends | conf bits] (compacted)."""
    counts = np.asarray(batch.counts, dtype=np.int32)
    n = len(counts)
    t = batch.labels.shape[1] if n else 0
    keep = (np.arange(t)[None, :] < counts[:, None]) if n else np.zeros((0, 0), bool)
    parts = [counts, np.asarray(olens, dtype=np.int32).reshape(-1)]
    for arr in (batch.labels, batch.starts, batch.ends, batch.confs.view(np.int32)):
        parts.append(np.asarray(arr, dtype=np.int32)[keep])
    return np.concatenate(parts).astype(np.int32)


def unpack_decoded(flat: np.ndarray, n: int, k: int) -> tuple[DecodedBatch, np.ndarray]:
    """Inverse of pack_decoded for n lines / k tuples; rows are padded to the longest line."""
    counts = flat[:n]
    olens = flat[n:2 * n]
    body = flat[2 * n:2 * n + 4 * k].reshape(4, k)
    t = int(counts.max()) if n else 0
    out = np.zeros((4, n, max(t, 1)), dtype=np.int32)
    keep = np.arange(max(t, 1))[None, :] < counts[:, None]
    for a in range(4):
        out[a][keep] = body[a]
    return DecodedBatch(out[0], out[1], out[2], out[3].view(np.float32), counts.copy()), olens.copy()


def concat_decoded(batches) -> tuple[DecodedBatch, np.ndarray]:
    """[(DecodedBatch, olens), ...] of one rank -> a single (DecodedBatch, olens): one exchange for all its batches."""
    t = max([b.labels.shape[1] for b, _ in batches] + [1])
    pad = lambda a: np.pad(a, ((0, 0), (0, t - a.shape[1])))   # noqa: E731
    fields = [np.concatenate([pad(getattr(b, f)) for b, _ in batches]) for f in ('labels', 'starts', 'ends', 'confs')]
    return (DecodedBatch(*fields, np.concatenate([np.asarray(b.counts) for b, _ in batches])),
            np.concatenate([np.asarray(o) for _, o in batches]))


def gather_decoded(batch, olens=None, group=None, force: bool = False) -> list[DecodedBatch]:
    """
    All ranks receive every rank's decoded lines, in rank order (`force`: run the collectives even alone).
    `batch` is one DecodedBatch with its `olens`, or a list of (DecodedBatch, olens) pairs -- all batches a rank decoded
    travel in ONE exchange, each packed compactly (only the tuples that exist, not the padded rows).
    """
    if isinstance(batch, (list, tuple)):
        parts = list(batch)
    else:
        parts = [(batch, olens)]
    if not td.is_initialized() or (td.get_world_size(group) == 1 and not force):
        return [concat_decoded(parts)[0]] if len(parts) != 1 else [parts[0][0]]
    world = td.get_world_size(group)
    backend = td.get_backend(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    # [counts | olens | labels | starts | ends | conf bits] of all parts, field by field
    counts = np.concatenate([np.asarray(b.counts, dtype=np.int32) for b, _ in parts])
    packs = [pack_decoded(b, o) for b, o in parts]
    ns = [len(b.counts) for b, _ in parts]
    ks = [int(np.sum(b.counts)) for b, _ in parts]
    n, k = int(sum(ns)), int(sum(ks))
    fields = [counts, np.concatenate([p[m:2 * m] for p, m in zip(packs, ns)])]
    for a in range(4):
        fields.append(np.concatenate([p[2 * m + a * kk:2 * m + (a + 1) * kk] for p, m, kk in zip(packs, ns, ks)]))
    flat = np.concatenate(fields).astype(np.int32)
    head = torch.tensor([n, k], dtype=torch.int64, device=dev)
    heads = [torch.empty_like(head) for _ in range(world)]
    td.all_gather(heads, head, group=group)
    sizes = [(int(h[0]), int(h[1])) for h in heads]
    longest = max(2 * a + 4 * b for a, b in sizes)
    buf = torch.zeros(max(longest, 1), dtype=torch.int32, device=dev)
    buf[:flat.size] = torch.from_numpy(flat).to(dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    td.all_gather(bufs, buf, group=group)
    return [unpack_decoded(b.cpu().numpy(), a, c)[0] for b, (a, c) in zip(bufs, sizes)]
