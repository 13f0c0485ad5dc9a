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

__all__ = ['init', 'shard_bounds', 'shard_indices', 'gather_decoded', 'pack_decoded', 'unpack_decoded', 'concat_decoded',
           'ShardedRecognizer', 'recognize_lines', 'GatheredBatch', 'parse_cpulist', 'device_numa_nodes', 'rank_cpu_block', 'pin_rank_to_cpus']


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


def parse_cpulist(text: str) -> list[int]:
    """'0-3,8,10-11' (the kernel's cpulist format) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus: list[int] = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_numa_nodes(pci_ids: Sequence[Optional[str]], sysfs: str = '/sys') -> list[Optional[int]]:
    """NUMA node of every device from /sys/bus/pci/devices/<dddd:bb:dd.f>/numa_node (None: unknown, -1 in sysfs, or no such file)."""
    nodes: list[Optional[int]] = []
    for pid in pci_ids:
        node = None
        if pid:
            try:
                with open(os.path.join(sysfs, 'bus/pci/devices', pid.lower(), 'numa_node')) as fh:
                    v = int(fh.read().strip())
                node = v if v >= 0 else None
            except (OSError, ValueError):
                node = None
        nodes.append(node)
    return nodes


def rank_cpu_block(local_rank: int, local_world: int, allowed: Sequence[int], nodes: Optional[Sequence[Optional[int]]] = None,
                   sysfs: str = '/sys') -> list[int]:
    """
    The CPUs rank `local_rank` (one process per GPU) keeps for its host side (the codec, record assembly, worker pools): the CPUs of
    ITS GPU's NUMA node -- the pinned tuple buffers and the launch path stay on the socket the device hangs on --, divided between
    the ranks whose GPUs share that node in rank order.  `nodes[r]` = NUMA node of rank r's device (device_numa_nodes); where it is
    unknown, or a node's cpulist holds none of the `allowed` CPUs, the rank falls back to the r-th contiguous block of `allowed`.
    Every rank computes the same partition, so blocks never overlap between ranks of one kind (NUMA blocks / fallback blocks).
    """
    allowed = sorted(allowed)
    per = max(1, len(allowed) // max(1, local_world))
    fallback = allowed[local_rank * per:(local_rank + 1) * per] or allowed
    node = nodes[local_rank] if nodes is not None and local_rank < len(nodes) else None
    if node is None:
        return fallback
    try:
        with open(os.path.join(sysfs, f'devices/system/node/node{node}/cpulist')) as fh:
            node_cpus = [c for c in parse_cpulist(fh.read()) if c in set(allowed)]
    except (OSError, ValueError):
        return fallback
    peers = [r for r in range(local_world) if r < len(nodes) and nodes[r] == node]
    share = len(node_cpus) // max(1, len(peers))
    if share < 1:
        return fallback
    k = peers.index(local_rank)
    return node_cpus[k * share:(k + 1) * share]


def _device_pci_ids(n: int) -> list[Optional[str]]:
    ids: list[Optional[str]] = []
    for i in range(n):
        try:
            p = torch.cuda.get_device_properties(i)
            ids.append(f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0')
        except Exception:       # no device i / a torch without the PCI fields
            ids.append(None)
    return ids


def pin_rank_to_cpus(local_rank: int, local_world: int, sysfs: str = '/sys') -> dict:
    """
    One rank per GPU must not mean N ranks x (intra-op threads + worker pools) on every core: the rank keeps rank_cpu_block's CPUs
    (`os.sched_setaffinity`) and caps torch's intra-op threads to them.  Returns {'cpus': count, 'numa_node': node or None,
    'first_cpu': ..}.
    """
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    nodes = device_numa_nodes(_device_pci_ids(local_world), sysfs) if torch.cuda.is_available() else [None] * local_world
    mine = rank_cpu_block(local_rank, local_world, allowed, nodes, sysfs)
    try:
        os.sched_setaffinity(0, mine)
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(max(1, min(8, len(mine))))
    return {'cpus': len(mine), 'numa_node': nodes[local_rank] if local_rank < len(nodes) else None, 'first_cpu': mine[0] if mine else None}


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


def _tuple_index(counts: np.ndarray, t: int) -> np.ndarray:
    """Flat index (row * t + column) of every existing tuple of a padded (n, t) result, line by line: O(tuples), not O(n * t)."""
    c = counts.astype(np.int64)
    k = int(c.sum())
    first = np.cumsum(c) - c                                   # position of a line's first tuple in the compact order
    return np.arange(k, dtype=np.int64) + np.repeat(np.arange(len(c), dtype=np.int64) * t - first, c)


def pack_decoded(batch: DecodedBatch, olens) -> np.ndarray:
    """DecodedBatch -> flat int32 [counts | olens | labels | starts | ends | conf bits] (compacted)."""
    counts = np.asarray(batch.counts, dtype=np.int32)
    n = len(counts)
    t = batch.labels.shape[1] if n else 0
    lin = _tuple_index(counts, t) if n else np.zeros(0, np.int64)
    k = len(lin)
    flat = np.empty(2 * n + 4 * k, dtype=np.int32)
    flat[:n] = counts
    flat[n:2 * n] = np.asarray(olens, dtype=np.int32).reshape(-1)
    for a, arr in enumerate((batch.labels, batch.starts, batch.ends, np.asarray(batch.confs).view(np.int32))):
        np.take(np.ascontiguousarray(arr).reshape(-1), lin, out=flat[2 * n + a * k:2 * n + (a + 1) * k])
    return flat


def unpack_decoded(flat: np.ndarray, n: int, k: int) -> tuple[DecodedBatch, np.ndarray]:
    """Inverse of pack_decoded for n lines / k tuples; rows are padded to the longest line."""
    counts = np.asarray(flat[:n], dtype=np.int32)
    olens = flat[n:2 * n]
    body = flat[2 * n:2 * n + 4 * k].reshape(4, k)
    t = max(int(counts.max()) if n else 0, 1)
    out = np.zeros((4, n * t), dtype=np.int32)
    if k:
        out[:, _tuple_index(counts, t)] = body
    out = out.reshape(4, n, t)
    return DecodedBatch(out[0], out[1], out[2], out[3].view(np.float32), counts.copy()), olens.copy()


class GatheredBatch(DecodedBatch):
    """
    One rank's lines as they arrived in the exchange: the compact int32 message (`pack_decoded`'s layout) plus `counts` and `olens`,
    which are slices of it.  The padded `(n, t)` arrays of a `DecodedBatch` -- `labels`, `starts`, `ends`, `confs` -- are built from
    the message on first access (`unpack_decoded`): a consumer that only routes or counts lines never pays for them, and on an N-rank
    job every rank would otherwise unpack N ranks' lines inside the exchange.
    """

    def __init__(self, flat: np.ndarray, n: int, k: int):
        self._flat, self._n, self._k, self._full = flat, int(n), int(k), None
        self.counts = np.asarray(flat[:n], dtype=np.int32).copy()
        self.olens = np.asarray(flat[n:2 * n], dtype=np.int32).copy()

    def _unpacked(self) -> DecodedBatch:
        if self._full is None:
            self._full = unpack_decoded(self._flat, self._n, self._k)[0]
        return self._full

    labels = property(lambda self: self._unpacked().labels)
    starts = property(lambda self: self._unpacked().starts)
    ends = property(lambda self: self._unpacked().ends)
    confs = property(lambda self: self._unpacked().confs)


def concat_decoded(batches) -> tuple[DecodedBatch, np.ndarray]:
    """[(DecodedBatch, olens), ...] of one rank -> a single (DecodedBatch, olens): one exchange for all its batches."""
    t = max([b.labels.shape[1] for b, _ in batches] + [1])
    pad = lambda a: np.pad(a, ((0, 0), (0, t - a.shape[1])))   # noqa: E731
    fields = [np.concatenate([pad(getattr(b, f)) for b, _ in batches]) for f in ('labels', 'starts', 'ends', 'confs')]
    return (DecodedBatch(*fields, np.concatenate([np.asarray(b.counts) for b, _ in batches])),
            np.concatenate([np.asarray(o) for _, o in batches]))


def gather_decoded(batch, olens=None, group=None, force: bool = False) -> list[DecodedBatch]:
    """
    All ranks receive every rank's decoded lines, in rank order (`force`: run the collectives even alone); each returned
    DecodedBatch carries the lines' valid output widths as ``.olens``.
    `batch` is one DecodedBatch with its `olens`, or a list of (DecodedBatch, olens) pairs -- all batches a rank decoded
    travel in ONE exchange, each packed compactly (only the tuples that exist, not the padded rows).
    """
    if isinstance(batch, (list, tuple)):
        parts = list(batch)
    else:
        parts = [(batch, olens)]
    if not parts:                                   # a rank without lines still takes part in the exchange
        z = np.zeros((0, 1), dtype=np.int32)
        parts = [(DecodedBatch(z, z.copy(), z.copy(), z.view(np.float32).copy(), np.zeros(0, np.int32)), np.zeros(0, np.int32))]
    if not td.is_initialized() or (td.get_world_size(group) == 1 and not force):
        b, o = concat_decoded(parts) if len(parts) != 1 else parts[0]
        b.olens = None if o is None else np.asarray(o)
        return [b]
    world = td.get_world_size(group)
    backend = td.get_backend(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    # [counts | olens | labels | starts | ends | conf bits] of all parts, field by field
    counts = np.concatenate([np.asarray(b.counts, dtype=np.int32) for b, _ in parts])
    # (ShardedRecognizer.stream packs a batch the moment it is collected, while the device works on the next ones)
    packs = [getattr(b, '_packed', None) if getattr(b, '_packed', None) is not None else pack_decoded(b, o) for b, o in parts]
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
    # The tuples are already on the host (the engine copies the compact result back for the codec), so a rank's message
    # goes host -> device -> xGMI -> device -> host: one extra PCIe round trip of <= 5 MB per exchange, against the
    # alternative of keeping every slot's device result buffers alive until the end of the run.
    buf = torch.zeros(max(longest, 1), dtype=torch.int32, device=dev)
    buf[:flat.size] = torch.from_numpy(flat).to(dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    td.all_gather(bufs, buf, group=group)
    # every rank's message is on this rank's host now; the padded arrays are built when a consumer asks for them (GatheredBatch);
    # the valid output steps per line travel with the tuples (`.olens`)
    host = torch.stack(bufs).cpu().numpy()
    return [GatheredBatch(host[r], a, c) for r, (a, c) in enumerate(sizes)]


class ShardedRecognizer:
    """
    Recognition of independent text lines on the GPUs of one node -- BASELINE config 3 ("lines sharded over xGMI, RCCL gather
    of decoded strings") as a product call.  One instance per rank (= per GPU, one process each): weights replicated, a
    pipelined ``RecognitionEngine`` per rank, no data-path collective; the single exchange is ``gather``.

        sr = ShardedRecognizer(model)                       # every rank, after dist.init()
        texts = sr.recognize_lines(lines)                   # the same list of (C, H, W) line tensors on every rank
                                                            # -> list of LineResult in INPUT order, on every rank

    ``engine_factory`` replaces the device engine in the CPU test-suite (the plumbing is identical).
    """

    def __init__(self, model, device: Optional[int] = None, batch: int = 256, slots: int = 3, max_width: int = 2400,
                 temperature: float = 1.0, group=None, engine_factory=None):
        self.model, self.batch, self.group = model, int(batch), group
        self.rank = td.get_rank(group) if td.is_initialized() else 0
        self.world = td.get_world_size(group) if td.is_initialized() else 1
        if engine_factory is None:
            from .engine import RecognitionEngine
            if device is None:
                device = torch.cuda.current_device()
            self.engine = RecognitionEngine(model, device=device, max_batch=self.batch, max_width=max_width, slots=slots,
                                            temperature=temperature)
        else:
            self.engine = engine_factory()
        self.gather_ms = 0.0

    # -- this rank's share of the work -----------------------------------------------------------------------------
    def stream(self, batches, on_batch=None) -> list:
        """
        Runs ``batches`` -- an iterable of ``x`` or ``(x, lens)``, device-resident or host tensors -- through the engine with
        all its slots in flight: a freed slot is resubmitted BEFORE the host turns the collected batch into text
        (``on_batch(decoded, olens)``, e.g. the codec).  Returns [(DecodedBatch, olens)] in submission order.
        """
        eng, done = self.engine, []

        prepack = td.is_initialized()          # an exchange will follow: pack now, under the device's work on the next batches

        def finish(item):
            if on_batch is not None:
                on_batch(*item)
            if prepack and item[1] is not None:
                item[0]._packed = pack_decoded(item[0], item[1])
            done.append(item)

        for b in batches:
            x, lens = b if isinstance(b, tuple) else (b, None)
            item = eng.collect() if eng.free_slots() == 0 else None
            eng.submit(x, lens)
            if item is not None:
                finish(item)
        while eng.free_slots() < len(eng.slots):
            finish(eng.collect())
        return done

    def gather(self, done, force: bool = False) -> list:
        """The exchange step: every rank's decoded lines to every rank, in rank order (one RCCL all_gather of compact tuples)."""
        import time
        t0 = time.perf_counter()
        out = gather_decoded(done, group=self.group, force=force)
        self.gather_ms = 1e3 * (time.perf_counter() - t0)
        return out

    # -- the whole job -------------------------------------------------------------------------------------------
    def recognize_lines(self, lines: Sequence, codec=None) -> list:
        """
        ``lines``: the job's line tensors ``(C, H, W_i)`` (host or device), the same list on every rank.  Lines are dealt to
        the ranks width-balanced (``shard_indices``), each rank width-sorts its share into batches of ``batch`` lines
        (zero-padded to the batch's widest line, true widths passed as ``lens``: masked kernels make a line's result
        independent of its batch mates), recognises them, and the decoded tuples of all ranks are gathered.  Returns one
        ``rpred.LineResult`` per input line, in input order, on every rank.
        """
        from .rpred import _decode_lines
        codec = codec or self.model.codec
        widths = np.asarray([int(t.shape[-1]) for t in lines], dtype=np.int64)
        shards = [shard_indices(widths, self.world, r) for r in range(self.world)]
        mine = shards[self.rank]
        order = [mine[np.argsort(widths[mine], kind='stable')]] if len(mine) else []
        batches = [order[0][lo:lo + self.batch] for lo in range(0, len(mine), self.batch)] if len(mine) else []

        def padded(idx):
            w = int(widths[idx].max())
            first = lines[int(idx[0])]
            x = torch.zeros((len(idx),) + tuple(first.shape[:-1]) + (w,), dtype=torch.float32, device=first.device)
            for j, i in enumerate(idx):
                x[j, ..., :int(widths[i])] = lines[int(i)]
            return x, widths[idx].astype(np.int32)

        done = self.stream(padded(idx) for idx in batches)
        parts = self.gather(done)                                   # rank r's lines, in ITS batch order
        results = [None] * len(lines)
        for r, part in enumerate(parts):
            rank_order = shards[r][np.argsort(widths[shards[r]], kind='stable')]
            if len(part.counts) != len(rank_order):
                raise RuntimeError(f'rank {r} returned {len(part.counts)} lines for a shard of {len(rank_order)}')
            olens = getattr(part, 'olens', None)
            for i, res in zip(rank_order, _decode_lines(codec, part, olens if olens is not None else np.zeros(len(rank_order), np.int32))):
                results[int(i)] = res
        return results

    def close(self):
        self.engine.close()


def recognize_lines(model, lines: Sequence, **kw) -> list:
    """One-call form of ``ShardedRecognizer(model, **kw).recognize_lines(lines)`` (initialises torch.distributed from the launcher's environment)."""
    if 'WORLD_SIZE' in os.environ and not td.is_initialized():
        init()
    sr = ShardedRecognizer(model, **kw)
    try:
        return sr.recognize_lines(lines)
    finally:
        sr.close()
