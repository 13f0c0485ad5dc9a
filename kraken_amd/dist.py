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

Failure rule (round 6; the reference turns a failing line into an empty record and keeps going,
kraken/lib/vgsl/rpred.py:104-113, kraken/rpred.py:200-223): every header carries a STATUS word.  A rank
whose engine raised still takes part in every exchange -- its failed batches travel as lines without
tuples, its error text behind them -- so the other ranks never wait for a peer that has left; what
the caller gets is the reference's empty records (``on_error='empty'``) or ONE exception on every
rank (``on_error='raise'``).  A peer that is gone altogether (killed process) ends the collective
with the process group's timeout (``init(timeout_s=...)``, default 300 s -- not NCCL's 10 minutes).
"""
import datetime
import os
import time
from collections import deque
from typing import Optional, Sequence

import numpy as np
import torch
import torch.distributed as td

from .vgsl import DecodedBatch

__all__ = ['init', 'shard_bounds', 'shard_indices', 'gather_decoded', 'pack_decoded', 'unpack_decoded', 'concat_decoded',
           'ShardedRecognizer', 'recognize_lines', 'GatheredBatch', 'parse_cpulist', 'device_numa_nodes', 'rank_cpu_block',
           'rank_cpu_blocks', 'pin_rank_to_cpus', 'pack_results', 'GatheredResults', 'ShardResults', 'ShardError', 'DEFAULT_TIMEOUT_S']

DEFAULT_TIMEOUT_S = 300.0     # a collective whose peer never arrives fails after this long (env KRAKEN_AMD_DIST_TIMEOUT)


class ShardError(RuntimeError):
    """Raised on EVERY rank (``on_error='raise'``) when some rank's share of a sharded job failed; ``.ranks`` = {rank: error text}."""

    def __init__(self, ranks: dict):
        self.ranks = dict(ranks)
        super().__init__('sharded recognition failed on rank(s) ' + '; '.join(f'{r}: {m}' for r, m in sorted(self.ranks.items())))


def init(backend: Optional[str] = None, timeout_s: Optional[float] = None):
    """
    Initialises torch.distributed from the torchrun environment (idempotent).  ``backend``: 'nccl' (= RCCL) is THE backend of the
    product -- picked whenever a HIP device is visible; 'gloo' exists for the host-only test-suite and has to be asked for (or no
    device is visible).  ``timeout_s``: how long a collective waits for a peer that never arrives.
    """
    if td.is_initialized():
        return
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if timeout_s is None:
        timeout_s = float(os.environ.get('KRAKEN_AMD_DIST_TIMEOUT', DEFAULT_TIMEOUT_S))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('RANK', '0')            # a plain `python bench.py --force-dist` is a one-rank job
    os.environ.setdefault('WORLD_SIZE', '1')
    # the host driver only supports dmabuf IPC (see the environment notes)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    td.init_process_group(backend=backend, init_method='env://', timeout=datetime.timedelta(seconds=float(timeout_s)))


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


def rank_cpu_blocks(local_world: int, allowed: Sequence[int], nodes: Optional[Sequence[Optional[int]]] = None,
                    sysfs: str = '/sys') -> list[list[int]]:
    """
    The CPUs every local rank (one process per GPU) keeps for its host side (the codec, record assembly, worker pools): the CPUs of
    ITS GPU's NUMA node -- the pinned tuple buffers and the launch path stay on the socket the device hangs on --, divided between
    the ranks whose GPUs share that node in rank order.  `nodes[r]` = NUMA node of rank r's device (device_numa_nodes).  The choice
    between the NUMA partition and plain contiguous blocks is made ONCE for all ranks (a per-rank fall-back could hand a
    rank CPUs a NUMA-pinned peer already holds): if any rank's node is unknown, or any node has fewer allowed CPUs than ranks, every
    rank takes the r-th contiguous block.  Both partitions deal out every allowed CPU (block sizes differ by at most one).
    """
    allowed = sorted(allowed)
    world = max(1, int(local_world))

    def deal(cpus, k):                       # k near-equal consecutive blocks covering all of `cpus`
        base, extra = divmod(len(cpus), k)
        out, lo = [], 0
        for i in range(k):
            hi = lo + base + (1 if i < extra else 0)
            out.append(list(cpus[lo:hi]))
            lo = hi
        return out

    contiguous = deal(allowed, world)
    if len(allowed) < world:                 # fewer CPUs than ranks: everybody keeps everything
        contiguous = [list(allowed) for _ in range(world)]
    if nodes is None or len(nodes) < world or any(nodes[r] is None for r in range(world)):
        return contiguous
    blocks: list = [None] * world
    aset = set(allowed)
    for node in sorted({nodes[r] for r in range(world)}):
        try:
            with open(os.path.join(sysfs, f'devices/system/node/node{node}/cpulist')) as fh:
                node_cpus = [c for c in parse_cpulist(fh.read()) if c in aset]
        except (OSError, ValueError):
            return contiguous
        peers = [r for r in range(world) if nodes[r] == node]
        if len(node_cpus) < len(peers):
            return contiguous
        for r, b in zip(peers, deal(node_cpus, len(peers))):
            blocks[r] = b
    return blocks


def rank_cpu_block(local_rank: int, local_world: int, allowed: Sequence[int], nodes: Optional[Sequence[Optional[int]]] = None,
                   sysfs: str = '/sys') -> list[int]:
    """``rank_cpu_blocks(...)[local_rank]``: every rank computes the same partition, so blocks never overlap."""
    return rank_cpu_blocks(local_world, allowed, nodes, sysfs)[local_rank]


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


def _empty_decoded(n: int) -> tuple[DecodedBatch, np.ndarray]:
    """n lines without a single tuple (a failed batch travels like this: the reference's empty records)."""
    z = np.zeros((n, 1), dtype=np.int32)
    return DecodedBatch(z, z.copy(), z.copy(), z.view(np.float32).copy(), np.zeros(n, np.int32)), np.zeros(n, np.int32)


def _exchange(flat: np.ndarray, head_words: Sequence[int], group=None, dst: Optional[int] = None):
    """
    The two collectives every exchange of this module consists of: an all_gather of a small int64 header per rank (its first word is
    the number of int32 words of the rank's message, the others are the caller's: line counts, a STATUS word) and one all_gather --
    or, with ``dst``, a gather to that rank -- of the messages, padded to the longest.  Returns (headers as a (world, words) array,
    list of per-rank int32 messages or None on ranks that receive nothing).
    """
    world = td.get_world_size(group)
    backend = td.get_backend(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    head = torch.tensor([int(flat.size)] + [int(w) for w in head_words], dtype=torch.int64, device=dev)
    heads = [torch.empty_like(head) for _ in range(world)]
    td.all_gather(heads, head, group=group)
    heads_np = torch.stack(heads).cpu().numpy()
    longest = int(heads_np[:, 0].max())
    # The payload is already on the host (the engine copies the compact result back for the codec), so a rank's message
    # goes host -> device -> xGMI -> device -> host: one extra PCIe round trip of <= 5 MB per exchange, against the
    # alternative of keeping every slot's device result buffers alive until the end of the run.
    buf = torch.zeros(max(longest, 1), dtype=torch.int32, device=dev)
    if flat.size:
        buf[:flat.size] = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.int32)).to(dev)
    if dst is None:
        bufs = [torch.empty_like(buf) for _ in range(world)]
        td.all_gather(bufs, buf, group=group)
    else:
        me = td.get_rank(group)
        bufs = [torch.empty_like(buf) for _ in range(world)] if me == dst else None
        td.gather(buf, bufs, dst=td.get_global_rank(group, dst) if group is not None else dst, group=group)
        if bufs is None:
            return heads_np, None
    host = torch.stack(bufs).cpu().numpy()
    return heads_np, [host[r, :int(heads_np[r, 0])] for r in range(world)]


def gather_decoded(batch, olens=None, group=None, force: bool = False, status: int = 0) -> list[DecodedBatch]:
    """
    All ranks receive every rank's decoded lines, in rank order (`force`: run the collectives even alone); each returned
    DecodedBatch carries the lines' valid output widths as ``.olens`` and the sender's ``.status`` word (0 = its share ran clean;
    a rank that failed passes ``status != 0`` and still takes part: nobody waits for a peer that raised).
    `batch` is one DecodedBatch with its `olens`, or a list of (DecodedBatch, olens) pairs -- all batches a rank decoded
    travel in ONE exchange, each packed compactly (only the tuples that exist, not the padded rows).
    """
    if isinstance(batch, (list, tuple)):
        parts = list(batch)
    else:
        parts = [(batch, olens)]
    if not parts:                                   # a rank without lines still takes part in the exchange
        parts = [_empty_decoded(0)]
    if not td.is_initialized() or (td.get_world_size(group) == 1 and not force):
        b, o = concat_decoded(parts) if len(parts) != 1 else parts[0]
        b.olens = None if o is None else np.asarray(o)
        b.status = int(status)
        return [b]
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
    heads, msgs = _exchange(flat, (n, k, int(status)), group=group)
    # every rank's message is on this rank's host now; the padded arrays are built when a consumer asks for them (GatheredBatch);
    # the valid output steps per line travel with the tuples (`.olens`)
    out = []
    for r in range(len(msgs)):
        g = GatheredBatch(msgs[r], int(heads[r, 1]), int(heads[r, 2]))
        g.status = int(heads[r, 3])
        out.append(g)
    return out


# ------------------------------------------------------------------------------------------- finished results on the wire
def pack_results(results: Sequence) -> np.ndarray:
    """
    ``rpred.LineResult``s (text, per-code-point first / last output step and confidence, the line's output width) -> one flat int32
    message: [n_chars(n) | out_width(n) | code points(K) | starts(K) | ends(K) | confidence bits(K)].  What a rank sends after it has
    DECODED its own shard: the receivers slice, they never run the codec on another rank's lines.
    """
    n = len(results)
    nc = np.fromiter((len(r.text) for r in results), dtype=np.int32, count=n)
    k = int(nc.sum())
    flat = np.empty(2 * n + 4 * k, dtype=np.int32)
    flat[:n] = nc
    flat[n:2 * n] = np.fromiter((int(r.out_width) for r in results), dtype=np.int32, count=n)
    if k:
        flat[2 * n:2 * n + k] = np.frombuffer(''.join(r.text for r in results).encode('utf-32-le'), dtype='<u4').astype(np.int64).astype(np.int32)
        for a, name in enumerate(('starts', 'ends')):
            flat[2 * n + (a + 1) * k:2 * n + (a + 2) * k] = np.concatenate([np.asarray(getattr(r, name), dtype=np.int32).reshape(-1) for r in results])
        flat[2 * n + 3 * k:2 * n + 4 * k] = np.concatenate([np.asarray(r.confs, dtype=np.float32).reshape(-1) for r in results]).view(np.int32)
    return flat


class ShardResults:
    """
    What ``recognize_lines`` returns: a read-only sequence, one entry per input line.  This rank's own lines are ``LineResult``s; a
    line another rank recognised is a slot in that rank's message and becomes a ``LineResult`` when it is asked for (so a rank's
    host work stays proportional to its shard however many ranks there are); ``None`` where nothing was received.
    """

    def __init__(self, n: int):
        self._own: list = [None] * n
        self._src: list = [None] * n           # (GatheredResults, index in it)

    def __len__(self):
        return len(self._own)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if self._own[i] is None and self._src[i] is not None:
            g, j = self._src[i]
            self._own[i] = g[j]
        return self._own[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class GatheredResults:
    """
    One rank's finished lines as they arrived (``pack_results``' layout); ``self[i]`` builds the i-th ``LineResult`` from slices of
    the message on access (no codec, no per-line work before somebody asks).
    """

    def __init__(self, flat: np.ndarray, n: int):
        self._flat, self._n = flat, int(n)
        nc = np.asarray(flat[:n], dtype=np.int64)
        self._first = np.concatenate([[0], np.cumsum(nc)])
        self._k = int(self._first[-1])
        self.out_widths = np.asarray(flat[n:2 * n], dtype=np.int32)

    def __len__(self):
        return self._n

    def __getitem__(self, i: int):
        from .rpred import LineResult
        if not 0 <= i < self._n:
            raise IndexError(i)
        n, k, lo, hi = self._n, self._k, int(self._first[i]), int(self._first[i + 1])
        body = self._flat[2 * n:]
        text = body[lo:hi].astype('<u4').tobytes().decode('utf-32-le')
        return LineResult(text, body[k + lo:k + hi].copy(), body[2 * k + lo:2 * k + hi].copy(),
                          body[3 * k + lo:3 * k + hi].copy().view(np.float32), int(self.out_widths[i]))

    def __iter__(self):
        return (self[i] for i in range(self._n))


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
        self.errors, self.status, self.rank_errors = [], 0, {}
        self.decoded_lines, self.host_us_per_line = 0, 0.0

    # -- this rank's share of the work -----------------------------------------------------------------------------
    def stream(self, batches, on_batch=None, on_error: str = 'raise') -> list:
        """
        Runs ``batches`` -- an iterable of ``x`` or ``(x, lens)``, device-resident or host tensors -- through the engine with
        all its slots in flight: a freed slot is resubmitted BEFORE the host turns the collected batch into text
        (``on_batch(decoded, olens)``, e.g. the codec).  Returns [(DecodedBatch, olens)] in submission order.

        ``on_error='empty'``: a batch whose submission or collection raises does not end the run (the reference's rule: a line that
        fails becomes an empty record, kraken/lib/vgsl/rpred.py:104-113): its lines come back without tuples, ``self.errors`` gets
        ``(batch index, text)``, ``self.status`` becomes 1 and the following batches run.  'raise' (the benchmark): the exception
        propagates -- call ``gather(..., status=1)`` in the handler if other ranks are waiting in the exchange.
        """
        eng = self.engine
        done: dict = {}
        pending: deque = deque()               # (batch index, lines) of what is in flight, oldest first (the engine collects FIFO)
        self.errors, self.status = [], 0
        prepack = td.is_initialized()          # an exchange will follow: pack now, under the device's work on the next batches

        def failed(idx, n, exc):
            if on_error != 'empty':
                raise exc
            self.errors.append((idx, f'{type(exc).__name__}: {exc}'))
            self.status = 1
            done[idx] = _empty_decoded(n)

        def finish(head, item):
            try:
                if on_batch is not None:
                    on_batch(*item)
                if prepack and item[1] is not None:
                    item[0]._packed = pack_decoded(item[0], item[1])
                done[head[0]] = item
            except Exception as exc:           # noqa: BLE001 -- the consumer raised: this batch is lost, the run is not
                failed(head[0], head[1], exc)

        def collect():
            head = pending.popleft()
            try:
                return head, eng.collect()
            except Exception as exc:           # noqa: BLE001 -- whatever the engine raised for this batch
                failed(head[0], head[1], exc)
                return head, None

        idx = -1
        for idx, b in enumerate(batches):
            x, lens = b if isinstance(b, tuple) else (b, None)
            head, item = collect() if pending and eng.free_slots() == 0 else (None, None)
            try:                               # the freed slot is resubmitted BEFORE the host works on the collected batch
                eng.submit(x, lens)
                pending.append((idx, int(x.shape[0])))
            except Exception as exc:           # noqa: BLE001
                failed(idx, int(x.shape[0]), exc)
            if item is not None:
                finish(head, item)
        while pending:
            head, item = collect()
            if item is not None:
                finish(head, item)
        return [done[i] for i in range(idx + 1)]

    def gather(self, done, force: bool = False, status: Optional[int] = None) -> list:
        """
        The exchange step: every rank's decoded lines to every rank, in rank order (one RCCL all_gather of compact tuples); the
        parts carry the sender's ``.status`` (default: what the last ``stream`` left in ``self.status``).
        """
        t0 = time.perf_counter()
        out = gather_decoded(done, group=self.group, force=force, status=self.status if status is None else status)
        self.gather_ms = 1e3 * (time.perf_counter() - t0)
        return out

    # -- the whole job -------------------------------------------------------------------------------------------
    def recognize_lines(self, lines: Sequence, codec=None, results: str = 'all', root: int = 0, on_error: str = 'empty') -> list:
        """
        ``lines``: the job's line tensors ``(C, H, W_i)`` (host or device), the same list on every rank.  Lines are dealt to
        the ranks width-balanced (``shard_indices``), each rank width-sorts its share into batches of ``batch`` lines
        (zero-padded to the batch's widest line, true widths passed as ``lens``: masked kernels make a line's result
        independent of its batch mates), recognises them and DECODES ITS OWN SHARD (codec + cuts: host work proportional to
        the shard, ``self.decoded_lines`` / ``self.host_us_per_line``); what travels is the finished results (``pack_results``).

        ``results``: 'all' -- every rank gets every line (one all_gather); 'root' -- one gather to rank ``root``, the other
        ranks get their own lines only (``None`` elsewhere); 'local' -- no payload exchange at all (own lines, ``None`` elsewhere).
        The status header is exchanged in every mode.  Returns one ``rpred.LineResult`` per input line, in input order.

        ``on_error``: 'empty' -- the lines of a batch that failed on some rank come back as EMPTY results (text '', no cuts: the
        reference's empty records) and ``self.rank_errors`` says which rank reported what; 'raise' -- after the exchange EVERY
        rank raises one ``ShardError``.  In neither case does a rank wait for a peer that has raised.
        """
        from .rpred import LineResult, _decode_lines
        codec = codec or self.model.codec
        if results not in ('all', 'root', 'local'):
            raise ValueError(f"results must be 'all', 'root' or 'local', not {results!r}")
        widths = np.asarray([int(t.shape[-1]) for t in lines], dtype=np.int64)
        shards = [shard_indices(widths, self.world, r) for r in range(self.world)]
        orders = [sh[np.argsort(widths[sh], kind='stable')] for sh in shards]       # rank r's lines in ITS batch order
        mine = orders[self.rank]
        batches = [mine[lo:lo + self.batch] for lo in range(0, len(mine), self.batch)]

        def padded(idx):
            w = int(widths[idx].max())
            first = lines[int(idx[0])]
            x = torch.zeros((len(idx),) + tuple(first.shape[:-1]) + (w,), dtype=torch.float32, device=first.device)
            for j, i in enumerate(idx):
                x[j, ..., :int(widths[i])] = lines[int(i)]
            return x, widths[idx].astype(np.int32)

        def empty(n):
            z = np.zeros(0, np.int32)
            return [LineResult('', z, z.copy(), z.view(np.float32).copy(), 0) for _ in range(n)]

        t0 = time.perf_counter()
        own: list = []
        try:
            done = self.stream((padded(idx) for idx in batches), on_error='empty')
            for part, olens in done:
                own.extend(_decode_lines(codec, part, olens))
        except Exception as exc:               # noqa: BLE001 -- preparing a batch or the codec failed: the rest of the shard is lost, the exchange is not
            self.errors.append((-1, f'{type(exc).__name__}: {exc}'))
            self.status = 2
            own.extend(empty(len(mine) - len(own)))
        self.decoded_lines = len(own)
        self.host_us_per_line = 1e6 * (time.perf_counter() - t0) / max(1, len(own))
        err = '; '.join(f'batch {i}: {m}' if i >= 0 else m for i, m in self.errors)[:2000]

        out = ShardResults(len(lines))
        for i, res in zip(mine, own):
            out._own[int(i)] = res
        self.rank_errors = {self.rank: err} if self.status else {}
        if td.is_initialized() and self.world > 1:
            t1 = time.perf_counter()
            eb = np.frombuffer(err.encode('utf-8') + b'\0' * (-len(err.encode('utf-8')) % 4), dtype=np.int32)
            payload = np.concatenate([pack_results(own), eb]) if results != 'local' else eb
            heads, msgs = _exchange(payload, (len(own), self.status, eb.size), group=self.group,
                                    dst=root if results == 'root' else None)
            self.gather_ms = 1e3 * (time.perf_counter() - t1)
            for r in range(self.world):
                n_r, st_r, ew_r = int(heads[r, 1]), int(heads[r, 2]), int(heads[r, 3])
                if n_r != len(orders[r]):
                    raise RuntimeError(f'rank {r} returned {n_r} lines for a shard of {len(orders[r])}')
                if st_r:
                    msg = ''
                    if msgs is not None:
                        msg = msgs[r][msgs[r].size - ew_r:].tobytes().rstrip(b'\0').decode('utf-8', 'replace')
                    self.rank_errors[r] = msg or self.rank_errors.get(r, '') or f'status {st_r}'
                if msgs is None or results == 'local' or r == self.rank:
                    continue
                got = GatheredResults(msgs[r][:msgs[r].size - ew_r], n_r)
                for j, i in enumerate(orders[r]):
                    out._src[int(i)] = (got, j)
        if self.rank_errors and on_error == 'raise':
            raise ShardError(self.rank_errors)
        return out

    def close(self):
        self.engine.close()


def recognize_lines(model, lines: Sequence, **kw) -> list:
    """One-call form of ``ShardedRecognizer(model, **kw).recognize_lines(lines)`` (initialises torch.distributed from the launcher's environment)."""
    if 'WORLD_SIZE' in os.environ and not td.is_initialized():
        init()
    call = {k: kw.pop(k) for k in ('codec', 'results', 'root', 'on_error') if k in kw}
    sr = ShardedRecognizer(model, **kw)
    try:
        return sr.recognize_lines(lines, **call)
    finally:
        sr.close()
