"""
The N>1 path on CPU: two processes, gloo backend -- sharding of lines and the gather of decoded
label sequences (kraken_amd/dist.py).  The same code runs over RCCL (backend "nccl") on GPUs.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from kraken_amd import dist as kdist
from kraken_amd.vgsl import DecodedBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_batch(rank, n, t=12):
    rng = np.random.RandomState(100 + rank)
    counts = rng.randint(0, t + 1, size=n).astype(np.int32)
    lab = rng.randint(1, 200, size=(n, t)).astype(np.int32)
    st = np.sort(rng.randint(0, 150, size=(n, t)), axis=1).astype(np.int32)
    en = st + rng.randint(0, 3, size=(n, t)).astype(np.int32)
    cf = rng.rand(n, t).astype(np.float32)
    return DecodedBatch(lab, st, en, cf, counts), rng.randint(1, 151, size=n).astype(np.int32)


def test_pack_unpack_roundtrip():
    for n in (0, 1, 7):
        b, ol = _fake_batch(0, n)
        flat = kdist.pack_decoded(b, ol)
        b2, ol2 = kdist.unpack_decoded(flat, n, int(b.counts.sum()))
        assert b2.tuples() == b.tuples()
        assert ol2.tolist() == ol.tolist()


def test_concat_decoded_keeps_every_line():
    parts = [_fake_batch(r, n, t) for r, (n, t) in enumerate(((5, 12), (0, 3), (7, 20)))]
    b, ol = kdist.concat_decoded(parts)
    assert b.tuples() == sum((p[0].tuples() for p in parts), [])
    assert ol.tolist() == sum((p[1].tolist() for p in parts), [])


def test_shard_bounds_cover_everything_once():
    for n, world in ((16384, 8), (10, 4), (3, 8), (0, 2)):
        seen = []
        for r in range(world):
            lo, hi = kdist.shard_bounds(n, world, r)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    widths = np.random.RandomState(0).randint(400, 2401, size=1000)
    parts = [kdist.shard_indices(widths, 8, r) for r in range(8)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(1000))
    means = [widths[p].mean() for p in parts]
    assert max(means) - min(means) < 20          # every rank sees the same width mix


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from kraken_amd import dist as kdist
from tests.test_dist_cpu import _fake_batch
kdist.init(backend='gloo')
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
lo, hi = kdist.shard_bounds(11, world, rank)
batch, olens = _fake_batch(rank, hi - lo)
got = kdist.gather_decoded(batch, olens)
assert len(got) == world
for r in range(world):
    l, h = kdist.shard_bounds(11, world, r)
    want, _ = _fake_batch(r, h - l)
    assert got[r].tuples() == want.tuples(), (rank, r)
# every batch a rank decoded travels in ONE exchange (bench.py: all steps of the timed region)
mine = [_fake_batch(10 * rank + j, n, t) for j, (n, t) in enumerate(((4, 12), (0, 5), (6, 30 + rank)))]
got = kdist.gather_decoded(mine)
assert len(got) == world
for r in range(world):
    want = [_fake_batch(10 * r + j, n, t) for j, (n, t) in enumerate(((4, 12), (0, 5), (6, 30 + r)))]
    assert got[r].tuples() == sum((w[0].tuples() for w in want), []), (rank, r)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print('rank', rank, 'ok')
'''


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.timeout(180)
def test_gather_decoded_two_ranks_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=170)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f'rank {rank} ok' in o


# ---------------------------------------------------------------------------- sharded recognition as a product call
SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from kraken_amd import dist as kdist
from kraken_amd.codec import PytorchCodec
from kraken_amd.vgsl import DecodedBatch
kdist.init(backend='gloo')
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])

class Stub:                                   # RecognitionEngine surface; a line decodes to chr(width % 50 + 1) x (1 + line sum % 3)
    def __init__(self):
        self.slots, self.q = [0, 0], []
    def free_slots(self):
        return 2 - len(self.q)
    def submit(self, x, lens=None):
        assert x.shape[0] == len(lens) and x.shape[-1] == int(max(lens))
        for j, l in enumerate(lens):          # zero padding to the right of every line, the line itself intact
            assert float(x[j, ..., l:].abs().sum()) == 0.0
        self.q.append((x.clone(), np.asarray(lens)))
    def collect(self):
        x, lens = self.q.pop(0)
        n = len(lens)
        reps = np.array([1 + int(round(float(x[j].sum()))) % 3 for j in range(n)], np.int32)
        lab = np.repeat((lens % 50 + 1).astype(np.int32)[:, None], 3, 1)
        st = np.tile(np.arange(3, dtype=np.int32) * 2, (n, 1))
        return DecodedBatch(lab, st, st + 1, np.full((n, 3), 0.25, np.float32), reps), (lens // 8).astype(np.int32)
    def close(self):
        pass

rng = np.random.RandomState(5)
widths = rng.randint(40, 400, size=int(sys.argv[2]))
lines = [torch.full((1, 4, int(w)), float(i % 7) / (4 * int(w))) for i, w in enumerate(widths)]     # line sum = i % 7
codec = PytorchCodec({chr(0x40 + k): [k] for k in range(1, 52)})
model = type('M', (), {'codec': codec})()
sr = kdist.ShardedRecognizer(model, batch=16, engine_factory=Stub)
res = sr.recognize_lines(lines)
assert len(res) == len(lines)
for i, (r, w) in enumerate(zip(res, widths)):
    assert r.text == chr(0x40 + w % 50 + 1) * (1 + (i % 7) % 3), (rank, i, r.text)
    assert r.out_width == w // 8 and r.starts.tolist() == [0, 2, 4][:len(r.text)]
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print('rank', rank, 'ok', len(res))
'''


@pytest.mark.timeout(180)
@pytest.mark.parametrize('n_lines', [101, 1])
def test_recognize_lines_sharded_over_two_ranks_returns_input_order(tmp_path, n_lines):
    """BASELINE config 3 as a product call: shard -> per-rank pipelined engine -> gather -> results in input order on every rank."""
    script = tmp_path / 'worker.py'
    script.write_text(SHARD_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(n_lines)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=170)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f'rank {rank} ok {n_lines}' in o


@pytest.mark.timeout(240)
def test_bench_launches_its_own_ranks_and_refuses_missing_devices():
    """
    `python bench.py --gpus N` without torchrun must be an N-rank job: the launcher, the per-rank streaming loop
    and the gather run here with 2 gloo ranks and a host stub in place of the device engine.
    """
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                        '--stub-engine', '--batch', '8', '--width', '64'], env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['n_gpus'] == 2 and out['ranks_in_collective'] == 2 and out['gathered_lines'] == 2 * 5 * 8
    assert out['config']['parallelism'] == 'dp2' and 'STUB' in out['data'] and out['gather_ms'] > 0
    # every rank keeps its own block of the host's CPUs (no 8 ranks x all cores)
    assert 1 <= out['host_cpus_per_rank']['cpus'] <= max(1, len(os.sched_getaffinity(0)) // 2)
    # ... and its own clock in the line (tools/scale_preflight.sh logs them per rank)
    assert len(out['per_rank']['lines_per_s']) == 2 and all(v > 0 for v in out['per_rank']['lines_per_s'])
    assert min(out['per_rank']['lines_per_s']) * 2 >= out['value'] * 0.999
    # without the stub there is no device here: the launcher refuses instead of running one rank labelled as two
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env,
                           capture_output=True, text=True, timeout=200)
        assert r.returncode != 0 and 'refusing' in r.stderr and not r.stdout.strip()
    # a launcher's WORLD_SIZE that contradicts --gpus is an error, not a relabelled run
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stub-engine'],
                       env=dict(env, WORLD_SIZE='1', RANK='0'), capture_output=True, text=True, timeout=200)
    assert r.returncode != 0


def _fake_sysfs(root, gpu_nodes, node_cpus):
    """A /sys stand-in: PCI devices 0000:<10+i>:00.0 with numa_node files, NUMA nodes with cpulists."""
    ids = []
    for i, node in enumerate(gpu_nodes):
        pid = f'0000:{0x10 + i:02x}:00.0'
        d = root / 'bus/pci/devices' / pid
        d.mkdir(parents=True)
        (d / 'numa_node').write_text(f'{node}\n')
        ids.append(pid)
    for node, text in node_cpus.items():
        d = root / f'devices/system/node/node{node}'
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(text + '\n')
    return ids


def test_rank_cpu_blocks_follow_the_numa_node_of_the_ranks_gpu(tmp_path):
    """Eight GPUs on two sockets (the MI355X node's shape: four GPUs per socket, hyper-thread siblings in the second half of the
    cpulist): every rank keeps CPUs of ITS socket, the four ranks of a socket share them without overlap, all CPUs are dealt."""
    from kraken_amd import dist as kdist
    assert kdist.parse_cpulist('0-3,8,10-11') == [0, 1, 2, 3, 8, 10, 11]
    ids = _fake_sysfs(tmp_path, [0, 0, 0, 0, 1, 1, 1, 1], {0: '0-63,128-191', 1: '64-127,192-255'})
    nodes = kdist.device_numa_nodes(ids, str(tmp_path))
    assert nodes == [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = list(range(256))
    blocks = [kdist.rank_cpu_block(r, 8, allowed, nodes, str(tmp_path)) for r in range(8)]
    node0 = set(kdist.parse_cpulist('0-63,128-191'))
    for r, b in enumerate(blocks):
        assert len(b) == 32 and (set(b) <= node0) == (r < 4)
    assert sorted(c for b in blocks for c in b) == allowed
    # a restricted affinity mask (a container that may use half of each socket) is respected
    half = [c for c in allowed if c % 2 == 0]
    blocks = [kdist.rank_cpu_block(r, 8, half, nodes, str(tmp_path)) for r in range(8)]
    assert all(len(b) == 16 and set(b) <= set(half) for b in blocks) and len({c for b in blocks for c in b}) == 128
    # an interleaved topology (GPU i on socket i % 2) still gives disjoint per-socket blocks
    ids2 = _fake_sysfs(tmp_path / 'b', [0, 1, 0, 1], {0: '0-7', 1: '8-15'})
    nodes2 = kdist.device_numa_nodes(ids2, str(tmp_path / 'b'))
    assert [kdist.rank_cpu_block(r, 4, range(16), nodes2, str(tmp_path / 'b')) for r in range(4)] == \
        [[0, 1, 2, 3], [8, 9, 10, 11], [4, 5, 6, 7], [12, 13, 14, 15]]


def test_rank_cpu_blocks_fall_back_to_contiguous_blocks(tmp_path):
    """numa_node -1 (a single-socket box, a VM), a missing sysfs entry, a device torch cannot describe: plain contiguous blocks."""
    from kraken_amd import dist as kdist
    ids = _fake_sysfs(tmp_path, [-1, -1], {})
    assert kdist.device_numa_nodes(ids + [None, '0000:ff:00.0'], str(tmp_path)) == [None, None, None, None]
    assert [kdist.rank_cpu_block(r, 2, range(8), [None, None], str(tmp_path)) for r in range(2)] == [[0, 1, 2, 3], [4, 5, 6, 7]]
    # a node whose cpulist holds none of the allowed CPUs
    ids = _fake_sysfs(tmp_path / 'c', [0, 0], {0: '100-103'})
    nodes = kdist.device_numa_nodes(ids, str(tmp_path / 'c'))
    assert kdist.rank_cpu_block(1, 2, range(8), nodes, str(tmp_path / 'c')) == [4, 5, 6, 7]
    info = kdist.pin_rank_to_cpus(0, 1, str(tmp_path))      # this process: one rank keeps everything it may run on
    assert info['cpus'] == len(os.sched_getaffinity(0))


@pytest.mark.timeout(300)
def test_scale_preflight_script_on_the_host_stub(tmp_path):
    """tools/scale_preflight.sh (what runs first on a multi-GPU node) with the host stub and gloo: N = 1 and 2 launched the way the
    driver launches them (torch.distributed.run), every line checked for its rank count and backend, per-rank rates logged."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(PREFLIGHT_STUB='1', PREFLIGHT_RANKS='2', PREFLIGHT_OUT=str(tmp_path), GRAFT_REPO_ROOT=ROOT)
    r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'scale_preflight.sh'), '3'], env=env, capture_output=True, text=True,
                       timeout=280)
    summary = (tmp_path / 'summary.txt').read_text()
    assert r.returncode == 0, r.stdout + r.stderr + summary
    assert 'N=1: ok' in summary and 'N=2: ok' in summary and 'preflight passed' in summary
    import json
    two = json.loads((tmp_path / 'scale_2.json').read_text().strip().splitlines()[-1])
    assert two['ranks_in_collective'] == 2 and two['collective_backend'] == 'gloo' and len(two['per_rank']['gather_ms']) == 2


def test_gathered_batches_unpack_lazily_and_prepacked_batches_are_reused():
    """The exchange hands every rank's lines over as the compact message (`GatheredBatch`): counts and olens at once, the padded
    arrays on first access -- equal to what was packed; a batch `ShardedRecognizer.stream` packed while the device worked is not
    packed again in the exchange."""
    from kraken_amd import dist as kdist
    parts = [_fake_batch(r, n) for r, n in enumerate((5, 1, 17))]
    flat = [kdist.pack_decoded(b, o) for b, o in parts]
    for (b, o), f in zip(parts, flat):
        g = kdist.GatheredBatch(f, len(b.counts), int(np.sum(b.counts)))
        assert g._full is None and g.counts.tolist() == list(b.counts) and g.olens.tolist() == list(o)
        assert g._full is None                      # counting lines did not build the padded arrays
        assert g.tuples() == b.tuples() and g._full is not None
        for i, c in enumerate(b.counts):
            assert np.array_equal(g.labels[i, :c], b.labels[i, :c]) and np.array_equal(g.confs[i, :c], b.confs[i, :c])
    # a pre-packed batch: gather_decoded must take the stored message (a poisoned one shows up in the result)
    b, o = parts[0]
    b._packed = flat[0].copy()
    b._packed[len(b.counts):2 * len(b.counts)] = 77                       # olens of the stored message
    import torch.distributed as td
    if not td.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29641')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        kdist.init('gloo')
    out = kdist.gather_decoded([(b, o)], force=True)
    assert out[0].olens.tolist() == [77] * len(b.counts) and out[0].tuples() == b.tuples()
    td.destroy_process_group()
