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
