"""N > 1 path of the batch front end (libacm_amd/batch.py): sharding, shard-table scatter, PCM gather.

CPU: two gloo ranks; the decode step is a stand-in built from the CPU oracle (tests may use it) so that
the plumbing around the GPU decoder is exercised with world_size 2.  GPU (-m gpu): the same front end with
the real GpuDecoder, single process."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

import oracle_api as O
from helpers import make_stream, oracle_pcm
from libacm_amd import batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def corpus():
    files = []
    for i in range(13):
        lv = [5, 7, 9, 3, 0][i % 5]
        files.append(make_stream(4000 + i, lv, [16, 3, 1][i % 3], 1 + (i * 5) % 7, channels=1 + i % 2, cut=i % 3))
    files.insert(4, b"not an acm file")
    return files


def oracle_decoder(files):
    """stand-in with GpuDecoder's return shape: (pcm tensor, offsets, words, statuses)"""
    parts, offsets, words, statuses, pos = [], [], [], [], 0
    for f in files:
        o = O.Oracle(f)
        if o.err < 0:
            offsets.append(0), words.append(0), statuses.append(o.err)
            continue
        o.close()
        pcm, st = O.Oracle.decode_all(f)
        pad = (-pcm.size) % 64
        parts.append(np.concatenate([pcm, np.zeros(pad, np.int16)]))
        offsets.append(pos), words.append(pcm.size), statuses.append(st)
        pos += pcm.size + pad
    flat = torch.from_numpy(np.concatenate(parts)) if parts else torch.zeros(0, dtype=torch.int16)
    return flat, offsets, words, statuses


def check(out, files):
    assert len(out) == len(files)
    for (st, pcm), f in zip(out, files):
        o = O.Oracle(f)
        if o.err < 0:
            assert st == o.err and pcm.size == 0
            continue
        want, wst = oracle_pcm(f)
        assert st == wst and np.array_equal(pcm, want)


def test_shard_longest_first_balances():
    w = [100, 1, 1, 1, 50, 50, 7, 0, 93]
    shards = batch.shard_longest_first(w, 3)
    assert sorted(i for s in shards for i in s) == list(range(len(w)))
    loads = [sum(w[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= 10
    assert batch.shard_longest_first(w, 1) == [sorted(range(len(w)), key=lambda k: (-w[k], k))]


def test_single_process_front_end():
    files = corpus()
    check(batch.decode_sharded(files, oracle_decoder), files)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        files = corpus() if rank == 0 else None
        out = batch.decode_sharded(files, oracle_decoder, dist=dist, root=0, device=torch.device("cpu"))
        if rank == 0:
            check(out, files)
            q.put("ok")
        else:
            assert out is None
    except Exception as e:      # surface the failure to the parent
        q.put("rank %d: %r" % (rank, e))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


@pytest.mark.gpu
def test_front_end_with_gpu_decoder(dev):
    files = corpus()
    check(batch.decode_sharded(files, batch.GpuDecoder(0)), files)
