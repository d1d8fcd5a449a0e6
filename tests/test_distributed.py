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


def host_parse_decoder(files):
    """GpuDecoder's shape with the PRODUCT's host half in it: probing, bit parsing and H1 resolution are libacm_hip.so's
    own (acm_stage_probe / acm_stage_file run without a GPU); only the HIP synthesis is replaced, by the oracle's
    juggle_block + writer over the staged form (value = idx * val, patches applied, decode.c:592-600)"""
    from libacm_amd import capi
    parts, offsets, words, statuses, pos = [], [], [], [], 0
    for f in files:
        rc, info = capi.probe(f)
        if rc != 0:
            offsets.append(0), words.append(0), statuses.append(rc)
            continue
        st = capi.stage_file(f)
        bl, level, rows = st.block_len, st.info.level, st.info.rows
        x = st.idx.astype(np.int64).reshape(st.info.blocks, bl) * st.hdr[:st.info.blocks, 0].astype(np.int64)[:, None]
        x = (x & 0xFFFFFFFF).astype(np.uint32).view(np.int32).reshape(-1).copy()
        if st.patches is not None:
            for p in st.patches:
                x[p.sample] = p.value
        wrap = np.zeros(max(1, 2 * (1 << level) - 2), dtype=np.int32)
        out = np.zeros(st.info.blocks * bl, dtype=np.int16)
        for b in range(st.info.blocks):
            blk = x[b * bl:(b + 1) * bl].copy()
            O.Oracle.juggle_block(level, rows, blk, wrap)
            out[b * bl:(b + 1) * bl] = O.Oracle.output(blk, level, 0, 1)[1].view(np.int16)
        w = st.words
        pad = (-w) % 64
        parts.append(np.concatenate([out[:w], np.zeros(pad, np.int16)]))
        offsets.append(pos), words.append(w), statuses.append(st.info.end_status)
        pos += w + pad
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


def test_host_parse_decoder_single_process():
    """the stand-in used by the two-rank test below is itself right (product parser + oracle synthesis == oracle)"""
    files = corpus()
    check(batch.decode_sharded(files, host_parse_decoder, chunks=3), files)


def test_corpus_shards_balance_at_8_ranks():
    """SURVEY 8e / configs[3]: the ~4000-file corpus over 8 GPUs, weights from the 14-byte headers (decode.c:734-736):
    greedy longest-first leaves the heaviest rank within 2 % of the mean"""
    from libacm_amd import workload
    w = [s["total_values"] for s in workload.corpus_shapes(4000)]
    shards = batch.shard_longest_first(w, 8)
    loads = [sum(w[i] for i in s) for s in shards]
    assert sorted(i for s in shards for i in s) == list(range(len(w)))
    assert max(loads) / (sum(loads) / 8) <= 1.02, loads


def check_pipeline_trace(trace, rank, nch, ring):
    """what decode_sharded's trace must look like on a rank with `nch` chunks (task 6): a sender posts the send
    of chunk k before it decodes chunk k + 1 and first waits for it when the buffer comes round again (behind decode k + 1);
    the root posts receives before its first own decode and never has more than `ring` of them outstanding"""
    at = {ev: n for n, ev in enumerate(trace)}
    if rank != 0:
        assert [e for e in trace if e[0] == "decode"] == [("decode", k) for k in range(nch)], trace
        for k in range(nch - 1):
            assert at[("send", k)] < at[("decode", k + 1)], trace
        for k in range(nch - 2):
            assert at[("decode", k + 1)] < at[("send_done", k)] < at[("decode", k + 2)], trace
    else:
        first_decode = min(n for n, e in enumerate(trace) if e[0] == "decode")
        assert any(e[0] == "recv_posted" for e in trace[:first_decode]), trace
        out = 0
        for e in trace:
            out += (e[0] == "recv_posted") - (e[0] == "recv_done")
            assert 0 <= out <= ring, trace
        posted = [e[1:] for e in trace if e[0] == "recv_posted"]
        done = [e[1:] for e in trace if e[0] == "recv_done"]
        assert posted == done and len(posted) >= 4, trace


def _worker_few(rank, world, port, q, nfiles):
    """fewer files than ranks / than chunks: a rank may get nothing at all, a chunk plan may be empty"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        files = corpus()[:nfiles]
        out = batch.decode_sharded(files, oracle_decoder, dist=dist, root=0, device=torch.device("cpu"), chunks=4, ring=1)
        if rank == 0:
            check(out, files)
            q.put("ok")
    except Exception as e:
        q.put("rank %d: %r" % (rank, e))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nfiles", [0, 1, 3])
def test_two_ranks_gloo_fewer_files_than_ranks_or_chunks(nfiles):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_few, args=(r, 2, port, q, nfiles)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


def _worker(rank, world, port, q, paths=None, product_parser=False, chunks=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if paths is not None:
            # the file LIST lives on rank 0 only; the other rank learns its paths from the shard table (C1)
            files = paths if rank == 0 else None
        else:
            # file images: every rank holds the same list, only ids are scattered
            files = corpus()
        trace = []
        nch = chunks or (2 if product_parser else 1)
        out = batch.decode_sharded(files, host_parse_decoder if product_parser else oracle_decoder, dist=dist, root=0,
                                   device=torch.device("cpu"), chunks=nch, ring=2, trace=trace)
        if chunks:
            check_pipeline_trace(trace, rank, nch, 2)
        if rank == 0:
            check(out, corpus())
            q.put("ok")
        else:
            assert out is None
    except Exception as e:      # surface the failure to the parent
        q.put("rank %d: %r" % (rank, e))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("by_path,product_parser,chunks", [(False, False, None), (True, False, None), (True, True, None), (True, True, 5)])
def test_two_ranks_gloo(by_path, product_parser, chunks, tmp_path):
    """chunks=5: the pipelined C2 - send of chunk k under way while chunk k + 1 decodes, two receives at most outstanding on
    the root - with the order of events asserted on both ranks (check_pipeline_trace)"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    paths = None
    if by_path:
        paths = []
        for k, f in enumerate(corpus()):
            p = tmp_path / ("f%02d.acm" % k)
            p.write_bytes(bytes(f))
            paths.append(str(p))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, paths, product_parser, chunks)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


def mini_corpus(n=40):
    """enough small files for three chunks and more on each of eight ranks"""
    # (about equal weights - the shards are cut longest first by weight, and every rank is to get five files - but every file its own shape)
    shapes = [(5, 16, 8), (7, 16, 2), (6, 8, 8), (8, 5, 3), (7, 3, 11)]
    return [make_stream(4500 + i, shapes[i % 5][0], shapes[i % 5][1], shapes[i % 5][2], channels=1 + i % 2, cut=i % 3) for i in range(n)]


def _worker8(rank, world, port, q, paths):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashlib
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        trace = []
        out = batch.decode_sharded(paths if rank == 0 else None, oracle_decoder, dist=dist, root=0, device=torch.device("cpu"), chunks=3, ring=2,
                                   trace=trace)
        nch = sum(1 for e in trace if e[0] == "decode")
        assert nch == 3, (rank, trace)                  # 40 files over 8 ranks: five each, three chunks
        check_pipeline_trace(trace, rank, nch, 2)
        if rank == 0:
            h = hashlib.sha256()
            for st, pcm in out:
                h.update(np.int32(st).tobytes() + pcm.tobytes())
            q.put(("digest", h.hexdigest()))
        else:
            assert out is None
            q.put(("ok", rank))
    except Exception as e:
        q.put(("error", "rank %d: %r" % (rank, e)))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_gloo(tmp_path):
    """the configs[3] shape in miniature, without hardware (task 4): eight ranks, the file list on rank 0 only, three chunks per
    rank through a receive ring of two on the root.  The concatenated (status, PCM) digest on the root equals the one-process digest, every
    file's PCM equals the oracle's, and the order of events holds on every rank: send of chunk k posted before decode k + 1 and first
    waited for behind it; receives posted before the root's first decode and never more than two outstanding"""
    import hashlib
    import torch.multiprocessing as mp
    files = mini_corpus()
    paths = []
    for k, f in enumerate(files):
        p = tmp_path / ("m%02d.acm" % k)
        p.write_bytes(bytes(f))
        paths.append(str(p))
    single = batch.decode_sharded(paths, oracle_decoder)
    check(single, files)
    h = hashlib.sha256()
    for st, pcm in single:
        h.update(np.int32(st).tobytes() + pcm.tobytes())
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q, paths)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = [q.get(timeout=5) for _ in range(8)]
    assert not [g for g in got if g[0] == "error"], got
    assert ("digest", h.hexdigest()) in got and sorted(g[1] for g in got if g[0] == "ok") == list(range(1, 8))


def test_bytes_on_root_only_is_refused():
    """file contents are never scattered: images passed on the root alone cannot reach the other ranks"""
    class FakeDist:
        def get_world_size(self): return 2
        def get_rank(self): return 1
        def scatter_object_list(self, out, src_list, src=0): out[0] = [(0, None)]
    with pytest.raises(ValueError):
        batch.decode_sharded(None, oracle_decoder, dist=FakeDist(), root=0, device=torch.device("cpu"))


@pytest.mark.gpu
def test_front_end_with_gpu_decoder(dev):
    files = corpus()
    check(batch.decode_sharded(files, batch.GpuDecoder(0)), files)


def _nccl_worker(rank, world, port, q, paths):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    ordinal = rank if torch.cuda.device_count() >= world else 0
    torch.cuda.set_device(ordinal)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", ordinal))
    try:
        trace = []
        out = batch.decode_sharded(paths if rank == 0 else None, batch.GpuDecoder(ordinal), dist=dist, root=0, chunks=4, ring=2, trace=trace)
        if world > 1:
            check_pipeline_trace(trace, rank, 4, 2)
        if rank == 0:
            check(out, corpus())
            q.put("ok")
    except Exception as e:
        q.put("rank %d: %r" % (rank, e))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_front_end_on_rccl(dev, tmp_path):
    """the N > 1 code path on RCCL with the real GPU decoder: backend nccl, world size 1 on a one-GPU box (scatter of the
    shard table, gather of the metadata; the PCM of rank 0 stays where it is), in a fresh process"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    paths = []
    for k, f in enumerate(corpus()):
        p = tmp_path / ("f%02d.acm" % k)
        p.write_bytes(bytes(f))
        paths.append(str(p))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(0, 1, port, q, paths))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


@pytest.mark.gpu
def test_front_end_on_rccl_two_ranks(dev, tmp_path):
    """two ranks on two GPUs over RCCL: the PCM of rank 1 really travels point-to-point (runs where the box has two GPUs;
    the one-GPU boxes of the pool skip it - the ordering it guards is described in decode_sharded's docstring)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    paths = []
    for k, f in enumerate(corpus()):
        p = tmp_path / ("f%02d.acm" % k)
        p.write_bytes(bytes(f))
        paths.append(str(p))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q, paths)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        if p.is_alive():
            p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"
