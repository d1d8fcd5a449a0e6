"""Batch-of-files front end across the GPUs of one node (BASELINE.json north_star).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).
ACM streams share no state, so the data path has NO collective:

    rank 0:  read the 14-byte headers -> weights (total_values) -> greedy longest-first shards
    C1    :  scatter of the shard TABLE - file ids and paths, a few KB; never file contents   [collective, control]
    rank r:  read its own files, parse them on the host, synthesise them on its GPU           [no communication]
    C2    :  per-stream status and offsets to rank 0 (gather_object), then the PCM (2 B/sample) point-to-point,
             exact size, straight from the HBM buffer the decoder filled (no compaction pass)      [results]

The decode itself is injected (`decoder`): the product decoder is GpuDecoder below (HIP kernels through
libacm_hip.so, device memory owned by torch); tests on CPU-only machines inject a stand-in so that
sharding, scatter and gather are exercised with world_size > 1.
"""
import heapq
import os

import numpy as np

from . import capi


def shard_longest_first(weights, n_ranks):
    """Greedy LPT: heaviest stream first onto the least loaded rank.  Returns n_ranks lists of indices."""
    shards = [[] for _ in range(n_ranks)]
    heap = [(0, r) for r in range(n_ranks)]
    heapq.heapify(heap)
    for i in sorted(range(len(weights)), key=lambda k: (-weights[k], k)):
        load, r = heapq.heappop(heap)
        shards[r].append(i)
        heapq.heappush(heap, (load + weights[i], r))
    return shards


def file_weight(data):
    """Decode cost proxy known from the 14-byte header alone: total_values (0 for non-ACM files)."""
    rc, info = capi.probe(data)
    return int(info.total_values) if rc == 0 else 0


def _is_path(f):
    return isinstance(f, (str, os.PathLike))


def _head(f, n=64):
    """the first bytes of a file (path) or file image (bytes): enough for the ACM / WAVC header"""
    if _is_path(f):
        with open(f, "rb") as fh:
            return fh.read(n)
    return bytes(f[:n])


def _load(f):
    if _is_path(f):
        with open(f, "rb") as fh:
            return fh.read()
    return f


class GpuDecoder:
    """Decode a list of file images on this rank's GPU; PCM stays in HBM as one torch int16 tensor.

    Device memory and the stream belong to torch (plumbing); parsing and synthesis are libacm_hip.so's.
    """

    def __init__(self, ordinal=None, fmt=capi.FMT_S16LE, parse=capi.PARSE_AUTO):
        import torch
        self.torch = torch
        if ordinal is None:
            ordinal = torch.cuda.current_device()
        self.ordinal = ordinal
        self.fmt = fmt
        self.parse = parse
        self.timing = None
        torch.cuda.set_device(ordinal)
        # the library runs on a stream of its own (torch's current stream is usually the null stream, which
        # acmhip_device_open does not adopt); __call__ orders the two around the torch-owned PCM tensor
        self.dev = capi.Device(ordinal)

    def __call__(self, files):
        """-> (pcm int16 tensor in HBM, per-file word offsets into it, per-file word counts, per-file statuses)"""
        torch = self.torch
        cap = capi.batch_pcm_words(files)
        d_pcm = torch.empty(max(cap, 1), dtype=torch.int16, device="cuda")
        # a block the caching allocator hands out may still be in use by torch work queued on torch's stream
        torch.cuda.current_stream().synchronize()
        # one call: threaded (or device-side) bit parsing, pipelined H2D, synthesis; the PCM stays in d_pcm
        statuses, words, offsets, self.timing = capi.batch_decode_device(
            self.dev, files, d_pcm.data_ptr(), cap, fmt=self.fmt, parse=self.parse)
        # acm_batch_decode returns with its stream drained: d_pcm is complete and visible to torch's streams
        return d_pcm, offsets, words, statuses


def decode_sharded(files, decoder, dist=None, root=0, device=None, chunks=1):
    """Decode `files` across all ranks of `dist`.

    `files`: list of paths (str / PathLike: every rank can open them; only `root` needs the list) and/or file images
    (bytes: every rank must pass the same list - contents are never sent, only ids).
    Returns on root: list of (status, np.uint16 array) in input order; on other ranks: None.
    With dist=None runs single-process.  `decoder(list_of_bytes)` -> (pcm 1-D int16 tensor, offsets, words, statuses).

    Order of communication, identical on every rank (RCCL runs the operations of one communicator in issue order, so a
    rank that sends before a collective while the root receives after it never completes):
        C1 scatter_object_list -> [local decode, no communication] -> gather_object(metadata) -> PCM point-to-point.
    Nothing is in flight while acm_batch_decode runs (its plan teardown frees device memory, which waits for every
    stream of the device, a pending send kernel included).  `chunks` only bounds the size of one decode call.
    """
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0

    # ---- C1: shard table scatter (control): ids + paths, never contents
    if rank == root:
        weights = [file_weight(_head(f)) for f in files]
        shards = shard_longest_first(weights, world)
        table = [[(i, os.fspath(files[i]) if _is_path(files[i]) else None) for i in shard] for shard in shards]
    else:
        table = None
    if dist is not None:
        mine = [None]
        dist.scatter_object_list(mine, table if rank == root else None, src=root)
        mine = mine[0]
    else:
        mine = table[0]
    for i, path in mine:
        if path is None and (files is None or i >= len(files)):
            raise ValueError("decode_sharded: file %d was passed as bytes on the root only; pass paths, or the same list on every rank" % i)

    # ---- local decode (no communication).  The PCM of a call stays where the decoder left it: one dense int16
    # tensor, stream k at offsets[k] (no compaction pass over HBM); what travels is its used prefix.
    nch = max(1, min(chunks, len(mine)))
    cuts = [len(mine) * k // nch for k in range(nch + 1)]
    pieces = []
    for k in range(nch):
        part = mine[cuts[k]:cuts[k + 1]]
        ids = [i for i, _ in part]
        pcm, offsets, words, statuses = decoder([_load(path if path is not None else files[i]) for i, path in part])
        if device is None:
            device = pcm.device
        used = max([int(o) + int(w) for o, w in zip(offsets, words) if w] or [0])
        pieces.append(((ids, [int(o) for o in offsets], [int(w) for w in words], [int(x) for x in statuses], used), pcm[:used]))

    # ---- C2: results to root
    if dist is None:
        metas = [[m for m, _ in pieces]]
        flats = [[f for _, f in pieces]]
    else:
        metas = [None] * world if rank == root else None
        dist.gather_object([m for m, _ in pieces], metas, dst=root)
        if rank == root:
            flats = [None] * world
            flats[root] = [f for _, f in pieces]
            reqs = []
            for r in range(world):
                if r == root:
                    continue
                flats[r] = []
                for m in metas[r]:
                    buf = torch.empty(2 * m[4], dtype=torch.uint8, device=device)
                    if m[4]:
                        reqs.append(dist.irecv(buf, src=r))
                    flats[r].append(buf.view(torch.int16))
        else:
            # neither RCCL nor gloo moves int16; bytes are bytes.  Exact size, straight from HBM.
            reqs = [dist.isend(f.contiguous().view(torch.uint8), dst=root) for _, f in pieces if f.numel()]
        for q in reqs:
            q.wait()
    if rank != root:
        return None

    # device -> host: every piece into its own pinned buffer, asynchronously, one synchronisation for all of them
    hosts = []
    for flats_r in flats:
        row = []
        for f in flats_r:
            if f.is_cuda:
                h = torch.empty(f.numel(), dtype=torch.int16, pin_memory=True)
                h.copy_(f, non_blocking=True)
            else:
                h = f
            row.append(h)
        hosts.append(row)
    if any(f.is_cuda for fl in flats for f in fl):
        torch.cuda.synchronize()
    out = [None] * len(files)
    for metas_r, hosts_r in zip(metas, hosts):
        for (ids_r, offs_r, words_r, st_r, _), h in zip(metas_r, hosts_r):
            host = h.numpy().view(np.uint16)
            for i, o, w, st in zip(ids_r, offs_r, words_r, st_r):
                out[i] = (st, host[o:o + w].copy())
    return out
