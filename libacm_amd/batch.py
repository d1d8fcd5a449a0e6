"""Batch-of-files front end across the GPUs of one node (BASELINE.json north_star).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).
ACM streams share no state, so the data path has NO collective:

    rank 0:  read the 14-byte headers -> weights (total_values) -> greedy longest-first shards
    C1    :  scatter of the shard TABLE - file ids and paths, a few KB; never file contents   [collective, control]
    C1b   :  every rank's chunk plan (PCM capacity per chunk, from headers and file lengths) to rank 0   [control]
    rank r:  read its own files, parse and synthesise them on its GPU chunk by chunk; chunk k travels to rank 0
             point-to-point (2 B/sample, straight from the buffer the decoder filled) while chunk k + 1 decodes
    rank 0:  decodes its own chunks and receives the others' into a ring of device buffers -> pinned host memory
    C3    :  per-stream status, offsets and word counts to rank 0 (gather_object) once the transfers are done  [results]

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


def pcm_capacity_words(head, file_len, force_chans=0):
    """16-bit words a stream can take in the dense PCM buffer of a decode call, from its header and its length alone - what
    acm_batch_pcm_words() adds up (acm_batch.cpp: blocks_possible): the blocks the header promises, but no more than the bytes
    of the file can hold (a block costs at least its 20-bit header and a 5-bit filler code per column, decode.c:491-502,
    586-589), padded to 64 words.  0 for a file that is not ACM."""
    rc, info = capi.probe(head, force_chans)
    if rc != 0:
        return 0
    bl = info.rows * info.cols
    promised = (info.total_values + bl - 1) // bl
    bits = max(0, file_len - info.header_bytes) * 8 + 8
    blocks = min(promised, bits // (20 + 5 * info.cols) + 1)
    return (blocks * bl + 63) // 64 * 64


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

    def empty(self, words):
        """a PCM buffer this decoder can decode into (decode_sharded keeps two of them per rank and reuses them)"""
        return self.torch.empty(max(words, 1), dtype=self.torch.int16, device="cuda:%d" % self.ordinal)

    def __call__(self, files, out=None):
        """-> (pcm int16 tensor in HBM, per-file word offsets into it, per-file word counts, per-file statuses).
        out: a tensor of at least acm_batch_pcm_words(files) words to decode into (else a new one is allocated)"""
        torch = self.torch
        cap = capi.batch_pcm_words(files)
        d_pcm = out if out is not None and out.numel() >= cap else torch.empty(max(cap, 1), dtype=torch.int16, device="cuda")
        # a block the caching allocator hands out may still be in use by torch work queued on torch's stream
        torch.cuda.current_stream().synchronize()
        # one call: threaded (or device-side) bit parsing, pipelined H2D, synthesis; the PCM stays in d_pcm.  In steady state
        # nothing in it allocates or frees device memory (grow-only arenas, plan tables from the handle's spare blocks), so a
        # transfer of the previous chunk that is still in flight is not waited for
        statuses, words, offsets, self.timing = capi.batch_decode_device(
            self.dev, files, d_pcm.data_ptr(), d_pcm.numel(), fmt=self.fmt, parse=self.parse)
        # acm_batch_decode returns with its stream drained: d_pcm is complete and visible to torch's streams
        return d_pcm, offsets, words, statuses


def _trace(trace, *ev):
    if trace is not None:
        trace.append(ev)


def decode_sharded(files, decoder, dist=None, root=0, device=None, chunks=1, ring=2, trace=None):
    """Decode `files` across all ranks of `dist`.

    `files`: list of paths (str / PathLike: every rank can open them; only `root` needs the list) and/or file images
    (bytes: every rank must pass the same list - contents are never sent, only ids).
    Returns on root: list of (status, np.uint16 array) in input order; on other ranks: None.
    With dist=None runs single-process.  `decoder(list_of_bytes, out=None)` -> (pcm 1-D int16 tensor, offsets, words, statuses).

    Order of communication, identical on every rank (RCCL runs the operations of one communicator in issue order):
        C1  scatter_object_list of the shard table (ids and paths)
        C1b gather_object of every rank's chunk plan: how many chunks, and the PCM capacity of each - known from headers and
            file lengths alone, so the root can post its receives before anything is decoded
        C2  point-to-point, pipelined: a rank decodes chunk k into one of TWO persistent buffers and sends it while chunk
            k + 1 decodes into the other (a buffer is reused once its send has completed); the root decodes its own chunks and
            meanwhile receives into a RING of `ring` buffers of one chunk each, copying every arrival to pinned host memory and
            posting the next receive into the freed slot - its HBM holds `ring` chunks of the others' PCM, not their shards
        C3  gather_object of the per-stream metadata (ids, offsets, words, statuses) once every transfer is done.
    `trace` (tests): a list that receives (event, ...) tuples in the order things happen on this rank.
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

    # ---- the chunk plan of this rank: whole files, capacities from headers + lengths
    nch = max(1, min(chunks, len(mine))) if mine else 0
    cuts = [len(mine) * k // nch for k in range(nch + 1)] if nch else [0]

    def capacity(i, path):
        if path is not None:
            return pcm_capacity_words(_head(path), os.path.getsize(path))
        return pcm_capacity_words(_head(files[i]), len(files[i]))
    caps = [sum(capacity(i, p) for i, p in mine[cuts[k]:cuts[k + 1]]) for k in range(nch)]

    # ---- C1b: every rank's chunk capacities to the root
    if dist is not None:
        all_caps = [None] * world if rank == root else None
        dist.gather_object(caps, all_caps, dst=root)
    else:
        all_caps = [caps]

    def decode_chunk(k, out):
        part = mine[cuts[k]:cuts[k + 1]]
        _trace(trace, "decode", k)
        pcm, offsets, words, statuses = decoder([_load(path if path is not None else files[i]) for i, path in part], out=out) \
            if out is not None else decoder([_load(path if path is not None else files[i]) for i, path in part])
        used = max([int(o) + int(w) for o, w in zip(offsets, words) if w] or [0])
        if used > caps[k]:
            raise RuntimeError("decode_sharded: chunk %d decoded %d words into a capacity of %d" % (k, used, caps[k]))
        meta = ([i for i, _ in part], [int(o) for o in offsets], [int(w) for w in words], [int(x) for x in statuses], used)
        return pcm, meta

    def new_buffer(words, like=None):
        if hasattr(decoder, "empty"):
            return decoder.empty(words)
        return torch.empty(max(words, 1), dtype=torch.int16, device=device if device is not None else (like.device if like is not None else "cpu"))

    def to_host(t, words):
        """the first `words` of a PCM tensor as a host tensor (pinned + asynchronous for device memory)"""
        if t.is_cuda:
            h = torch.empty(max(words, 1), dtype=torch.int16, pin_memory=True)
            h[:words].copy_(t[:words], non_blocking=True)
            return h
        return t[:words].clone()

    # ---- a rank that is not the root: decode chunk k, send it, decode chunk k + 1 meanwhile
    if dist is not None and rank != root:
        bufs = [new_buffer(max(caps) if caps else 1), new_buffer(max(caps) if caps else 1)] if nch else []
        inflight = [None, None]
        metas = []
        for k in range(nch):
            b = k % 2
            if inflight[b] is not None:
                inflight[b].wait()              # this buffer's previous chunk has left
                _trace(trace, "send_done", k - 2)
            pcm, meta = decode_chunk(k, bufs[b] if hasattr(decoder, "empty") else None)
            if pcm.data_ptr() != bufs[b].data_ptr():
                bufs[b][:meta[4]].copy_(pcm[:meta[4]])          # a decoder with buffers of its own (the CPU stand-ins of the tests)
            metas.append(meta)
            if caps[k]:
                # neither RCCL nor gloo moves int16; bytes are bytes.  The chunk's capacity, not what was used: the root posted
                # its receive before this chunk was decoded
                inflight[b] = dist.isend(bufs[b][:caps[k]].view(torch.uint8), dst=root)
                _trace(trace, "send", k)
        for b, q in enumerate(inflight):
            if q is not None:
                q.wait()
        dist.gather_object(metas, None, dst=root)
        return None

    # ---- the root (or the only process): its own chunks, and meanwhile the ring of receives
    pending = [(r, k, all_caps[r][k]) for k in range(max([len(c) for c in all_caps] or [0]))
               for r in range(world) if r != root and k < len(all_caps[r]) and all_caps[r][k]]
    ring_words = max([c for _, _, c in pending] or [0])
    slots = [new_buffer(ring_words) for _ in range(min(max(1, ring), len(pending)))] if pending else []
    free_slots = list(range(len(slots)))
    posted = []                         # (request, slot, rank, chunk, words) in posting order
    arrived = {}                        # (rank, chunk) -> host tensor
    nxt = 0

    def post_receives():
        nonlocal nxt
        while nxt < len(pending) and free_slots:
            r, k, words = pending[nxt]
            j = free_slots.pop(0)
            posted.append((dist.irecv(slots[j][:words].view(torch.uint8), src=r), j, r, k, words))
            _trace(trace, "recv_posted", r, k)
            nxt += 1

    def collect(block):
        """arrivals in posting order (a communicator completes them in that order): copy out, free the slot"""
        while posted and (block or posted[0][0].is_completed()):
            q, j, r, k, words = posted.pop(0)
            q.wait()
            arrived[(r, k)] = to_host(slots[j], words)
            if slots[j].is_cuda:
                torch.cuda.current_stream().synchronize()       # the slot is about to be written again
            _trace(trace, "recv_done", r, k)
            free_slots.append(j)
            post_receives()
            if block:
                break

    own = []
    for k in range(nch):
        post_receives()
        pcm, meta = decode_chunk(k, None)
        if device is None:
            device = pcm.device
        own.append((meta, to_host(pcm, meta[4])))
        collect(False)
    post_receives()
    while posted:
        collect(True)
    if dist is not None:
        metas = [None] * world
        dist.gather_object([m for m, _ in own], metas, dst=root)
    else:
        metas = [[m for m, _ in own]]
    if torch.cuda.is_available() and any(h.is_pinned() for _, h in own):
        torch.cuda.synchronize()
    out = [None] * len(files)
    for r in range(world):
        for k, (ids_r, offs_r, words_r, st_r, _) in enumerate(metas[r]):
            h = own[k][1] if r == root else arrived.get((r, k))
            host = h.numpy().view(np.uint16) if h is not None else np.zeros(0, np.uint16)
            for i, o, w, st in zip(ids_r, offs_r, words_r, st_r):
                out[i] = (st, host[o:o + w].copy())
    return out
