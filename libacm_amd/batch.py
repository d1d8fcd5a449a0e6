"""Batch-of-files front end across the GPUs of one node (BASELINE.json north_star).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).
ACM streams share no state, so the data path has NO collective:

    rank 0:  read headers -> weights (total_values) -> greedy longest-first shards
    C1    :  scatter of the shard table (a few KB of metadata)            [collective, control]
    rank r:  parse its files on the host, synthesise them on its GPU       [no communication]
    C2    :  gather of PCM (2 B/sample) + per-stream status to rank 0      [collective, results]

The decode itself is injected (`decoder`): the product decoder is GpuDecoder below (HIP kernels through
libacm_hip.so, device memory and stream owned by torch); tests on CPU-only machines inject a stand-in so that
sharding, scatter and gather are exercised with world_size > 1.
"""
import heapq

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
        self.dev = capi.Device(ordinal, torch.cuda.current_stream().cuda_stream)

    def __call__(self, files):
        """-> (pcm int16 tensor in HBM, per-file word offsets into it, per-file word counts, per-file statuses)"""
        torch = self.torch
        cap = capi.batch_pcm_words(files)
        d_pcm = torch.empty(max(cap, 1), dtype=torch.int16, device="cuda")
        # one call: threaded (or device-side) bit parsing, pipelined H2D, synthesis; the PCM stays in d_pcm
        statuses, words, offsets, self.timing = capi.batch_decode_device(
            self.dev, files, d_pcm.data_ptr(), cap, fmt=self.fmt, parse=self.parse)
        return d_pcm, offsets, words, statuses


def decode_sharded(files, decoder, dist=None, root=0, device=None):
    """Decode `files` (list of bytes; only rank `root` needs to hold them) across all ranks of `dist`.

    Returns on root: list of (status, np.uint16 array) in input order; on other ranks: None.
    With dist=None runs single-process.  `decoder(list_of_bytes)` -> (pcm 1-D int16 tensor, offsets, words, statuses).
    """
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0

    # ---- C1: shard table scatter (control)
    if rank == root:
        weights = [file_weight(f) for f in files]
        shards = shard_longest_first(weights, world)
        payload = [[(i, files[i]) for i in shard] for shard in shards]
    else:
        payload = None
    if dist is not None:
        mine = [None]
        dist.scatter_object_list(mine, payload if rank == root else None, src=root)
        mine = mine[0]
    else:
        mine = payload[0]

    # ---- local decode (no communication)
    ids = [i for i, _ in mine]
    pcm, offsets, words, statuses = decoder([f for _, f in mine])
    if device is None:
        device = pcm.device
    # compact this rank's PCM into one contiguous run (drop the alignment padding between streams)
    parts = [pcm[o:o + w] for o, w in zip(offsets, words) if w]
    flat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int16, device=device)

    # ---- C2: gather (results)
    meta = (ids, words, statuses)
    if dist is None:
        metas, flats = [meta], [flat]
    else:
        metas = [None] * world if rank == root else None
        dist.gather_object(meta, metas, dst=root)
        n = torch.tensor([flat.numel()], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n)
        cap = int(max(int(s.item()) for s in sizes))
        padded = torch.zeros(max(cap, 1), dtype=torch.int16, device=device)
        padded[:flat.numel()] = flat
        wire = padded.view(torch.uint8)              # neither RCCL nor gloo moves int16; bytes are bytes
        bufs = [torch.empty_like(wire) for _ in range(world)] if rank == root else None
        dist.gather(wire, bufs, dst=root)
        flats = [b.view(torch.int16)[:int(s.item())] for b, s in zip(bufs, sizes)] if rank == root else None
    if rank != root:
        return None

    out = [None] * len(files)
    for (ids_r, words_r, st_r), flat_r in zip(metas, flats):
        host = flat_r.cpu().numpy().view(np.uint16)
        pos = 0
        for i, w, st in zip(ids_r, words_r, st_r):
            out[i] = (st, host[pos:pos + w].copy())
            pos += w
    return out
