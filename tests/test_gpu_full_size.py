"""BASELINE.json configs[2] and configs[4] at their FULL size on one MI355X, every stream checked against the CPU oracle
(task 4: until now only miniatures of them ran under `-m gpu`).  configs[1] at full size: test_gpu_parity.py.

Reference semantics: the whole decode path, /root/reference/src/decode.c:580-677 (decode_block + output_values) per stream.
"""
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_api as O
from libacm_amd import capi, workload

pytestmark = pytest.mark.gpu


def oracle_crc(f, words):
    want = O.Oracle.decode_all(f.tobytes() if hasattr(f, "tobytes") else f)[0]
    assert want.size >= words
    return zlib.crc32(want.view(np.uint8)[:2 * words])


def test_full_size_config_2(dev):
    """configs[2]: the Fallout-2-shaped corpus - 4000 files, mono / stereo, levels 7-9, 1-60 s at 22 050 Hz, ragged ends
    (1.9 Gsamples) - (a) staged by the host parser and decoded by ONE plan, (b) as file images through acm_batch_decode with the
    bits parsed on the device: every one of the 4000 streams' PCM against the oracle's, by CRC-32"""
    threads = max(4, min(64, workload.usable_cpus()))
    b = workload.build_corpus(4000, keep_files=1 << 30, threads=threads)
    assert len(b.descs) == 4000 and {d.level for d in b.descs} == {7, 8, 9} and b.samples > 1_800_000_000
    with ThreadPoolExecutor(max_workers=threads) as ex:
        want = list(ex.map(lambda k: oracle_crc(b.files[k], b.descs[k].n_emit), range(4000)))
    # (a) one plan over the staged arena: with the byte-plane form of the whole tiles bound (first pass on the matrix cores; the
    # ragged tails stay int16), then on the int16 form
    bufs = b.upload(dev)
    mf = capi.mform_streams(b.idx, b.descs, threads=threads)
    mf_ptrs = mf.upload(dev)
    try:
        plan = capi.Plan(dev, b.descs, packed=mf.streams)
        st = plan.stats()
        assert st.fused_streams == 4000 and st.stagewise_streams == 0 and st.samples == b.samples and st.mform_tiles > 100000
        # the byte-plane form (the headline kernel family) FIRST, and the WHOLE PCM arena poisoned in front of every form's launch:
        # a kernel that skipped a tile would leave 0xA5A5 there, not the previous form's correct PCM (Weak 1a)
        for bind in (mf_ptrs, (None, None)):
            plan.bind_mform(*bind)
            dev.memset(bufs[2], 0xA5, 2 * b.pcm_words)
            plan.launch(*bufs)
            dev.sync()
            host = np.empty(b.pcm_words, dtype=np.uint16)
            dev.download(host, bufs[2])
            raw = host.view(np.uint8)
            with ThreadPoolExecutor(max_workers=threads) as ex:
                got = list(ex.map(lambda k: zlib.crc32(raw[2 * b.descs[k].pcm_off: 2 * (b.descs[k].pcm_off + b.descs[k].n_emit)]), range(4000)))
            bad = [k for k in range(4000) if got[k] != want[k]]
            assert not bad, (bind[0] is not None, bad[:10])
            del host, raw
        plan.destroy()
    finally:
        for p in bufs + mf_ptrs:
            dev.free(p)
    del mf
    # (b) the batch front end, device-side bit parsing
    files = [f.tobytes() for f in b.files]
    res, tm = capi.batch_decode(dev, files, parse=capi.PARSE_DEVICE)
    assert tm.samples == b.samples and tm.device_parsed + tm.host_parsed >= 4000 and tm.device_parsed >= 3900
    with ThreadPoolExecutor(max_workers=threads) as ex:
        got = list(ex.map(lambda k: (res[k][0], zlib.crc32(res[k][1].view(np.uint8))), range(4000)))
    bad = [k for k in range(4000) if got[k] != (0, want[k]) or res[k][1].size != b.descs[k].n_emit]
    assert not bad, bad[:10]


def test_full_size_config_4(dev):
    """configs[4], the stress shape: 65 536 stereo streams, level 11, 64 rows, 2 blocks each = 17.2 Gsamples in ONE plan (34 GB of
    PCM in HBM).  BASELINE allows 1024 distinct streams replicated x 64: the replicas are stream descriptors that name the
    same staged input and their own output.  Every distinct stream against the oracle, every replica against its original."""
    threads = max(4, min(64, workload.usable_cpus()))
    distinct, reps = 1024, 64
    b = workload.build_uniform(distinct, 11, 64, 2, channels=2, keep_files=distinct, seed0=4 << 20, threads=threads)
    per = b.descs[0].n_emit
    pad = (per + 63) // 64 * 64
    assert per == 2 * 64 * 2048 and distinct * reps == 65536
    descs = []
    for r in range(reps):
        for d in b.descs:
            descs.append(capi.StreamDesc(idx_off=d.idx_off, hdr_off=d.hdr_off, pcm_off=len(descs) * pad, n_emit=d.n_emit,
                                         level=d.level, rows=d.rows, nrows=d.nrows, row_begin=0))
    with ThreadPoolExecutor(max_workers=threads) as ex:
        want = list(ex.map(lambda k: oracle_crc(b.files[k], per), range(distinct)))
    d_idx, d_hdr, d_pcm0 = b.upload(dev)
    dev.free(d_pcm0)
    d_pcm = dev.malloc(len(descs) * pad * 2)
    # the byte-plane form of the 1024 distinct streams; the replicas name the same pair-table entries
    mf = capi.mform_streams(b.idx, b.descs, threads=threads)
    mf_ptrs = mf.upload(dev)
    try:
        plan = capi.Plan(dev, descs, packed=[mf.streams[k % distinct] for k in range(len(descs))])
        st = plan.stats()
        assert st.samples == 65536 * per and st.fused_streams == 65536 and st.mform_tiles == 65536 * 128 // capi.lib().acmhip_mform_tile_rows(11)
        for bind in (mf_ptrs, (None, None)):                  # the byte-plane form first, then the int16 form (vector-ALU first pass)
            plan.bind_mform(*bind)
            dev.memset(d_pcm, 0xA5, len(descs) * pad * 2)     # all 34 GB: nothing of another launch's PCM survives (Weak 1a)
            plan.launch(d_idx, d_hdr, d_pcm)
            dev.sync()
            # read back replica by replica (1024 streams = 537 MB each)
            slab = np.empty(distinct * pad, dtype=np.uint16)
            for r in range(reps):
                dev.download(slab, d_pcm + 2 * r * distinct * pad)
                raw = slab.view(np.uint8)
                with ThreadPoolExecutor(max_workers=threads) as ex:
                    got = list(ex.map(lambda k: zlib.crc32(raw[2 * k * pad: 2 * (k * pad + per)]), range(distinct)))
                bad = [k for k in range(distinct) if got[k] != want[k]]
                assert not bad, (bind[0] is not None, r, bad[:10])
        plan.destroy()
    finally:
        for p in (d_idx, d_hdr, d_pcm) + mf_ptrs:
            dev.free(p)


def test_batch_read_back_into_pinned_buffers_is_not_slower(dev):
    """VERDICT r3 Weak 5: on one box the driver saw the device-parse batch with pinned caller buffers (ACM_BATCH_PCM_PINNED, what
    acmtool -B uses) take twice as long as with pageable ones.  A batch of a gigasample, three calls each, best of each: within
    1.3 x of each other (single calls do show 1.5-2 x outliers in either mode on these shared hosts: profiles/pinned_out_probe.py)"""
    from libacm_amd import synth
    with ThreadPoolExecutor(max_workers=16) as ex:
        files = list(ex.map(lambda i: synth.generate(seed=synth.BASE_SEED + 900000 + i, level=9, rows=16, nblocks=250), range(512)))
    best = {}
    for rep in range(4):
        for pinned in (False, True):
            t0 = time.perf_counter()
            res, tm = capi.batch_decode(dev, files, parse=capi.PARSE_DEVICE, pinned=pinned)
            assert tm.samples == 512 * 250 * 16 * 512 and all(s == 0 for s, _ in res)
            del res
            if rep:             # the first round sizes the arenas
                best[pinned] = min(best.get(pinned, 1e9), tm.total_s)
    assert best[True] <= 1.3 * best[False] and best[False] <= 1.3 * best[True], best
