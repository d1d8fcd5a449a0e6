#!/usr/bin/env python3
"""int16 staged form against the packed staged form on one box, one process, same plan (GPU box only).

  python3 profiles/packed_probe.py [--level 9 --rows 16 --blocks 250 --streams 1024] [--rounds 3] [--steps 60]

Stages the workload with the library's host stager (bit parser + packer), uploads both forms, builds ONE plan with the packed
records, then times launches with the packed arenas bound / unbound in interleaved rounds (acmhip_plan_time: HIP events on
the launch stream).  The PCM of both is compared word for word and, for the first streams, with the CPU oracle.
"""
import argparse
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--blocks", type=int, default=250)
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--verify", type=int, default=16)
    ap.add_argument("--pwr-max", type=int, default=12)
    ap.add_argument("--power", type=float, default=0.0, help="seconds of rocm-smi power / clock sampling per form")
    a = ap.parse_args()
    from libacm_amd import capi, workload
    dev = capi.Device(0)
    kw = {} if a.pwr_max == 12 else dict(pwr_max=a.pwr_max, pwr_min=min(4, a.pwr_max))
    b = workload.build_uniform(a.streams, a.level, a.rows, a.blocks, keep_files=a.verify, threads=workload.usable_cpus(), **kw)
    t0 = time.perf_counter()
    pk = capi.pack_streams(b.idx, b.descs, threads=workload.usable_cpus())
    t_pack = time.perf_counter() - t0
    kinds = np.bincount(pk.chunks["kind"], minlength=5)
    print("packed: %d chunk slots, %d used (zero %d nibble %d byte %d word %d), blob %.1f MB + chunk table %.1f MB = %.3f B/sample (int16: 2), "
          "packer %.2f s" % (pk.chunks.size, pk.chunks.size - kinds[0], kinds[1], kinds[2], kinds[3], kinds[4], pk.blob.nbytes / 1e6, pk.chunks.nbytes / 1e6,
                             pk.nbytes / b.samples, t_pack), flush=True)
    bufs = b.upload(dev)
    ptrs = pk.upload(dev)
    plan = capi.Plan(dev, b.descs, packed=pk.streams)
    st = plan.stats()
    print("plan: %d tiles, %d with packed records, %d launches" % (st.tiles, st.packed_tiles, st.launches), flush=True)

    def pcm_crc():
        host = np.empty(b.pcm_words, dtype=np.uint16)
        dev.download(host, bufs[2])
        return zlib.crc32(host.view(np.uint8)), host
    plan.bind_packed(None, None)
    plan.launch(*bufs)
    dev.sync()
    crc16, host16 = pcm_crc()
    dev.upload(bufs[2], np.zeros(1 << 20, dtype=np.uint16))
    plan.bind_packed(*ptrs)
    plan.launch(*bufs)
    dev.sync()
    crcpk, hostpk = pcm_crc()
    print("PCM int16 form %08x, packed form %08x: %s" % (crc16, crcpk, "identical" if crc16 == crcpk else "DIFFERENT"), flush=True)
    if crc16 != crcpk:
        bad = np.nonzero(host16 != hostpk)[0]
        print("  %d words differ, first at %d (stream %d, sample %d)" % (bad.size, bad[0], bad[0] // (b.pcm_words // a.streams), bad[0] % (b.pcm_words // a.streams)))
    if b.files:
        import oracle_api as O
        ok = 0
        for k, f in enumerate(b.files):
            want = O.Oracle.decode_all(f.tobytes())[0].view(np.uint16)
            d = b.descs[k]
            ok += bool(np.array_equal(want[:d.n_emit], hostpk[d.pcm_off:d.pcm_off + d.n_emit]))
        print("oracle: %d of %d streams identical" % (ok, len(b.files)), flush=True)
    res = {"int16": [], "packed": []}
    for _ in range(30):
        plan.launch(*bufs)
    dev.sync()
    for r in range(a.rounds):
        for name, bind in (("int16", (None, None)), ("packed", ptrs)):
            plan.bind_packed(*bind)
            for _ in range(5):
                plan.launch(*bufs)
            res[name].append(plan.time(*bufs, reps=a.steps) / a.steps)
    for name, ms in res.items():
        m = sorted(ms)[len(ms) // 2]
        print("%-7s median %.4f ms  frac(4 B/sample) %.4f  rounds %s" % (name, m, b.samples * 4 / (m * 1e-3) / 8e12, " ".join("%.4f" % x for x in ms)), flush=True)
    m16, mpk = sorted(res["int16"])[len(res["int16"]) // 2], sorted(res["packed"])[len(res["packed"]) // 2]
    print("packed / int16: %+.1f %%" % ((m16 / mpk - 1) * 100))
    if a.power:
        import bench
        for name, bind in (("int16", (None, None)), ("packed", ptrs)):
            plan.bind_packed(*bind)
            sm = bench.PowerSampler()
            sm.start()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < a.power:
                for _ in range(20):
                    plan.launch(*bufs)
                dev.sync()
            print("%-7s power %s" % (name, sm.stop()), flush=True)
    plan.destroy()
    dev.close()


if __name__ == "__main__":
    main()
