#!/usr/bin/env python3
"""int16 staged form (first pass on the vector ALU) against the byte-plane form (first pass on the matrix cores): one box, one
process, same plan (GPU box only).

  python3 profiles/mform_probe.py [--level 9 --rows 16 --blocks 250 --streams 1024] [--rounds 3] [--steps 60] [--matrix]

--matrix first runs a parity matrix (levels 7-12 x several acm_rows, incl. odd ones and 1) against the CPU oracle.
Then stages the workload, uploads both forms, builds ONE plan with the byte-plane records and times launches with the arena bound /
unbound in interleaved rounds (acmhip_plan_time: HIP events on the launch stream).  The PCM of both is compared word for word and,
for the first streams, with the CPU oracle.
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


def parity_matrix(dev):
    from libacm_amd import capi, workload
    import oracle_api as O
    bad = 0
    for level in range(7, 13):
        tr = capi.lib().acmhip_mform_tile_rows(level)
        for rows in (16, 1, 3, 6, 17, 64):
            nrows = 5 * tr + 3
            blocks = (nrows + rows - 1) // rows
            for seed in range(2):
                b = workload.build_uniform(3, level, rows, blocks, keep_files=3, seed0=100 * level + rows + seed, threads=1)
                staged = [capi.stage_file(f.tobytes()) for f in b.files]
                got, st = capi.synth(dev, staged, mform=True, return_stats=True)
                for k, f in enumerate(b.files):
                    want = O.Oracle.decode_all(f.tobytes())[0].view(np.uint16)
                    ok = np.array_equal(want[:got[k].size], got[k]) and want.size == got[k].size
                    if not ok:
                        bad += 1
                        d = np.nonzero(want[:got[k].size] != got[k])[0]
                        print("  MISMATCH level %d rows %d seed %d stream %d: %d words differ, first at %d (row %d col %d)" % (
                            level, rows, seed, k, d.size, d[0] if d.size else -1, (d[0] >> level) if d.size else -1, (d[0] & ((1 << level) - 1)) if d.size else -1), flush=True)
                if seed == 0:
                    print("level %2d rows %2d: %d tiles on the matrix cores of %d" % (level, rows, st.mform_tiles, st.tiles), flush=True)
    print("parity matrix: %s" % ("all identical" if not bad else "%d streams differ" % bad), flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--blocks", type=int, default=250)
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--verify", type=int, default=16)
    ap.add_argument("--matrix", action="store_true")
    ap.add_argument("--power", type=float, default=0.0, help="seconds of rocm-smi power / clock sampling per form")
    ap.add_argument("--pwr-min", type=int, default=None, help="range of the blocks' pwr (their index width); default: the benchmark's")
    ap.add_argument("--pwr-max", type=int, default=None)
    a = ap.parse_args()
    from libacm_amd import capi, workload
    dev = capi.Device(0)
    if a.matrix:
        parity_matrix(dev)
    kw = {}
    if a.pwr_max is not None:
        kw = dict(pwr_max=a.pwr_max, pwr_min=a.pwr_min if a.pwr_min is not None else min(4, a.pwr_max))
    b = workload.build_uniform(a.streams, a.level, a.rows, a.blocks, keep_files=a.verify, threads=workload.usable_cpus(), **kw)
    t0 = time.perf_counter()
    mf = capi.mform_streams(b.idx, b.descs, threads=workload.usable_cpus())
    cc = mf.class_counts()
    print("byte-plane form: %.1f MB = %.3f B/sample (int16 form %.1f MB), row pairs at 4 / 8 / 16 bits: %d / %d / %d, stager %.2f s" % (
        mf.nbytes / 1e6, mf.nbytes / b.samples, b.idx.nbytes / 1e6, cc[1], cc[2], cc[3], time.perf_counter() - t0), flush=True)
    bufs = b.upload(dev)
    d_mf = mf.upload(dev)
    plan = capi.Plan(dev, b.descs, packed=mf.streams)
    st = plan.stats()
    print("plan: %d tiles, %d with a byte-plane form, %d launches" % (st.tiles, st.mform_tiles, st.launches), flush=True)

    def pcm_crc():
        host = np.empty(b.pcm_words, dtype=np.uint16)
        dev.download(host, bufs[2])
        return zlib.crc32(host.view(np.uint8)), host
    plan.bind_mform(None, None)
    plan.launch(*bufs)
    dev.sync()
    crc16, host16 = pcm_crc()
    dev.upload(bufs[2], np.zeros(1 << 20, dtype=np.uint16))
    plan.bind_mform(*d_mf)
    plan.launch(*bufs)
    dev.sync()
    crcmf, hostmf = pcm_crc()
    print("PCM int16 form %08x, byte-plane form %08x: %s" % (crc16, crcmf, "identical" if crc16 == crcmf else "DIFFERENT"), flush=True)
    if crc16 != crcmf:
        bad = np.nonzero(host16 != hostmf)[0]
        per = b.pcm_words // a.streams
        print("  %d words differ, first at %d (stream %d, row %d, col %d)" % (bad.size, bad[0], bad[0] // per, (bad[0] % per) >> a.level, bad[0] & ((1 << a.level) - 1)))
    if b.files:
        import oracle_api as O
        ok = 0
        for k, f in enumerate(b.files):
            want = O.Oracle.decode_all(f.tobytes())[0].view(np.uint16)
            d = b.descs[k]
            ok += bool(np.array_equal(want[:d.n_emit], hostmf[d.pcm_off:d.pcm_off + d.n_emit]))
        print("oracle: %d of %d streams identical" % (ok, len(b.files)), flush=True)
    res = {"int16": [], "mform": []}
    for _ in range(30):
        plan.launch(*bufs)
    dev.sync()
    for r in range(a.rounds):
        for name, bind in (("int16", (None, None)), ("mform", d_mf)):
            plan.bind_mform(*bind)
            for _ in range(5):
                plan.launch(*bufs)
            res[name].append(plan.time(*bufs, reps=a.steps) / a.steps)
    for name, ms in res.items():
        m = sorted(ms)[len(ms) // 2]
        print("%-7s median %.4f ms  frac(4 B/sample) %.4f  rounds %s" % (name, m, b.samples * 4 / (m * 1e-3) / 8e12, " ".join("%.4f" % x for x in ms)), flush=True)
    m16, mmf = sorted(res["int16"])[len(res["int16"]) // 2], sorted(res["mform"])[len(res["mform"]) // 2]
    print("byte-plane / int16: %+.1f %%" % ((m16 / mmf - 1) * 100))
    if a.power:
        import bench
        for name, bind in (("int16", (None, None)), ("mform", d_mf)):
            plan.bind_mform(*bind)
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
