"""Narrow staged tiles (acmhip_plan_attach_narrow) against the int16 form: the same plan on the same box, interleaved rounds.

  python3 profiles/narrow_probe.py [level streams rows blocks [pwr_max]] ...

Per configuration: launch time with the int8 plane detached ("wide") / attached ("narrow": the plan decides per level whether the
narrow build of the kernel pays; "forced": ACM_NARROW=1, the narrow build whenever a tile is narrow) and the tiles read from
the int8 plane (HIP events over 20 launches, best and median of the rounds); PCM of the forms compared word for word on the
device's output (CRC-32 of the arena)."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libacm_amd import capi, workload  # noqa: E402


def run(dev, level, streams, rows, blocks, pwr_max=None, pwr_min=None):
    kw = {} if pwr_max is None else {"pwr_min": min(4, pwr_max) if pwr_min is None else pwr_min, "pwr_max": pwr_max}
    b = workload.build_uniform(streams, level, rows, blocks, seed0=0, keep_files=0, **kw)
    bufs = b.upload(dev)
    plan = capi.Plan(dev, b.descs)
    host = np.empty(b.pcm_words, dtype=np.uint16)

    def crc():
        plan.launch(*bufs)
        dev.download(host, bufs[2])
        dev.sync()
        return zlib.crc32(host.view(np.uint8))

    wide_crc = crc()
    n = plan.attach_narrow(bufs[0])
    narrow_crc = crc()
    forms = ("wide", "narrow", "forced")
    times = {f: [] for f in forms}
    counts = {}
    for _ in range(6):
        for form in forms:
            if form == "forced":
                os.environ["ACM_NARROW"] = "1"            # the narrow build even where the plan would not choose it
            else:
                os.environ.pop("ACM_NARROW", None)
            plan.attach_narrow(None if form == "wide" else bufs[0])
            st = plan.stats()
            counts[form] = (st.narrow_tiles, st.narrow_front_tiles, st.tiles)
            for _ in range(30):
                plan.launch(*bufs)
            dev.sync()
            times[form].append(plan.time(*bufs, reps=20) / 20)
    w = np.array(times["wide"])
    print("level %2d  %d x %d blocks of %d rows%s: %d tiles, wide %.4f ms (median %.4f, %.1f Gs/s); PCM %s" % (
        level, streams, blocks, rows, "" if pwr_max is None else " pwr<=%d" % pwr_max, counts["wide"][2], w.min(), np.median(w),
        b.samples / w.min() / 1e6, "identical" if wide_crc == narrow_crc else "DIFFERENT"))
    for form in forms[1:]:
        t = np.array(times[form])
        print("    %-8s %6d narrow tiles, %6d of them with the rows in front: %.4f ms (median %.4f)  %+.1f %% (median %+.1f %%)" % (
            form, counts[form][0], counts[form][1], t.min(), np.median(t), (w.min() / t.min() - 1) * 100,
            (np.median(w) / np.median(t) - 1) * 100), flush=True)
    plan.destroy()
    for p in bufs:
        dev.free(p)
    return wide_crc == narrow_crc


if __name__ == "__main__":
    cfgs = [a.split(",") for a in sys.argv[1:]] or [["9", "1024", "16", "250"], ["9", "1024", "16", "250", "7"]]
    ok = True
    with capi.Device(0) as dev:
        for c in cfgs:
            ok &= run(dev, *[int(x) for x in c])
    sys.exit(0 if ok else 1)
