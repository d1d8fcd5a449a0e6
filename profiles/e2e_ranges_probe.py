"""End-to-end device-parse batch (file bytes in host memory -> PCM in host memory) by block-range count, same box, interleaved.
usage: python3 profiles/e2e_ranges_probe.py [streams level rows blocks]   (GPU box; E2E_RANGES=1,8,16,32 picks the counts)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from libacm_amd import capi, synth
n, level, rows, blocks = [int(x) for x in (sys.argv[1:5] or [1024, 9, 16, 250])]
dev = capi.Device(0)
with ThreadPoolExecutor(32) as ex:
    files = list(ex.map(lambda i: synth.generate(seed=synth.BASE_SEED + i, level=level, rows=rows, nblocks=blocks), range(n)))
ref = None
for rep in range(3):
    for R in os.environ.get("E2E_RANGES", "1,4,8,16").split(","):
        os.environ["ACM_BATCH_RANGES"] = R
        for pinned in (False, True):
            res, tm = capi.batch_decode(dev, files, parse=capi.PARSE_DEVICE, pinned=pinned)
            if ref is None:
                ref = res
            elif rep == 0:
                assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(ref, res))
            if rep:
                print("ranges %2s %-10s total %.4f s  %6.0f Msamples/s  (stage %.3f h2d %.3f kernel %.4f d2h %.3f)"
                      % (R, "pinned-out" if pinned else "pageable", tm.total_s, tm.samples / tm.total_s / 1e6, tm.stage_s, tm.h2d_s, tm.kernel_s, tm.d2h_s), flush=True)
            del res
