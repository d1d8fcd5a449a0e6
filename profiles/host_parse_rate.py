import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libacm_amd import capi, workload
import numpy as np
for lv, rows, blocks in ((9, 16, 250), (11, 64, 16), (11, 16, 64), (12, 64, 8), (7, 16, 1000), (11, 64, 2)):
    b = workload.build_uniform(4, lv, rows, blocks, keep_files=1 << 30)
    f = b.files[0].tobytes()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); s = capi.stage_file(f); dt = time.perf_counter() - t
        best = min(best, dt)
    n = blocks * rows << lv
    t = time.perf_counter(); s2 = capi.stage_file_mform(f); dt2 = time.perf_counter() - t
    print("level %2d rows %2d blocks %4d: stage_file %.1f Msamples/s, stage_file_mform %.1f" % (lv, rows, blocks, n / best / 1e6, n / dt2 / 1e6))
