"""Multi-chunk check of acm_batch_decode with ACM_BATCH_STAGE_BYTEPLANE: big uniform batches (several pipeline chunks) at levels 9, 11, 7, 13 -
same statuses and PCM as the default staging, first / middle / last stream against the oracle, upload bytes and wall time of both.  GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from libacm_amd import capi, workload
dev = capi.Device(0)
for level, rows, blocks, n in ((9, 16, 250, 512), (11, 64, 16, 256), (7, 16, 1000, 256), (13, 16, 16, 128)):
    b = workload.build_uniform(n, level, rows, blocks, keep_files=n, threads=16)
    files = [f.tobytes() for f in b.files]
    plain, tm0 = capi.batch_decode(dev, files, threads=0)
    bp, tm = capi.batch_decode(dev, files, threads=0, byteplane=True)
    bad = sum(1 for (s0, p0), (s1, p1) in zip(plain, bp) if s0 != s1 or not np.array_equal(p0, p1))
    print("level %d: %d streams, byte-plane staged %d, h2d %.2f -> %.2f GB, total %.3f -> %.3f s, differing streams: %d" % (
        level, n, tm.packed_streams, tm0.h2d_bytes / 1e9, tm.h2d_bytes / 1e9, tm0.total_s, tm.total_s, bad), flush=True)
    import oracle_api as O
    for k in (0, n // 2, n - 1):
        want = O.Oracle.decode_all(files[k])[0].view(np.uint16)
        assert np.array_equal(want, bp[k][1]), (level, k)
print("ok")
