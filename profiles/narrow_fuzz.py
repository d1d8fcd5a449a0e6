"""One-off soak of the narrow rows: random batches (levels 6-14, block heights 1-70, pwr ranges that mix narrow and 16-bit blocks,
mono / stereo, ragged ends) through acmhip_plan_attach_narrow with the narrow build forced on, every stream against the CPU oracle.
usage: python3 profiles/narrow_fuzz.py [batches [seed]]   (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ACM_NARROW"] = "1"
os.environ["ACM_K2"] = "1"
from helpers import make_stream, oracle_pcm  # noqa: E402
from libacm_amd import capi  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261003)
bad = streams = narrow = tiles = 0
with capi.Device(0) as dev:
    for b in range(batches):
        files = []
        for _ in range(int(rng.integers(1, 24))):
            lv = int(rng.integers(6, 15))
            rows = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 17, 33, 64, 70]))
            tr = max(2, 16384 >> lv)
            nb = max(1, int(rng.integers(1, 6) * tr // rows) + int(rng.integers(0, 4)))
            lo = int(rng.integers(0, 9))
            hi = int(rng.integers(lo, 13))
            files.append(make_stream(int(rng.integers(1, 1 << 30)), lv, rows, min(nb, 2000), channels=int(rng.integers(1, 3)),
                                     cut=int(rng.integers(0, 7)), pwr_min=lo, pwr_max=hi, val_max=int(rng.choice([255, 65535]))))
        staged = [capi.stage_file(f) for f in files]
        fmt = int(rng.integers(0, 4))
        got, st = capi.synth(dev, staged, fmt=fmt, return_stats=True, narrow=True)
        narrow += st.narrow_tiles
        tiles += st.tiles
        for k, (f, g) in enumerate(zip(files, got)):
            want, _ = oracle_pcm(f, 0, fmt & 1, 0 if fmt & 2 else 1)
            streams += 1
            if g.size != want.size or not np.array_equal(g, want):
                bad += 1
                print("MISMATCH batch %d stream %d level %d rows %d" % (b, k, staged[k].info.level, staged[k].info.rows), flush=True)
print("%d batches, %d streams, %d of %d tiles' worth of rows narrow: %d mismatches" % (batches, streams, narrow, tiles, bad))
sys.exit(1 if bad else 0)
