"""One-off soak of acm_batch_decode: random batches (levels 0-14, block heights 1-70, mono / stereo, ragged ends, truncated files,
files with out-of-range indices (H1), junk) through host parsing, device parsing with a random number of block ranges (striped
upload) and prestaged parsing, pinned or pageable outputs - every stream's status and PCM against the CPU oracle.
usage: python3 profiles/batch_fuzz.py [batches [seed]]   (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as O  # noqa: E402
from helpers import make_stream, oracle_pcm  # noqa: E402
from libacm_amd import capi  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
bad = streams = flagged = 0
with capi.Device(0) as dev:
    for b in range(batches):
        if os.environ.get("ACM_FUZZ_TRACE"):
            print("batch", b, flush=True)
        files = []
        for _ in range(int(rng.integers(1, 40))):
            kind = rng.random()
            lv = int(rng.integers(0, 15))
            rows = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 33, 64, 70]))
            nb = int(rng.integers(1, max(2, min(60, (1 << 19) // (rows << lv) + 2))))
            kw = dict(channels=int(rng.integers(1, 3)), cut=int(rng.integers(0, 7)))
            if kind < 0.08:
                kw.update(mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
            f = make_stream(int(rng.integers(1, 1 << 30)), lv, rows, nb, **kw)
            if 0.08 <= kind < 0.2:
                f = f[:int(rng.integers(1, len(f) + 1))]
            elif 0.2 <= kind < 0.23:
                f = bytes(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8))
            files.append(f)
        mode = int(rng.integers(0, 3))
        R = int(rng.choice([1, 2, 3, 5, 8, 16]))
        capi.BATCH_EXTRA = capi.batch_ranges(R)
        res, tm = capi.batch_decode(dev, files, threads=int(rng.integers(1, 9)), pinned=bool(rng.integers(0, 2)),
                                    parse=capi.PARSE_DEVICE if mode == 1 else capi.PARSE_HOST, prestage=mode == 2)
        flagged += tm.host_parsed if mode == 1 else 0
        for k, f in enumerate(files):
            streams += 1
            o = O.Oracle(f)
            if o.err < 0:
                ok = res[k][0] == o.err and res[k][1].size == 0
            else:
                want, wst = oracle_pcm(f)
                ok = np.array_equal(res[k][1], want) and (res[k][0] == wst or (wst == 0 and res[k][0] < 0 and want.size == res[k][1].size))
            if not ok:
                bad += 1
                print("MISMATCH batch %d stream %d mode %d ranges %d status %d len %d" % (b, k, mode, R, res[k][0], len(f)), flush=True)
print("%d batches, %d streams (%d re-parsed by the host behind the device parser): %d mismatches" % (batches, streams, flagged, bad))
sys.exit(1 if bad else 0)
