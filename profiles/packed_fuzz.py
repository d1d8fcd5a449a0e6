"""One-off soak of the packed staged form: random batches (levels 5-10 around the levels that have the form, block heights 1-70,
pwr ranges from "every index a nibble" to full 16 bits, mono / stereo, ragged ends, truncated files, H1 streams, junk) through
(a) the plan API with the packed tables bound (capi.synth(packed=True): host packer + acm_tile2p, ACM_K2=1 so that small plans take
the lean kernels too) and (b) acm_batch_decode with ACM_BATCH_STAGE_PACKED (host parsing, optionally prestaged, pinned or pageable
output) - every stream's PCM against the CPU oracle.
usage: python3 profiles/packed_fuzz.py [batches [seed]]   (GPU box)"""
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
bad = streams = packed_tiles = packed_streams = 0
capi.PLAN_EXTRA = capi.PLAN_LEAN_ALWAYS            # (what ACM_K2=1 used to ask for: the lean kernels for small plans too)
with capi.Device(0) as dev:
    for b in range(batches):
        files = []
        for _ in range(int(rng.integers(1, 30))):
            kind = rng.random()
            lv = int(rng.integers(5, 11))
            rows = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 33, 64, 70]))
            tr = max(1, 8192 >> lv)
            nb = int(rng.integers(1, max(2, min(400, (int(rng.integers(1, 9)) * tr) // rows + 3))))
            pm = int(rng.choice([3, 5, 7, 9, 12, 15]))
            kw = dict(channels=int(rng.integers(1, 3)), cut=int(rng.integers(0, 7)), pwr_min=min(4, pm), pwr_max=pm,
                      val_max=65535 if rng.random() < 0.3 else 255)
            if kind < 0.06:
                kw.update(mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
            elif kind < 0.12:
                kw.update(mix=2, single_code=int(rng.choice([0, 3, 8, 16, 17, 19, 22, 26, 29])))
                if 3 <= kw["single_code"] <= 16:
                    kw.update(pwr_min=15, pwr_max=15)
            f = make_stream(int(rng.integers(1, 1 << 30)), lv, rows, nb, **kw)
            if 0.12 <= kind < 0.2:
                f = f[:int(rng.integers(15, len(f) + 1))]
            elif 0.2 <= kind < 0.22:
                f = bytes(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8))
            files.append(f)
        good = [f for f in files if O.Oracle(f).err >= 0]
        # (a) plan API
        if good:
            staged = [capi.stage_file(f) for f in good]
            fmt = int(rng.integers(0, 4))
            got, st = capi.synth(dev, staged, fmt=fmt, return_stats=True, packed=True)
            packed_tiles += st.packed_tiles
            for f, g in zip(good, got):
                want, _ = oracle_pcm(f, 0, fmt & 1, 0 if fmt & 2 else 1)
                streams += 1
                if not np.array_equal(g, want):
                    bad += 1
                    print("batch %d (plan API, fmt %d): stream differs" % (b, fmt), flush=True)
        # (b) batch front end
        res, tm = capi.batch_decode(dev, files, threads=int(rng.integers(1, 9)), pinned=bool(rng.integers(0, 2)), prestage=bool(rng.integers(0, 2)),
                                    packed=True)
        packed_streams += tm.packed_streams
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
                print("batch %d (acm_batch_decode): stream %d differs (status %d)" % (b, k, res[k][0]), flush=True)
print("%d batches, %d stream decodes, %d tiles through acm_tile2p, %d streams staged packed by acm_batch_decode: %d mismatches"
      % (batches, streams, packed_tiles, packed_streams, bad))
