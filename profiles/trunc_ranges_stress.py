"""Device-parse batches in 2 ... 16 block ranges over mid-sized streams of the chunk kernel's levels, half of them truncated at random
places, a few with H1 indices: status and PCM of every stream against the CPU oracle (GPU box).  usage: [rounds [seed]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_stream, oracle_pcm
from libacm_amd import capi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
bad = n = 0
with capi.Device(0) as dev:
    for rd in range(rounds):
        lv = int(rng.choice([8, 9, 10, 11, 12]))
        t2 = {8: 32, 9: 16, 10: 8, 11: 4, 12: 4}[lv]
        rows = int(rng.choice([t2, 2 * t2, 4 * t2, 2, 6, 16]))
        files = []
        for k in range(int(rng.integers(8, 48))):
            nb = int(rng.integers(4, max(5, (1 << 21) // (rows << lv) // 4)))
            kw = dict(pwr_max=int(rng.choice([7, 9, 12])), channels=int(rng.integers(1, 3)), cut=int(rng.integers(0, 5)))
            if rng.random() < 0.08:
                kw.update(mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
            f = make_stream(int(rng.integers(1, 1 << 30)), lv, rows, nb, **kw)
            if rng.random() < 0.5:
                f = f[:int(rng.integers(20, len(f)))]
            files.append(f)
        R = int(rng.choice([2, 3, 5, 8, 16]))
        os.environ["ACM_BATCH_RANGES"] = str(R)
        res, tm = capi.batch_decode(dev, files, threads=int(rng.integers(1, 9)), parse=capi.PARSE_DEVICE, pinned=bool(rng.integers(0, 2)))
        for k, f in enumerate(files):
            want, wst = oracle_pcm(f)
            n += 1
            if not np.array_equal(res[k][1], want):
                bad += 1
                print("round %d stream %d (level %d rows %d, %d ranges): differs" % (rd, k, lv, rows, R), flush=True)
        print("round %d: level %d rows %d, %d streams, %d ranges, %d staged as byte planes, %d redone by the host" % (rd, lv, rows, len(files), R, tm.packed_streams, tm.host_parsed), flush=True)
print("%d rounds, %d streams: %d mismatches" % (rounds, n, bad))
sys.exit(1 if bad else 0)
