"""One-off: extreme acm_rows (1 and 4095) at several levels through every kernel family and both parse modes vs the oracle."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from libacm_amd import capi, synth
import oracle_api as O
files = []
for level, rows, nb in ((5, 4095, 3), (7, 4095, 2), (9, 4095, 2), (3, 4095, 2), (0, 4095, 3), (7, 1, 3000), (9, 1, 700), (2, 1, 5000), (11, 1, 40), (12, 3, 4)):
    files.append(synth.generate(seed=synth.BASE_SEED + 88000 + level * 7 + rows, level=level, rows=rows, nblocks=nb,
                                channels=1 + level % 2, total_values=nb * (rows << level) - 11))
wants = [O.Oracle.decode_all(f)[0].view(np.uint16) for f in files]
dev = capi.Device(0)
for carry in ("0", "1"):
    os.environ["ACM_K1_CARRY"] = carry
    for mode in (capi.PARSE_HOST, capi.PARSE_DEVICE):
        for flags in (capi.PLAN_AUTO, capi.PLAN_STAGEWISE):
            res, tm = capi.batch_decode(dev, files, parse=mode, flags=flags)
            bad = [k for k, ((st, pcm), w) in enumerate(zip(res, wants)) if st != 0 or not np.array_equal(pcm, w)]
            print("carry", carry, "parse", mode, "flags", flags, "mismatches", bad, "device_parsed", tm.device_parsed, flush=True)
            assert not bad
print("extreme rows ok")
