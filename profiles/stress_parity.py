"""One-off large-scale parity run (GPU box): ~1.5 Gsamples of random-shaped streams (levels 0-15: register, tile and prefix
kernels; rows 1-64, mono/stereo, ragged), both tile-kernel flavours x both parse modes of acm_batch_decode against the
CPU oracle.  usage: python profiles/stress_parity.py [seed]"""
import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from libacm_amd import capi, synth
import oracle_api as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
shapes = []
total = 0
while total < 1.5e9:
    level = int(rng.integers(0, 16)); rows = int(rng.integers(1, 65)); bl = rows << level
    nb = int(rng.integers(max(1, 2000000 // bl // 4), max(2, 2000000 // bl)))
    shapes.append((level, rows, nb, int(rng.integers(1, 3)), int(rng.integers(0, bl))))
    total += nb * bl
print(len(shapes), "streams", total / 1e9, "Gsamples", flush=True)
def gen(a):
    i, (level, rows, nb, ch, cut) = a
    return synth.generate(seed=synth.BASE_SEED + 77000 + i, level=level, rows=rows, nblocks=nb, channels=ch, total_values=max(ch, nb * (rows << level) - cut))
with ThreadPoolExecutor(32) as ex:
    files = list(ex.map(gen, enumerate(shapes)))
def ref(f):
    pcm, st = O.Oracle.decode_all(f)
    return pcm.view(np.uint16)
t = time.time()
with ThreadPoolExecutor(32) as ex:
    wants = list(ex.map(ref, files))
print("oracle %.1f s" % (time.time() - t), flush=True)
dev = capi.Device(0)
for carry in ("0", "1"):
    os.environ["ACM_K1_CARRY"] = carry
    for mode in (capi.PARSE_HOST, capi.PARSE_DEVICE):
        res, tm = capi.batch_decode(dev, files, parse=mode)
        bad = [k for k, ((st, pcm), w) in enumerate(zip(res, wants)) if st != 0 or not np.array_equal(pcm, w)]
        print("carry", carry, "parse", mode, "mismatches:", bad[:10], "device_parsed", tm.device_parsed, flush=True)
        assert not bad
print("stress parity ok")
