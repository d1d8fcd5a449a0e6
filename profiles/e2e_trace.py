import sys; sys.path.insert(0, ".")
from concurrent.futures import ThreadPoolExecutor
from libacm_amd import capi, synth
dev = capi.Device(0)
with ThreadPoolExecutor(32) as ex:
    files = list(ex.map(lambda i: synth.generate(seed=synth.BASE_SEED + i, level=9, rows=16, nblocks=250), range(1024)))
capi.batch_decode(dev, files, parse=capi.PARSE_DEVICE)
print("---- second call", file=sys.stderr, flush=True)
capi.batch_decode(dev, files, parse=capi.PARSE_DEVICE)
