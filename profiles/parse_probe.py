"""Host-pool vs device-lane bit parsing in acm_batch_decode, at several batch shapes (run on the GPU box)."""
import sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
import numpy as np
from libacm_amd import capi, synth

dev = capi.Device(0)
import os
extra = {}
if os.environ.get("SINGLE_CODE"):       # every column the same filler (isolates one parser path)
    extra = dict(mix=synth.MIX_SINGLE, single_code=int(os.environ["SINGLE_CODE"]), pwr_min=12, pwr_max=12)
shapes = [(int(a), int(b), int(c), int(d)) for a, b, c, d in
          (s.split("x") for s in (sys.argv[1:] or ["1024x7x16x250", "8192x7x16x32", "32768x7x16x8", "65536x5x8x16"]))]
for n, level, rows, nblocks in shapes:
    with ThreadPoolExecutor(32) as ex:
        files = list(ex.map(lambda i: synth.generate(seed=synth.BASE_SEED + i, level=level, rows=rows, nblocks=nblocks, **extra), range(n)))
    ref = None
    for parse, name in ((capi.PARSE_HOST, "host"), (capi.PARSE_DEVICE, "device")):
        for rep in range(2):
            res, tm = capi.batch_decode(dev, files, parse=parse)
        if ref is None:
            ref = res
        else:
            assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(ref, res)), "device parse != host parse"
        print("%6d x L%d R%d B%d  %-6s: stage %.4f s (%.0f Msamples/s) h2d %.4f kernel %.4f d2h %.4f total %.4f  dev %d host %d"
              % (n, level, rows, nblocks, name, tm.stage_s, tm.samples / tm.stage_s / 1e6, tm.h2d_s, tm.kernel_s, tm.d2h_s,
                 tm.total_s, tm.device_parsed, tm.host_parsed), flush=True)
