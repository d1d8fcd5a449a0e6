import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from libacm_amd import capi, synth
f = synth.generate(seed=1, level=7, rows=16, nblocks=1000)
t=time.perf_counter(); st = capi.stage_file(f); dt=time.perf_counter()-t
print("1 thread numpy dst: %.1f Msamples/s" % (st.idx.size/dt/1e6))
files=[synth.generate(seed=i, level=7, rows=16, nblocks=1000) for i in range(256)]
dev=capi.Device(0)
for th in (8, 32, 64, 128, 256):
    res, tm = capi.batch_decode(dev, files, threads=th)
    print("threads %3d: alloc %.3f parse %.3f s (%.0f Msamples/s aggregate), h2d %.3f kernel %.4f d2h %.3f total %.3f" % (th, tm.alloc_s, tm.stage_s, tm.samples/tm.stage_s/1e6, tm.h2d_s, tm.kernel_s, tm.d2h_s, tm.total_s))
