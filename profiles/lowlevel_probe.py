"""Kernel-only rate of the levels outside the fused tile kernel's range (stage-wise kernels): python profiles/lowlevel_probe.py"""
import sys
sys.path.insert(0, '.')
from libacm_amd import capi, workload
dev = capi.Device(0)
for level, rows, blocks, streams in ((0, 16, 20000, 512), (2, 16, 20000, 512), (3, 16, 16000, 512), (4, 16, 8000, 512), (12, 16, 8, 512), (13, 4, 8, 256)):
    b = workload.build_uniform(streams, level, rows, blocks, seed0=level << 12)
    bufs = b.upload(dev)
    plan = capi.Plan(dev, b.descs)
    for _ in range(2):
        plan.launch(*bufs)
    ms = plan.time(*bufs, reps=5) / 5
    st = plan.stats()
    print("level %2d rows %2d: %8.1f Gsamples/s  (%d launches per step, %.1f Msamples)" % (level, rows, b.samples / ms / 1e6, st.launches, b.samples / 1e6), flush=True)
    plan.destroy()
    for p in bufs:
        dev.free(p)
