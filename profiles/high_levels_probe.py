"""Kernel-only rate of levels 12-15 (prefix stages + plane tile kernel) at a batch big enough to fill the chip:
python profiles/high_levels_probe.py [level rows blocks streams]...   (run under rocprofv3 --kernel-trace for the per-kernel split)"""
import sys
sys.path.insert(0, '.')
from libacm_amd import capi, workload
dev = capi.Device(0)
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(12, 16, 32, 1024), (13, 16, 16, 1024), (14, 8, 16, 1024), (15, 8, 8, 1024)]
for level, rows, blocks, streams in shapes:
    b = workload.build_uniform(streams, level, rows, blocks, seed0=level << 12)
    bufs = b.upload(dev)
    plan = capi.Plan(dev, b.descs)
    for _ in range(3):
        plan.launch(*bufs)
    ms = plan.time(*bufs, reps=10) / 10
    st = plan.stats()
    print("level %2d rows %2d blocks %3d streams %d: %8.1f Gsamples/s  %.3f ms (%d launches per step, %.1f Msamples)"
          % (level, rows, blocks, streams, b.samples / ms / 1e6, ms, st.launches, b.samples / 1e6), flush=True)
    plan.destroy()
    for p in bufs:
        dev.free(p)
