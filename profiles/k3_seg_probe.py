"""EXPERIMENT: the chunk kernel's table in time-major order (ACM_K3_SEG=S, acm_hip_api.cpp) against the plain stream-major order, on the
same arenas, for a few fresh allocations (GPU box; library built with ACM_TUNING=1, profiles/build_variant.sh).  PCM compared with the plain order's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libacm_amd import capi, workload
level, rows, blocks = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (9, 16, 250)
dev = capi.Device(0)
b = workload.build_uniform(1024, level, rows, blocks, keep_files=0)
mf = capi.mform_streams(b.idx, b.descs, threads=workload.usable_cpus())
NB, NP = mf.data.nbytes, b.idx.nbytes
d_idx = dev.malloc(b.idx.nbytes)
d_hdr = dev.malloc(b.hdr.nbytes)
d_pairs = dev.malloc(mf.pairs.nbytes)
dev.upload(d_hdr, b.hdr)
dev.upload(d_pairs, mf.pairs)
for o in range(0, b.idx.nbytes, 1 << 28):
    dev.upload(d_idx + o, b.idx.view(np.uint8)[o:o + (1 << 28)])
junk = []
ref = None
probe = [(0, 32 << 20), (NP // 2 // 4096 * 4096, 32 << 20), (NP - (32 << 20), 32 << 20)]
for trial in range(3):
    blob = dev.malloc(NB)
    pcm = dev.malloc(NP)
    for o in range(0, NB, 1 << 28):
        dev.upload(blob + o, mf.data[o:o + (1 << 28)])
    for S in (0, 8, 16, 32, 64, 128, 0):
        if S:
            os.environ["ACM_K3_SEG"] = str(S)
        else:
            os.environ.pop("ACM_K3_SEG", None)
        plan = capi.Plan(dev, b.descs, packed=mf.streams)
        plan.bind_mform(blob, d_pairs)
        for _ in range(40):
            plan.launch(d_idx, d_hdr, pcm)
        ms = plan.time(d_idx, d_hdr, pcm, reps=100) / 100
        got = []
        for at, n in probe:
            o = np.empty(n, dtype=np.uint8)
            dev.download(o, pcm + at)
            got.append(o)
        if ref is None:
            ref = got
        same = all(np.array_equal(a, c) for a, c in zip(ref, got))
        print("trial %d S %4d tiles %8d: %.4f ms frac %.4f %s" % (trial, S, plan.stats().mform_tiles, ms, b.samples * 4 / ms / 1e6 / 8000, "same PCM" if same else "PCM DIFFERS"), flush=True)
        plan.destroy()
    dev.free(pcm)
    dev.free(blob)
    junk.append(dev.malloc((137 + 311 * trial) << 20))
