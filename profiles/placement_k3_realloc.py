"""Headline workload on the byte-plane form, the arenas allocated afresh (hipMalloc) between timings inside ONE process, junk allocations of
changing size kept alive in between: does the launch time follow the allocation (GPU box)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libacm_amd import capi, workload
dev = capi.Device(0)
b = workload.build_uniform(1024, 9, 16, 250, keep_files=0)
mf = capi.mform_streams(b.idx, b.descs, threads=workload.usable_cpus())
NB, NP = mf.data.nbytes, b.idx.nbytes
d_idx = dev.malloc(b.idx.nbytes)
d_hdr = dev.malloc(b.hdr.nbytes)
d_pairs = dev.malloc(mf.pairs.nbytes)
dev.upload(d_hdr, b.hdr)
dev.upload(d_pairs, mf.pairs)
plan = capi.Plan(dev, b.descs, packed=mf.streams)
import ctypes as C
cb = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libcopybw.so"))
cb.acm_copy_between_gbs.restype = C.c_double
cb.acm_copy_between_gbs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
junk = []
for trial in range(8):
    blob = dev.malloc(NB)
    pcm = dev.malloc(NP)
    for o in range(0, NB, 1 << 28):
        dev.upload(blob + o, mf.data[o:o + (1 << 28)])
    plan.bind_mform(blob, d_pairs)
    res = []
    for _ in range(3):
        for _ in range(60):
            plan.launch(d_idx, d_hdr, pcm)
        res.append(plan.time(d_idx, d_hdr, pcm, reps=100) / 100)
    dev.sync()
    cp = cb.acm_copy_between_gbs(pcm, blob, NB // 4096 * 4096)        # (garbage into the PCM arena: timed, not looked at)
    print("trial %d blob %#x pcm %#x: %s ms, frac %.4f; copy blob -> pcm %.0f GB/s" % (trial, blob, pcm, " ".join("%.4f" % r for r in res), b.samples * 4 / min(res) / 1e6 / 8000, cp), flush=True)
    dev.free(pcm)
    dev.free(blob)
    junk.append(dev.malloc((37 + 211 * trial) << 20))
