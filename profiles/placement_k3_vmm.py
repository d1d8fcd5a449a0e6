"""As placement_k3_realloc.py, the two arenas obtained in different ways: hipMalloc, or the virtual-memory API with physical chunks of 2 MB,
64 MB, 1 GB or one piece (profiles/ubench/copy_bw.hip: acm_vmm_alloc).  Three fresh allocations per way (GPU box)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libacm_amd import capi, workload
cb = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libcopybw.so"))
cb.acm_copy_between_gbs.restype = C.c_double
cb.acm_copy_between_gbs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
cb.acm_vmm_alloc.restype = C.c_void_p
cb.acm_vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]
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
gran = C.c_size_t()
ways = (("hipMalloc", None), ("vmm 2 MB", 2 << 20), ("vmm 64 MB", 64 << 20), ("vmm 1 GB", 1 << 30), ("vmm one piece", 0), ("hipMalloc", None))
if len(sys.argv) > 1 and sys.argv[1] == "short":
    ways = (("hipMalloc", None), ("one hipMalloc", -1), ("vmm one piece", 0), ("hipMalloc", None), ("one hipMalloc", -1), ("vmm one piece", 0))
slab_gap = 2 << 20
for way, chunk in ways:
    for trial in range(3 if len(ways) == 6 and ways[1][0] != "one hipMalloc" else 2):
        slab = None
        if chunk is None:
            blob, pcm = dev.malloc(NB), dev.malloc(NP)
        elif chunk == -1:
            up = (NB + slab_gap - 1) // slab_gap * slab_gap
            slab = dev.malloc(up + NP)
            blob, pcm = slab, slab + up
        else:
            blob, pcm = cb.acm_vmm_alloc(NB, chunk, C.byref(gran)), cb.acm_vmm_alloc(NP, chunk, C.byref(gran))
            if not blob or not pcm:
                print(way, "allocation failed", flush=True)
                break
        for o in range(0, NB, 1 << 28):
            dev.upload(blob + o, mf.data[o:o + (1 << 28)])
        plan.bind_mform(blob, d_pairs)
        for _ in range(60):
            plan.launch(d_idx, d_hdr, pcm)
        ms = plan.time(d_idx, d_hdr, pcm, reps=100) / 100
        dev.sync()
        cp = cb.acm_copy_between_gbs(pcm, blob, NB // 4096 * 4096)
        print("%-14s trial %d (granularity %d KB): %.4f ms frac %.4f; copy blob -> pcm %.0f GB/s" % (way, trial, gran.value >> 10, ms, b.samples * 4 / ms / 1e6 / 8000, cp), flush=True)
        if chunk is None:
            dev.free(pcm); dev.free(blob)
        elif slab is not None:
            dev.free(slab)
