"""Headline workload on the byte-plane form: the blob and the PCM arena at chosen places inside one big allocation, every
combination timed twice (GPU box).  Is the run-to-run spread of the level-9 launch (1.57 against 1.66 ms between processes on
one box) a matter of where the two arenas sit?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libacm_amd import capi, workload
dev = capi.Device(0)
b = workload.build_uniform(1024, 9, 16, 250, keep_files=0)
mf = capi.mform_streams(b.idx, b.descs, threads=workload.usable_cpus())
NB, NP = mf.data.nbytes, b.idx.nbytes
big = dev.malloc(NB + 2 * NP + (1 << 30))
d_idx = dev.malloc(b.idx.nbytes)
d_hdr = dev.malloc(b.hdr.nbytes)
d_pairs = dev.malloc(mf.pairs.nbytes)
dev.upload(d_hdr, b.hdr)
dev.upload(d_pairs, mf.pairs)
for o in range(0, b.idx.nbytes, 1 << 28):
    dev.upload(d_idx + o, b.idx.view(np.uint8)[o:o + (1 << 28)])
plan = capi.Plan(dev, b.descs, packed=mf.streams)
print("big %#x idx %#x hdr %#x pairs %#x; blob %d MB pcm %d MB" % (big, d_idx, d_hdr, d_pairs, NB >> 20, NP >> 20), flush=True)
def put_blob(at):
    for o in range(0, NB, 1 << 28):
        dev.upload(big + at + o, mf.data[o:o + (1 << 28)])
def t(tag, a_at, p_at):
    plan.bind_mform(big + a_at, d_pairs)
    for _ in range(60):
        plan.launch(d_idx, d_hdr, big + p_at)
    ms = plan.time(d_idx, d_hdr, big + p_at, reps=100) / 100
    print("%-34s blob +%5d MB pcm +%5d MB (delta %% 1 MB = %7d B): %.4f ms frac %.4f" %
          (tag, a_at >> 20, p_at >> 20, (p_at - a_at) % (1 << 20), ms, b.samples * 4 / ms / 1e6 / 8000), flush=True)
for rnd in range(2):
    put_blob(0)
    up = (NB + (1 << 21) - 1) & ~((1 << 21) - 1)
    for gap in (0, 4096, 64 << 10, 1 << 20, (1 << 20) + 8192, 34 << 20, 512 << 20):
        t("blob below pcm, gap %d KB" % (gap >> 10), 0, up + gap)
    hi = up + NP + (600 << 20)
    put_blob(hi)
    for gap in (0, 8192, 34 << 20):
        t("blob above pcm, gap %d KB" % (gap >> 10), hi, gap)
