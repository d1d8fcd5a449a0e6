"""One big allocation, the staged-index and PCM arenas at chosen places inside it (GPU box): order and distance."""
import sys
sys.path.insert(0, '.')
import numpy as np
from libacm_amd import capi, workload
level, blocks = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7, 1000)
dev = capi.Device(0)
b = workload.build_uniform(1024, level, 16, blocks, seed0=1 << 20)
flat = b.idx.view(np.uint8)
N = b.idx.nbytes
big = dev.malloc(3 * N + (256 << 20))
d_hdr = dev.malloc(b.hdr.nbytes)
dev.upload(d_hdr, b.hdr)
plan = capi.Plan(dev, b.descs)
def put_idx(at):
    for o in range(0, flat.size, 1 << 28):
        dev.upload(big + at + o, flat[o:o + (1 << 28)])
def t(tag, i_at, p_at):
    for _ in range(40):
        plan.launch(big + i_at, d_hdr, big + p_at)
    ms = plan.time(big + i_at, d_hdr, big + p_at, reps=100) / 100
    print("%-40s idx +%5d MB  pcm +%5d MB: %.4f ms frac %.4f" % (tag, i_at >> 20, p_at >> 20, ms, b.samples * 4 / ms / 1e6 / 8000), flush=True)
print("big %#x, arena %d MB" % (big, N >> 20))
put_idx(0)
for gap in (0, 2 << 20, 34 << 20, 128 << 20):
    t("idx below pcm, gap %d MB" % (gap >> 20), 0, N + gap)
t("idx below pcm, far", 0, 2 * N + (128 << 20))
put_idx(2 * N + (128 << 20))
for gap in (0, 2 << 20, 34 << 20, 128 << 20):
    t("idx above pcm, gap %d MB" % (gap >> 20), 2 * N + (128 << 20), N + (128 << 20) - gap)
t("idx above pcm, far", 2 * N + (128 << 20), 0)
