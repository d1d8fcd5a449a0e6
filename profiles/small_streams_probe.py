"""acm_batch_decode over many small streams (configs[4]'s shape, an eighth of its count), host parsing: byte-plane staging (the default)
against int16 staging - transfer calls per stream are what this looks for (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libacm_amd import capi, workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
b = workload.build_uniform(n, 11, 64, 2, channels=2, keep_files=1 << 30)
files = [f.tobytes() for f in b.files]
dev = capi.Device(0)
for rep in range(3):
    for bp in (None, False):
        for mode, name in ((capi.PARSE_HOST, "host"), (capi.PARSE_DEVICE, "device")):
            res, tm = capi.batch_decode(dev, files, threads=0, parse=mode, byteplane=bp)
            print("%-6s parsing, %-10s staging: parse %.3f h2d %.3f kernel %.4f total %.3f s  upload %.2f GB  second-form streams %d" %
                  (name, "byte-plane" if bp is None else "int16", tm.stage_s, tm.h2d_s, tm.kernel_s, tm.total_s, tm.h2d_bytes / 1e9, tm.packed_streams), flush=True)
            del res
