"""host-parsed batch with and without ACM_BATCH_STAGE_BYTEPLANE on the headline workload: parse_s / total_s of acm_batch_decode
(VERDICT r4 task 3: the byte-plane staging within 5 % of the int16 staging).  python3 profiles/host_stage_probe.py [level rows blocks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libacm_amd import capi, workload
lv, rows, blocks = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (9, 16, 250)
b = workload.build_uniform(1024, lv, rows, blocks, keep_files=1 << 30)
files = [f.tobytes() for f in b.files]
dev = capi.Device(0)
for mode in ("int16", "byteplane", "int16", "byteplane", "int16", "byteplane"):
    res, tm = capi.batch_decode(dev, files, threads=0, parse=capi.PARSE_HOST, byteplane=mode == "byteplane")
    print("%-10s parse %.3f s  h2d %.3f s  total %.3f s  upload %.2f GB  second-form streams %d" % (mode, tm.stage_s, tm.h2d_s, tm.total_s, tm.h2d_bytes / 1e9, tm.packed_streams), flush=True)
    del res
