"""device-parse batch of the headline workload, int16 or byte-plane staging (argv[1] = 0|1), a few calls: run under rocprofv3 --kernel-trace --stats"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libacm_amd import capi, workload
bp = len(sys.argv) > 1 and sys.argv[1] == "1"
b = workload.build_uniform(1024, 9, 16, 250, keep_files=1 << 30)
files = [f.tobytes() for f in b.files]
dev = capi.Device(0)
for _ in range(4):
    res, tm = capi.batch_decode(dev, files, threads=0, parse=capi.PARSE_DEVICE, byteplane=bp)
    print("byteplane=%s parse %.3f total %.3f packed %d" % (bp, tm.stage_s, tm.total_s, tm.packed_streams), flush=True)
    del res
