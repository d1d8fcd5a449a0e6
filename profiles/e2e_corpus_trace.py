"""The configs[2] corpus (4000 ragged files, levels 7-9) through the device-parse batch: second call traced (ACM_BATCH_TRACE).
usage: python3 profiles/e2e_corpus_trace.py   (GPU box; under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import sys, os
sys.path.insert(0, ".")
from libacm_amd import capi, workload
shapes = workload.corpus_shapes(4000)
b = workload.build_corpus(4000, shapes=shapes, seed0=0, keep_files=1 << 30, threads=16)
files = [f.tobytes() for f in b.files]
dev = capi.Device(0)
capi.batch_decode(dev, files, parse=capi.PARSE_DEVICE)
print("---- second call", file=sys.stderr, flush=True)
os.environ["ACM_BATCH_TRACE"] = "1"
res, tm = capi.batch_decode(dev, files, parse=capi.PARSE_DEVICE)
print("total %.3f parse %.3f h2d %.3f d2h %.3f device_parsed %d host_parsed %d" % (tm.total_s, tm.stage_s, tm.h2d_s, tm.d2h_s, tm.device_parsed, tm.host_parsed), file=sys.stderr)
