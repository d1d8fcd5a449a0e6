import os, sys, subprocess
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
code = r'''
import os, sys
sys.path.insert(0, "%(root)s"); sys.path.insert(0, "%(root)s/tests")
import numpy as np
from helpers import make_stream, oracle_pcm
from libacm_amd import capi
lv, rows, R = %(lv)d, %(rows)d, %(R)d
os.environ["ACM_BATCH_RANGES"] = str(R)
files = [make_stream(31000 + k, lv, rows, max(3, (6 * capi.lib().acmk_tile2_rows(lv) + rows - 1) // rows + 1 + k %% 3), channels=1 + k %% 2, cut=k %% 4, pwr_max=12) for k in range(6)]
dev = capi.Device(0)
res, tm = capi.batch_decode(dev, files, threads=2, parse=capi.PARSE_DEVICE, byteplane=True)
bad = sum(0 if np.array_equal(res[k][1], oracle_pcm(f)[0]) else 1 for k, f in enumerate(files))
print("lv", lv, "rows", rows, "R", R, "packed", tm.packed_streams, "devparsed", tm.device_parsed, "bad", bad)
'''
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for lv in (9, 10, 11, 12, 8):
    for rows in (16, 64, 700, 2):
        for R in (3,):
            r = subprocess.run([sys.executable, "-c", code % dict(root=root, lv=lv, rows=rows, R=R)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            out = r.stdout.strip().splitlines()[-1:] or [""]
            err = [l for l in r.stderr.splitlines() if "fault" in l or "Error" in l]
            print(out[0] if r.returncode == 0 else "lv %d rows %d R %d: rc %d %s" % (lv, rows, R, r.returncode, err[:1]), flush=True)
