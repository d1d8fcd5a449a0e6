#!/usr/bin/env python3
"""profiles/<tag>_summary.txt files -> profiles/r6_traffic.json: HBM bytes per launch of the tile kernel from the PMC passes
(FETCH_SIZE x 2 + WRITE_SIZE, KiB units -> bytes; MI355X_MICROARCH.md "HBM"), stamped with the sha256 of the kernel source
the profile was taken on.  bench.py reports `traffic` only while the source still has that hash.
usage: python3 profiles/traffic_json.py key=summary.txt [key=summary.txt ...]"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha       # comments and white space stripped: what the compiler sees
sha = kernel_source_sha()
out = {"kernel_source_sha16": sha, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; per dispatch of the dominant kernel; "
       "FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 B); the counters are in KiB"}
for arg in sys.argv[1:]:
    key, path = arg.split("=", 1)
    if not os.path.exists(path):
        continue
    vals = {}
    for line in open(path):
        m = re.search(r"(acm_tile2|acm_fused|acm_chunk)[^\n]*?\s+(FETCH_SIZE|WRITE_SIZE)\s+per-dispatch\s+([0-9.]+)", line)
        if m:
            vals.setdefault(m.group(2), 0.0)
            vals[m.group(2)] = max(vals[m.group(2)], float(m.group(3)))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        out[key] = {"fetch_size_kib_raw": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
                    "hbm_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), "summary": os.path.basename(path)}
json.dump(out, open(os.path.join(ROOT, "profiles", "r6_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
