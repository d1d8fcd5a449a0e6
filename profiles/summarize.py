#!/usr/bin/env python3
"""Condense a profiles/run_profile.sh output directory into a small text summary (kernel stats + PMC means)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print("# profile summary for", root)
stamp = os.path.join(root, "BUILD_STAMP.txt")
if os.path.exists(stamp):
    print("## build stamp (one collection = one build; run_profile.sh refuses a directory that already holds one)")
    for line in open(stamp):
        print("  " + line.rstrip())
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("  %-60s calls %6s  avg_ns %12s  total_ns %14s  pct %6s" % (
                row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"),
                row.get("Percentage")))
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    seen = {}
    for r in rows:
        k = r.get("Kernel_Name", "")
        if k not in seen:
            seen[k] = r
    print("## launch shapes")
    for k, r in seen.items():
        print("  %-60s grid %s wg %s lds %s vgpr %s sgpr %s" % (k[:60], r.get("Grid_Size"), r.get("Workgroup_Size"),
              r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("SGPR_Count")))
print("## PMC (mean per dispatch of the dominant kernel, summed over XCDs/instances as rocprofv3 reports)")
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: [0.0, 0])
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if not any(k in r.get("Kernel_Name", "") for k in ("fused", "acm_sw", "acm_tile", "acm_chunk")):
                continue
            key = (r["Kernel_Name"][:40], r["Counter_Name"])
            acc[key][0] += float(r["Counter_Value"])
            acc[key][1] += 1
    # rocprofv3 emits one row per (dispatch, counter[, dimension]); normalise by dispatch count
    disp = defaultdict(set)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            disp[r["Kernel_Name"][:40]].add(r["Dispatch_Id"])
    for (k, c), (tot, n) in sorted(acc.items()):
        nd = max(1, len(disp[k]))
        print("  %-42s %-24s per-dispatch %16.1f   (dispatches %d)" % (k, c, tot / nd, nd))
