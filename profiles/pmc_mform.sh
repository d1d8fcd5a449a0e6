#!/bin/bash
# PMC passes over profiles/mform_probe.py (GPU box): both forms run in one process; the matrix-core build of acm_tile2 has
# MFORM = true in its template arguments ("Lb1E" in the mangled name), so one pass gives both.
# usage: profiles/pmc_mform.sh <tag> [probe args...]
set -u
TAG=${1:-x}; shift || true
OUT=gpurun_out/pmcm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- python3 profiles/mform_probe.py --rounds 1 --steps 4 --verify 0 "$@" > $OUT/$name.log 2> $OUT/$name.err
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = None
        if "acm_tile2<" in k or "acm_tile2I" in k:
            name = "acm_tile2 matrix-core build" if ("true" in k.split("acm_tile2")[1][:120] or "Lb1E" in k) else "acm_tile2"
        if name:
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        print("  %-28s %16.0f  (%d dispatches)" % (c, sum(v) / len(v), len(v)))
PY
