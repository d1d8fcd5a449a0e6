#!/bin/bash
# kernel trace of the device-parse batch path: profiles/parse_trace.sh <tag> <shape...>
set -u
TAG=${1:-parse}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/parse_probe.py "$@" > $OUT/probe.txt 2> $OUT/trace.err
cat $OUT/probe.txt
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'parse' in r['Kernel_Name']:
        print("%-40s %9.3f ms grid %s x %s" % (r['Kernel_Name'].split('(')[1][-30:] if False else r['Kernel_Name'][:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r['Grid_Size_X'], r['Grid_Size_Y']))
PY
