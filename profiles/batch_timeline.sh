#!/bin/bash
# Timeline of one device-parse acm_batch_decode (GPU box): kernel and copy start/end times relative to the first event
#   profiles/batch_timeline.sh <tag> [shape, default 1024x9x16x250]
set -u
TAG=${1:-tl}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 profiles/parse_probe.py ${1:-1024x9x16x250} > $OUT/probe.txt 2> $OUT/trace.err
cat $OUT/probe.txt
python3 - $OUT <<'PY'
import csv, sys, glob
ev = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44], r.get('Queue_Id', '')))
for f in glob.glob(sys.argv[1] + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), "copy " + r.get('Direction', r.get('Name', '')), ''))
ev.sort()
# the last device-parse call: from the last but one walk cluster on
walks = [e for e in ev if 'parse_scan' in e[2]]
t0 = walks[-4][0] - 30_000_000 if len(walks) >= 4 else ev[0][0]
for s, e, name, q in ev:
    if s >= t0 and (e - s) > 200_000:
        print("%9.3f .. %9.3f ms  (%8.3f)  %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, name, q))
PY
