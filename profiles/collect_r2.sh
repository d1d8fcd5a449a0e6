#!/bin/bash
# After `gpurun -- 'profiles/run_profile.sh r2_level9; profiles/run_profile.sh r2_level7 --level 7 --blocks 1000'`:
# copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked) and rebuild r2_traffic.json.
set -e
cd "$(dirname "$0")/.."
for t in r2_level9 r2_level7; do
  src=gpurun_out/prof_$t
  cp $src/summary.txt profiles/${t}_summary.txt
  cp $(ls -t $src/trace/*/*kernel_stats.csv | head -1) profiles/${t}_kernel_stats.csv       # newest: gpurun merges every call into gpurun_out/
  cp $src/bench_trace.json profiles/${t}_bench_profiled.json
  cp $src/bench_unprofiled.json profiles/${t}_bench.json
done
python3 profiles/traffic_json.py level9_1024x250blocks_rows16=profiles/r2_level9_summary.txt level7_1024x1000blocks_rows16=profiles/r2_level7_summary.txt
