#!/bin/bash
# After `gpurun -- 'for t in ...; do profiles/run_profile.sh r3_<tag> <bench args>; done'` (see the tags below):
# copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked) and rebuild r3_traffic.json.
set -e
cd "$(dirname "$0")/.."
for t in r3_level9 r3_level7 r3_level11 r3_config5 r3_level13 r3_level14; do
  src=gpurun_out/prof_$t
  [ -d $src ] || continue
  cp $src/summary.txt profiles/${t}_summary.txt
  cp $(ls -t $src/trace/*/*kernel_stats.csv | head -1) profiles/${t}_kernel_stats.csv       # newest: gpurun merges every call into gpurun_out/
  cp $src/bench_trace.json profiles/${t}_bench_profiled.json
  cp $src/bench_unprofiled.json profiles/${t}_bench.json
done
python3 profiles/traffic_json.py level9_1024x250blocks_rows16=profiles/r3_level9_summary.txt level7_1024x1000blocks_rows16=profiles/r3_level7_summary.txt \
  level11_1024x16blocks_rows64=profiles/r3_level11_summary.txt level11_65536x2blocks_rows64_ch2=profiles/r3_config5_summary.txt \
  level13_1024x16blocks_rows16=profiles/r3_level13_summary.txt level14_1024x16blocks_rows8=profiles/r3_level14_summary.txt
