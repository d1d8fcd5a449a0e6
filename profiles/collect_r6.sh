#!/bin/bash
# After `gpurun -- 'profiles/run_profile.sh r6_level9; profiles/run_profile.sh r6_level9_int16 --form int16; ...'` (tags below):
# copy the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked) and rebuild r6_traffic.json.
# Tags without a suffix are bench.py's default staged form for their level (the byte-plane form: acm_chunk at levels 9-12).
set -e
cd "$(dirname "$0")/.."
for t in r6_level9 r6_level9_int16 r6_level7 r6_level8 r6_level10 r6_level11 r6_level12 r6_level13 r6_level14 r6_config5; do
  src=gpurun_out/prof_$t
  [ -d $src ] || continue
  python3 profiles/summarize.py $src > profiles/${t}_summary.txt
  cp $(ls -t $src/trace/*/*kernel_stats.csv | head -1) profiles/${t}_kernel_stats.csv       # newest: gpurun merges every call into gpurun_out/
  cp $src/bench_trace.json profiles/${t}_bench_profiled.json
  cp $src/bench_unprofiled.json profiles/${t}_bench.json
done
python3 profiles/traffic_json.py level9_1024x250blocks_rows16_byteplane=profiles/r6_level9_summary.txt level9_1024x250blocks_rows16=profiles/r6_level9_int16_summary.txt \
  level7_1024x1000blocks_rows16_byteplane=profiles/r6_level7_summary.txt level8_1024x500blocks_rows16_byteplane=profiles/r6_level8_summary.txt level10_1024x125blocks_rows16_byteplane=profiles/r6_level10_summary.txt \
  level11_1024x16blocks_rows64_byteplane=profiles/r6_level11_summary.txt level12_1024x8blocks_rows64_byteplane=profiles/r6_level12_summary.txt \
  level13_1024x4blocks_rows64_byteplane=profiles/r6_level13_summary.txt level14_1024x16blocks_rows8_byteplane=profiles/r6_level14_summary.txt \
  level11_65536x2blocks_rows64_ch2_byteplane=profiles/r6_config5_summary.txt
