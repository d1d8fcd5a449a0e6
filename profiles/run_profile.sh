#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + PMC passes for one bench.py workload.
# usage: profiles/run_profile.sh <tag> [bench args...]      (default workload = bench.py's default: level 9)
# Counters are collected in passes of their own (never together with the trace), as MI355X_MICROARCH.md prescribes;
# FETCH_SIZE is doubled by summarize.py / traffic_json.py (gfx950 counts 128-byte requests at 64 B).
set -u
# PROFILE_LIGHT=1: kernel trace + the two traffic passes + the clock pass only (big workloads whose staging takes minutes)
TAG=${1:-r3}; shift || true
OUT=gpurun_out/prof_$TAG
# one summary = one build: refuse a directory that already holds a collection (round 5 merged four builds into one summary that way)
if [ -d "$OUT" ] && [ -n "$(ls -A "$OUT" 2>/dev/null)" ]; then
  echo "run_profile.sh: $OUT is not empty - pick another tag or remove it (a summary must come from ONE build)" >&2; exit 2
fi
mkdir -p $OUT
# what was profiled: hash of the kernel source and of the library on this box
{ echo "tag $TAG"; echo "args $*"; date -u +"utc %Y-%m-%dT%H:%M:%SZ"; sha256sum libacm_amd/csrc/acm_kernels.hip libacm_amd/csrc/acm_hip_api.cpp libacm_amd/lib/libacm_hip.so bench.py; } > $OUT/BUILD_STAMP.txt 2>&1
export TMPDIR=/tmp
ARGS="--steps 100 --warmup 20 --no-cpu --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
python3 bench.py $ARGS > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
if [ -n "${PROFILE_LIGHT:-}" ]; then GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"); else GROUPS_=(); fi
for grp in "${GROUPS_[@]}"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-extra --no-verify $* > /dev/null 2> $OUT/pmc_$name.err
done
[ -n "${PROFILE_LIGHT:-}" ] || for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extra $* > /dev/null 2> $OUT/pmc_$name.err
done
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
