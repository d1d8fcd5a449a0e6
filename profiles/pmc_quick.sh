#!/bin/bash
# Quick PMC comparison of tile-kernel variants (GPU box): profiles/pmc_quick.sh <tag> [bench args...]
# Environment (ACM_K1_VARIANT, ACM_K1_CARRY ...) is inherited by the profiled bench.py.
set -u
TAG=${1:-x}; shift || true
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu --no-extra $*"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 bench.py $ARGS > $OUT/bench_$name.json 2> $OUT/pmc_$name.err
done
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
grep -h "per-dispatch" $OUT/summary.txt | awk '{printf "%-24s %16.0f\n", $(NF-4), $(NF-2)}'
grep -ho '"value": [0-9.]*' $OUT/bench_GRBM*.json
