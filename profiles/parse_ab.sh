#!/bin/bash
# A/B of the device walk kernels on one box (library built with ACM_TUNING=1, which keeps round 1's scalar walk):
#   profiles/parse_ab.sh <tuning-lib.so> <shape...>      ->  gpurun_out/parse_ab.txt
# ACM_PARSE_SCAN: 0 = one stream per lane, 1 = round 1's one stream per wavefront on the scalar unit, 2 = wave-per-stream walk
LIB=$1; shift
export ACM_HIP_LIB=$LIB ACM_PARSE_WAVE_MAX=1000000
mkdir -p gpurun_out
{
for sc in mix 0 8 29 17 24; do
  for m in 0 1 2; do
    if [ $sc = mix ]; then unset SINGLE_CODE; else export SINGLE_CODE=$sc; fi
    echo "== columns: $sc   ACM_PARSE_SCAN=$m"
    ACM_PARSE_SCAN=$m timeout 300 bash profiles/parse_trace.sh ab_${sc}_$m "$@" 2>&1 | grep "acm_parse_scan\| device" | awk '!seen[$1 $2 $3 $4 $5 $6]++ || /device/'
  done
done
} > gpurun_out/parse_ab.txt 2>&1
cat gpurun_out/parse_ab.txt
