#!/bin/bash
# A/B of experimental builds of libacm_hip.so on one box: profiles/ab_libs.sh "<bench args>" lib1.so lib2.so ...
# (each library is timed twice, interleaved, kernel-only)
ARGS=$1; shift
for rep in 1 2; do
  for so in "$@"; do
    v=$(ACM_HIP_LIB=$so python3 bench.py $ARGS --steps 100 --warmup 20 --no-extra --no-cpu $ABFLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])")
    echo "$(basename $so) $v"
  done
done
