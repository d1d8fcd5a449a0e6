#!/bin/bash
# A/B of libraries AND of the chunk kernel switch on one box: profiles/ab_k3.sh "<bench args>" lib1.so[:K3] lib2.so[:K3] ...
# (":0" runs the library with ACM_K3=0 = acm_tile2's matrix build; each entry is timed twice, interleaved, kernel-only)
ARGS=$1; shift
for rep in 1 2; do
  for ent in "$@"; do
    so=${ent%%:*}; k3=1; [[ "$ent" == *:* ]] && k3=${ent##*:}
    v=$(ACM_K3=$k3 ACM_HIP_LIB=$so python3 bench.py $ARGS --steps 100 --warmup 20 --no-extra --no-cpu --no-packed --no-verify $ABFLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])")
    echo "$(basename $so) K3=$k3 $v"
  done
done
