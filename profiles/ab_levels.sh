#!/bin/bash
# A/B of two builds of libacm_hip.so over the levels on one box, interleaved (how the phase priorities were measured):
#   build the variants into libacm_amd/lib/exp/{base,prio}.so, then  gpurun -- 'bash profiles/ab_levels.sh'
# r2 result (frac of 8 TB/s, base -> prio): L7 0.580 -> 0.592, L8 0.575 -> 0.594, L9 0.543 -> 0.579, L10 0.481 -> 0.524,
# L11 0.449 -> 0.472, L12 0.372 -> 0.362 (priorities switched off there), corpus 0.557 -> 0.569
for cfg in "--level 6 --blocks 2000" "--level 7 --blocks 1000" "--level 8 --blocks 500" "--level 9 --blocks 250" "--level 10 --blocks 125" "--level 11 --rows 64 --blocks 16" "--level 12 --rows 16 --blocks 32" "--workload corpus"; do
  for so in base prio base prio; do
    v=$(ACM_HIP_LIB=libacm_amd/lib/exp/$so.so python3 bench.py $cfg --steps 100 --warmup 20 --no-extra --no-cpu --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['frac'])")
    echo "$cfg $so $v"
  done
done
