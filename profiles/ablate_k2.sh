#!/bin/bash
# Timing-only ablations of acm_tile2 at level 9 (library built with ACM_ABLATION=1): what each part of the tile loop costs.
# masks: 1 no staged-index loads, 2 no butterflies in the LDS passes, 4 none in the first pass, 8 no LDS passes at all,
#        16 no PCM stores, 32 no barriers inside the LDS passes (sums combine)
for m in 0 1 2 4 6 8 16 17 23 25 31 32; do
  v=$(ACM_K2_ABL=$m python3 bench.py --level 9 --blocks 250 --steps 60 --warmup 10 --no-extra --no-cpu --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_ms'], d['value'])")
  echo "mask $m: launch_ms Msamples/s = $v"
done
