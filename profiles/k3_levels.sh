#!/bin/bash
# the chunk kernel against acm_tile2's matrix build, level by level: profiles/k3_levels.sh "9 10 11" [bench args]
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
LEVELS=${1:-"9 10 11"}; shift || true
for lv in $LEVELS; do
  case $lv in 7) R=16; B=1000;; 8) R=16; B=500;; 9) R=16; B=250;; 10) R=16; B=125;; 11) R=64; B=16;; 12) R=64; B=8;; *) R=64; B=4;; esac
  for k3 in 1 0; do
    ACM_K3=$k3 timeout 600 python bench.py --level $lv --rows $R --blocks $B --steps 20 --warmup 5 --no-extra --no-packed --no-cpu "$@" 2>gpurun_out/k3_lv${lv}_$k3.err | tail -1 > gpurun_out/k3_lv${lv}_$k3.json
    python - <<PY
import json
try:
    j=json.load(open("gpurun_out/k3_lv${lv}_$k3.json"))
    o=[(x["form"][:5], x["frac"]) for x in j.get("other_staged_forms",[])]
    print("level $lv K3=$k3", j["ms_per_step"], j["roofline"]["frac"], "verified", j.get("verified_streams"), o)
except Exception as e:
    print("level $lv K3=$k3 failed", e); print(open("gpurun_out/k3_lv${lv}_$k3.err").read()[-2000:])
PY
  done
done
