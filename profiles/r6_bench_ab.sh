# the whole bench.py of the round-5 tree (.r5tree, built in place) against this tree's, alternating, on ONE box: bash profiles/r6_bench_ab.sh [bench args]
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6f
ARGS="$*"
for rep in ${REPS:-1 2 3}; do
  for t in r5 r6; do
    if [ $t = r5 ]; then d=.r5tree; else d=.; fi
    ( cd $d && timeout 600 python3 bench.py --no-extra --no-cpu --steps 100 --warmup 20 $ARGS 2>/dev/null | tail -1 ) > gpurun_out/r6f/${t}_$rep.json
    python3 - <<PY
import json
j=json.load(open("gpurun_out/r6f/${t}_$rep.json"))
print("$t rep $rep [$ARGS]: ms_per_step", j["ms_per_step"], "launch_ms", j["roofline"]["launch_ms"], "frac", j["roofline"]["frac"], "verified", j["verified_streams"])
PY
  done
done 2>&1 | tee -a gpurun_out/r6f/summary_levels.txt
