cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6z
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6z/bench_default.json 2> gpurun_out/r6z/bench_default.err; echo "default bench wall seconds: $SECONDS" > gpurun_out/r6z/wall.txt
for t in "level9" "level9_int16 --form int16" "level12 --level 12 --rows 64 --blocks 8" "level13 --level 13 --rows 64 --blocks 4" "level14 --level 14 --rows 8 --blocks 16" "level7 --level 7 --rows 16 --blocks 1000" "level8 --level 8 --rows 16 --blocks 500" "level10 --level 10 --rows 16 --blocks 125" "level11 --level 11 --rows 64 --blocks 16"; do
  set -- $t; tag=$1; shift
  bash profiles/run_profile.sh r6_$tag "$@" > gpurun_out/r6_prof_$tag.log 2>&1
done
cat gpurun_out/r6z/wall.txt
python3 - <<PY
import json
j=json.loads(open("gpurun_out/r6z/bench_default.json").read().strip().split("\n")[-1])
r=j["roofline"]
print("value", j["value"], "ms_per_step", j["ms_per_step"], "frac", r["frac"], "launch_ms", r["launch_ms"], "traffic", r["traffic"], "copy", r.get("d2d_copy_gbs"), r.get("d2d_copy_same_arenas_gbs"))
print({k:(v["frac"],v["launch_ms"]) for k,v in r.get("other_configs",{}).items()})
PY
for t in 9 8 12 13; do grep -E "acm_chunk|acm_tile2" gpurun_out/prof_r6_level$t/summary.txt | grep -E "calls|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_WAIT_ANY|SQ_WAVE_CYCLES" | cut -c1-140; done
