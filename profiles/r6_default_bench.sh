cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6k
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6k/bench_default.json 2> gpurun_out/r6k/bench_default.err
echo "wall seconds: $SECONDS"; tail -c 300 gpurun_out/r6k/bench_default.err
python3 - <<PY
import json
j=json.loads(open("gpurun_out/r6k/bench_default.json").read().strip().split("\n")[-1])
r=j["roofline"]
print("value", j["value"], "ms_per_step", j["ms_per_step"], "frac", r["frac"], "launch_ms", r["launch_ms"], "traffic", r["traffic"], "copy", r.get("d2d_copy_gbs"), r.get("d2d_copy_same_arenas_gbs"))
print(json.dumps(r.get("other_configs"), indent=1))
print(json.dumps(j.get("cpu_baseline")))
print(json.dumps(j.get("power")))
print(json.dumps(j.get("end_to_end"))[:1500])
PY
