#!/bin/bash
# first contact of the chunk kernel (acm_chunk) with the GPU: parity of the byte-plane suite, then the headline bench with and without it
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_byteplane.py -x -q 2>&1 | tail -15 > gpurun_out/k3_tests.txt
cat gpurun_out/k3_tests.txt
for k3 in 1 0 1 0; do
  ACM_K3=$k3 timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-packed 2>gpurun_out/k3_bench_err_$k3.txt | tail -1 > gpurun_out/k3_bench_$k3.json
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/k3_bench_$k3.json"))
    print("K3=$k3", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("kernel"), j.get("verified_streams"), j.get("power"))
except Exception as e:
    print("K3=$k3 failed", e); print(open("gpurun_out/k3_bench_err_$k3.txt").read()[-3000:])
PY
done
