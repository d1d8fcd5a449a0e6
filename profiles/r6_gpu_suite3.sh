cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r6_pytest_h.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r6_smoke_h.txt 2>&1
: > gpurun_out/r6_fuzz_i.txt
for seed in 66001 66002; do
  timeout 900 python3 profiles/byteplane_fuzz.py 800 $seed 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r6_fuzz_i.txt
done
timeout 900 python3 profiles/batch_fuzz.py 600 1121 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r6_fuzz_i.txt
cat gpurun_out/r6_pytest_h.txt gpurun_out/r6_smoke_h.txt gpurun_out/r6_fuzz_i.txt
