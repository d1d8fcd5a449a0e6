cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r6_pytest_d.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r6_smoke_d.txt 2>&1
timeout 900 python3 profiles/byteplane_fuzz.py 600 9363 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r6_fuzz_d.txt
cat gpurun_out/r6_pytest_d.txt gpurun_out/r6_smoke_d.txt gpurun_out/r6_fuzz_d.txt
