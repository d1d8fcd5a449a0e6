cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/r6_fuzz_final.txt
timeout 2400 python3 profiles/byteplane_fuzz.py 4000 424242 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r6_fuzz_final.txt
timeout 1800 python3 profiles/batch_fuzz.py 2500 515151 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r6_fuzz_final.txt
cat gpurun_out/r6_fuzz_final.txt
