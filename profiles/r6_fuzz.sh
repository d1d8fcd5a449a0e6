cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6n
timeout 2400 python3 profiles/byteplane_fuzz.py 4000 9262 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r6n/byteplane_fuzz.txt
timeout 1500 python3 profiles/batch_fuzz.py 2000 717 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r6n/batch_fuzz.txt
cat gpurun_out/r6n/byteplane_fuzz.txt gpurun_out/r6n/batch_fuzz.txt
