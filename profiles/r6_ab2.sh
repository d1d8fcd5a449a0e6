cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6r
( timeout 900 python -m pytest tests/test_gpu_byteplane.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r6r/pytest.txt 2>&1
timeout 900 python3 profiles/byteplane_fuzz.py 500 9464 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r6r/fuzz.txt
for lv in "8 500" "9 250"; do set -- $lv
timeout 1200 python3 profiles/ab_kernels.py --form byteplane --own-form --level $1 --rows 16 --blocks $2 --rounds 4 --steps 60 libacm_amd/lib/exp/r5full.so libacm_amd/lib/libacm_hip.so 2>&1 | grep -v "amdgpu.ids\|own byte\|first launch"
done > gpurun_out/r6r/ab.txt 2>&1
cat gpurun_out/r6r/pytest.txt gpurun_out/r6r/fuzz.txt gpurun_out/r6r/ab.txt
