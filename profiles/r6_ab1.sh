cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6c
( timeout 1200 python -m pytest tests/test_gpu_byteplane.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r6c/pytest.txt 2>&1
for lv in 9; do
timeout 600 python3 profiles/ab_kernels.py --form byteplane --level $lv --rounds 4 --steps 60 libacm_amd/lib/libacm_hip.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6c/ab9.txt 2>&1
timeout 300 python3 profiles/mform_probe.py --level 9 --rounds 3 --steps 60 --verify 4 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6c/probe9.txt
cat gpurun_out/r6c/pytest.txt gpurun_out/r6c/ab9.txt gpurun_out/r6c/probe9.txt
