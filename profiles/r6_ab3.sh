cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6o
( timeout 900 python -m pytest tests/test_gpu_byteplane.py -x -q -m gpu -k "13 or 14" 2>&1 | tail -5 ) > gpurun_out/r6o/pytest.txt 2>&1
timeout 900 python3 profiles/ab_kernels.py --form byteplane --level 13 --rows 64 --blocks 4 --rounds 3 --steps 40 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/l13r1.so libacm_amd/lib/exp/l13r1b.so 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6o/ab13.txt
timeout 900 python3 profiles/ab_kernels.py --form byteplane --level 14 --rows 8 --blocks 16 --rounds 3 --steps 40 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/l14r1.so libacm_amd/lib/exp/l14r2.so 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6o/ab14.txt
cat gpurun_out/r6o/pytest.txt gpurun_out/r6o/ab13.txt gpurun_out/r6o/ab14.txt
