# the whole-range class in FirstPassZW (levels 13 / 14): parity, and the launches of the usual workload against the library before it
# (libacm_amd/lib/exp/prev.so = profiles/build_rev.sh 7160965 prev)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6w
( timeout 1200 python -m pytest tests/test_gpu_byteplane.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r6w/pytest.txt 2>&1
timeout 900 python3 profiles/ab_kernels.py --form byteplane --own-form --level 13 --rows 64 --blocks 4 --rounds 3 --steps 40 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/prev.so 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6w/ab13.txt
timeout 900 python3 profiles/ab_kernels.py --form byteplane --own-form --level 14 --rows 8 --blocks 16 --rounds 3 --steps 40 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/prev.so 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6w/ab14.txt
cat gpurun_out/r6w/pytest.txt gpurun_out/r6w/ab13.txt gpurun_out/r6w/ab14.txt
