cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6l
timeout 900 python3 profiles/ab_kernels.py --form byteplane --level 13 --rows 64 --blocks 4 --rounds 3 --steps 40 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/l13a.so libacm_amd/lib/exp/l13b.so libacm_amd/lib/exp/l13c.so 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6l/ab13.txt
timeout 900 python3 profiles/ab_kernels.py --form byteplane --level 14 --rows 8 --blocks 16 --rounds 3 --steps 40 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/l14a.so libacm_amd/lib/exp/l14b.so 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6l/ab14.txt
cat gpurun_out/r6l/ab13.txt gpurun_out/r6l/ab14.txt
