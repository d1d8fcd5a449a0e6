cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6w
timeout 900 python3 profiles/ab_kernels.py --form byteplane --own-form --level 13 --rows 64 --blocks 4 --rounds 5 --steps 40 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/prev.so 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6w/ab13b.txt
timeout 900 python3 profiles/ab_kernels.py --form byteplane --own-form --level 14 --rows 8 --blocks 16 --rounds 5 --steps 40 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/prev.so 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6w/ab14b.txt
cat gpurun_out/r6w/ab13b.txt gpurun_out/r6w/ab14b.txt | grep -v "^#"
