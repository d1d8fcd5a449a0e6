cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6s
bash profiles/r6_copy_probe.sh > gpurun_out/r6s/copy.txt 2>&1
timeout 1200 python3 profiles/ab_kernels.py --form byteplane --own-form --level 9 --rounds 3 --steps 60 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/tuning.so libacm_amd/lib/exp/tuning.so@ACM_K3_SEG=4 libacm_amd/lib/exp/tuning.so@ACM_K3_SEG=8 libacm_amd/lib/exp/tuning.so@ACM_K3_SEG=16 libacm_amd/lib/exp/tuning.so@ACM_K3_SEG=32 2>&1 | grep -v "amdgpu.ids\|own byte\|first launch" > gpurun_out/r6s/seg.txt 2>&1
cat gpurun_out/r6s/copy.txt gpurun_out/r6s/seg.txt
