cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6e
timeout 1200 python3 profiles/ab_kernels.py --form byteplane --own-form --level 9 --rounds 4 --steps 60 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/ldnt.so libacm_amd/lib/exp/ldsc1.so libacm_amd/lib/exp/ldsc0sc1.so libacm_amd/lib/exp/ldsc1nt.so libacm_amd/lib/exp/ldsc0sc1nt.so libacm_amd/lib/exp/ldsc0.so 2>&1 | grep -v "amdgpu.ids\|own byte-plane" > gpurun_out/r6e/ab9.txt 2>&1
cat gpurun_out/r6e/ab9.txt
