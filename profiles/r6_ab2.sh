cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6t
timeout 1500 python3 profiles/ab_kernels.py --form byteplane --level 9 --rounds 4 --steps 60 libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/stnt.so libacm_amd/lib/exp/stplain.so libacm_amd/lib/exp/stsc1.so libacm_amd/lib/exp/stsc0sc1.so libacm_amd/lib/exp/stsc1nt.so 2>&1 | grep -v "amdgpu.ids\|own byte\|first launch" > gpurun_out/r6t/st.txt 2>&1
cat gpurun_out/r6t/st.txt
