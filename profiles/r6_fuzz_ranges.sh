cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/r6_fuzz_f.txt
for seed in 88001 88002 88003; do
  timeout 900 python3 profiles/byteplane_fuzz.py 800 $seed 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r6_fuzz_f.txt
done
timeout 900 python3 profiles/batch_fuzz.py 600 818 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/r6_fuzz_f.txt
cat gpurun_out/r6_fuzz_f.txt
