cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6a
for r in "4 12" "12 12" "5 7" "4 9" "8 11"; do set -- $r; echo "== pwr $1..$2"; timeout 300 python3 profiles/mform_probe.py --level 9 --pwr-min $1 --pwr-max $2 --rounds 3 --steps 60 --verify 2 2>&1 | grep -v "^int16\|amdgpu.ids"; done > gpurun_out/r6a/pwr_sens.txt 2>&1
for p in 0 33 56 100; do timeout 120 ./profiles/ubench/phases_k3.bin 9 16 $p; done > gpurun_out/r6a/phases9.txt 2>&1
timeout 120 ./profiles/ubench/phases_k3.bin 12 64 56 >> gpurun_out/r6a/phases9.txt 2>&1
cat gpurun_out/r6a/pwr_sens.txt gpurun_out/r6a/phases9.txt
