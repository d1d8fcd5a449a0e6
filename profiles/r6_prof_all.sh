cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
bash profiles/run_profile.sh r6_level9 > gpurun_out/r6_prof_level9.log 2>&1
bash profiles/run_profile.sh r6_level9_int16 --form int16 > gpurun_out/r6_prof_level9_int16.log 2>&1
bash profiles/run_profile.sh r6_level12 --level 12 --rows 64 --blocks 8 > gpurun_out/r6_prof_level12.log 2>&1
bash profiles/run_profile.sh r6_level13 --level 13 --rows 64 --blocks 4 > gpurun_out/r6_prof_level13.log 2>&1
bash profiles/run_profile.sh r6_level14 --level 14 --rows 8 --blocks 16 > gpurun_out/r6_prof_level14.log 2>&1
bash profiles/run_profile.sh r6_level7 --level 7 --rows 16 --blocks 1000 > gpurun_out/r6_prof_level7.log 2>&1
bash profiles/run_profile.sh r6_level10 --level 10 --rows 16 --blocks 125 > gpurun_out/r6_prof_level10.log 2>&1
bash profiles/run_profile.sh r6_level11 --level 11 --rows 64 --blocks 16 > gpurun_out/r6_prof_level11.log 2>&1
for t in 9 12 13 14; do echo "== level $t"; grep -E "kernel stats|acm_chunk|acm_tile2" gpurun_out/prof_r6_level$t/summary.txt | head -40; done
