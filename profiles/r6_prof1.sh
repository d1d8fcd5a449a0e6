cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r6_pytest_a.txt 2>&1
bash profiles/run_profile.sh r6_level9 > gpurun_out/r6_prof_level9.log 2>&1
tail -5 gpurun_out/r6_pytest_a.txt; tail -60 gpurun_out/prof_r6_level9/summary.txt
