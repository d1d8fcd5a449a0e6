cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r6_pytest_c.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r6_smoke_c.txt 2>&1
bash profiles/r6_cli_small.sh > gpurun_out/r6_cli_small.txt 2>&1
cat gpurun_out/r6_pytest_c.txt gpurun_out/r6_smoke_c.txt gpurun_out/r6_cli_small.txt
