cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_byteplane.py -x -q -m gpu -k "device_parser or batch" 2>&1 | tail -25 ) > gpurun_out/r6_devparse_pytest.txt 2>&1
cat gpurun_out/r6_devparse_pytest.txt
