R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/pytest_gpu.txt 2>&1
tail -40 gpurun_out/pytest_gpu.txt
