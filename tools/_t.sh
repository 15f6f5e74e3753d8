cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -8
