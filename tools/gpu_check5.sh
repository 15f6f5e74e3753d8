cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or hipgraph or edge" 2>&1 | tail -5
