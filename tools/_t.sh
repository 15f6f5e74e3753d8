cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -3
for i in 1 2 3; do python tools/bench_train.py fb15k237_cpg | tail -1; done
python tools/bench_train.py fb15k237_plain | tail -1
python tools/bench_train.py wn18rr_cpg | tail -1
bash tools/trace_train.sh 2>&1 | grep -i "pack\|gemm\|span\|reduce"
