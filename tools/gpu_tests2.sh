cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_synth10m.py tests/test_gpu_train.py -x -q --durations=8 > gpurun_out/pytest_gpu2.txt 2>&1
tail -30 gpurun_out/pytest_gpu2.txt
for w in fb15k237_cpg wn18rr_cpg fb15k237_plain; do timeout 300 python tools/bench_train.py $w 2>&1 | tail -1; done | tee gpurun_out/bench_train_r02a.txt
