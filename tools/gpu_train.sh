#!/bin/bash
# training-step parity tests, timings and the rocprofv3 kernel breakdown (run on the GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/pytest_train.txt 2>&1; tail -4 gpurun_out/pytest_train.txt
for w in fb15k237_cpg wn18rr_cpg fb15k237_plain; do timeout 300 python tools/bench_train.py $w 2>&1 | tail -1; done | tee gpurun_out/bench_train.txt
bash tools/prof_train.sh 2>&1 | head -16
