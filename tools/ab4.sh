cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "topk" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab4 -- python3 $GRAFT_REPO_ROOT/tools/ab_score.py fb15k237_cpg 20480 40 2>&1 | tail -1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_ab4 -name "*kernel_stats.csv" | head -1); head -8 $f
} > gpurun_out/ab4.txt 2>&1
cat gpurun_out/ab4.txt
