cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size or bf16x3 or topk or sharded" 2>&1 | tail -3
for i in 1 2; do
COPER_HIP_LIB=$PWD/build/ab/lib_clk.so timeout 300 python tools/ab_score.py fb15k237_cpg 20480 100 2>&1 | tail -2
COPER_SCORE_V1=1 timeout 300 python tools/ab_score.py fb15k237_cpg 20480 100 2>&1 | tail -1
done
COPER_HIP_LIB=$PWD/build/ab/lib_clk.so timeout 300 python tools/ab_score.py synth10m_cpg 128 30 2>&1 | tail -2
COPER_SCORE_V1=1 timeout 300 python tools/ab_score.py synth10m_cpg 128 30 2>&1 | tail -1
COPER_HIP_LIB=$PWD/build/ab/lib_clk.so timeout 300 python tools/ab_score.py synth10m_cpg 4096 8 2>&1 | tail -2
COPER_SCORE_V1=1 timeout 300 python tools/ab_score.py synth10m_cpg 4096 8 2>&1 | tail -1
COPER_HIP_LIB=$PWD/build/ab/lib_clk.so timeout 300 python tools/ab_score.py wn18rr_cpg 3072 100 2>&1 | tail -2
COPER_SCORE_V1=1 timeout 300 python tools/ab_score.py wn18rr_cpg 3072 100 2>&1 | tail -1
} > gpurun_out/ab8.txt 2>&1
cat gpurun_out/ab8.txt
