cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size or bf16x3 or topk or sharded" 2>&1 | tail -5
for v1 in 0 1; do
  if [ $v1 = 1 ]; then export COPER_SCORE_V1=1; else unset COPER_SCORE_V1; fi
  timeout 300 python tools/ab_score.py fb15k237_cpg 20480 40 2>&1 | tail -1
  timeout 300 python tools/ab_score.py wn18rr_cpg 3072 40 2>&1 | tail -1
  timeout 300 python tools/ab_score.py synth10m_cpg 128 20 2>&1 | tail -1
  timeout 300 python tools/ab_score.py synth10m_cpg 4096 5 2>&1 | tail -1
done
} > gpurun_out/ab3.txt 2>&1
cat gpurun_out/ab3.txt
