cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in clk c_nogl c_nolds c_noepi c_none c_pd9 c_pd3 clk; do
  COPER_HIP_LIB=$PWD/build/ab/lib_$v.so timeout 300 python tools/ab_score.py fb15k237_cpg 20480 100 2>&1 | tail -2
done
for v in clk c_pd9 c_pd3 c_nogl; do
  COPER_HIP_LIB=$PWD/build/ab/lib_$v.so timeout 300 python tools/ab_score.py synth10m_cpg 128 30 2>&1 | tail -2
done
} > gpurun_out/ab6.txt 2>&1
cat gpurun_out/ab6.txt
