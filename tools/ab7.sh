cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in clk c_r0 c_r0_nolds c_r0_nogl c_r0_none; do
  COPER_HIP_LIB=$PWD/build/ab/lib_$v.so COPER_MODE_EQ=0 timeout 300 python tools/ab_score.py fb15k237_cpg 20480 100 2>&1 | tail -2
done
} > gpurun_out/ab7.txt 2>&1
cat gpurun_out/ab7.txt
