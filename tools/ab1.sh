cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in default nomfma nogl pd5 pd2 nt; do
  if [ $v = default ]; then unset COPER_HIP_LIB; else export COPER_HIP_LIB=$PWD/build/ab/lib_$v.so; fi
  timeout 300 python tools/ab_score.py synth10m_cpg 128 20 2>&1 | tail -1
done > gpurun_out/ab1_10m.txt 2>&1
for v in default nomfma nogl; do
  if [ $v = default ]; then unset COPER_HIP_LIB; else export COPER_HIP_LIB=$PWD/build/ab/lib_$v.so; fi
  timeout 300 python tools/ab_score.py fb15k237_cpg 20480 40 2>&1 | tail -1
done > gpurun_out/ab1_fb.txt 2>&1
cat gpurun_out/ab1_10m.txt gpurun_out/ab1_fb.txt
