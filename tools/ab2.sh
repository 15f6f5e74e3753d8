cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 build/mb/l2stream > gpurun_out/l2stream.txt 2>&1
for v in default nomfma nomfma_nogl nomfma_nolds nomfma_none skipgl skiplds skipboth skipboth_noep; do
  if [ $v = default ]; then unset COPER_HIP_LIB; else export COPER_HIP_LIB=$PWD/build/ab/lib_$v.so; fi
  timeout 300 python tools/ab_score.py fb15k237_cpg 20480 40 2>&1 | tail -1
done > gpurun_out/ab2_fb.txt 2>&1
unset COPER_HIP_LIB
timeout 900 python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err
cat gpurun_out/l2stream.txt gpurun_out/ab2_fb.txt; tail -5 gpurun_out/bench_r02a.err; cat gpurun_out/bench_r02a.json
