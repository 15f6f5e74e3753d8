cd $GRAFT_REPO_ROOT
for v in default wfb default wfb; do
  if [ $v != default ]; then export COPER_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib_$v.so; else unset COPER_HIP_LIB; fi
  echo "== $v (default: k-step-major weight planes; wfb: feature-block-major)"; python tools/ab_tail.py 2>&1 | grep avg | tr '\n' ' '; echo
done
unset COPER_HIP_LIB
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
