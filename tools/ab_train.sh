#!/bin/bash
# A/B of builds on one box, training step: tools/ab_train.sh <workload> <libA.so> <libB.so> ... -- alternating runs of tools/bench_train.py, 3 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
W=$1; shift
for i in 1 2 3; do
  for L in "$@"; do
    COPER_HIP_LIB=$L python tools/bench_train.py $W 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$L $W: %.4f ms/step' % d['ms_per_step'])"
  done
done
