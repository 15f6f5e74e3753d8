#!/bin/bash
# A/B of builds on one box: tools/ab_bench.sh <workload> <libA.so> <libB.so> ... -- alternating runs of the main line (no extras), 3 rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
W=$1; shift
for i in 1 2 3; do
  for L in "$@"; do
    COPER_HIP_LIB=$L python bench.py --no-cpu-baseline --no-extras --workload $W --steps 200 --warmup 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('$L $W: %.4f ms/step; dominant %.4f ms; other %s' % (d['ms_per_step'], r.get('avg_launch_ms',0), r.get('other_launches_ms')))"
  done
done
