#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block cites (run on the GPU box via gpurun):
#   pass 1: --kernel-trace --stats            -> per-kernel average duration
#   pass 2/3: --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass, no other trace domains)
# usage: tools/collect_profiles.sh <tag> [bench.py args...]; results under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_$C.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT
