#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block cites (run on the GPU box via gpurun):
#   pass 1: --kernel-trace --stats                                  -> per-kernel average duration
#   pass 2/3: --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE          -> HBM bytes per launch (one counter per pass)
#   pass 4: --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -> matrix-pipe busy fraction, effective clock
# Counter passes carry --kernel-trace only (no other trace domains).  Kernels of the 10M-entity blocks are other template
# instantiations (16 k-steps) than the FB15k-237-shaped pass (13), so one bench.py command yields separate rows for both.
# (every pass under `timeout`: a rocprofv3 run without --output-format csv once sat in its post-processing for half an hour)
# usage: tools/collect_profiles.sh <tag> [bench.py args...]; results under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_$C.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_BUSY -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_BUSY.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT
