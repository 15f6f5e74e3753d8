cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or full_size" 2>&1 | tail -2
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-scale --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.3fM ms/step %.4f dense %.4f score %.4f tail %.3f' % (d['value']/1e6, d['ms_per_step'], r['all_kernels']['k_dense_fused_bf16x3']['avg_launch_ms'], r['avg_launch_ms'], r['tail_frac']))"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tail2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale --no-extras --steps 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_tail2 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    m=re.search(r"(k_\w+)", r["Name"]); print("%-40s calls %5s avg %8.1f us" % (m.group(1) if m else r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
