cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-scale --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused tail: value %.3fM ms/step %.4f tail %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['tail_frac']))"
COPER_TAIL_UNFUSED=1 timeout 300 python bench.py --no-cpu-baseline --no-scale --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused tail: value %.3fM ms/step %.4f tail %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['tail_frac']))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tail -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale --no-extras --steps 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_tail -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    m=re.search(r"(k_\w+)", r["Name"]); print("%-40s calls %5s avg %8.1f us" % (m.group(1) if m else r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
