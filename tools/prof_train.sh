R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python3 $R/tools/bench_train.py fb15k237_cpg > $R/gpurun_out/prof_train.log 2>&1
cd $R
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-70s calls %4s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 gpurun_out/prof_train.log
