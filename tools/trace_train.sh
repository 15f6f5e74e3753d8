R=${GRAFT_REPO_ROOT:-/root/repo}
# per-dispatch durations of one training step (the last one of the run): `bash tools/trace_train.sh [workload]`
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_train -- python3 $R/tools/bench_train.py ${1:-fb15k237_cpg} > $R/gpurun_out/trace_train.log 2>&1
cd $R
f=$(find gpurun_out/trace_train -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last k_tr_conv_fwd on
idx = max(i for i, r in enumerate(rows) if "k_tr_conv_fwd" in r["Kernel_Name"])
# the step starts a few launches earlier (zero list); take from the previous amsgrad + 1
prev = max(i for i, r in enumerate(rows[:idx]) if "k_tr_amsgrad" in r["Kernel_Name"]) + 1
step = rows[prev:]
t0 = int(step[0]["Start_Timestamp"])
tot = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%8.1f  %7.1f us  grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", "?"), name[-60:]))
print("span %.1f us, kernels %.1f us" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e3, tot / 1e3))
PY
tail -1 gpurun_out/trace_train.log
