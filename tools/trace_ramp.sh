# per-kernel durations of the LAST timed region's 20 passes of bench.py, pass by pass (rocprofv3 kernel trace): does a region start slow?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trace_ramp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_ramp -- python3 $R/bench.py --no-cpu-baseline --no-scale --no-extras --steps 20 --warmup 5 > $R/gpurun_out/trace_ramp.log 2>&1
cd $R
f=$(find gpurun_out/trace_ramp -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_rel_hist_scan" in r["Kernel_Name"]]
idx = idx[-20:] + [len(rows)]
print("pass: period_us | " + "per-kernel us")
for a, b in zip(idx, idx[1:]):
    t0 = int(rows[a]["Start_Timestamp"])
    period = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3 if b < len(rows) else float("nan")
    out = []
    for r in rows[a:b]:
        n = r["Kernel_Name"]; n = n[n.find("k_"):] if "k_" in n else n[:20]
        out.append("%s %.1f" % (n.split("(")[0].split("<")[0][2:14], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print("%7.1f | %s" % (period, "  ".join(out)))
PY
