# per-launch timeline of ONE ranking pass inside bench.py's timed region (rocprofv3 kernel trace):  bash tools/trace_pass.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trace_pass
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_pass -- python3 $R/bench.py --no-cpu-baseline --no-scale --no-extras --steps 20 --warmup 3 $TRACE_ARGS > $R/gpurun_out/trace_pass.log 2>&1
cd $R
f=$(find gpurun_out/trace_pass -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# passes start with the grouping kernel -- or, grouped ahead (coper_group_next), with the encoder; take the last complete one
tail = rows[-60:]
first = "k_rel_hist_scan" if sum("k_rel_hist_scan" in r["Kernel_Name"] for r in tail) >= 3 else "k_dense_fused"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
print("one pass of bench.py (FB15k-237-shaped, 20,480 queries), start offsets and durations in us:")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]; n = n[n.find("k_"):] if "k_" in n else n
    print("%8.1f  %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n.split("(")[0][:70]))
print("pass period (start of this pass to start of the next): %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
