"""Reduce tools/collect_profiles.sh output to the two files committed under profiles/:
kernel_stats.csv (copied) and pmc_traffic.json (HBM bytes per launch per kernel).
Bytes = counter x 1024 (FETCH_SIZE / WRITE_SIZE count KiB); FETCH_SIZE is doubled on gfx950 as
/opt/skills/guides/MI355X_MICROARCH.md prescribes; WRITE_SIZE is taken as is."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def counter_means(d, counter):
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[name] += float(r["Counter_Value"])
            n[name] += 1
    return {k: acc[k] / n[k] for k in acc}


def main(out):
    fetch = counter_means(os.path.join(out, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = counter_means(os.path.join(out, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": int(round((2.0 * f + w) * 1024))}
    line = None
    for ln in open(os.path.join(out, "bench_stats.log")):
        if ln.startswith("{"):
            line = json.loads(ln)
    doc = {"_doc": "HBM-side traffic per launch from rocprofv3 PMC passes (one counter per pass, --kernel-trace only). "
                   "Bytes = counter x 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide "
                   "coalesced read); WRITE_SIZE exact.",
           "command": "tools/collect_profiles.sh (rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv "
                      "-- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline)",
           "kernels": kernels}
    if line:
        import re
        wl = line["config"]["workload"]
        doc["workload"] = wl.split(":")[0]
        doc["queries"] = int(re.search(r"Q=(\d+)", wl).group(1))
        doc["score_mode"] = line["config"]["score_mode"].split(" ")[0]
        doc["bench_line"] = line
    json.dump(doc, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(out, "kernel_stats.csv"))
    print("wrote", os.path.join(out, "pmc_traffic.json"), len(kernels), "kernels")


if __name__ == "__main__":
    main(sys.argv[1])
