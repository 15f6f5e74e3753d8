#!/bin/bash
# Per-kernel SQ / cache counters of one python command (run on the GPU box): several rocprofv3 --pmc passes, --kernel-trace only
# (one small counter group per pass; the program itself follows "--": python3 <script> ...), then the means per kernel of every
# counter, for the kernels whose name contains <filter>.
#   usage: tools/pmc_kernel.sh <tag> <kernel-name-filter> <script.py> [args...]      -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; FILT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
G4="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_READ_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
G5="FETCH_SIZE"
G6="WRITE_SIZE"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -- python3 "$@" > $OUT/g$i.log 2>&1
done
python3 - "$OUT" "$FILT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, filt = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if filt in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if filt in name:
            dur[name].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for name in sorted(acc):
        lines = ["%s: %d launches traced, avg %.1f us under the counters" % (name, len(dur[name]), sum(dur[name]) / max(1, len(dur[name])))]
        for c in sorted(acc[name]):
            v = acc[name][c]
            lines.append("   %-32s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
        fh.write("\n".join(lines) + "\n")
        print("\n".join(lines))
PY
