cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
COPER_HIP_LIB=$PWD/build/ab/lib_clk.so timeout 300 python tools/ab_score.py fb15k237_cpg 20480 200 2>&1 | tail -2
COPER_HIP_LIB=$PWD/build/ab/lib_clk.so timeout 300 python tools/ab_score.py synth10m_cpg 128 50 2>&1 | tail -2
COPER_HIP_LIB=$PWD/build/ab/lib_clk.so timeout 300 python tools/ab_score.py synth10m_cpg 4096 10 2>&1 | tail -2
build/mb/mfma_rate 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
for v1 in 0 1; do
  if [ $v1 = 1 ]; then export COPER_SCORE_V1=1; else unset COPER_SCORE_V1; fi
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ab5_v1_$v1 -- python3 $GRAFT_REPO_ROOT/tools/ab_score.py fb15k237_cpg 20480 20 2>&1 | tail -1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_ab5_v1_0", "gpurun_out/pmc_ab5_v1_1"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "score_count" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, {c: sum(x) / len(x) for c, x in v.items()})
PY
} > gpurun_out/ab5.txt 2>&1
cat gpurun_out/ab5.txt
