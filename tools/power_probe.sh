#!/bin/bash
# Socket power / shader clock (rocm-smi, every 0.2 s) while a kernel runs in a loop:  bash tools/power_probe.sh   (GPU box)
#  (a) the count kernel at FB15k-237 shapes, (b) the same kernel in its HBM regime (10M x 256, 128 queries),
#  (c) the bare bf16 MFMA rate microbenchmark on random data, (d) the fused conv + dense kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
rocm-smi --showmaxpower 2>&1 | grep -i "power (W)" | head -2
python3 - > gpurun_out/pp_samples.txt <<'PY' &
import subprocess, time, re, sys
t0 = time.time()
while time.time() - t0 < 150:
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    m = re.search(r"Power \(W\): ([0-9.]+)", out)
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    print("%.2f %s %s" % (time.time(), m.group(1) if m else "nan", c.group(1) if c else "nan"), flush=True)
    time.sleep(0.15)
PY
SAMPLER=$!
mark() { python3 -c "import time; print('MARK %.2f $1' % time.time())" >> gpurun_out/pp_marks.txt; }
rm -f gpurun_out/pp_marks.txt
if true; then hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_rate.hip -o /tmp/mfma_rate 2>/dev/null; fi
mark "start mfma_rate, random data"; /tmp/mfma_rate 6 > gpurun_out/pp_c.log 2>&1; mark "end mfma_rate, random data"
mark "start mfma_rate, zeros"; /tmp/mfma_rate 6 zeros >> gpurun_out/pp_c.log 2>&1; mark "end mfma_rate, zeros"
COPER_PP_MARK=gpurun_out/pp_marks.txt python3 tools/ab_score.py fb15k237_cpg 20480 30000 > gpurun_out/pp_a.log 2>&1
COPER_PP_MARK=gpurun_out/pp_marks.txt python3 tools/ab_score.py synth10m_cpg 128 4000 > gpurun_out/pp_b.log 2>&1
COPER_PP_MARK=gpurun_out/pp_marks.txt python3 tools/ab_dense.py fb15k237_cpg 20480 20000 > gpurun_out/pp_d.log 2>&1
kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
cat gpurun_out/pp_c.log; grep -v clock gpurun_out/pp_a.log | tail -1; grep -v clock gpurun_out/pp_b.log | tail -1; tail -1 gpurun_out/pp_d.log
python3 - <<'PY'
marks = [l.split(None, 2) for l in open("gpurun_out/pp_marks.txt")]
S = [tuple(float(x) for x in l.split()) for l in open("gpurun_out/pp_samples.txt") if "nan" not in l.split()[1]]
ev = {}
for _, t, label in marks:
    kind, name = label.strip().split(" ", 1)
    ev.setdefault(name, {})[kind] = float(t)
for name, d in ev.items():
    if "start" in d and "end" in d:
        w = [s for s in S if d["start"] + 0.5 <= s[0] <= d["end"] - 0.2]
        if w:
            p = sorted(x[1] for x in w); c = sorted(x[2] for x in w)
            print("%-34s %5.1f s  power W: median %4.0f  p90 %4.0f  max %4.0f   sclk MHz median %4.0f  (%d samples)" % (name, d["end"] - d["start"], p[len(p) // 2], p[int(len(p) * 0.9)], p[-1], c[len(c) // 2], len(w)))
        else:
            print(name, "no samples in window", d)
PY
