#!/usr/bin/env python3
"""Prints VGPR/SGPR/LDS/spill/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "coper_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.startswith("kernels_"))
for f in files:
    out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", "-DCOPER_BUILD",
                          "-mllvm", "-amdgpu-mfma-vgpr-form", "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, f), "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            if cur: print(cur)
            name = t.split(":", 1)[1].strip()
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            cur = {"kernel": name}
        elif ":" in t:
            k, v = t.split(":", 1)
            if k.strip() in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPRs Spill", "SGPRs Spill"):
                cur[k.strip().replace(" [bytes/lane]","").replace(" [waves/SIMD]","").replace(" [bytes/block]","")] = v.strip()
    if cur: print(cur)
